#!/usr/bin/env python3
"""Per-kernel roofline table for the HBM-bound helper kernels of the path: algorithmic
bytes / HIP-event time vs the 8 TB/s HBM peak, at the DLRM-v3 layer shape (1024 users -- or argv[1] -- L ~ U[180,200), D = 512, bf16).
Prints one JSON object (also the torch CPU time of the same op on a bounded sample, as the CPU leg).
Run on the GPU box:  python tools/bench_ops.py > gpurun_out/bench_ops.json"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from generative_recommenders_amd.ops import _launch

PEAK = 8000.0
dev = "cuda"
torch.manual_seed(0)
B, N, D, H = (int(sys.argv[1]) if len(sys.argv) > 1 else 1024), 200, 512, 4   # 8192 users: every working set is far beyond the 256 MiB Infinity Cache
lengths = torch.randint(180, 200, (B,), device=dev)
off = _launch.complete_cumsum(lengths)
L = int(off[-1])
bf = torch.bfloat16
x = torch.randn(L, D, device=dev, dtype=bf)
dy = torch.randn(L, D, device=dev, dtype=bf)
w = torch.ones(D, device=dev, dtype=bf); b = torch.zeros(D, device=dev, dtype=bf)
gw = torch.ones(H, device=dev, dtype=bf); gb = torch.zeros(H, device=dev, dtype=bf)
u = torch.randn(L, D, device=dev, dtype=bf)
dy3 = torch.randn(L, 3 * D, device=dev, dtype=bf)
uvqk = torch.randn(L, 4 * D, device=dev, dtype=bf)
lens_r = torch.randint(1, 21, (B,), device=dev)
off_r = _launch.complete_cumsum(lens_r)
xr = torch.randn(int(off_r[-1]), D, device=dev, dtype=bf)


def timed(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


es = 2
y_ln, mean, rstd = _launch.layer_norm_fwd(x, w, b, 1e-6)
y_nm, m2, r2 = _launch.norm_mul_fwd(x, u, gw, gb, 1e-6, H, D // H, True, True)
cat = _launch.concat_2d_jagged(x, xr, off, off_r, N, 20, N + 20)
dense = _launch.jagged_to_padded_dense(x, off, N)
rows = {
    "complete_cumsum (B=1024, int64)": (lambda: _launch.complete_cumsum(lengths), 2 * B * 8),
    "layer_norm_fwd": (lambda: _launch.layer_norm_fwd(x, w, b, 1e-6), 2 * L * D * es),
    "layer_norm_bwd": (lambda: _launch.layer_norm_bwd(dy, x, w, mean, rstd), 3 * L * D * es),
    "swish_layer_norm_fwd": (lambda: _launch.swish_layer_norm_fwd(x, w, b, 1e-5), 2 * L * D * es),
    "swish_layer_norm_bwd": (lambda: _launch.swish_layer_norm_bwd(dy, x, w, b, mean, rstd), 3 * L * D * es),
    "norm_mul_fwd (group norm, concat [u, attn, y])": (lambda: _launch.norm_mul_fwd(x, u, gw, gb, 1e-6, H, D // H, True, True), 5 * L * D * es),
    "norm_mul_bwd (group norm, concat)": (lambda: _launch.norm_mul_bwd(dy3, x, u, gw, gb, m2, r2, H, D // H, True, True), 7 * L * D * es),
    "silu_fwd on the u slice of uvqk": (lambda: _launch.silu_fwd(uvqk[:, :D]), 2 * L * D * es),
    "concat_2d_jagged (history + targets)": (lambda: _launch.concat_2d_jagged(x, xr, off, off_r, N, 20, N + 20), 2 * (L + xr.shape[0]) * D * es),
    "split_2d_jagged": (lambda: _launch.split_2d_jagged(cat, L, xr.shape[0], off, off_r, N, 20, N + 20), 2 * (L + xr.shape[0]) * D * es),
    "jagged_to_padded_dense": (lambda: _launch.jagged_to_padded_dense(x, off, N), L * D * es + B * N * D * es),
    "dense_to_jagged": (lambda: _launch.dense_to_jagged(dense, off, L), 2 * L * D * es),
}
# timestamp / position encoder (the step before the STU stack)
from generative_recommenders_amd.ops.position import add_timestamp_positional_embeddings, _table_grad
ts_all = torch.cat([torch.sort(torch.randint(0, 3 * 10**6, (int(l),), device=dev)).values for l in lengths.tolist()])
pos_w = torch.randn(8192, D, device=dev) * 0.1
ts_w = torch.randn(2049, D, device=dev) * 0.1
nt = torch.randint(1, 21, (B,), device=dev)
def _enc():
    return add_timestamp_positional_embeddings(alpha=D**0.5, max_seq_len=N, max_contextual_seq_len=0, position_embeddings_weight=pos_w,
        timestamp_embeddings_weight=ts_w, seq_offsets=off, seq_lengths=lengths, seq_embeddings=x, timestamps=ts_all, num_targets=nt,
        interleave_targets=False)
pidx = torch.randint(0, 200, (L,), device=dev, dtype=torch.int32)
rows["add_timestamp_positional_embeddings fwd"] = (_enc, 2 * L * D * es + L * 8)
rows["position-table gradient (sort + segment sum)"] = (lambda: _table_grad(dy, pidx, 8192), L * D * es + 8192 * D * 4)
rows["l2_norm_fwd (output postprocessor)"] = (lambda: _launch.l2_norm_fwd(x, 1e-6), 2 * L * D * es)
rows["l2_norm_bwd"] = (lambda: _launch.l2_norm_bwd(dy, x, 1e-6), 3 * L * D * es)
# sampled-softmax loss at the Amazon-Books shape: 128 users x <= 50 positions, 512 negatives per
# position, D = 64, 695,762 items, l2 norm, T = 0.05 (configs/amzn-books/hstu-sampled-softmax-n512-final.gin)
from generative_recommenders_amd.research.modeling.sequential.losses.sampled_softmax import sampled_softmax_row_loss
SS_N, SS_R, SS_D, SS_V = 4096, 512, 64, 695762
ss = {}
for ss_dt, ss_es in ((torch.float32, 4), (torch.bfloat16, 2)):
    tab = (torch.randn(SS_V, SS_D, device=dev) * 0.1).to(ss_dt).requires_grad_()
    sq = torch.randn(SS_N, SS_D, device=dev).to(ss_dt).requires_grad_()
    sids = torch.randint(0, SS_V, (SS_N,), device=dev)
    spos = tab.detach()[sids].clone().requires_grad_()
    srows = torch.randint(0, SS_V, (SS_N, SS_R), device=dev)
    sg = torch.full((SS_N,), 1.0 / SS_N, device=dev)
    fwd = lambda: sampled_softmax_row_loss(sq, spos, tab, sids, srows, srows, 0.05, True, True, 1e-6)
    rl = fwd()
    lse = rl.grad_fn.saved_tensors[-1] if False else None
    nb_f = SS_N * SS_R * SS_D * ss_es + 2 * SS_N * SS_D * ss_es + 8 * SS_N * SS_R
    tag = "fp32" if ss_es == 4 else "bf16"
    rows[f"sampled_softmax fwd ({tag}, 4096 rows x 512 negatives x D=64)"] = (fwd, nb_f)
    row_loss_k, lse_k = _launch.sampled_softmax_fwd(sq.detach(), spos.detach(), sids, srows, srows, tab.detach(), 0.05, True, True, 1e-6)
    bwd = (lambda q_=sq.detach(), p_=spos.detach(), t_=tab.detach(), i_=sids, r_=srows, l_=lse_k:
           _launch.sampled_softmax_bwd(sg, l_, q_, p_, i_, r_, r_, t_, 0.05, True, True, 1e-6))
    # backward: the gather again + read-modify-write of the same rows in the fp32 gradient table + its zero fill
    nb_b = nb_f + 2 * SS_N * SS_R * SS_D * 4 + SS_V * SS_D * 4 + 2 * SS_N * SS_D * ss_es
    rows[f"sampled_softmax bwd ({tag}; incl. zeroing the {SS_V}x{SS_D} fp32 gradient table)"] = (bwd, nb_b)
    if ss_es == 4:
        # the reference's algorithm composed from torch ops on the same GPU (sampled_softmax.py:60-93)
        def torch_ref():
            ne = tab[srows]
            ne = ne / torch.clamp(torch.linalg.norm(ne, ord=2, dim=-1, keepdim=True), min=1e-6)
            pe = spos / torch.clamp(torch.linalg.norm(spos, ord=2, dim=-1, keepdim=True), min=1e-6)
            lp = (sq * pe).sum(-1, keepdim=True) / 0.05
            ln = torch.bmm(ne, sq.unsqueeze(2)).squeeze(2)
            ln = torch.where(sids.unsqueeze(1) == srows, -5e4, ln / 0.05)
            return (-torch.nn.functional.log_softmax(torch.cat([lp, ln], dim=1), dim=1)[:, 0] * sg).sum()
        def torch_ref_fb():
            tab.grad = None; sq.grad = None; spos.grad = None
            torch_ref().backward()
        def ours_fb():
            tab.grad = None; sq.grad = None; spos.grad = None
            (fwd() * sg).sum().backward()
        ss = {"torch_composition_fwd_us": round(timed(torch_ref, 5) * 1e6, 1), "torch_composition_fwd_bwd_us": round(timed(torch_ref_fb, 5) * 1e6, 1),
              "fused_fwd_bwd_us (autograd, incl. table-gradient cast)": round(timed(ours_fb, 5) * 1e6, 1)}
out = {"shape": {"users": B, "rows": L, "D": D, "heads": H, "dtype": "bf16"}, "peak_GBps": PEAK, "kernels": {}}
out["sampled_softmax_vs_torch_composition_fp32"] = ss
for name, (fn, nbytes) in rows.items():
    t = timed(fn)
    out["kernels"][name] = {"us": round(t * 1e6, 1), "algorithmic_bytes": nbytes, "GBps": round(nbytes / t / 1e9, 1),
                            "frac_of_hbm_peak": round(nbytes / t / 1e9 / PEAK, 3)}
# the two projections (MFMA-bound rows a7 / a8): hipBLASLt through torch, fwd + data gradient + weight gradient
from generative_recommenders_amd.ops.mm import weight_grad_mm
MFMA_PEAK = 2500.0   # dense bf16 TFLOP/s (MI355X_MICROARCH.md)
Wu = torch.randn(D, 4 * D, device=dev, dtype=bf) * 0.02
bu = torch.zeros(4 * D, device=dev, dtype=bf)
Wo = torch.randn(3 * D, D, device=dev, dtype=bf) * 0.02
g4 = torch.randn(L, 4 * D, device=dev, dtype=bf)
gemms = {
    "uvqk fwd  (L x 512)(512 x 2048) addmm": (lambda: torch.addmm(bu, x, Wu), 2.0 * L * D * 4 * D),
    "uvqk dgrad (L x 2048)(2048 x 512)": (lambda: torch.mm(g4, Wu.t()), 2.0 * L * D * 4 * D),
    "uvqk wgrad (512 x L)(L x 2048) split-K": (lambda: weight_grad_mm(x, g4), 2.0 * L * D * 4 * D),
    "output fwd (L x 1536)(1536 x 512)": (lambda: torch.mm(dy3, Wo), 2.0 * L * 3 * D * D),
    "output dgrad (L x 512)(512 x 1536)": (lambda: torch.mm(dy, Wo.t()), 2.0 * L * 3 * D * D),
    "output wgrad (1536 x L)(L x 512) split-K": (lambda: weight_grad_mm(dy3, dy), 2.0 * L * 3 * D * D),
}
out["projections"] = {"mfma_peak_TFLOPs": MFMA_PEAK, "gemms": {}}
tot_f = tot_t = 0.0
for name, (fn, fl) in gemms.items():
    t = timed(fn)
    tot_f += fl; tot_t += t
    out["projections"]["gemms"][name] = {"us": round(t * 1e6, 1), "TFLOPs": round(fl / t / 1e12, 1), "frac_of_mfma_peak": round(fl / t / 1e12 / MFMA_PEAK, 3)}
out["projections"]["all_six"] = {"us": round(tot_t * 1e6, 1), "TFLOPs": round(tot_f / tot_t / 1e12, 1), "frac_of_mfma_peak": round(tot_f / tot_t / 1e12 / MFMA_PEAK, 3)}

# CPU leg: the torch CPU equivalents of the two heaviest row kernels on a 128-user sample
torch.set_num_threads(min(32, os.cpu_count() or 1))
n = int(off[128])
xc, uc = x[:n].float().cpu(), u[:n].float().cpu()
t0 = time.perf_counter()
for _ in range(3):
    yc = torch.nn.functional.layer_norm(xc, (D,), eps=1e-6)
t_ln = (time.perf_counter() - t0) / 3
t0 = time.perf_counter()
for _ in range(3):
    g = torch.nn.functional.group_norm(xc.view(n, H, D // H).transpose(1, 2).reshape(n, D // H * H)[:, :, None].reshape(n, D, 1)[:, :, 0].view(n, H, D // H), H) if False else torch.nn.functional.layer_norm(xc.view(n, H, D // H), (D // H,), eps=1e-6).view(n, D)
    yc = torch.cat([uc, xc, uc * g], dim=1)
t_nm = (time.perf_counter() - t0) / 3
out["cpu_reference"] = {"sample_rows": n, "threads": torch.get_num_threads(),
                        "layer_norm_fwd_GBps": round(2 * n * D * 4 / t_ln / 1e9, 2),
                        "norm_mul_fwd_GBps": round(5 * n * D * 4 / t_nm / 1e9, 2)}
print(json.dumps(out, indent=1))
