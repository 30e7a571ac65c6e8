#!/usr/bin/env python3
"""The hand-written projection GEMM (csrc/gemm_ops.hip) against hipBLASLt (torch) on the layer's forward shapes:
correctness (max abs / relative Frobenius error vs an fp32 reference) and TFLOP/s."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import torch
from generative_recommenders_amd import _lib as hl
HERE = os.path.dirname(os.path.abspath(__file__))
_g = C.CDLL(os.path.join(HERE, "gemm_exp", "libgemm_exp.so"))        # tools/gemm_exp/build.sh
_g.hstu_gemm.restype = C.c_int
_g.hstu_gemm.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64,
                         C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int, C.c_void_p]
dev = "cuda"; bf = torch.bfloat16


def gemm(a, b, bias=None, residual=None, silu_cols=0):
    M, K = a.shape
    N = b.shape[1]
    c = torch.empty(M, N, dtype=a.dtype, device=a.device)
    rc = _g.hstu_gemm(a.data_ptr(), a.stride(0), b.data_ptr(), b.stride(0), c.data_ptr(), c.stride(0),
                      bias.data_ptr() if bias is not None else None, residual.data_ptr() if residual is not None else None,
                      residual.stride(0) if residual is not None else 0, M, N, K, silu_cols, hl.torch_dtype_code(a.dtype),
                      hl.current_stream_ptr(a.device))
    assert rc == 0, rc
    return c
torch.manual_seed(0)


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


out = {}
L = int(os.environ.get("ROWS", "194560"))
for name, K, N, silu in (("uvqk fwd (bias, SiLU on 512 cols)", 512, 2048, 512), ("output fwd (residual)", 1536, 512, 0)):
    x = torch.randn(L, K, device=dev, dtype=bf)
    W = (torch.randn(K, N, device=dev) * 0.05).to(bf)
    b = (torch.randn(N, device=dev) * 0.1).to(bf) if silu else None
    r = torch.randn(L, N, device=dev, dtype=bf) if not silu else None
    mine = gemm(x, W, bias=b, residual=r, silu_cols=silu)
    n_chk = min(L, 4096)
    ref = x[:n_chk].float() @ W.float()
    if b is not None:
        ref = ref + b.float()
        ref[:, :silu] = torch.nn.functional.silu(ref[:, :silu])
    if r is not None:
        ref = ref.to(bf).float() + r[:n_chk].float()       # the kernel rounds the product to bf16 before the residual add
    err = (mine[:n_chk].float() - ref).norm() / ref.norm()
    tail = (mine[-300:].float() - (x[-300:].float() @ W.float() + (b.float() if b is not None else 0))).abs().max() if silu == 0 and r is None else None

    def lib():
        y = torch.addmm(b, x, W) if b is not None else torch.mm(x, W)
        if silu:
            y[:, :silu] = torch.nn.functional.silu(y[:, :silu])
        if r is not None:
            y = y + r
        return y
    fl = 2.0 * L * K * N
    t_m, t_l, t_l0 = timed(lambda: gemm(x, W, bias=b, residual=r, silu_cols=silu)), timed(lib), timed(lambda: torch.mm(x, W))
    out[name] = {"rel_frobenius_err": float(err), "hand_TFLOPs": round(fl / t_m / 1e12, 1), "hand_us": round(t_m * 1e6, 1),
                 "hipblaslt_with_epilogue_ops_us": round(t_l * 1e6, 1), "hipblaslt_gemm_only_TFLOPs": round(fl / t_l0 / 1e12, 1),
                 "hipblaslt_gemm_only_us": round(t_l0 * 1e6, 1)}
print(json.dumps(out, indent=1))
