"""RCCL and the gradient reducer's stream discipline on the ONE GPU a test box has.

A one-rank ``nccl`` (= RCCL on ROCm) process group is the only communicator a single MI355X can host, and it is enough to put on
the hardware what gloo on CPU cannot test: the reducer's backward hooks run on autograd's thread, stage a layer's gradients into
the bucket with one multi-tensor copy on the COMPUTE stream, launch an asynchronous all-reduce on RCCL's own stream, and hand the
results back as ``p.grad`` views that an optimizer step then reads on the compute stream.  With ``single_rank_collectives=True``
all of that runs exactly as it does with 8 ranks (a one-rank all-reduce is the identity), so the gradients and the parameters
after the optimizer steps must equal, bit for bit, those of a run without any reducer.  The worker runs in its own process: a
process group is global state the other tests must not see.
"""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import json, os, sys
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
sys.path.insert(0, %(root)r)
import torch
import torch.distributed as dist
from generative_recommenders_amd import data_parallel as dp
from generative_recommenders_amd.modules.stu import STULayer, STULayerConfig, STUStack

os.environ.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=%(port)r)
assert dp.init_from_env(backend="nccl", single_rank_group=True) == (0, 0, 1)
assert dist.is_initialized() and dist.get_backend() == "nccl" and dist.get_world_size() == 1
dev = torch.device("cuda", 0)
info = dp.describe_ranks(dev)

N, H, d = 64, 2, 32
D = H * d
def make():
    torch.manual_seed(11)
    st = STUStack([STULayer(STULayerConfig(embedding_dim=D, num_heads=H, hidden_dim=d, attention_dim=d, output_dropout_ratio=0.0,
                                           use_group_norm=True)) for _ in range(3)]).to(dev)
    st.train()
    return st
model, ref = make(), make()
red = dp.GradientAllReducer(None, buckets=[l.parameters() for l in model._stu_layers], overlap=True, single_rank_collectives=True,
                            check_every=10)
opt_m = torch.optim.SGD(model.parameters(), lr=1e-4, momentum=0.9)
opt_r = torch.optim.SGD(ref.parameters(), lr=1e-4, momentum=0.9)
g = torch.Generator(device=dev).manual_seed(3)
steps, unequal, finite = 50, 0, True
for step in range(steps):
    lengths = torch.randint(1, N + 1, (12,), generator=g, device=dev)
    off = dp.local_offsets(lengths)
    L = int(off[-1])
    x = torch.randn(L, D, device=dev, dtype=torch.bfloat16, generator=g)
    gy = 0.1 * torch.randn(L, D, device=dev, dtype=torch.bfloat16, generator=g)
    half = L // 2
    def run(m, accumulate):
        for p in m.parameters():
            p.grad = None
        if accumulate:                          # two micro-batches: the first only accumulates
            ctx = red.no_sync() if m is model else None
            if ctx is not None:
                with ctx:
                    m(x=x, x_lengths=lengths, x_offsets=off, max_seq_len=N, num_targets=None).backward(gy)
            else:
                m(x=x, x_lengths=lengths, x_offsets=off, max_seq_len=N, num_targets=None).backward(gy)
        m(x=x, x_lengths=lengths, x_offsets=off, max_seq_len=N, num_targets=None).backward(gy)
    acc = step %% 3 == 2
    run(model, acc)
    red.reduce()
    run(ref, acc)
    for p, q in zip(model.parameters(), ref.parameters()):
        assert p.grad is not None and q.grad is not None
        finite = finite and bool(torch.isfinite(p.grad).all()) and bool(torch.isfinite(q.grad).all())
        unequal += 0 if torch.equal(p.grad, q.grad) else 1
    opt_m.step()
    opt_r.step()
torch.cuda.synchronize()
params_equal = all(torch.equal(p, q) for p, q in zip(model.parameters(), ref.parameters()))
# the probe bench.py reports at --gpus 1: a 22 MB all-reduce on the communicator
t = torch.ones((22 << 20) // 4, device=dev)
dist.all_reduce(t)
torch.cuda.synchronize()
ok = bool((t == 1).all())
print("RESULT " + json.dumps({"unequal_grads": unequal, "finite": finite, "params_equal": params_equal, "allreduce_identity": ok, "calls": red._calls,
                              "backend": info["backend"], "rccl_version": info.get("rccl_version"), "steps": steps}))
dist.destroy_process_group()
'''


@pytest.mark.gpu
def test_one_rank_rccl_group_runs_the_reducer_bit_for_bit(tmp_path):
    script = tmp_path / "rccl_worker.py"
    script.write_text(WORKER % {"root": ROOT, "port": str(29300 + os.getpid() % 200)})
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("RESULT ")]
    assert line, r.stdout[-2000:]
    res = json.loads(line[-1][7:])
    assert res["backend"] == "nccl" and res["rccl_version"], res
    assert res["finite"] and res["unequal_grads"] == 0 and res["params_equal"] and res["allreduce_identity"], res
    assert res["calls"] == res["steps"] == 50, res
    try:
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, "gpurun_out", "rccl_single_rank.json"), "w") as f:
            json.dump(res, f)
    except OSError:
        pass
