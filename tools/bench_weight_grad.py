import sys, time; sys.path.insert(0, "/root/repo")
import torch
from generative_recommenders_amd.ops.mm import weight_grad_mm
torch.manual_seed(0)
L, K, N = 154321, 512, 2048
x = torch.randn(L, K, device="cuda", dtype=torch.bfloat16); dy = torch.randn(L, N, device="cuda", dtype=torch.bfloat16)
ref = (x.float().t() @ dy.float())
for name, fn in (("mm", lambda: torch.mm(x.t(), dy)), ("split", lambda: weight_grad_mm(x, dy))):
    out = fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10): out = fn()
    torch.cuda.synchronize()
    err = (out.float() - ref).abs().max().item() / ref.abs().max().item()
    print(name, f"{(time.perf_counter()-t0)/10*1e3:.3f} ms", "rel max err", f"{err:.2e}")
y = torch.randn(L, 1536, device="cuda", dtype=torch.bfloat16); do = torch.randn(L, 512, device="cuda", dtype=torch.bfloat16)
for name, fn in (("mm", lambda: torch.mm(y.t(), do)), ("split", lambda: weight_grad_mm(y, do))):
    out = fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10): out = fn()
    torch.cuda.synchronize(); print("out-proj", name, f"{(time.perf_counter()-t0)/10*1e3:.3f} ms")
