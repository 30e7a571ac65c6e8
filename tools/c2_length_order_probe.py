import sys, argparse, torch
sys.path.insert(0, '/root/repo')
import bench
dev = torch.device('cuda')
def run(sort_mode):
    orig = bench.make_lengths
    def ml(wl, B, N, gen, device):
        l = orig(wl, B, N, gen, device)
        if sort_mode == 'desc': l = torch.sort(l, descending=True).values
        if sort_mode == 'full': l = torch.full_like(l, int(l.float().pow(2).mean().sqrt().item()))
        return l
    bench.make_lengths = ml
    a = argparse.Namespace(workload='C2', max_seq_len=211, heads=4, head_dim=64, users_per_gpu=8192, steps=20, warmup=5, sort_by_length=False,
                           parity_users=0, prewarm_s=0.3)
    att = bench.attention_section(a, 0, 1, dev)
    bench.make_lengths = orig
    print(sort_mode, 'fwd', round(att['fwd_ms'], 4), 'bwd', round(att['bwd_ms'], 4), att['kernels'])
for m in ('none', 'desc', 'none', 'desc', 'full'):
    run(m)
