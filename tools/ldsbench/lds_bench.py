#!/usr/bin/env python3
"""Cycles per LDS wave-instruction for the access patterns of the attention kernels under candidate
tile swizzles (run on the GPU box: python tools/ldsbench/lds_bench.py)."""
import ctypes as C, os, sys
import torch
HERE = os.path.dirname(os.path.abspath(__file__))
lib = C.CDLL(os.path.join(HERE, "liblds_bench.so"))
lib.lds_bench_run.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]

def run(addrs, kind, threads=256, iters=200):
    a = torch.tensor(addrs, dtype=torch.int32, device="cuda")
    out = torch.zeros(64, dtype=torch.int64, device="cuda")
    for _ in range(2):
        lib.lds_bench_run(a.data_ptr(), kind, iters, threads, out.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    w = threads // 64
    return out[:w].max().item() / (iters * 16)

SWZ = {
    "old r&15": lambda r: r & 15,
    "new (r&3)<<2|(r>>2)&3": lambda r: ((r & 3) << 2) | ((r >> 2) & 3),
    "none": lambda r: 0,
    "r&3<<2": lambda r: (r & 3) << 2,
    "(r>>2)&3": lambda r: (r >> 2) & 3,
    "(r&3)<<2 | (r>>3)&1 | ... r2<<1": lambda r: ((r & 3) << 2) | ((r >> 3) & 1) | (((r >> 2) & 1) << 1),
    "r&7": lambda r: r & 7,
    "(r&1)<<3|(r>>1)&7": lambda r: ((r & 1) << 3) | ((r >> 1) & 7),
    "(r&7)<<1|(r>>3)&1": lambda r: ((r & 7) << 1) | ((r >> 3) & 1),
}

def tile_off(r, u, f, upr=16):
    return (r * upr + (u ^ f(r))) * 16

def tr_pattern(f, rowA, colblk, hf_rows=4):
    """lds_col_frag 'a' read: lane -> row rowA(hf) + i16>>2, 8 bytes at col"""
    out = []
    for lane in range(64):
        i16, hf = lane & 15, lane >> 5
        col = colblk + (((lane >> 4) & 1) << 4) + ((i16 & 3) << 2)
        r = rowA + hf_rows * hf + (i16 >> 2)
        out.append(tile_off(r, col >> 3, f) + ((col & 7) << 1))
    return out

def row_pattern(f, unit_of_hf):
    out = []
    for lane in range(64):
        n32, hf = lane & 31, lane >> 5
        out.append(tile_off(n32, unit_of_hf(hf), f))
    return out

if __name__ == "__main__":
    for threads in (256, 512):
        print(f"== {threads} threads")
        for name, f in SWZ.items():
            tr = [run(tr_pattern(f, ra, cb), 0, threads) for ra in (0, 16) for cb in (0, 32, 64, 96)]
            tr8 = [run(tr_pattern(f, ra, cb, 8), 0, threads) for ra in (0, 16) for cb in (0, 64)]
            rw = [run(row_pattern(f, lambda hf, kg=kg: hf * 8 + kg), 1, threads) for kg in (0, 3, 7)]
            print(f"{name:40s} tr(rows 4hf) {min(tr):6.1f}..{max(tr):6.1f}  tr(rows 8hf) {min(tr8):6.1f}..{max(tr8):6.1f}   b128 row {min(rw):6.1f}..{max(rw):6.1f}")


def fold_patterns():
    """access patterns of the folded backward's dQ GEMM (16x16x32 fragments)"""
    swz = lambda r: ((r & 3) << 2) | ((r >> 2) & 3)
    def ds_off(row, chunk): return (row << 6) + ((chunk ^ ((row >> 1) & 7)) << 3)
    pats = {}
    for wave in (0, 1, 5):
        for hi in (0, 4):
            a = []
            for lane in range(64):
                i16, g = lane & 15, lane >> 4
                row = 8 * g + (i16 >> 2) + hi
                col = 16 * wave + 4 * (i16 & 3)
                a.append((row * 16 + ((col >> 3) ^ swz(row))) * 16 + ((col & 7) << 1))
            pats[f"K^T wave{wave} hi{hi}"] = a
    for qb in (0, 1):
        for hi in (0, 4):
            a = []
            for lane in range(64):
                i16, g = lane & 15, lane >> 4
                row = 8 * g + (i16 >> 2) + hi
                a.append(ds_off(row, 4 * qb + (i16 & 3)))
            pats[f"dS qb{qb} hi{hi}"] = a
    # publish writes (ds_write_b64): lane (n32, hf), chunk hf + 2 rq
    for rq in (0, 3):
        pats[f"publish rq{rq} (write b64)"] = [ds_off(l & 31, (l >> 5) + 2 * rq) for l in range(64)]
    return pats


if __name__ == "__main__":
    print("== folded backward dQ GEMM patterns (cycles per wave-instruction, 4 / 8 waves)")
    for name, a in fold_patterns().items():
        kind = 2 if "write" in name else 0
        print(f"{name:32s} {run(a, kind, 256):6.1f} {run(a, kind, 512):6.1f}")
