#!/usr/bin/env python3
"""Statistics of the fused dropout's counter-based generator (csrc/norm_kernels.inc drop_hash, restated bit for bit by
oracle/hstu_oracle.py::dropout_keep_mask) on the CPU: keep rate per tensor / row / column, neighbour / row / seed correlations as
z-scores, chi-square of the 16-bit uniforms' top byte.  Everything below ~4.5 is what independent draws give (the row and column
figures are maxima over thousands of rows / columns).

    python tools/dropout_hash_stats.py [--rows 4096] [--stride 1536]"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import hstu_oracle as O  # noqa: E402


def z_scores(seed, rows, stride, p):
    keep, scale = O.dropout_keep_mask(seed, rows, stride, p)
    pe = 1.0 - 1.0 / scale
    keep = keep.astype(np.float64)
    n = keep.size
    z = {"mean": abs(keep.mean() - (1 - pe)) / (pe * (1 - pe) / n) ** 0.5,
         "column max": np.abs(keep.mean(0) - (1 - pe)).max() / (pe * (1 - pe) / rows) ** 0.5,
         "row max": np.abs(keep.mean(1) - (1 - pe)).max() / (pe * (1 - pe) / stride) ** 0.5}
    k = keep - keep.mean()
    var = (k * k).mean()
    for name, a, c in (("lag 1", k[:, :-1], k[:, 1:]), ("lag 2", k[:, :-2], k[:, 2:]), ("lag 3", k[:, :-3], k[:, 3:]),
                       ("next row", k[:-1], k[1:]), ("row + 2", k[:-2], k[2:])):
        z[name] = abs((a * c).mean() / var) * a.size ** 0.5
    for name, other in (("seed + 1", seed + 1), ("high seed word + 1", seed ^ (1 << 32))):
        k2 = O.dropout_keep_mask(other, rows, stride, p)[0].astype(np.float64) - keep.mean()
        z[name] = abs((k * k2).mean() / var) * n ** 0.5
    return z


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=4096)
    ap.add_argument("--stride", type=int, default=1536)
    a = ap.parse_args()
    worst = 0.0
    for seed in (42, 0, 2 ** 40 + 7, 123456789012345):
        for p in (0.1, 0.3, 0.5):
            z = z_scores(seed, a.rows, a.stride, p)
            worst = max(worst, max(z.values()))
            print(f"seed {seed} p {p}: " + ", ".join(f"{k} {v:.2f}" for k, v in z.items()))
    print(f"worst z {worst:.2f}")


if __name__ == "__main__":
    main()
