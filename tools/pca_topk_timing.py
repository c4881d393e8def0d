#!/usr/bin/env python3
"""Wall time of the top-k eigen solve on a device-resident covariance (block Krylov over panel dgemms):
tools/pca_topk_timing.py [N] [L] [k]"""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
from snprelate_amd import multigpu  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
L = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
k = int(sys.argv[3]) if len(sys.argv) > 3 else 32
rng = np.random.default_rng(1)
p = rng.uniform(0.1, 0.9, size=(L, 1)).astype(np.float32)
pop = (np.arange(N) * 4 // N)
shift = rng.normal(0, 0.1, size=(L, 4)).astype(np.float32)
blocks = []
for b0 in range(0, L, 4096):
    pp = np.clip(p[b0:b0 + 4096] + shift[b0:b0 + 4096][:, pop], 0.02, 0.98)
    g = (rng.random((len(pp), N), dtype=np.float32) < pp).astype(np.uint8) + (rng.random((len(pp), N), dtype=np.float32) < pp).astype(np.uint8)
    blocks.append(g)
torch.cuda.synchronize()
t0 = time.perf_counter()
r = multigpu.pca_distributed(blocks, N, eigen_cnt=k, max_block_snps=4096)
torch.cuda.synchronize()
t1 = time.perf_counter()
print("N=%d L=%d k=%d: covariance + top-%d eigen solve %.2f s; info %s" % (N, L, k, k, t1 - t0, r["info"]))
print("eigenval[:6]", r["eigenval"][:6].cpu().numpy().round(3))
