#!/usr/bin/env python3
"""Step time of one accumulator kind on synthetic blocks (N = 10 000, 65 536 SNPs per block):
    python tools/kind_timing.py INDIV_BETA 0.05      python tools/kind_timing.py KING_HOMO 0.0"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from snprelate_amd import _lib

n, B, steps = 10000, 65536, 20
kind = sys.argv[1] if len(sys.argv) > 1 else "INDIV_BETA"
miss = float(sys.argv[2]) if len(sys.argv) > 2 else 0.05
rb = (n + 3) // 4
blk = torch.empty((B, rb), dtype=torch.uint8, device="cuda")
_lib.synth_block(blk.data_ptr(), n, 0, B, 20240601, miss, 0, False, 0)
a = _lib.Accumulator(getattr(_lib, kind), n, max_block_snps=B)
for _ in range(5):
    a.feed_device(blk.data_ptr(), B, fmt=_lib.GENO_PACKED2) if hasattr(_lib, "GENO_PACKED2") else a.feed_device(blk.data_ptr(), B)
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(steps):
    a.feed_device(blk.data_ptr(), B, fmt=_lib.GENO_PACKED2) if hasattr(_lib, "GENO_PACKED2") else a.feed_device(blk.data_ptr(), B)
a.sync() if hasattr(a, "sync") else torch.cuda.synchronize()
torch.cuda.synchronize()
dt = (time.perf_counter() - t) / steps
print(kind + ": %.3f ms per %d-SNP block at N = %d, %.0f %% missing -> %.3e pair-genotypes/s" % (dt * 1e3, B, n, miss * 100, n * n / 2 * B / dt))
a.close()
