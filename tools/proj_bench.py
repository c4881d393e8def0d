#!/usr/bin/env python3
"""Timing of the PCA projection kernels (fp64, O(N B k)) on synthetic blocks:  tools/proj_bench.py [N] [B] [k]"""
import ctypes
import sys
import time

import numpy as np

sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
import torch  # noqa: E402  (device buffers only)

from snprelate_amd import _lib  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
B = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
k = int(sys.argv[3]) if len(sys.argv) > 3 else 32
L = _lib.lib()
dev = torch.device("cuda", 0)
g = torch.randint(0, 3, (B, (N + 3) // 4), dtype=torch.uint8, device=dev)     # 2-bit packed rows (random codes)
ev = torch.randn(k, N, dtype=torch.float64, device=dev)
out = torch.empty(B, k, dtype=torch.float64, device=dev)
af = torch.empty(B, dtype=torch.float64, device=dev)
sc = torch.empty(B, dtype=torch.float64, device=dev)
h = ctypes.c_void_p()
o = _lib.Opts(device=0, bayesian=0, row_begin=0, row_end=0, max_block_snps=B, stream=None)
_lib.check(L.snpgpu_proj_create(N, k, ctypes.byref(o), ctypes.byref(h)))
_lib.check(L.snpgpu_proj_set_eigvec(h, ctypes.c_void_p(ev.data_ptr()), _lib.DEVICE))
P = lambda t: ctypes.c_void_p(t.data_ptr())


def timed(fn, reps=5):
    fn(); L.snpgpu_proj_sync(h)
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    L.snpgpu_proj_sync(h)
    return (time.perf_counter() - t0) / reps


t_corr = timed(lambda: _lib.check(L.snpgpu_proj_snp_corr(h, P(g), B, _lib.GENO_PACKED2, _lib.DEVICE, P(out), _lib.DEVICE)))
t_load = timed(lambda: _lib.check(L.snpgpu_proj_snp_loading(h, P(g), B, _lib.GENO_PACKED2, _lib.DEVICE, 0, P(out), P(af), P(sc), _lib.DEVICE)))
t_samp = timed(lambda: _lib.check(L.snpgpu_proj_samp_loading_feed(h, P(g), B, _lib.GENO_PACKED2, _lib.DEVICE, P(out), P(af), P(sc), _lib.DEVICE)))
fl = 2.0 * N * B * k
print("N=%d B=%d k=%d" % (N, B, k))
print("  snp_corr     %8.2f ms/block  %6.2f TFLOP/s fp64 (3 FMA per sample, SNP, eigenvector)" % (t_corr * 1e3, 3 * fl / t_corr / 1e12))
print("  snp_loading  %8.2f ms/block  %6.2f TFLOP/s fp64" % (t_load * 1e3, fl / t_load / 1e12))
print("  samp_loading %8.2f ms/block  %6.2f TFLOP/s fp64" % (t_samp * 1e3, fl / t_samp / 1e12))
print("  genotype bytes per block %.1f MB -> %.2f TB/s at the snp_loading rate" % (N * B / 4 / 1e6, N * B / 4 / t_load / 1e12))
L.snpgpu_proj_destroy(h)
