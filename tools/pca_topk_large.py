#!/usr/bin/env python3
"""Covariance + top-k eigenvectors beyond the dense solver's reach (n > 46 340) on ONE MI355X: structured synthetic
genotypes (four populations) generated on the device, streamed through the PCA covariance accumulator, then the block-Krylov
solver over the resident fp64 panel (snprelate_amd/eigen.py).  Prints timings and the solver's residual.
    python tools/pca_topk_large.py [N=150000] [L=32768] [k=32]"""
import json
import sys
import time

import torch

sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
from snprelate_amd import _lib  # noqa: E402
from snprelate_amd.eigen import PanelOperator, topk_eigen  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 150000
L = int(sys.argv[2]) if len(sys.argv) > 2 else 32768
k = int(sys.argv[3]) if len(sys.argv) > 3 else 32
B = 8192
dev = torch.device("cuda", 0)
gen = torch.Generator(device=dev).manual_seed(11)
pop = (torch.arange(N, device=dev) * 4 // N)


def block(nb):
    """2-bit packed [nb][N/4] with allele frequencies shifted per population"""
    out = torch.empty((nb, N // 4), dtype=torch.uint8, device=dev)
    step = 1024
    for s in range(0, nb, step):
        e = min(nb, s + step)
        p = torch.rand((e - s, 1), generator=gen, device=dev) * 0.8 + 0.1
        shift = torch.randn((e - s, 4), generator=gen, device=dev) * 0.1
        pp = (p + shift[:, pop]).clamp_(0.02, 0.98)
        g = (torch.rand((e - s, N), generator=gen, device=dev) < pp).to(torch.uint8)
        g += (torch.rand((e - s, N), generator=gen, device=dev) < pp).to(torch.uint8)
        g = g.view(e - s, N // 4, 4)
        out[s:e] = g[:, :, 0] | (g[:, :, 1] << 2) | (g[:, :, 2] << 4) | (g[:, :, 3] << 6)
    return out


acc = _lib.Accumulator(_lib.PCA_COV, N, max_block_snps=B)
t_acc = 0.0
for _ in range(L // B):
    blk = block(B)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    acc.feed_device(blk.data_ptr(), B)
    acc.sync()
    t_acc += time.perf_counter() - t0
    del blk
torch.cuda.synchronize()
t0 = time.perf_counter()
op = PanelOperator([acc], N, dev)
w, v, info = topk_eigen(op, k, tol=1e-9)
torch.cuda.synchronize()
t_eig = time.perf_counter() - t0
free, total = torch.cuda.mem_get_info()
print(json.dumps({"n": N, "snps": L, "k": k, "accumulate_s": t_acc, "pair_genotypes_per_s": N * N / 2 * L / t_acc,
                  "eigen_s": t_eig, "panel_products": info["matmuls"], "restarts": info["restarts"],
                  "max_rel_residual": info["max_rel_residual"], "ms_per_panel_product_incl_algebra": t_eig / info["matmuls"] * 1e3,
                  "eigenval_head": [float(x) for x in w[:6].cpu()], "hbm_used_GiB": (total - free) / 2**30}))
acc.close()
