"""Multi-GPU drivers: one process per GPU (torch.distributed, backend "nccl" = RCCL over xGMI),
the output triangle cut into equal-area row panels, no collective on the data path, one gather
(or, for PCA, the eigen solver's all-reduces) at the end.  Launch with
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 script.py

`blocks` is any iterable of genotype blocks (uint8 [b][n_samp] or 2-bit packed [b][ceil(n/4)]);
every rank iterates the same stream (in an R deployment: the kept GDS reader opened per rank).
"""
import numpy as np

from . import _lib
from .dist import gather_slabs, panel_rows, slab_range


def _env(group=None):
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(group), dist.get_world_size(group)
    return 0, 1


def _panel_ctx(kind, n, rank, world, device_index, max_block_snps, **kw):
    b = panel_rows(n, world)
    r0, r1 = b[rank], b[rank + 1]
    if r1 <= r0:
        return None, b
    full = (r0 == 0 and r1 == n)
    return _lib.Accumulator(kind, n, device=device_index, row_begin=0 if full else r0,
                            row_end=0 if full else r1, max_block_snps=max_block_snps, **kw), b


def _stream(acc, blocks):
    for blk in blocks:
        if acc is not None:
            acc.feed(blk)


def grm_distributed(blocks, n, method="GCTA", device_index=0, max_block_snps=16384, group=None, dst=0):
    """snpgdsGRM(method = "GCTA" | "Eigenstrat") across the ranks of `group`.
    Returns the packed upper triangle (torch float64 tensor on the device) on rank `dst`, else None."""
    import torch
    import torch.distributed as dist
    rank, world = _env(group)
    dev = torch.device("cuda", device_index)
    kind = _lib.GRM_GCTA if method == "GCTA" else _lib.PCA_COV
    acc, bounds = _panel_ctx(kind, n, rank, world, device_index, max_block_snps)
    _stream(acc, blocks)
    lo, hi = slab_range(n, bounds[rank], bounds[rank + 1])
    slab = torch.empty(hi - lo, dtype=torch.float64, device=dev)
    if method == "GCTA":
        if acc is not None:
            acc.grm_gcta(packed=True, out_ptr=slab.data_ptr())
    elif method == "Eigenstrat":
        tr = torch.tensor([acc.pca_panel_trace() if acc is not None else 0.0], dtype=torch.float64, device=dev)
        if world > 1:
            dist.all_reduce(tr, group=group)
        if acc is not None:
            acc.pca_cov(packed=True, normalize=True, trace_in=float(tr.item()), out_ptr=slab.data_ptr())
    else:
        raise ValueError("Invalid 'method'!")
    torch.cuda.synchronize(dev)
    if acc is not None:
        acc.close()
    if world == 1:
        return slab
    return gather_slabs(slab, n, bounds, rank, world, group=group, dst=dst)


def king_distributed(blocks, n, family=None, device_index=0, max_block_snps=16384, group=None, dst=0):
    """snpgdsIBDKING(type="KING-robust"): (IBS0, kinship) packed triangles on rank `dst`."""
    import torch
    rank, world = _env(group)
    dev = torch.device("cuda", device_index)
    acc, bounds = _panel_ctx(_lib.KING_ROBUST, n, rank, world, device_index, max_block_snps)
    _stream(acc, blocks)
    lo, hi = slab_range(n, bounds[rank], bounds[rank + 1])
    a = torch.empty(hi - lo, dtype=torch.float64, device=dev)
    b = torch.empty(hi - lo, dtype=torch.float64, device=dev)
    if acc is not None:
        acc.king_robust(family=family, packed=True, out_ptrs=(a.data_ptr(), b.data_ptr()))
        acc.close()
    torch.cuda.synchronize(dev)
    if world == 1:
        return a, b
    ga = gather_slabs(a, n, bounds, rank, world, group=group, dst=dst)
    gb = gather_slabs(b, n, bounds, rank, world, group=group, dst=dst)
    return ga, gb


def pca_distributed(blocks, n, eigen_cnt=32, bayesian=False, device_index=0, max_block_snps=16384,
                    group=None, tol=1e-9):
    """snpgdsPCA(algorithm="exact") across ranks: covariance panels stay distributed, the top
    `eigen_cnt` eigenpairs come from the block-Krylov solver (snprelate_amd/eigen.py).
    Every rank returns dict(eigenval, eigenvect [n, k], varprop, TraceXTX)."""
    import torch
    from .eigen import PanelOperator, topk_eigen
    rank, world = _env(group)
    dev = torch.device("cuda", device_index)
    acc, _ = _panel_ctx(_lib.PCA_COV, n, rank, world, device_index, max_block_snps, bayesian=bayesian)
    _stream(acc, blocks)
    op = PanelOperator([acc] if acc is not None else [], n, dev, group=group)
    w, v, info = topk_eigen(op, eigen_cnt, tol=tol)
    if acc is not None:
        acc.close()
    return dict(eigenval=w, eigenvect=v, varprop=w / (n - 1), TraceXTX=op.trace_xtx, info=info)
