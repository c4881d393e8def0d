"""Multi-GPU drivers: one process per GPU (torch.distributed, backend "nccl" = RCCL over xGMI),
the output triangle cut into equal-area row panels, no collective on the data path, one gather
(or, for PCA, the eigen solver's all-reduces) at the end.  Launch with
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 script.py

`blocks` is any iterable of genotype blocks (uint8 [b][n_samp] or 2-bit packed [b][ceil(n/4)]);
every rank iterates the same stream (in an R deployment: the kept GDS reader opened per rank).
"""
import numpy as np

from . import _lib
from .dist import gather_plan, panel_plan, slab_range


def _env(group=None):
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(group), dist.get_world_size(group)
    return 0, 1


def _panel_ctxs(kind, n, rank, world, device_index, max_block_snps, panels_per_rank=1, **kw):
    """This rank's accumulators (one per owned panel, None for an empty panel) + the plan.
    panels_per_rank > 1 balances accumulator memory across GPUs (dist.panel_plan)."""
    bounds, owned = panel_plan(n, world, panels_per_rank)
    accs = []
    for p in owned[rank]:
        r0, r1 = bounds[p], bounds[p + 1]
        if r1 <= r0:
            accs.append(None)
            continue
        full = (r0 == 0 and r1 == n)
        accs.append(_lib.Accumulator(kind, n, device=device_index, row_begin=0 if full else r0,
                                     row_end=0 if full else r1, max_block_snps=max_block_snps, **kw))
    return accs, bounds, owned


def _stream(accs, blocks):
    for blk in blocks:
        for acc in accs:
            if acc is not None:
                acc.feed(blk)


def _slab(n, bounds, p, dev, dtype):
    import torch
    lo, hi = slab_range(n, bounds[p], bounds[p + 1])
    return torch.empty(hi - lo, dtype=dtype, device=dev)


def grm_distributed(blocks, n, method="GCTA", device_index=0, max_block_snps=16384, group=None, dst=0,
                    panels_per_rank=1):
    """snpgdsGRM(method = "GCTA" | "Eigenstrat") across the ranks of `group`.
    Returns the packed upper triangle (torch float64 tensor on the device) on rank `dst`, else None."""
    import torch
    import torch.distributed as dist
    if method not in ("GCTA", "Eigenstrat"):
        raise ValueError("Invalid 'method'!")
    rank, world = _env(group)
    dev = torch.device("cuda", device_index)
    kind = _lib.GRM_GCTA if method == "GCTA" else _lib.PCA_COV
    accs, bounds, owned = _panel_ctxs(kind, n, rank, world, device_index, max_block_snps, panels_per_rank)
    _stream(accs, blocks)
    slabs = [_slab(n, bounds, p, dev, torch.float64) for p in owned[rank]]
    if method == "GCTA":
        for acc, slab in zip(accs, slabs):
            if acc is not None:
                acc.grm_gcta(packed=True, out_ptr=slab.data_ptr())
    else:
        tr = torch.tensor([sum(a.pca_panel_trace() for a in accs if a is not None)], dtype=torch.float64, device=dev)
        if world > 1:
            dist.all_reduce(tr, group=group)
        for acc, slab in zip(accs, slabs):
            if acc is not None:
                acc.pca_cov(packed=True, normalize=True, trace_in=float(tr.item()), out_ptr=slab.data_ptr())
    torch.cuda.synchronize(dev)
    for acc in accs:
        if acc is not None:
            acc.close()
    return gather_plan(slabs, n, bounds, owned, rank, world, group=group, dst=dst) if world > 1 else _join(slabs, n, bounds, owned)


def _join(slabs, n, bounds, owned):
    """world == 1: concatenate this rank's slabs (it owns every panel, in index order)."""
    import torch
    return slabs[0] if len(slabs) == 1 else torch.cat(slabs)


def king_distributed(blocks, n, family=None, device_index=0, max_block_snps=16384, group=None, dst=0,
                     panels_per_rank=1):
    """snpgdsIBDKING(type="KING-robust"): (IBS0, kinship) packed triangles on rank `dst`."""
    import torch
    rank, world = _env(group)
    dev = torch.device("cuda", device_index)
    accs, bounds, owned = _panel_ctxs(_lib.KING_ROBUST, n, rank, world, device_index, max_block_snps, panels_per_rank)
    _stream(accs, blocks)
    sa = [_slab(n, bounds, p, dev, torch.float64) for p in owned[rank]]
    sb = [_slab(n, bounds, p, dev, torch.float64) for p in owned[rank]]
    for acc, a, b in zip(accs, sa, sb):
        if acc is not None:
            acc.king_robust(family=family, packed=True, out_ptrs=(a.data_ptr(), b.data_ptr()))
            acc.close()
    torch.cuda.synchronize(dev)
    if world == 1:
        return _join(sa, n, bounds, owned), _join(sb, n, bounds, owned)
    ga = gather_plan(sa, n, bounds, owned, rank, world, group=group, dst=dst)
    gb = gather_plan(sb, n, bounds, owned, rank, world, group=group, dst=dst)
    return ga, gb


def pca_distributed(blocks, n, eigen_cnt=32, bayesian=False, device_index=0, max_block_snps=16384,
                    group=None, tol=1e-9, panels_per_rank=1):
    """snpgdsPCA(algorithm="exact") across ranks: covariance panels stay distributed, the top
    `eigen_cnt` eigenpairs come from the block-Krylov solver (snprelate_amd/eigen.py).
    Every rank returns dict(eigenval, eigenvect [n, k], varprop, TraceXTX)."""
    import torch
    from .eigen import PanelOperator, topk_eigen
    rank, world = _env(group)
    dev = torch.device("cuda", device_index)
    accs, _, _ = _panel_ctxs(_lib.PCA_COV, n, rank, world, device_index, max_block_snps, panels_per_rank,
                             bayesian=bayesian)
    _stream(accs, blocks)
    op = PanelOperator([a for a in accs if a is not None], n, dev, group=group)
    w, v, info = topk_eigen(op, eigen_cnt, tol=tol)
    for acc in accs:
        if acc is not None:
            acc.close()
    return dict(eigenval=w, eigenvect=v, varprop=w / (n - 1), TraceXTX=op.trace_xtx, info=info)
