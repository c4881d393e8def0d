"""Multi-GPU sharding of the N x N output triangle.

The path shards over OUTPUTS: every (i, j) pair is independent given all SNPs,
so the packed upper triangle is cut into contiguous row blocks ("panels") of
(nearly) equal area, one per rank/GPU.  Every rank consumes the same genotype
block stream and accumulates only its panel; there is no collective on the data
path.  The only exchange is the final gather of the finished slabs (RCCL over
xGMI with backend "nccl", gloo on CPU), because a row block of the packed
triangle is one contiguous slab.

The reference's own partitioner (Array_SplitJobs, src/dGenGWAS.cpp:2202-2216)
splits the same packed index range into equal-count contiguous ranges across
pthreads; `panel_rows` is its row-aligned analogue across devices.
"""
import numpy as np

ALIGN = 256  # snpgpu_opts.row_begin must be a multiple of this (include/snpgpu.h)


def tri_offset(n, i):
    """Packed-triangle index of (i, i)."""
    return i * n - i * (i - 1) // 2


def panel_rows(n, world, align=ALIGN):
    """Row boundaries [b_0=0, ..., b_world=n] of `world` panels with equal pair counts,
    interior boundaries rounded to multiples of `align` (empty panels possible when
    n is small)."""
    total = n * (n + 1) // 2
    bounds = [0]
    for r in range(1, world):
        target = total * r / world
        # rows [0, b) hold b*n - b(b-1)/2 pairs: solve for b
        b = (2 * n + 1 - np.sqrt((2 * n + 1) ** 2 - 8 * target)) / 2
        b = int(round(b / align)) * align
        b = min(max(b, bounds[-1]), n // align * align)
        bounds.append(b)
    bounds.append(n)
    return bounds


def slab_range(n, row_begin, row_end):
    return tri_offset(n, row_begin), tri_offset(n, row_end)


def gather_slabs(slab, n, bounds, rank, world, group=None, dst=0):
    """Gather the per-rank packed slabs (torch tensors) on `dst` into the full packed
    triangle.  Slabs are padded to the largest slab so one fixed-size gather is used
    (equal-area panels => padding is small)."""
    import torch
    import torch.distributed as dist
    sizes = [slab_range(n, bounds[r], bounds[r + 1]) for r in range(world)]
    lens = [b - a for a, b in sizes]
    m = max(lens)
    send = torch.zeros(m, dtype=slab.dtype, device=slab.device)
    send[: lens[rank]] = slab
    recv = [torch.empty_like(send) for _ in range(world)] if rank == dst else None
    dist.gather(send, recv, dst=dst, group=group)
    if rank != dst:
        return None
    out = torch.empty(n * (n + 1) // 2, dtype=slab.dtype, device=slab.device)
    for r in range(world):
        out[sizes[r][0]: sizes[r][1]] = recv[r][: lens[r]]
    return out
