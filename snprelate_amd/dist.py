"""Multi-GPU sharding of the N x N output triangle.

The path shards over OUTPUTS: every (i, j) pair is independent given all SNPs,
so the packed upper triangle is cut into contiguous row blocks ("panels") of
(nearly) equal TIME (area + the pre-pass over the panel's columns), one per rank/GPU.  Every rank consumes the same genotype
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


PLAN_ALPHA = 512.0   # pair-equivalents of one transposed column (see panel_rows)


def plan_alpha():
    """Cost of the pre-pass per column of a panel, in pairs (SNPGPU_PLAN_ALPHA overrides; 0 = equal-area panels)."""
    import os
    try:
        v = float(os.environ.get("SNPGPU_PLAN_ALPHA", PLAN_ALPHA))
        return v if v >= 0 else PLAN_ALPHA
    except ValueError:
        return PLAN_ALPHA


def panel_rows(n, world, align=ALIGN, alpha=None):
    """Row boundaries [b_0=0, ..., b_world=n] of `world` panels of equal TIME, interior boundaries rounded to multiples of
    `align` (empty panels possible when n is small).

    A panel [r0, r1) costs   pairs(r0, r1) + alpha * (n - r0):   its pair kernel works on its pairs, its pre-pass transposes the
    columns r0 .. n of every block, so the first panel (all n columns) pays the most for it.  Measured on MI355X at N = 100 000,
    8 equal-AREA panels, GRM: 42.5 ms per 32 768-SNP block for panel 0 against 41.2 ms for panel 7 (tools/northstar_share.py): the
    pre-pass costs ~2.4e-5 ms per column, a pair ~4.6e-8 ms -> alpha ~ 500 pairs per column (IBS / KING: ~360).  With alpha = 512
    panel 0 gives ~7 % of its area to the others at N = 100 000 (1.6 % at 500 000) and every rank finishes a block at the same time.
    The boundaries solve   cost(panel) = T for all panels   by bisection on T (100 steps on doubles -- snpgpu_multi's plan_rows in
    multi.hip repeats the same arithmetic, tests/test_gpu_multi_device.py compares the two)."""
    if alpha is None:
        alpha = plan_alpha()
    nf = float(n)
    # the model is calibrated at N >= 1e5; where the pre-pass term would exceed a quarter of a panel's pairs (small n, many panels)
    # it is capped, so that no panel of the plan comes out empty for that reason
    alpha = min(float(alpha), nf / (4.0 * world))

    def tri(b):
        return b * nf - b * (b - 1.0) / 2.0

    def ends(T):
        b = [0.0]
        for _ in range(world):
            r0 = b[-1]
            budget = T - alpha * (nf - r0)
            x = r0
            if budget > 0.0:
                disc = (2.0 * nf + 1.0) * (2.0 * nf + 1.0) - 8.0 * (tri(r0) + budget)
                x = nf if disc <= 0.0 else ((2.0 * nf + 1.0) - np.sqrt(disc)) / 2.0
                x = min(max(x, r0), nf)
            b.append(x)
        return b

    lo, hi = 0.0, tri(nf) + alpha * nf
    for _ in range(100):
        mid = 0.5 * (lo + hi)
        if ends(mid)[-1] >= nf:
            hi = mid
        else:
            lo = mid
    cont = ends(hi)
    bounds = [0]
    for r in range(1, world):
        b = int(round(cont[r] / align)) * align
        b = min(max(b, bounds[-1]), n // align * align)
        bounds.append(b)
    bounds.append(n)
    return bounds


def panel_cost(n, row_begin, row_end, alpha=None):
    """The time model of panel_rows: pairs + alpha * columns (0 for an empty panel)."""
    if row_end <= row_begin:
        return 0.0
    if alpha is None:
        alpha = plan_alpha()
    return float(tri_offset(n, row_end) - tri_offset(n, row_begin)) + float(alpha) * (n - row_begin)


def panel_storage(n, row_begin, row_end, align=ALIGN):
    """Elements of the rectangular accumulator of panel [row_begin, row_end) x [row_begin, n)
    (padded as in snpgpu_create)."""
    up = lambda x: (x + align - 1) // align * align
    return up(row_end - row_begin) * up(n - row_begin)


def panel_plan(n, world, panels_per_rank=1, align=ALIGN):
    """Cut the triangle into P = world * panels_per_rank equal-area panels and deal them to the ranks:
    returns (bounds, owned) with owned[rank] = sorted panel indices (possibly of different lengths).

    A panel's accumulator is a rectangle rows x (n - row_begin): the first panel is thin and wide
    (storage ~ its pair count), the LAST one a square holding a triangle (storage ~ twice its pair
    count).  With one panel per rank the last rank therefore needs about twice the memory of the others
    (N = 500 000 on 8 GPUs: 120 ... 136 GiB of fp64, but 233 GiB on the last GPU).  With several panels
    per rank, each given largest-first to the least loaded rank, the square panel shrinks and the worst
    rank approaches the mean (N^2/2)(1 + 1/(2P)) / world elements: 176 GiB at 2, 146 GiB at 4, 132 GiB at 8
    panels per rank (mean 122 GiB)."""
    P = world * panels_per_rank
    bounds = panel_rows(n, P, align)
    if panels_per_rank == 1:
        return bounds, [[r] for r in range(world)]
    size = [panel_storage(n, bounds[p], bounds[p + 1], align) if bounds[p + 1] > bounds[p] else 0 for p in range(P)]
    owned = [[] for _ in range(world)]
    load = [0] * world
    for p in sorted(range(P), key=lambda q: (-size[q], q)):
        r = min(range(world), key=lambda x: (load[x], len(owned[x]), x))
        owned[r].append(p)
        load[r] += size[p]
    return bounds, [sorted(o) for o in owned]


def slab_range(n, row_begin, row_end):
    return tri_offset(n, row_begin), tri_offset(n, row_end)


def _gather_ranges(pieces, ranges_of, n, rank, world, group, dst):
    """The final exchange of the row-block partition (north_star: "a final RCCL gather over xGMI"): every slab is a contiguous
    range of the packed triangle, so the destination receives each one STRAIGHT into its place -- point-to-point sends of the
    exact sizes (torch.distributed.batch_isend_irecv: one RCCL group on the "nccl" backend), no padding to the largest slab and
    no staging copy of the triangle on the destination (the fixed-size `dist.gather` this replaces held world x max-slab
    there next to the output).  pieces: this rank's slabs; ranges_of[r]: the (begin, end) ranges of rank r's slabs."""
    import torch
    import torch.distributed as dist
    out = None
    ops = []
    if rank == dst:
        out = torch.empty(n * (n + 1) // 2, dtype=pieces[0].dtype, device=pieces[0].device)
        for (a, b), piece in zip(ranges_of[rank], pieces):
            out[a:b] = piece
        for r in range(world):
            if r != dst:
                for a, b in ranges_of[r]:
                    if b > a:
                        ops.append(dist.P2POp(dist.irecv, out[a:b], dist.get_global_rank(group, r) if group is not None else r, group))
    else:
        peer = dist.get_global_rank(group, dst) if group is not None else dst
        for (a, b), piece in zip(ranges_of[rank], pieces):
            if b > a:
                ops.append(dist.P2POp(dist.isend, piece.contiguous(), peer, group))
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()
    return out


def gather_plan(slabs, n, bounds, owned, rank, world, group=None, dst=0):
    """gather_slabs for a panel_plan: `slabs` = this rank's packed slabs in the order of owned[rank]."""
    ranges_of = [[slab_range(n, bounds[p], bounds[p + 1]) for p in owned[r]] for r in range(world)]
    return _gather_ranges(list(slabs), ranges_of, n, rank, world, group, dst)


def gather_slabs(slab, n, bounds, rank, world, group=None, dst=0):
    """Gather the per-rank packed slabs (torch tensors) on `dst` into the full packed triangle."""
    ranges_of = [[slab_range(n, bounds[r], bounds[r + 1])] for r in range(world)]
    a, b = ranges_of[rank][0]
    return _gather_ranges([slab[: b - a]], ranges_of, n, rank, world, group, dst)


def pass_plan(n, world, panels_per_rank=1, passes=1, bytes_per_element=8.0, align=ALIGN):
    """Several PASSES over the SNP stream, each with its own set of resident panels (output-stationary): for
    accumulators that do not fit the node at once -- KING-robust holds five uint32 counters per pair
    (TS_KINGRobust, src/genKING.cpp:274-281: 20 B, 2.5 TB at N = 500 000) against 8 x 288 GB.

    The triangle is cut into P = world * panels_per_rank * passes equal-area panels; every (pass, rank) slot takes
    panels_per_rank of them, largest storage first into the least loaded slot.  Returns (bounds, owned) with
    owned[pass][rank] = sorted panel indices, and the largest slot's accumulator bytes."""
    P = world * panels_per_rank * passes
    bounds = panel_rows(n, P, align)
    size = [panel_storage(n, bounds[p], bounds[p + 1], align) if bounds[p + 1] > bounds[p] else 0 for p in range(P)]
    slots = [(q, r) for q in range(passes) for r in range(world)]
    owned = {s: [] for s in slots}
    load = {s: 0 for s in slots}
    for p in sorted(range(P), key=lambda k: (-size[k], k)):
        s = min((x for x in slots if len(owned[x]) < panels_per_rank), key=lambda x: (load[x], len(owned[x]), x))
        owned[s].append(p)
        load[s] += size[p]
    out = [[sorted(owned[(q, r)]) for r in range(world)] for q in range(passes)]
    return bounds, out, max(load.values()) * bytes_per_element


def passes_needed(n, world, bytes_per_element, budget_bytes, panels_per_rank=1, max_passes=64, align=ALIGN):
    """Smallest number of passes whose largest (pass, rank) slot fits `budget_bytes` of accumulators."""
    for q in range(1, max_passes + 1):
        if pass_plan(n, world, panels_per_rank, q, bytes_per_element, align)[2] <= budget_bytes:
            return q
    raise ValueError("accumulators do not fit: raise panels_per_rank or the memory budget")


def snp_share(n_snp, rank, world):
    """Rank `rank`'s share [lo, hi) of a block's SNP rows for the shared per-SNP statistics (contiguous, sizes differing by at
    most one; Array_SplitJobs, src/dGenGWAS.cpp:2202-2216, over SNPs instead of pairs)."""
    base, extra = divmod(int(n_snp), int(world))
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def stats_ranks(bounds, owned, world):
    """The ranks that own at least one non-empty panel: the only ones that have a context to scan a share of a block with.  A rank
    outside this list (n small against world x panels: 256-row boundaries leave panels empty) still JOINS every all-gather of the
    block statistics -- a collective is a collective -- with an empty share (such a rank used to skip the call and the
    others hung)."""
    return [r for r in range(world) if any(bounds[p + 1] > bounds[p] for p in owned[r])]


def allgather_block_stats(sum_t, num_t, n_snp, rank, world, group=None, active=None, active_index=None):
    """sum_t / num_t: int32 tensors [n_snp] of which this rank filled its share -- snp_share(n_snp, i, len(active)) with i = this
    rank's position in `active` (default: every rank) --; on return every rank holds the whole arrays (8 bytes per SNP over the
    wire: ONE torch.distributed all_gather_into_tensor of the padded shares -- "nccl" = RCCL on the GPUs, gloo in the CPU tests --
    enqueued on the current torch stream, no host synchronisation on the nccl backend; active_index: `active` as a device tensor built
    once by the caller).  In place; returns (sum_t, num_t)."""
    import torch
    import torch.distributed as dist
    if world == 1:
        return sum_t, num_t
    active = list(range(world)) if active is None else list(active)
    na = len(active)
    width = -(-int(n_snp) // na)
    mine = torch.zeros((2, width), dtype=torch.int32, device=sum_t.device)
    if rank in active:
        lo, hi = snp_share(n_snp, active.index(rank), na)
        mine[0, : hi - lo] = sum_t[lo:hi]
        mine[1, : hi - lo] = num_t[lo:hi]
    parts = torch.empty((world, 2, width), dtype=torch.int32, device=sum_t.device)
    dist.all_gather_into_tensor(parts.view(-1), mine.view(-1), group=group)
    # shares are contiguous and differ by at most one SNP: the first `extra` active ranks hold `width`, the others width - 1
    base, extra = divmod(int(n_snp), na)
    if na == world:
        sel = parts                                           # every rank scans: the parts are in share order already
    else:
        # (a host list turned into a device tensor is a synchronous copy -- it would wait for everything queued on the stream: callers
        # that run per block hand in the index tensor they built once, SharedStats does)
        idx = active_index if active_index is not None else torch.tensor(active, dtype=torch.long, device=sum_t.device)
        sel = parts.index_select(0, idx)                      # [na][2][width] in share order
    for k, dst in ((0, sum_t), (1, num_t)):
        if extra:
            dst[: extra * (base + 1)] = sel[:extra, k, : base + 1].reshape(-1)
        if base:
            dst[extra * (base + 1): n_snp] = sel[extra:, k, :base].reshape(-1)
    return sum_t, num_t
