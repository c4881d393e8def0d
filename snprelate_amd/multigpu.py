"""Multi-GPU drivers: one process per GPU (torch.distributed, backend "nccl" = RCCL over xGMI),
the output triangle cut into equal-area row panels, no collective on the data path, one gather
(or, for PCA, the eigen solver's all-reduces) at the end.  Launch with
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 script.py

`blocks` is an iterable of genotype blocks (uint8 [b][n_samp] or 2-bit packed [b][ceil(n/4)]), or -- needed
whenever the stream is walked more than once (KING with several passes) -- a callable returning a fresh
iterable; every rank iterates the same stream (in an R deployment: the kept GDS reader opened per rank).
A block may also be a (device_pointer, n_snp) pair of 2-bit rows already resident on this rank's GPU.

Where results go (`gather` / `sink`):
  gather=True (default, N up to ~100 000)  rank `dst` receives the whole packed triangle (every slab sent straight into its range: RCCL point-to-point)
  gather=False                             every rank keeps its own slabs: {panel: tensor}
  sink=SlabSink                            every finished slab is handed to the sink and released -- at N = 500 000 the
                                           triangle is 1 TB of doubles and no rank may hold it; the reference's answer at
                                           that size is the same: rows appended to a file (grm_save_to_gds,
                                           src/genPCA.cpp:1571-1584).
"""
import json
import os

import numpy as np

from . import _lib
from .dist import allgather_block_stats, gather_plan, panel_plan, pass_plan, passes_needed, slab_range, snp_share


def _env(group=None):
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(group), dist.get_world_size(group)
    return 0, 1


class SlabSink:
    """Receives finished slabs: put(name, panel, row_begin, row_end, tensor) with `tensor` the packed rows
    [row_begin, row_end) of the triangle (device, float64).  The tensor is released by the caller afterwards."""

    def put(self, name, panel, row_begin, row_end, tensor):
        raise NotImplementedError

    def close(self):
        return None


class FileSlabSink(SlabSink):
    """Raw little-endian float64 files `<dir>/<name>.rows<r0>-<r1>.f64` plus one index per rank
    (`index.rank<k>.json`): concatenated in row order they are the packed triangle (CdMatTri order)."""

    def __init__(self, directory, rank=0, chunk_elems=1 << 26):
        self.dir, self.rank, self.chunk = directory, rank, int(chunk_elems)
        os.makedirs(directory, exist_ok=True)
        self.entries = []

    def put(self, name, panel, row_begin, row_end, tensor):
        fn = "%s.rows%d-%d.f64" % (name, row_begin, row_end)
        with open(os.path.join(self.dir, fn), "wb") as f:
            for lo in range(0, tensor.numel(), self.chunk):          # bounded host staging
                tensor[lo: lo + self.chunk].cpu().numpy().tofile(f)
        self.entries.append(dict(name=name, panel=int(panel), row_begin=int(row_begin), row_end=int(row_end),
                                 elements=int(tensor.numel()), file=fn))

    def close(self):
        with open(os.path.join(self.dir, "index.rank%d.json" % self.rank), "w") as f:
            json.dump(self.entries, f, indent=1)
        return self.entries


def read_file_slabs(directory, name, n):
    """Reassemble the packed triangle of `name` from a FileSlabSink directory (small N: tests, inspection)."""
    ent = []
    for fn in sorted(os.listdir(directory)):
        if fn.startswith("index.rank") and fn.endswith(".json"):
            ent += [e for e in json.load(open(os.path.join(directory, fn))) if e["name"] == name]
    out = np.full(n * (n + 1) // 2, np.nan)
    for e in ent:
        lo, hi = slab_range(n, e["row_begin"], e["row_end"])
        out[lo:hi] = np.fromfile(os.path.join(directory, e["file"]), dtype="<f8")
    return out


def _blocks_iter(blocks):
    return blocks() if callable(blocks) else blocks


def _make_ctxs(kind, n, bounds, panels, device_index, max_block_snps, **kw):
    accs = []
    for p in panels:
        r0, r1 = bounds[p], bounds[p + 1]
        if r1 <= r0:
            accs.append(None)
            continue
        full = (r0 == 0 and r1 == n)
        accs.append(_lib.Accumulator(kind, n, device=device_index, row_begin=0 if full else r0,
                                     row_end=0 if full else r1, max_block_snps=max_block_snps, **kw))
    return accs


def _panel_ctxs(kind, n, rank, world, device_index, max_block_snps, panels_per_rank=1, **kw):
    """This rank's accumulators (one per owned panel, None for an empty panel) + the plan.
    panels_per_rank > 1 balances accumulator memory across GPUs (dist.panel_plan)."""
    bounds, owned = panel_plan(n, world, panels_per_rank)
    return _make_ctxs(kind, n, bounds, owned[rank], device_index, max_block_snps, **kw), bounds, owned


class SharedStats:
    """Per-SNP statistics of every block computed ONCE per node instead of once per rank (the reference computes them once per block,
    src/genPCA.cpp:84-142): each rank that owns a panel scans its share of the block's SNP rows over all samples
    (snpgpu_block_stats), ONE all-gather makes the arrays whole on every rank (8 bytes per SNP), every context takes the block with
    snpgpu_feed_stats.  Integer statistics: results are bit-identical to the per-rank form.

    Round 6: no host synchronisation.  The contexts are created on `stream` (a torch stream handed to snpgpu_opts.stream), and the
    zero fill, the scan, the all-gather (on "nccl" = RCCL the collective is ordered after the current torch stream and that stream
    waits for it on the device) and the feeds are all enqueued on it in program order.  Ranks whose panels are all empty still join
    the collective with an empty share (dist.stats_ranks)."""

    def __init__(self, rank, world, group, dev, bounds, owned):
        import torch
        from .dist import stats_ranks
        self.rank, self.world, self.group, self.dev = rank, world, group, dev
        self.active = stats_ranks(bounds, owned, world)
        self.stream = torch.cuda.Stream(device=dev)
        self._active_index = torch.tensor(self.active, dtype=torch.long, device=dev) if len(self.active) != world else None   # built ONCE
        self._keep = []

    def block(self, live, ptr, n_snp, fmt, rowb):
        """statistics of one block resident at device address `ptr` -> (sum, num) device tensors, whole on every rank"""
        import torch
        with torch.cuda.stream(self.stream):
            st = torch.zeros((2, n_snp), dtype=torch.int32, device=self.dev)
            if live and self.rank in self.active:
                lo, hi = snp_share(n_snp, self.active.index(self.rank), len(self.active))
                if hi > lo:
                    live[0].block_stats_device(ptr + lo * rowb, hi - lo, st[0, lo:].data_ptr(), st[1, lo:].data_ptr(), fmt)
            allgather_block_stats(st[0], st[1], n_snp, self.rank, self.world, self.group, active=self.active, active_index=self._active_index)
        # the tensor must outlive the kernels that read it: keep the last few blocks' (the caching allocator would otherwise hand
        # the memory to the next torch allocation on this stream, which is ordered behind those kernels anyway -- belt and braces)
        self._keep = self._keep[-3:] + [st]
        return st


def _stream(accs, blocks, shared=None):
    """Feed every block of the stream to every live context.  shared: a SharedStats (the contexts must have been created on
    shared.stream) or None = every context scans the block itself."""
    live = [a for a in accs if a is not None]
    if shared is None:
        if not live:
            return
        for blk in _blocks_iter(blocks):
            for acc in live:
                if isinstance(blk, tuple):
                    acc.feed_device(blk[0], blk[1])
                else:
                    acc.feed(blk)
        return
    import torch
    n = live[0].n if live else None
    keep = []
    for blk in _blocks_iter(blocks):
        if isinstance(blk, tuple):
            ptr, n_snp = blk[0], int(blk[1])
            fmt, rowb = _lib.GENO_PACKED2, ((n + 3) // 4 if n else 0)
        else:
            g = np.ascontiguousarray(blk, dtype=np.uint8)
            n_snp = g.shape[0]
            if live:
                fmt = _lib.GENO_U8 if g.shape[1] == n else _lib.GENO_PACKED2
                with torch.cuda.stream(shared.stream):
                    t = torch.from_numpy(g).to(shared.dev)        # ordered on the contexts' stream; the host copy is synchronous
                keep = keep[-2:] + [t]                            # (pageable memory), the device copy lives until two blocks later
                ptr, rowb = t.data_ptr(), g.shape[1]
            else:
                ptr, fmt, rowb = 0, _lib.GENO_U8, 0
        st = shared.block(live, ptr, n_snp, fmt, rowb)
        for acc in live:
            acc.feed_device_stats(ptr, n_snp, st[0].data_ptr(), st[1].data_ptr(), fmt)
    for acc in live:
        acc.sync()


def _slab(n, bounds, p, dev, dtype):
    import torch
    lo, hi = slab_range(n, bounds[p], bounds[p + 1])
    return torch.empty(hi - lo, dtype=dtype, device=dev)


def _deliver(names, slab_sets, n, bounds, owned, rank, world, group, dst, gather, sink):
    """slab_sets[k][i] = slab of result `names[k]` for panel owned[rank][i] (None once handed to a sink)."""
    if sink is not None:
        return sink.close()
    if not gather:
        return tuple({p: s for p, s in zip(owned[rank], slabs)} for slabs in slab_sets) if len(names) > 1 else \
            {p: s for p, s in zip(owned[rank], slab_sets[0])}
    outs = []
    for slabs in slab_sets:
        outs.append(gather_plan(slabs, n, bounds, owned, rank, world, group=group, dst=dst) if world > 1
                    else _join(slabs, n, bounds, owned))
    return tuple(outs) if len(names) > 1 else outs[0]


SHARED_STATS_MIN_WORLD = 4


def grm_distributed(blocks, n, method="GCTA", device_index=0, max_block_snps=16384, group=None, dst=0,
                    panels_per_rank=1, gather=True, sink=None, shared_stats=None):
    """snpgdsGRM(method = "GCTA" | "Eigenstrat") across the ranks of `group`.
    gather=True: the packed upper triangle (torch float64 tensor on the device) on rank `dst`, else None.
    gather=False: {panel index: slab} of this rank.  sink: every slab goes to sink.put("grm", ...) and is released;
    returns sink.close().  Panels are finalised one at a time, so only one slab is resident next to the accumulators."""
    import torch
    import torch.distributed as dist
    if method not in ("GCTA", "Eigenstrat"):
        raise ValueError("Invalid 'method'!")
    rank, world = _env(group)
    dev = torch.device("cuda", device_index)
    kind = _lib.GRM_GCTA if method == "GCTA" else _lib.PCA_COV
    # shared_stats=None: on when the group has at least SHARED_STATS_MIN_WORLD ranks (the scan is 0.7 of the ~5 ms of pre-pass every
    # rank repeats per 65 536-SNP block at N = 100 000 -- it bounds the 8-GPU efficiency)
    if shared_stats is None:
        shared_stats = world >= SHARED_STATS_MIN_WORLD
    shared = None
    if shared_stats and world > 1:
        bounds_, owned_ = panel_plan(n, world, panels_per_rank)
        shared = SharedStats(rank, world, group, dev, bounds_, owned_)
    accs, bounds, owned = _panel_ctxs(kind, n, rank, world, device_index, max_block_snps, panels_per_rank,
                                      **({"stream": shared.stream.cuda_stream} if shared else {}))
    _stream(accs, blocks, shared)
    tr = None
    if method == "Eigenstrat":
        tr = torch.tensor([sum(a.pca_panel_trace() for a in accs if a is not None)], dtype=torch.float64, device=dev)
        if world > 1:
            dist.all_reduce(tr, group=group)
    slabs = []
    for acc, p in zip(accs, owned[rank]):
        slab = _slab(n, bounds, p, dev, torch.float64)
        if acc is not None:
            if method == "GCTA":
                acc.grm_gcta(packed=True, out_ptr=slab.data_ptr())
            else:
                acc.pca_cov(packed=True, normalize=True, trace_in=float(tr.item()), out_ptr=slab.data_ptr())
            acc.close()                        # the accumulators of this panel are released before the next slab is made
        if sink is not None:
            torch.cuda.synchronize(dev)
            sink.put("grm", p, bounds[p], bounds[p + 1], slab)
            slab = None
        slabs.append(slab)
    torch.cuda.synchronize(dev)
    return _deliver(("grm",), [slabs], n, bounds, owned, rank, world, group, dst, gather, sink)


def _join(slabs, n, bounds, owned):
    """world == 1: concatenate this rank's slabs (it owns every panel, in index order)."""
    import torch
    return slabs[0] if len(slabs) == 1 else torch.cat(slabs)


KING_BYTES_PER_PAIR_SLOT = 5 * 4      # TS_KINGRobust: five uint32 counters per element of the panel rectangle


def king_distributed(blocks, n, family=None, device_index=0, max_block_snps=16384, group=None, dst=0,
                     panels_per_rank=1, passes=None, mem_budget=None, gather=True, sink=None):
    """snpgdsIBDKING(type="KING-robust"): (IBS0, kinship) packed triangles.

    The five uint32 counters per pair (20 B) do not fit the node at N = 500 000 (2.5 TB against 8 x 288 GB): the
    panels are processed in `passes` groups, each group resident for one walk over the whole SNP stream
    (output-stationary; `blocks` must then be a callable).  passes=None picks the smallest number whose largest slot
    fits `mem_budget` bytes of counters (default: 70 % of the device's free memory).  Results as in grm_distributed
    (a sink receives "IBS0" and "kinship" slabs)."""
    import torch
    rank, world = _env(group)
    dev = torch.device("cuda", device_index)
    if passes is None:
        if mem_budget is None:
            free, _ = torch.cuda.mem_get_info(dev)
            mem_budget = 0.7 * free
        # the finished IBS0 / kinship slabs of earlier passes stay on the device next to the counters unless a sink takes
        # them: 2 x 8 B per pair of this rank's share of the triangle come off the budget
        if sink is None:
            mem_budget = mem_budget - 16.0 * (n * (n + 1) / 2) / world
            if mem_budget <= 0:
                raise ValueError("king_distributed: the result slabs alone exceed the memory budget; pass a sink")
        passes = passes_needed(n, world, KING_BYTES_PER_PAIR_SLOT, mem_budget, panels_per_rank)
    # the plan must be the SAME on every rank (bounds, ownership, gather slots): ranks with different free memory would
    # otherwise derive different pass counts -- take the largest
    if world > 1:
        import torch.distributed as dist
        t = torch.tensor([int(passes)], dtype=torch.int64, device=dev if dist.get_backend(group) == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
        passes = int(t.item())
    if passes > 1 and not callable(blocks):
        raise ValueError("king_distributed: %d passes over the SNP stream need `blocks` to be a callable" % passes)
    bounds, owned_q, _ = pass_plan(n, world, panels_per_rank, passes, KING_BYTES_PER_PAIR_SLOT)
    mine, sa, sb = [], [], []
    for q in range(passes):
        accs = _make_ctxs(_lib.KING_ROBUST, n, bounds, owned_q[q][rank], device_index, max_block_snps)
        _stream(accs, blocks)
        for acc, p in zip(accs, owned_q[q][rank]):
            a, b = _slab(n, bounds, p, dev, torch.float64), _slab(n, bounds, p, dev, torch.float64)
            if acc is not None:
                acc.king_robust(family=family, packed=True, out_ptrs=(a.data_ptr(), b.data_ptr()))
                acc.close()
            if sink is not None:
                torch.cuda.synchronize(dev)
                sink.put("IBS0", p, bounds[p], bounds[p + 1], a)
                sink.put("kinship", p, bounds[p], bounds[p + 1], b)
                a = b = None
            mine.append(p); sa.append(a); sb.append(b)
    torch.cuda.synchronize(dev)
    # one flat ownership table over all passes for the gather
    owned = [sorted(p for q in range(passes) for p in owned_q[q][r]) for r in range(world)]
    order = np.argsort(mine)
    sa, sb = [sa[i] for i in order], [sb[i] for i in order]
    return _deliver(("IBS0", "kinship"), [sa, sb], n, bounds, owned, rank, world, group, dst, gather, sink)


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
