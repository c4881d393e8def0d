"""Top-k eigenpairs of the (trace-normalised) PCA covariance -- or of a GCTA GRM / an EIGMIX coancestry matrix finalised
in place -- held as fp64 row panels on one or several GPUs: a thin caller of the C ABI.

The solver itself (thick-restarted block Krylov + Rayleigh-Ritz, replacing the reference's dense LAPACK call CalcEigen ->
dspevx, src/genPCA.cpp:1262-1346) is C++ / HIP behind `snpgpu_panels_topk_eigen` (snprelate_amd/csrc/eigen.hip), so the R
`.Call` shim reaches it as well.  What stays here is the one-process-per-GPU plumbing: every product Y = C Q is formed by the
library in a buffer this module owns, and the library calls back to have it summed over the ranks -- one
`torch.distributed.all_reduce` (backend "nccl" = RCCL over xGMI; gloo in the CPU tests of the plan logic).
"""
import ctypes

import numpy as np

from . import _lib


def _world(group):
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        return dist.get_world_size(group)
    return 1


class PanelOperator:
    """The symmetric matrix `scale * C` distributed as row panels.

    `panels`: list of snprelate_amd._lib.Accumulator living on this rank's device (PCA_COV, or GRM_GCTA / EIGMIX after
    finalize_inplace).  Across ranks the panels must tile [0, N).  normalize=True: scale = (n - 1) / trace (gnrPCA,
    src/genPCA.cpp:1386-1390; PCA_COV panels only), else scale = 1."""

    def __init__(self, panels, n, device, group=None, normalize=True):
        import torch
        self.panels, self.n, self.device, self.group = list(panels), int(n), device, group
        self.trace_xtx = None
        self.scale = 1.0
        if normalize:
            tr = torch.zeros(1, dtype=torch.float64, device=device)
            for p in self.panels:
                tr += p.pca_panel_trace()
            if _world(group) > 1:
                import torch.distributed as dist
                dist.all_reduce(tr, group=group)
            self.trace_xtx = float(tr.item())
            self.scale = (self.n - 1) / self.trace_xtx

    def matmul(self, q):
        """q: (b, n) contiguous float64 on the device -> scale * C q, (b, n) (building block; measurement tools)."""
        import torch
        assert q.is_contiguous() and q.dtype == torch.float64 and q.shape[1] == self.n
        y = torch.zeros_like(q)
        torch.cuda.synchronize(self.device)
        for p in self.panels:
            p.pca_panel_matmul(self.scale, q.data_ptr(), q.shape[0], y.data_ptr())
        if _world(self.group) > 1:
            import torch.distributed as dist
            dist.all_reduce(y, group=self.group)
        return y


def topk_eigen(op, k, block=None, depth=0, tol=1e-9, max_restarts=60, seed=20240601, fp32_until=0.0):
    """Largest-k eigenpairs of the operator through snpgpu_panels_topk_eigen.  Returns (eigenvalues [k] descending,
    eigenvectors [n, k], info dict) as torch tensors on the operator's device."""
    import torch
    n = op.n
    k = int(min(k, n))
    world = _world(op.group)
    b = int(block or min(n, (k + 8 + 15) // 16 * 16))
    b = min(max(b, k), n)
    keep = []
    opts = _lib.EigOpts(tol=float(tol), block=b, depth=int(depth), max_restarts=int(max_restarts), seed=int(seed),
                        y_buf=None, reduce=_lib.REDUCE_FN(), user=None, fp32_until=float(fp32_until))
    if world > 1:
        import torch.distributed as dist
        y = torch.zeros((b, n), dtype=torch.float64, device=op.device)

        def _reduce(_user):
            try:
                dist.all_reduce(y, group=op.group)
                torch.cuda.synchronize(op.device)
                return 0
            except Exception:  # pragma: no cover - reported through the library's error path
                return 1
        cb = _lib.REDUCE_FN(_reduce)
        keep += [y, cb]
        opts.y_buf = y.data_ptr()
        opts.reduce = cb
    w = np.empty(k, np.float64)
    v = torch.empty((k, n), dtype=torch.float64, device=op.device)        # column-major n x k
    handles = (ctypes.c_void_p * len(op.panels))(*[p._h for p in op.panels])
    info = _lib.EigInfo()
    torch.cuda.synchronize(op.device)
    _lib.check(_lib.lib().snpgpu_panels_topk_eigen(handles, len(op.panels), float(op.scale), k, ctypes.byref(opts),
                                                   _lib._ptr(w), ctypes.c_void_p(v.data_ptr()), _lib.DEVICE,
                                                   ctypes.byref(info)))
    return torch.from_numpy(w).to(op.device), v.T.contiguous(), {
        "restarts": info.restarts, "matmuls": info.matmuls, "max_rel_residual": info.max_rel_residual,
        "block": info.block, "depth": info.depth, "matmuls_fp32": info.matmuls_fp32}
