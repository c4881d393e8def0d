"""Top-k eigenpairs of the (trace-normalised) PCA covariance held as row panels on one or
several GPUs: the scalable replacement for the reference's dense LAPACK call
(CalcEigen -> dspevx('V','I',IL=1..IU=k), src/genPCA.cpp:1262-1346), which is impossible at
N = 500 000.

Method: thick-restarted block Krylov (block Lanczos with full re-orthogonalisation) +
Rayleigh-Ritz.  The only O(N^2) work, Y = C Q, is done by libsnpgpu on each rank's panel
(`snpgpu_pca_panel_matmul`: two rocBLAS dgemms on the fp64 panel accumulator) followed by one
all-reduce of the N x b block over the ranks (RCCL over xGMI; gloo in the CPU tests); the small
dense algebra (QR of N x b blocks, eigh of the projected matrix) runs replicated on every rank
through torch.  Eigenvector signs are arbitrary, as with LAPACK.
"""
import torch


def _allreduce(t, group):
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(t, group=group)
    return t


class PanelOperator:
    """y = scale * C @ q for a symmetric matrix distributed as row panels.

    `panels`: list of snprelate_amd._lib.Accumulator (kind PCA_COV) living on this rank's device
    (normally one).  Across ranks the panels must tile [0, N)."""

    def __init__(self, panels, n, device, group=None, normalize=True):
        self.panels, self.n, self.device, self.group = list(panels), int(n), device, group
        tr = torch.zeros(1, dtype=torch.float64, device=device)
        for p in self.panels:
            tr += p.pca_panel_trace()
        _allreduce(tr, group)
        self.trace_xtx = float(tr.item())
        # (n-1)/trace scaling of gnrPCA, src/genPCA.cpp:1386-1390
        self.scale = (self.n - 1) / self.trace_xtx if normalize else 1.0

    def matmul(self, q):
        """q: (b, n) contiguous float64 on the device (b vectors of length n) -> (b, n)."""
        assert q.is_contiguous() and q.dtype == torch.float64 and q.shape[1] == self.n
        y = torch.zeros_like(q)
        torch.cuda.synchronize(self.device) if q.is_cuda else None
        for p in self.panels:
            p.pca_panel_matmul(self.scale, q.data_ptr(), q.shape[0], y.data_ptr())
        return _allreduce(y, self.group)


def _orth(x):
    """Orthonormalise the rows of x (b, n) -> rows span the same space."""
    q, _ = torch.linalg.qr(x.T, mode="reduced")
    return q.T.contiguous()


def topk_eigen(op, k, block=None, depth=12, tol=1e-9, max_restarts=60, seed=20240601, matmul=None):
    """Largest-k eigenpairs of the operator.  Returns (eigenvalues [k] descending,
    eigenvectors [n, k], info dict).  `matmul` overrides op.matmul (used by CPU tests)."""
    n = op.n
    mm = matmul or op.matmul
    k = int(min(k, n))
    b = int(block or min(n, k + 8))
    depth = int(max(2, min(depth, max(2, n // b))))
    gen = torch.Generator(device="cpu").manual_seed(seed)       # identical start on every rank
    x0 = torch.randn(b, n, generator=gen, dtype=torch.float64).to(op.device)
    q0 = _orth(x0)
    theta = vecs = None
    n_mm = 0
    for restart in range(max_restarts):
        K, W = [q0], []
        for j in range(depth):
            w = mm(K[-1])
            n_mm += 1
            W.append(w)
            if j + 1 == depth:
                break
            basis = torch.cat(K, 0)
            r = w.clone()
            for _ in range(2):                                  # full re-orthogonalisation, twice
                r -= (r @ basis.T) @ basis
            nr = torch.linalg.norm(r, dim=1)
            if float(nr.max()) < 1e-12 * max(1.0, float(torch.linalg.norm(w))):
                break                                           # invariant subspace found
            # a (numerically) rank-deficient remainder makes QR return directions that are not
            # orthogonal to the basis: orthonormalise, project out the basis once more, repeat
            qn = _orth(r)
            qn -= (qn @ basis.T) @ basis
            qn = _orth(qn)
            qn -= (qn @ basis.T) @ basis
            K.append(_orth(qn))
        basis = torch.cat(K[:len(W)], 0)                        # (m, n)
        cw = torch.cat(W, 0)                                    # C * basis
        t = basis @ cw.T
        t = 0.5 * (t + t.T)
        ev, s = torch.linalg.eigh(t)
        idx = torch.argsort(ev, descending=True)[:max(k, min(b, ev.numel()))]
        ev, s = ev[idx], s[:, idx]
        ritz = s.T @ basis                                      # (b', n) Ritz vectors
        cr = s.T @ cw
        res = torch.linalg.norm(cr - ev[:, None] * ritz, dim=1)
        theta, vecs = ev[:k], ritz[:k]
        rel = float((res[:k] / ev[:k].abs().clamp_min(1e-300)).max())
        if rel < tol:
            break
        q0 = _orth(ritz[:b])                                    # thick restart with the best Ritz vectors
        if q0.shape[0] < b:
            extra = torch.randn(b - q0.shape[0], n, generator=gen, dtype=torch.float64).to(op.device)
            q0 = _orth(torch.cat([q0, extra], 0))
    return theta, vecs.T.contiguous(), {"restarts": restart + 1, "matmuls": n_mm, "max_rel_residual": rel}
