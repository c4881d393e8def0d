"""The one error norm of the floating-point results (GRM / PCA covariance), and two diagnostics.

CONTRACT:   |got - ref| <= 1e-5 * |ref| + 1e-5 * median(diag(ref))
  i.e.  contract = max |got - ref| / (|ref| + median(diag))  must be <= 1e-5.
  north_star says "matrix entries matching reference to 1e-5 relative"; entries of a relatedness matrix are sums of
  ~L terms of either sign, and an off-diagonal entry can be arbitrarily close to zero by cancellation, so a
  per-entry relative error is bounded only with an absolute floor.  The floor is tied to the matrix's own scale,
  the median diagonal entry (~1 for a GCTA GRM).

DIAGNOSTICS (reported by the full-size tests, asserted only where a test says so):
  offdiag = max |got - ref| / (|ref| + median |ref|)   floor = the typical OFF-diagonal magnitude (~1/sqrt(L) of the
            diagonal): ~1000x tighter than the contract at L = 1e6.  The kernels are engineered to this figure.
  strict  = per-entry |got - ref| / |ref| with no floor: its maximum, and the fraction of entries above 1e-5.  It
            cannot be held by any kernel that accumulates in fp32 between fp64 promotions (the north_star fp32-MFMA
            tile included): the absolute error of an entry is ~6e-8 * sqrt(K L) / 5.6 sum units (K = SNPs per fp32
            run), so entries below ~0.1 of the off-diagonal scale exceed 1e-5 relative.
"""
import numpy as np

TOL = 1e-5


def error_figures(got, ref, diag_scale):
    got = np.asarray(got, np.float64).ravel()
    ref = np.asarray(ref, np.float64).ravel()
    fin = np.isfinite(ref)
    d = np.abs(got[fin] - ref[fin])
    a = np.abs(ref[fin])
    med = float(np.median(a)) if a.size else 1.0
    with np.errstate(divide="ignore", invalid="ignore"):
        strict = np.where(a > 0, d / a, np.where(d > 0, np.inf, 0.0))
    return {
        "contract": float(np.max(d / (a + diag_scale))) if d.size else 0.0,
        "offdiag": float(np.max(d / (a + med))) if d.size else 0.0,
        "strict_max": float(np.max(strict)) if d.size else 0.0,
        "strict_frac_above_tol": float(np.mean(strict > TOL)) if d.size else 0.0,
        "max_abs": float(np.max(d)) if d.size else 0.0,
        "median_abs_ref": med, "diag_scale": float(diag_scale), "entries": int(d.size),
    }


def tri_diag_scale(ref_packed, n):
    """median diagonal entry of a packed upper triangle (row-major with diagonal)."""
    i = np.arange(n, dtype=np.int64)
    d = np.asarray(ref_packed)[i * n - i * (i - 1) // 2]
    d = d[np.isfinite(d)]
    return float(np.median(np.abs(d))) if d.size else 1.0


def contract_err(got, ref, diag_scale):
    return error_figures(got, ref, diag_scale)["contract"]
