#!/usr/bin/env python3
"""Numerical model (numpy, CPU, exact fp64 arithmetic with fp16-rounded operands) of the ONE-product route for GRM / PCA blocks WITH
missing calls that VERDICT r04 #1 asked to build and DESIGN.md had costed: the single-product kernel's integer-centred operands with
MEAN IMPUTATION of the missing cells (row value fp16((avg - c_a) u), column value fp16((avg - c_b) v)) + the imputation residual as
low-precision products.  It answers, before any kernel is written, what every stage leaves in the off-diagonal error figure
(max / rms of |x - ref| / (|ref| + median |ref|), the figure tools/panel_error_distribution.py reports, independent of L):

  stage (a)  imputation alone                                                         rms 3.2e-5, max 1.3e-4   (budget: ~0.3e-6 rms)
  residual corrected EXACTLY (fp64)                                                   back to the weight error alone: the algebra holds
  residual as fp6 (e2m3) x integer products, scale per 32 SNPs or per block           rms 3.1e-6  -- of which
  ... the same with the coefficients kept in fp64 (no quantisation at all)            rms 2.9e-6: the term  kappa (n - avg) mu_i mu_j  that
      appears because a 4-bit genotype operand cannot carry the mean as its missing-cell value (n = round(avg)); it does not shrink
      with L (both-missing cells ~ f^2 L, signal ~ sqrt(L): the ratio is f) and needs a THIRD weighted product;
      the fp6 quantisation of the coefficients is the other 1.0e-6 (3 % of the residual: three mantissa bits)
  coefficient split multiplicatively over two 4-bit significands (fp6 x fp6)          2 % of the residual = 6.6e-7 rms, 2.8e-6 max -- and
      both operands are then 6-bit streams from memory (0.75 B per SNP and sample each): ~47 B/clk/CU from L2, 84 % of its measured peak

i.e. the residual must be held to ~1 % of itself (3.2e-5 -> 3e-7), which neither a single fp6 operand (three mantissa bits: ~3 %) next
to a 4-bit genotype operand nor the product of two fp6 operands (~2 %) can do, and the two-sided form is bound by L2 bandwidth
instead of the matrix pipe on top of it.  DESIGN.md 8 (round 5)
records the decision: two fp16 products per SNP stay for blocks with missing calls.

    python tools/sim_imputation_residual.py            (about two minutes on 8 cores)
"""
import numpy as np

rng = np.random.default_rng(1)
N, L, f = 1500, 30016, 0.02
p = rng.uniform(0.05, 0.95, L)
g = (rng.random((L, N)) < p[:, None]).astype(np.int8) + (rng.random((L, N)) < p[:, None]).astype(np.int8)
mu = rng.random((L, N)) < f
m = ~mu
num = m.sum(1)
s = (g * m).sum(1)
avg = s / num
pp = avg / 2
t = 1 / (pp * (1 - pp))
z = (g - avg[:, None]) * m
Cref = (z * t[:, None]).T @ z


def f16(x):
    return np.float16(x).astype(np.float64)


# the weight as a product of two fp16 numbers, six targets t / f_q (kernels_prep.hip: uv_factor_kernel)
best_err = np.full(L, 1e9)
U = np.zeros(L)
V = np.zeros(L)
F = np.ones(L)
for q in range(6):
    fq = 1 - q / 4096
    tt = t / fq
    e = np.floor(np.log2(np.sqrt(tt)))
    for mnt in range(1024):
        u = (1 + mnt / 1024) * 2.0 ** e
        v = f16(tt / u)
        err = np.abs(u * v - tt) / tt
        sel = err < best_err
        best_err[sel] = err[sel]; U[sel] = u[sel]; V[sel] = v[sel]; F[sel] = fq
print("weight error rms", np.sqrt(np.mean(best_err ** 2)))
ca = np.rint(avg)
cb = ca.copy()
da = avg - ca
db = avg - cb
ra = f16(da * U)                      # the imputed operands of a missing cell: the one real-valued entry of the tables
rb = f16(db * V)
A = np.where(mu, ra[:, None], (g - ca[:, None]) * U[:, None])
B = np.where(mu, rb[:, None], (g - cb[:, None]) * V[:, None])
P = (A * F[:, None]).T @ B
w = U * V * F
rho_a = ra / U
rho_b = rb / V
gim_a = np.where(mu, (ca + rho_a)[:, None], g)
gim_b = np.where(mu, (cb + rho_b)[:, None], g)
# product = w (g~_i - c_a)(g^_j - c_b) with (g~_i - c_a) = z_i + d_a + mu_i delta_a:   w z_i z_j = P - row - column + constant - residual
Rr = ((w * db)[:, None] * (gim_a - ca[:, None])).sum(0)
Qr = ((w * da)[:, None] * (gim_b - cb[:, None])).sum(0)
Kr = (w * da * db).sum()
C1 = P - Rr[:, None] - Qr[None, :] + Kr
wt_err = ((w - t)[:, None] * z).T @ z
off = ~np.eye(N, dtype=bool)
scale = np.median(np.abs(Cref[off]))


def fig(D):
    return float(np.max(np.abs(D[off]) / (np.abs(Cref[off]) + scale))), float(np.sqrt(np.mean((D[off] / scale) ** 2)))


print("median |off-diagonal entry|", scale, " 2 sqrt(L)", 2 * np.sqrt(L))
print("(a) imputation, residual ignored        max, rms:", fig(C1 - Cref))
print("    weight error alone                           :", fig(wt_err))
delta_a = rho_a - da
delta_b = rho_b - db
E = ((w * delta_b)[:, None] * z).T @ mu.astype(float)
E = E + ((w * delta_a)[:, None] * mu).astype(float).T @ z
E2 = ((w * delta_a * delta_b)[:, None] * mu).T @ mu.astype(float)
print("    residual corrected exactly                   :", fig(C1 - E - E2 - Cref), "(= the weight error: the algebra holds)")
# the implementable form: genotype operand = g (called) / n = round(avg) (a 2-bit code in place of the mean), coefficient side fp6
n = np.rint(avg)
val = np.where(mu, n[:, None], g).astype(float)
kb = w * delta_b
ka = w * delta_a


def q_fp6(x, group=None):
    """e2m3 with one power-of-two scale per group of 32 SNPs (or per block)"""
    x = np.asarray(x, float)
    if group is None:
        sc = np.full_like(x, 2.0 ** np.ceil(np.log2(np.max(np.abs(x)) / 7.5)))
    else:
        xx = np.abs(x).reshape(-1, group).max(1)
        sc = np.repeat(2.0 ** np.ceil(np.log2(np.maximum(xx, 1e-300) / 7.5)), group)
    y = np.abs(x) / sc
    step = np.where(y < 2, 0.125, np.where(y < 4, 0.25, 0.5))
    return np.sign(x) * np.minimum(np.round(y / step) * step, 7.5) * sc


def residual(kbq, kaq):
    # z_i = val_i - avg - (n - avg) mu_i   ->   kb mu_j z_i = kb mu_j val_i - kb avg mu_j - kb (n - avg) mu_i mu_j
    return val.T @ (kbq[:, None] * mu) + (kaq[:, None] * mu).T @ val - ((kbq * avg)[:, None] * mu).sum(0)[None, :] - \
        ((kaq * avg)[:, None] * mu).sum(0)[:, None]


for grp in (None, 32):
    print("(b) fp6 coefficients, scale per %-5s            :" % ("block" if grp is None else "32"), fig(C1 - residual(q_fp6(kb, grp), q_fp6(ka, grp)) - Cref))
print("    exact coefficients, 2-bit genotype operand   :", fig(C1 - residual(kb, ka) - Cref), "(the mu_i mu_j leftover)")
left = (((kb + ka) * (n - avg))[:, None] * mu).T @ mu.astype(float)
print("    the leftover  kappa (n - avg) mu_i mu_j  alone:", fig(left), " both-missing SNPs per pair", float((mu.T.astype(float) @ mu)[off].mean()))
sig = 1 + np.arange(8) / 8
prods = np.unique(np.outer(sig, sig).ravel())
prods = np.concatenate([prods / 4, prods / 2, prods, prods * 2, prods * 4])


def q_mul(x, group=32):
    """coefficient = (4-bit significand) x (4-bit significand): fp6 on BOTH operands"""
    x = np.asarray(x, float)
    xx = np.abs(x).reshape(-1, group).max(1)
    sc = np.repeat(2.0 ** np.ceil(np.log2(np.maximum(xx, 1e-300) / 7.0)), group)
    y = np.abs(x) / sc
    idx = np.abs(y[:, None] - prods[None, :]).argmin(1)
    return np.sign(x) * np.where(y < prods[0] / 2, 0.0, prods[idx]) * sc


err_q = ((q_mul(kb) - kb)[:, None] * z).T @ mu.astype(float) + ((q_mul(ka) - ka)[:, None] * mu).astype(float).T @ z
print("    fp6 x fp6 coefficient quantisation alone     :", fig(err_q), "(with an exact mean in the genotype operand)")
print("rms of the coefficients kappa", float(np.sqrt(np.mean(kb ** 2))), " rms |avg - centre|", float(np.sqrt(np.mean(da ** 2))))
