"""Accuracy of the GRM SYRK variants on allele-frequency spectra other than the bench's U(0.05, 0.95): array-like
(MAF ~ U(0.01, 0.5)) and rare-variant heavy (MAF = 0.5 u^3, floor 2/N) data without missing calls, against the fp64
oracle.  Prints max |err| / max(|ref|, median |ref|) [strict] and max |err| / (|ref| + median |ref|) [tests' metric].
python tools/err_check_maf.py [L]"""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle as orc
from snprelate_amd import _lib
L = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
n, blk = 1200, 16384
rng = np.random.default_rng(11)
for name, maf in (("uniform 0.05-0.95", None), ("array 0.01-0.5", rng.uniform(0.01, 0.5, L)),
                  ("rare 0.5u^3", np.maximum(0.5 * rng.random(L) ** 3, 2.0 / n))):
    p = rng.uniform(0.05, 0.95, L) if maf is None else np.where(rng.random(L) < 0.5, maf, 1 - maf)
    g = ((rng.random((L, n)) < p[:, None]).astype(np.uint8) + (rng.random((L, n)) < p[:, None]).astype(np.uint8))
    ref = orc.grm_gcta(g)
    med = np.median(np.abs(ref))
    for be in ("f16", "h3", "f32"):
        os.environ["SNPGPU_SYRK"] = be
        with _lib.Accumulator(_lib.GRM_GCTA, n, max_block_snps=blk) as a:
            for i in range(0, L, blk):
                a.feed(g[i:i + blk])
            got = a.grm_gcta(packed=True)
        d = np.abs(got - ref)
        print("%-18s %-3s strict %.3e  tests' metric %.3e  finite %s" % (name, be, float(np.max(d / np.maximum(np.abs(ref), med))),
              float(np.max(d / (np.abs(ref) + med))), bool(np.isfinite(got).all())))
