"""Accuracy of the exact-row-side SYRK (blocks without missing calls) against the fp64 oracle, GRM GCTA:
max over entries of |err| / max(|ref|, median |ref|).   python tools/err_check_nomiss.py [L]"""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle as orc
from oracle.synth import synth_geno
from snprelate_amd import _lib
L = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
n, blk = 1200, 16384
g = synth_geno(n, L, missing=0.0, seed=n, special=False)
ref = orc.grm_gcta(g)
floor = np.median(np.abs(ref))
for be in ("f16", "h3"):
    os.environ["SNPGPU_SYRK"] = be
    with _lib.Accumulator(_lib.GRM_GCTA, n, max_block_snps=blk) as a:
        for i in range(0, L, blk):
            a.feed(g[i:i + blk])
        got = a.grm_gcta(packed=True)
    print(n, L, be, "max rel err %.3e" % float(np.max(np.abs(got - ref) / np.maximum(np.abs(ref), floor))))
