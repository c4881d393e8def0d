#!/usr/bin/env python3
"""Error figures (tests/norms.py) of the default GRM path over the parity-test sizes, three seeds each, with and without
missing calls (GPU):  python tools/err_margins.py"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import oracle as orc
from oracle.synth import synth_geno
from snprelate_amd import _lib
from norms import error_figures, tri_diag_scale
SIZES = [(37, 301, 100), (279, 1000, 333), (600, 2500, 1024), (1030, 4100, 4096)]
for n, L, blk in SIZES:
    for missing in (0.0, 0.05):
        for seed_off in (3, 103, 203):
            g = synth_geno(n, L, missing=missing, seed=n + seed_off)
            ref = orc.grm_gcta(g)
            with _lib.Accumulator(_lib.GRM_GCTA, n, max_block_snps=4096) as a:
                for i in range(0, L, blk): a.feed(g[i:i + blk])
                got = a.grm_gcta(packed=True)
            f = error_figures(got, ref, tri_diag_scale(ref, n))
            print(n, L, missing, seed_off, "contract %.2e offdiag %.2e" % (f["contract"], f["offdiag"]))
