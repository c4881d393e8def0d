import os, sys, itertools
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import oracle as orc
from oracle.synth import synth_geno
from snprelate_amd import _lib
n, L = int(sys.argv[1]) if len(sys.argv) > 1 else 3000, 20000
g = synth_geno(n, L, missing=0.0, seed=3, special=False)
ref = orc.grm_gcta(g)
med = np.median(np.abs(ref))
for uv, pr, blk in itertools.product(("0", "1"), ("1024", "2048", "4096", "8192", "16384", "32768"), (16384, 32768)):
    os.environ["SNPGPU_SYRK_UV"] = uv
    os.environ["SNPGPU_H3_PROMOTE"] = pr
    with _lib.Accumulator(_lib.GRM_GCTA, n, max_block_snps=blk) as a:
        for i in range(0, L, blk):
            a.feed(g[i:i + blk])
        got = a.grm_gcta(packed=True)
    err = np.max(np.abs(got - ref) / (np.abs(ref) + med))
    print("uv=%s promote=%s blk=%d offdiag=%.3e" % (uv, pr, blk, err), flush=True)
