import time, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
t0=time.time()
import oracle as orc
from oracle.synth import synth_geno
from snprelate_amd import _lib
print("import", time.time()-t0)
n,L=1030,4100
g=synth_geno(n,L,missing=0.05,seed=3)
for rep in range(2):
    t=time.time(); ref=orc.grm_gcta(g); print("oracle grm", time.time()-t, "threads", orc.num_threads())
    t=time.time(); a=_lib.Accumulator(_lib.GRM_GCTA,n,max_block_snps=4096); print("create", time.time()-t)
    t=time.time()
    for i in range(0,L,4096): a.feed(g[i:i+4096])
    a.sync(); print("feed", time.time()-t)
    t=time.time(); got=a.grm_gcta(packed=True); print("final", time.time()-t)
    t=time.time(); a.close(); print("close", time.time()-t)
    t=time.time(); c=orc.ibs_count(g); print("oracle ibs", time.time()-t)
