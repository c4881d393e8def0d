import os, sys, faulthandler
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
faulthandler.dump_traceback_later(60, exit=True)
import numpy as np, torch, torch.distributed as dist
from oracle.synth import synth_geno
from snprelate_amd import multigpu
dist.init_process_group("gloo")
rank = dist.get_rank()
n, L = 1300, 2100
g = synth_geno(n, L, missing=0.03, seed=17)
blocks = lambda: (g[i:i + 1024] for i in range(0, L, 1024))
print("rank", rank, "start", flush=True)
pca = multigpu.pca_distributed(blocks(), n, eigen_cnt=8, max_block_snps=1024)
print("rank", rank, "done", pca["info"], flush=True)
dist.destroy_process_group()
