#!/usr/bin/env python3
"""Shader clock / socket power while the eigen solver's building block (snpgpu_pca_panel_matmul, 48 columns) runs back to back
on rank 0's panel of the N = 500 000 plan:  python tools/matmul_clock.py [reps]"""
import os, re, subprocess, sys, threading, time, statistics as st
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from snprelate_amd import _lib
from snprelate_amd.dist import panel_plan

n, m = 500000, 48
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 150
bounds, owned = panel_plan(n, 8, 1)
r0, r1 = bounds[owned[0][0]], bounds[owned[0][0] + 1]
a = _lib.Accumulator(_lib.PCA_COV, n, row_begin=int(r0), row_end=int(r1), max_block_snps=4096)
rb = (n + 3) // 4
blk = torch.empty((1024, rb), dtype=torch.uint8, device="cuda")
_lib.synth_block(blk.data_ptr(), n, 0, 1024, 1, 0.0, 0, False, 0)
a.feed_device(blk.data_ptr(), 1024)
q = torch.randn((m, n), dtype=torch.float64, device="cuda")
y = torch.zeros((m, n), dtype=torch.float64, device="cuda")
a.pca_panel_matmul(1.0, q.data_ptr(), m, y.data_ptr())
torch.cuda.synchronize()
samples, stop = [], False
def watch():
    while not stop:
        out = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--csv"], capture_output=True, text=True).stdout
        mm = re.findall(r"\((\d+)Mhz\)", out)
        pw = re.findall(r",(\d+\.\d+)\s*$", out.strip(), re.M)
        if len(mm) >= 3 and pw:
            samples.append((int(mm[2]), float(pw[-1])))
        time.sleep(0.15)
t = threading.Thread(target=watch); t.start()
t0 = time.perf_counter()
for _ in range(reps):
    a.pca_panel_matmul(1.0, q.data_ptr(), m, y.data_ptr())
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / reps
stop = True; t.join()
busy = [s for s in samples if s[1] > 0.8 * max(x[1] for x in samples)] if samples else []
print("panel product: %.2f ms | sclk MHz median %s | power W median %s (%d samples)" % (
    dt * 1e3, st.median([s[0] for s in busy]) if busy else None, st.median([s[1] for s in busy]) if busy else None, len(busy)))
