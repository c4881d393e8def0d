#!/bin/bash
# single-product SYRK: one workgroup per CU walking the work list (default) vs one workgroup per tile
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -k "grm or pca or several or ragged" 2>&1 | tail -3
bash tools/bench_env.sh "--no-sub-results --steps 8 --warmup 2" "SNPGPU_X1_PERSIST=0" "SNPGPU_X1_PERSIST=1"
