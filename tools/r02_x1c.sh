#!/bin/bash
python -m pytest tests/test_gpu_parity.py -q -x -k "grm_gcta or pca_cov or several_fp32 or ragged" 2>&1 | tail -3
bash tools/bench_lib.sh "--no-sub-results --steps 8 --warmup 2" libsnpgpu_prev.so libsnpgpu.so
