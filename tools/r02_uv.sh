#!/bin/bash
# A/B: single-product SYRK for blocks without missing calls (syrk_uv_kernel, default) vs the exact-row kernel (SNPGPU_SYRK_UV=0)
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_api_golden.py -q -x -k "grm or pca or GRM or PCA or syrk or ragged or several" 2>&1 | tail -5
bash tools/bench_env.sh "--no-sub-results --no-cpu-baseline --steps 8 --warmup 2" "SNPGPU_SYRK_UV=0" "SNPGPU_SYRK_UV=1"
