#!/bin/bash
# A/B: exact-row SYRK with two waves per SIMD (syrk_h3_kernel<2, true>) vs one wave per SIMD (syrk_x1_kernel, SNPGPU_SYRK_X1=1)
SNPGPU_SYRK_X1=1 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_api_golden.py -q -x -k "grm or pca or GRM or PCA or config2 or config3 or syrk or ragged" 2>&1 | tail -5
bash tools/bench_env.sh "--no-sub-results --steps 8 --warmup 2" "SNPGPU_SYRK_X1=0" "SNPGPU_SYRK_X1=1"
bash tools/bench_env.sh "--no-sub-results --steps 6 --warmup 2 --missing 0.02" "SNPGPU_SYRK_X1=0" "SNPGPU_SYRK_X1=1"
