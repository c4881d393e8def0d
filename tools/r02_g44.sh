#!/bin/bash
python -m pytest tests/test_gpu_parity.py tests/test_gpu_api_golden.py -q -x -k "grm or GRM" 2>&1 | tail -3
SNPGPU_LIB=$PWD/snprelate_amd/libsnpgpu_g44.so python -m pytest tests/test_gpu_parity.py -q -x -k "grm" 2>&1 | tail -3
bash tools/bench_lib.sh "--no-sub-results --workload grm --missing 0.02 --steps 6 --warmup 2" libsnpgpu.so libsnpgpu_g44.so
