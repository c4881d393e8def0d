#!/bin/bash
SNPGPU_LIB=$PWD/snprelate_amd/libsnpgpu_d4.so python -m pytest tests/test_gpu_parity.py -q -x -k "king or beta" 2>&1 | tail -2
bash tools/bench_lib.sh "--no-sub-results --workload king --steps 50 --warmup 20" libsnpgpu.so libsnpgpu_d4.so
