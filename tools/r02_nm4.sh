#!/bin/bash
# A/B of the binary pair kernel's per-wave tile: 64 x 64 at 2 waves per SIMD (libsnpgpu.so) vs 128 x 64 at 1 wave per SIMD with AGPR accumulators
python -m pytest tests/test_gpu_parity.py -q -x -k "ibs_counts or king_robust" 2>&1 | tail -3
SNPGPU_LIB=$PWD/snprelate_amd/libsnpgpu_nm4.so python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -q -x -k "ibs or king" 2>&1 | tail -3
bash tools/bench_lib.sh "--no-sub-results --workload ibs --steps 50 --warmup 20" libsnpgpu.so libsnpgpu_nm4.so
bash tools/bench_lib.sh "--no-sub-results --workload ibs --block 16384 --steps 100 --warmup 40" libsnpgpu.so libsnpgpu_nm4.so
bash tools/bench_lib.sh "--no-sub-results --workload king --missing 0 --steps 50 --warmup 20" libsnpgpu.so libsnpgpu_nm4.so
