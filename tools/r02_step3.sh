#!/bin/bash
set -u
OUT=$PWD/gpurun_out/${1:-r02e}
mkdir -p "$OUT"
python -m pytest tests -m gpu -q 2>&1 | tail -12 | tee "$OUT/pytest.txt"
python tools/northstar_share.py --rank 0 --world 8 --steps 4 2>&1 | tail -1 | tee "$OUT/northstar_pca_rank0.json"
python tools/northstar_share.py --rank 0 --world 8 --steps 4 --kind GRM_GCTA --matmul-cols 0 2>&1 | tail -1 | tee "$OUT/northstar_grm_rank0.json"
python tools/northstar_share.py --rank 0 --world 8 --steps 4 --kind KING_ROBUST --missing 0.05 2>&1 | tail -1 | tee "$OUT/northstar_king_rank0.json"
python tools/northstar_share.py --rank 7 --world 8 --panels-per-rank 2 --steps 4 2>&1 | tail -1 | tee "$OUT/northstar_pca_rank7_ppr2.json"
SNPGPU_SYRK=f32 python bench.py --no-cpu-baseline --no-sub-results --steps 3 --warmup 1 2>/dev/null | tail -1 | tee "$OUT/bench_f32.json" | cut -c1-400
