#!/bin/bash
set -u
OUT=$PWD/gpurun_out/${1:-r02k}
mkdir -p "$OUT"
python tools/northstar_share.py --rank 0 --world 8 --steps 4 2>&1 | tail -1 | tee "$OUT/northstar_pca_rank0.json"
python tools/northstar_share.py --rank 0 --world 8 --steps 4 --kind GRM_GCTA --matmul-cols 0 2>&1 | tail -1 | tee "$OUT/northstar_grm_rank0.json"
python tools/northstar_share.py --rank 0 --world 8 --steps 4 --kind KING_ROBUST --missing 0.05 2>&1 | tail -1 | tee "$OUT/northstar_king_rank0.json"
python tools/northstar_share.py --rank 7 --world 8 --panels-per-rank 2 --steps 4 2>&1 | tail -1 | tee "$OUT/northstar_pca_rank7_ppr2.json"
python tools/northstar_share.py --rank 3 --world 8 --steps 4 --kind GRM_GCTA --missing 0.02 --matmul-cols 0 2>&1 | tail -1 | tee "$OUT/northstar_grm_rank3_missing.json"
