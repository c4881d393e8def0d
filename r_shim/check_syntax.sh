#!/bin/bash
# Syntax check of the R-side binding where R and gdsfmt headers exist (they do not in this repository's build image):
#   R_HOME=/usr/lib/R GDSFMT_INC=<path to gdsfmt/include> SNPRELATE_SRC=<path to SNPRelate/src> r_shim/check_syntax.sh
# The shim's libsnpgpu call sequences are what tests/test_gpu_shim_order.py replays through ctypes.
set -e
: "${R_HOME:?set R_HOME}"; : "${GDSFMT_INC:?set GDSFMT_INC}"; : "${SNPRELATE_SRC:?set SNPRELATE_SRC}"
here=$(cd "$(dirname "$0")" && pwd)
g++ -std=gnu++14 -fsyntax-only -I"$R_HOME/include" -I"$GDSFMT_INC" -I"$SNPRELATE_SRC" -I"$here/../include" "$here/gpu_shim.cpp" && echo "gpu_shim.cpp: syntax ok"
