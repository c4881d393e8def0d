#!/bin/bash
# libsnpgpu variant with extra -D flags for kernels_pair.hip (timing ablations):  tools/build_variant.sh <name> <flags...>
#   -> snprelate_amd/libsnpgpu_<name>.so (use with SNPGPU_LIB / tools/bench_lib.sh)
name=$1; shift
cd snprelate_amd/csrc && /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function "$@" -c kernels_pair.hip -o /tmp/kernels_pair_$name.o &&
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libsnpgpu_$name.so kernels_prep.o /tmp/kernels_pair_$name.o kernels_final.o kernels_proj.o kernels_eig.o api.o proj.o workspace.o eigen.o multi.o diag.o -L/opt/rocm/lib -lhipsolver -lrocblas -ldl && echo built libsnpgpu_$name.so
