#!/bin/bash
# registers / scratch / LDS of every kernel of one .hip file, plus instruction counts of interest per kernel:
#   tools/kernel_resources.sh kernels_pair.hip
f=${1:-kernels_pair.hip}
cd /tmp && /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Rpass-analysis=kernel-resource-usage \
    -c /root/repo/snprelate_amd/csrc/$f -o /tmp/kres.o --save-temps=obj 2>&1 |
    grep -E "Function Name|VGPRs:|AGPRs:|ScratchSize|LDS Size|Occupancy" | sed 's/.*remark: [^ ]* *//; s/ \[-Rpass.*//; s/^ *//' |
    awk '/^Function Name:/ {if (n) print line; n=1; cmd="echo " $3 " | c++filt"; cmd | getline d; close(cmd); line=substr(d,1,64)} !/^Function Name:/ {line=line " | " $0} END {print line}'
s=/tmp/${f%.hip}-hip-amdgcn-amd-amdhsa-gfx950.s
awk '/^_Z[A-Za-z0-9_]*:/ {k=$1} /ds_read_b64/ {b64[k]++} /ds_read_b32/ {b32[k]++} /v_mfma/ {m[k]++} /scratch_/ {sc[k]++} /s_barrier/ {bar[k]++} /v_perm_b32/ {pm[k]++}
     END {for (k in m) printf "%s mfma=%d ds_read_b32=%d ds_read_b64=%d v_perm=%d scratch_ops=%d barriers=%d\n", substr(k,1,60), m[k], b32[k], b64[k], pm[k], sc[k], bar[k]}' $s | sort
