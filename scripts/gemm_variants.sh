#!/bin/bash
# development: build libmis_hip variants with -DMIS_GEMM_DBG_CT=<n> (gemm.hip ablations) into gpurun-visible files
#   scripts/gemm_variants.sh 1 2 8 ...   -> cv-ssl-mis_amd/mis_hip/libmis_hip_g<n>.so
cd "$(dirname "$0")/../cv-ssl-mis_amd/csrc" || exit 1
for n in "$@"; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -ffp-contract=fast \
        -DMIS_GEMM_DBG_CT=$n -c gemm.hip -o /tmp/gemm_g$n.o || exit 1
    objs=$(ls *.o | grep -v "^gemm.o$")
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../mis_hip/libmis_hip_g$n.so $objs /tmp/gemm_g$n.o || exit 1
done
