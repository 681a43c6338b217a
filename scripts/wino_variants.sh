#!/bin/bash
# development: build libmis_hip variants with -DMIS_WINO_DBG_CT=<n> (and any extra -D flags) into gpurun-visible files
#   scripts/wino_variants.sh "1" "2 -DFOO=1" ...   -> cv-ssl-mis_amd/mis_hip/libmis_hip_v<i>.so
cd "$(dirname "$0")/../cv-ssl-mis_amd/csrc" || exit 1
i=0
for v in "$@"; do
    set -- $v
    n=$1; shift
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -ffp-contract=fast \
        -DMIS_WINO_DBG_CT=$n "$@" -c conv_wino.hip -o /tmp/conv_wino_v$i.o || exit 1
    objs=$(ls *.o | grep -v "^conv_wino.o$")
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../mis_hip/libmis_hip_v$i.so $objs /tmp/conv_wino_v$i.o || exit 1
    i=$((i + 1))
done
