#!/bin/bash
# Builds copies of the library with the z-ring weight-gradient kernel's cycle counters compiled in (-DMIS_WR_PROF=1) under /tmp
# and runs scripts/wgrad_prof.py against them; `all` also builds the ablations.  The product library is not touched.
set -e
cd "$(dirname "$0")/../cv-ssl-mis_amd/csrc"
OBJS=$(ls *.o | grep -v '^conv_wino_wgrad.o$')
one() {   # tag, extra flags
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast -DMIS_WR_PROF=1 $2 -c conv_wino_wgrad.hip -o /tmp/wg_prof.o
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/libmis_hip_prof.so $OBJS /tmp/wg_prof.o
    (cd ../.. && MIS_WR_TAG="$1" MIS_HIP_LIB=/tmp/libmis_hip_prof.so python scripts/wgrad_prof.py)
}
if [ "$1" = "all" ]; then
    one full ""
    one "no DMA" -DMIS_WR_ABL=1
    one "no transforms" -DMIS_WR_ABL=2
    one "no patch loads" -DMIS_WR_ABL=4
    one "no barrier" -DMIS_WR_ABL=8
    one "no MFMA" -DMIS_WR_ABL=16
    one "no DMA, no transforms" -DMIS_WR_ABL=3
    one "MFMA only" -DMIS_WR_ABL=15
    one "full (again)" ""
else
    one "${MIS_WR_TAG:-full}" "$MIS_WR_EXTRA"
fi
