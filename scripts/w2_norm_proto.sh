#!/bin/bash
# Review item 5 (round 4) prototype: BatchNorm scale / shift + LeakyReLU applied on the LDS -> register read of the 2-D
# Winograd forward kernel (-DMIS_W2_NORM=1, conv_wino2d.hip) instead of in a pass of its own.  Builds a side library under
# /tmp and runs scripts/w2_norm_proto.py against it: parity with the two-kernel form, and what the fused read costs per layer
# next to the pass it would remove.  The product library is not touched.
set -e
cd "$(dirname "$0")/../cv-ssl-mis_amd/csrc"
OBJS=$(ls *.o | grep -v '^conv_wino2d.o$')
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast -DMIS_W2_NORM=1 -c conv_wino2d.hip -o /tmp/w2_norm.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/libmis_hip_w2norm.so $OBJS /tmp/w2_norm.o
cd ../..
MIS_HIP_LIB=/tmp/libmis_hip_w2norm.so python scripts/w2_norm_proto.py
