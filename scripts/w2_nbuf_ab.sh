# A/B of the ring depth of the 16-output-channel 2-D Winograd forward kernel (development tool): builds side libraries under /tmp
set -e
cd "$(dirname "$0")/../cv-ssl-mis_amd/csrc"
OBJS=$(ls *.o | grep -v '^conv_wino2d.o$')
for nb in 3 4; do
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast -DMIS_W2V0_NBUF=$nb -c conv_wino2d.hip -o /tmp/w2_nb.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/libmis_w2nb.so $OBJS /tmp/w2_nb.o
echo "== NBUF $nb"
(cd ../.. && MIS_HIP_LIB=/tmp/libmis_w2nb.so python scripts/wino2d_bench.py 2>&1 | grep -E "^N(48|24) (16|32)->16 256")
done
