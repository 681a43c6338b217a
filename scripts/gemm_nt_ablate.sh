#!/bin/bash
# Timing-only ablations of the staged NT GEMM (pre-split bf16x3 form) over the 20 forward / dX shapes of a SwinUnet step:
#   scripts/gemm_variants.sh 2 8 32 64      (here, builds cv-ssl-mis_amd/mis_hip/libmis_hip_g<n>.so; results of those builds are WRONG)
#   gpurun -- 'bash scripts/gemm_nt_ablate.sh > gpurun_out/gemm_nt_ablation.txt'
# bits: 2 no DMA in the k-loop, 8 no epilogue, 32 no split of the A fragments, 64 no MFMAs.  Remove the variant libraries afterwards.
lib=cv-ssl-mis_amd/mis_hip
export MIS_GEMM_REGA=0
for v in base 64 2 32 8; do
    if [ $v = base ]; then unset MIS_HIP_LIB; else export MIS_HIP_LIB=$PWD/$lib/libmis_hip_g$v.so; fi
    echo "== variant $v"
    python scripts/gemm_nt_bench.py --split 2>/dev/null | cut -c1-64
done
