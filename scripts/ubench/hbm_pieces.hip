// Micro-benchmark: what HBM delivers when a kernel reads (and writes) its tensor in contiguous PIECES of P bytes that are
// `stride` bytes apart, instead of as one stream -- the access pattern of the tiled convolution / window-attention kernels:
// a 16 x 16-pixel box of the 2-D UNet reads 96-byte row pieces 1 KB apart (conv_wino2d.hip), window attention 128-byte head rows
// 1152 bytes apart (attention.hip), the streaming normalisation passes whole rows.
// Every 16-byte lane access is a float4; a group of P / 16 consecutive lanes covers one piece, consecutive groups take pieces
// `stride` bytes apart inside a row block, and the next "column" of pieces starts P bytes further -- so the whole buffer is read
// exactly once, only the ORDER (the piece size seen by one wave at a time) changes.  Buffers of 512 MB (beyond the 256 MB infinity
// cache), read + write of the same size.
// hipcc --offload-arch=gfx950 -O3 -w hbm_pieces.hip -o hbm_pieces.bin && ./hbm_pieces.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

// element index (in float4 units) of access a: pieces of PQ float4s, rows of RQ float4s (the stride between pieces)
__device__ __forceinline__ size_t map(size_t a, int PQ, int RQ, int rows_per_block) {
    // block = rows_per_block rows of RQ float4; inside a block walk piece-column by piece-column, rows inside a column
    const size_t per_block = (size_t)RQ * rows_per_block;
    const size_t blk = a / per_block, r0 = a - blk * per_block;
    const size_t col = r0 / ((size_t)PQ * rows_per_block), r1 = r0 - col * ((size_t)PQ * rows_per_block);
    const size_t row = r1 / PQ, q = r1 - row * PQ;
    return blk * per_block + row * RQ + col * PQ + q;
}

__global__ __launch_bounds__(256) void copy_pieces(const float4* __restrict__ src, float4* __restrict__ dst, size_t n4, int PQ, int RQ,
                                                   int rows_per_block, int write_pieces) {
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t a = (size_t)blockIdx.x * 256 + threadIdx.x; a < n4; a += stride) {
        const size_t e = map(a, PQ, RQ, rows_per_block);
        const float4 v = src[e];
        dst[write_pieces ? e : a] = v;
    }
}

int main() {
    const size_t bytes = 512ull << 20, n4 = bytes / 16;
    float4 *src, *dst;
    hipMalloc(&src, bytes); hipMalloc(&dst, bytes);
    hipMemset(src, 1, bytes); hipMemset(dst, 0, bytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int strides[] = {1024, 1152, 4096};
    const int pieces[] = {64, 96, 128, 256, 512, 1024};
    printf("copy of 512 MB (read + write): pieces of P bytes, `stride` bytes apart, 64 rows per block\n");
    for (int wp = 0; wp < 2; ++wp)
        for (int st : strides)
            for (int P : pieces) {
                if (P > st || st % P) { if (!(P == 96 && st % 96 == 0) && !(st == 1152 && (P == 128 || P == 64 || P == 96))) continue; }
                if (st % P) continue;
                const int PQ = P / 16, RQ = st / 16, rows = 64;
                if (n4 % ((size_t)RQ * rows)) continue;
                for (int w = 0; w < 3; ++w) copy_pieces<<<256 * 8, 256>>>(src, dst, n4, PQ, RQ, rows, wp);
                hipEventRecord(e0);
                const int reps = 10;
                for (int r = 0; r < reps; ++r) copy_pieces<<<256 * 8, 256>>>(src, dst, n4, PQ, RQ, rows, wp);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                printf("  %s  stride %5d  piece %5d B: %7.1f us  %.2f TB/s (read + write)\n", wp ? "pieces in, pieces out" : "pieces in, stream out",
                       st, P, ms / reps * 1e3, 2.0 * bytes / (ms / reps * 1e-3) / 1e12);
            }
    return 0;
}
