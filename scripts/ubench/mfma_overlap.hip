// Micro-benchmark: how many independent VALU / SALU / LDS instructions issue under one v_mfma_f32_16x16x4_f32?
// hipcc --offload-arch=gfx950 -O3 -w mfma_overlap.hip -o mfma_overlap.bin && ./mfma_overlap.bin   (the .bin travels with gpurun)
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int N, int KIND>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
    __shared__ float lds[4096];
    typedef int i32x4 __attribute__((ext_vector_type(4)));
    i32x4 rs;
    {
        const unsigned long long b = (unsigned long long)out;
        rs[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)b);
        rs[1] = __builtin_amdgcn_readfirstlane((int)(unsigned)(b >> 32)) & 0xffff;
        rs[2] = 1 << 20; rs[3] = 0x00020000;
    }
    const unsigned ldsb = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long long)((__attribute__((address_space(3))) float*)lds) + (threadIdx.x >> 6) * 1024u * 0u);
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{0, 0, 0, 0};
    float a = threadIdx.x, b = 1.f;
    f32x2 v[8];
    for (int i = 0; i < 8; ++i) v[i] = f32x2{(float)i, (float)threadIdx.x};
    lds[threadIdx.x] = a;
    __syncthreads();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(acc[m]) : "v"(a), "v"(b));
#pragma unroll
            for (int j = 0; j < N; ++j) {
                if (KIND == 0) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(v[j % 8]) : "v"(v[(j + 1) % 8]));
                if (KIND == 1) asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[j % 8][0]) : "v"(v[(j + 1) % 8][1]));
                if (KIND == 2) asm volatile("s_nop 0");
                if (KIND == 3) { float t; asm volatile("ds_read_b32 %0, %1" : "=v"(t) : "v"(threadIdx.x * 4)); }
                if (KIND == 4) asm volatile("s_mov_b32 s40, s41" ::: "s40");
                if (KIND == 5) {   // LDS-DMA of 1 KiB (64 lanes x 16 B) from an L2-resident buffer, as the conv kernels issue it
                    unsigned keep;
                    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\t"
                                 "buffer_load_dwordx4 %2, %3, 0 offen lds\n\ts_mov_b32 m0, %0"
                                 : "=&s"(keep) : "s"(ldsb), "v"(threadIdx.x * 16u + (unsigned)((it * 8 + m) & 63) * 4096u), "s"(rs) : "memory");
                }
                if (KIND == 6) {   // the same as a single dword per lane
                    unsigned keep;
                    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\t"
                                 "buffer_load_dword %2, %3, 0 offen lds\n\ts_mov_b32 m0, %0"
                                 : "=&s"(keep) : "s"(ldsb), "v"(threadIdx.x * 4u + (unsigned)((it * 8 + m) & 63) * 4096u), "s"(rs) : "memory");
                }
            }
        }
        if (KIND == 3) asm volatile("s_waitcnt lgkmcnt(0)");
        if (KIND >= 5) asm volatile("s_waitcnt vmcnt(0)");
    }
    asm volatile("s_nop 15\n s_nop 15");
    float s = 0;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + v[i][0] + v[i][1];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int N, int KIND>
void run(const char* name) {
    float* out; hipMalloc(&out, 4 << 20);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    for (int blocks : {256}) {
        k<N, KIND><<<blocks, 256>>>(out, 100);
        hipEventRecord(e0);
        k<N, KIND><<<blocks, 256>>>(out, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%s n=%d: %.1f ns per MFMA\n", name, N, ms * 1e6 / (iters * 8.0));
    }
    hipFree(out);
}

int main() {
    run<0, 0>("pk_add"); run<2, 0>("pk_add"); run<4, 0>("pk_add"); run<6, 0>("pk_add"); run<7, 0>("pk_add"); run<8, 0>("pk_add"); run<12, 0>("pk_add");
    run<4, 1>("v_add"); run<6, 1>("v_add"); run<8, 1>("v_add");
    run<4, 2>("s_nop"); run<7, 2>("s_nop"); run<12, 2>("s_nop");
    run<2, 3>("ds_read"); run<4, 3>("ds_read");
    run<4, 4>("s_mov"); run<7, 4>("s_mov");
    run<1, 5>("dma_x4 (n per MFMA)"); run<2, 5>("dma_x4");
    run<1, 6>("dma_x1"); run<2, 6>("dma_x1");
    return 0;
}
