// Micro-benchmark: what does one global->LDS transfer cost the issuing wave when it is NOT throughput bound?
// One wave per SIMD (256 threads, 1 workgroup per CU), a loop of 64 independent fp32 MFMAs with D transfers of 1 KiB per
// wave spread evenly between them.  Variants: LDS-DMA dwordx4, LDS-DMA dword, plain buffer_load_dwordx4 into VGPRs
// (consumed by a ds_write_b128 one iteration later), plain loads never written.
// hipcc --offload-arch=gfx950 -O3 -w dma_issue.hip -o dma_issue.bin && ./dma_issue.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

enum { DMA_X4 = 0, DMA_X1 = 1, VLOAD_WRITE = 2, VLOAD_ONLY = 3, DS_WRITE_ONLY = 4 };

template <int D, int KIND>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void k(float* out, const float* src, int iters) {
    extern __shared__ float lds[];
    i32x4 rs;
    {
        const unsigned long long b = (unsigned long long)src;
        rs[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)b);
        rs[1] = __builtin_amdgcn_readfirstlane((int)(unsigned)(b >> 32)) & 0xffff;
        rs[2] = 64 << 20; rs[3] = 0x00020000;
    }
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned ldsb = (unsigned)(unsigned long long)((__attribute__((address_space(3))) float*)lds) + wave * 16384u;
    f32x4 acc[16];
    for (int i = 0; i < 16; ++i) acc[i] = f32x4{0, 0, 0, 0};
    float a = threadIdx.x, b = 1.f;
    f32x4 st[16];
    for (int i = 0; i < 16; ++i) st[i] = f32x4{0, 0, 0, 0};
    const unsigned lane_off = (threadIdx.x & 63) * 16u;
    // every workgroup walks its own 1 MiB window of the source (L2 resident after the first pass)
    const unsigned base = (blockIdx.x & 31) * (1u << 20) + wave * (1u << 18);
    for (int it = 0; it < iters; ++it) {
        const unsigned soff = base + (unsigned)((it & 15) * 16384);
#pragma unroll
        for (int m = 0; m < 64; ++m) {
            asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(acc[m % 16]) : "v"(a), "v"(b));
            if (D > 0 && m % (64 / (D > 0 ? D : 1)) == 1) {
                constexpr int dummy = 0;
                const int j = m / (64 / (D > 0 ? D : 1));
                if (KIND == DMA_X4) {
                    unsigned keep;
                    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\t"
                                 "buffer_load_dwordx4 %2, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
                                 : "=&s"(keep) : "s"(ldsb + j * 1024u), "v"(lane_off + j * 1024u), "s"(rs), "s"(soff) : "memory");
                }
                if (KIND == DMA_X1) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        unsigned keep;
                        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\t"
                                     "buffer_load_dword %2, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
                                     : "=&s"(keep) : "s"(ldsb + j * 1024u + q * 256u), "v"((threadIdx.x & 63) * 4u + j * 1024u + q * 256u), "s"(rs), "s"(soff) : "memory");
                    }
                }
                if (KIND == VLOAD_WRITE) {
                    // write what was loaded one iteration ago, then reload the register
                    asm volatile("s_waitcnt vmcnt(%2)\n\tds_write_b128 %0, %1" :: "v"(lane_off + j * 1024u + wave * 16384u), "v"(st[j]), "n"(D - 1) : "memory");
                    asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(st[j]) : "v"(lane_off + j * 1024u), "s"(rs), "s"(soff) : "memory");
                }
                if (KIND == VLOAD_ONLY) {
                    asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(st[j]) : "v"(lane_off + j * 1024u), "s"(rs), "s"(soff) : "memory");
                }
                if (KIND == DS_WRITE_ONLY) {
                    asm volatile("ds_write_b128 %0, %1" :: "v"(lane_off + j * 1024u + wave * 16384u), "v"(st[j]) : "memory");
                }
            }
        }
        if (KIND == DMA_X4 || KIND == DMA_X1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (KIND == DS_WRITE_ONLY) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n s_nop 15\n s_nop 15" ::: "memory");
    float s = lds[threadIdx.x];
    for (int i = 0; i < 16; ++i) s += acc[i][0] + st[i][0];
    if (s == 12345.678f) out[threadIdx.x] = s;
}

template <int D, int KIND>
void run(const char* name, float* out, float* src) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 4000;
    hipFuncSetAttribute((const void*)&k<D, KIND>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    k<D, KIND><<<256, 256, 65536>>>(out, src, 200);
    hipEventRecord(e0);
    k<D, KIND><<<256, 256, 65536>>>(out, src, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-14s D=%2d: %7.1f ns per 64-MFMA iteration  (%.1f ns per MFMA)\n", name, D, ms * 1e6 / iters, ms * 1e6 / iters / 64);
}

int main() {
    setvbuf(stdout, nullptr, _IONBF, 0);
    float *out, *src;
    hipMalloc(&out, 4 << 20); hipMalloc(&src, 64 << 20);
    hipMemset(src, 0, 64 << 20);
    run<0, DMA_X4>("none", out, src);
    run<2, DMA_X4>("dma_x4", out, src); run<4, DMA_X4>("dma_x4", out, src); run<8, DMA_X4>("dma_x4", out, src); run<16, DMA_X4>("dma_x4", out, src);
    run<2, DMA_X1>("4 x dma_x1", out, src); run<4, DMA_X1>("4 x dma_x1", out, src); run<8, DMA_X1>("4 x dma_x1", out, src);
    run<2, VLOAD_WRITE>("vload+dswrite", out, src); run<4, VLOAD_WRITE>("vload+dswrite", out, src); run<8, VLOAD_WRITE>("vload+dswrite", out, src); run<16, VLOAD_WRITE>("vload+dswrite", out, src);
    run<4, VLOAD_ONLY>("vload only", out, src); run<8, VLOAD_ONLY>("vload only", out, src); run<16, VLOAD_ONLY>("vload only", out, src);
    run<4, DS_WRITE_ONLY>("dswrite only", out, src); run<8, DS_WRITE_ONLY>("dswrite only", out, src); run<16, DS_WRITE_ONLY>("dswrite only", out, src);
    return 0;
}
