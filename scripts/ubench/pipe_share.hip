// Micro-benchmark: do a wave's fp32 MFMAs and ANOTHER wave's VALU work on the same SIMD overlap, or do they share a pipe?
// 512-thread workgroups: waves w and w + 4 sit on the same SIMD.  Waves 0-3 run role A, waves 4-7 role B; each wave
// times itself with the 100 MHz wall clock.  Reported: A alone, B alone, A with B (both durations).
// hipcc --offload-arch=gfx950 -O3 -w pipe_share.hip -o pipe_share.bin && ./pipe_share.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

enum { NONE = 0, MFMA_F32 = 1, MFMA_BF16 = 2, PK_ADD = 3, V_ADD = 4, ACC_READ = 5, DS_READ = 6, DS_WRITE = 7, VLOAD = 8, PK_FMA = 9, V_FMA = 10, MFMA_F32_32 = 11 };

template <int KIND>
__device__ __forceinline__ void body(int iters, float* out, float* gsrc) {
    __shared__ float lds[8192];
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{0, 0, 0, 0};
    float a = threadIdx.x, b = 1.f;
    f32x2 v[8];
    for (int i = 0; i < 8; ++i) v[i] = f32x2{(float)i, (float)threadIdx.x};
    bf16x8 ha, hb;
    for (int i = 0; i < 8; ++i) { ha[i] = (short)(threadIdx.x + i); hb[i] = (short)i; }
    f32x4 t4 = {1, 2, 3, 4};
    typedef f32x4 f32x16v[4];
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            if (KIND == MFMA_F32) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(acc[m]) : "v"(a), "v"(b));
            if (KIND == MFMA_BF16) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[m]) : "v"(ha), "v"(hb));
            if (KIND == PK_ADD) {
#pragma unroll
                for (int j = 0; j < 8; ++j) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(v[j % 8]) : "v"(v[(j + 1) % 8]));
            }
            if (KIND == PK_FMA) {
#pragma unroll
                for (int j = 0; j < 8; ++j) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(v[j % 8]) : "v"(v[(j + 1) % 8]));
            }
            if (KIND == V_ADD) {
#pragma unroll
                for (int j = 0; j < 8; ++j) asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[j % 8][0]) : "v"(v[(j + 1) % 8][1]));
            }
            if (KIND == V_FMA) {
#pragma unroll
                for (int j = 0; j < 8; ++j) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(v[j % 8][0]) : "v"(v[(j + 1) % 8][1]));
            }
            if (KIND == ACC_READ) {
#pragma unroll
                for (int j = 0; j < 8; ++j) asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(v[j][0]) : "a"(acc[j][0]));
            }
            if (KIND == DS_READ) {
#pragma unroll
                for (int j = 0; j < 4; ++j) asm volatile("ds_read_b128 %0, %1" : "=v"(t4) : "v"((threadIdx.x & 63) * 16 + j * 1024));
            }
            if (KIND == DS_WRITE) {
#pragma unroll
                for (int j = 0; j < 4; ++j) asm volatile("ds_write_b128 %0, %1" :: "v"((threadIdx.x & 63) * 16 + j * 1024), "v"(t4));
            }
            if (KIND == VLOAD) {
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(t4) : "v"(gsrc + (threadIdx.x & 63) * 4 + ((it * 16 + m * 2 + j) & 255) * 256));
            }
        }
        if (KIND == DS_READ || KIND == DS_WRITE) asm volatile("s_waitcnt lgkmcnt(0)");
        if (KIND == VLOAD) asm volatile("s_waitcnt vmcnt(0)");
    }
    asm volatile("s_nop 15\n s_nop 15");
    float s = t4[0] + lds[threadIdx.x];
    for (int i = 0; i < 8; ++i) s += acc[i][0] + v[i][0] + v[i][1];
    if (s == 12345.678f) out[threadIdx.x] = s;
}

template <int A, int B>
__global__ __launch_bounds__(512) void k(float* out, float* gsrc, int itA, int itB, long long* times) {
    const int wave = threadIdx.x >> 6;
    __syncthreads();
    const long long t0 = wall_clock64();
    if (wave < 4) body<A>(itA, out, gsrc); else body<B>(itB, out, gsrc);
    const long long t1 = wall_clock64();
    if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) times[wave] = t1 - t0;
}

template <int A, int B>
void run(const char* name, int itA, int itB, int perA, int perB) {
    float *out, *gsrc; long long* times;
    hipMalloc(&out, 4 << 20); hipMalloc(&gsrc, 4 << 20); hipMalloc(&times, 64);
    hipMemset(gsrc, 0, 4 << 20);
    long long h[8];
    auto once = [&](int a, int b) {
        k<A, B><<<256, 512>>>(out, gsrc, 10, 10, times);
        k<A, B><<<256, 512>>>(out, gsrc, a, b, times);
        hipDeviceSynchronize();
        hipMemcpy(h, times, 64, hipMemcpyDeviceToHost);
    };
    once(itA, 0);  const double a_alone = h[0] * 10.0;
    once(0, itB);  const double b_alone = h[4] * 10.0;
    once(itA, itB); const double a_both = h[0] * 10.0, b_both = h[4] * 10.0;
    printf("%-28s A alone %8.1f us (%.2f ns/instr)  B alone %8.1f us (%.2f ns/instr)  together A %8.1f B %8.1f  sum %8.1f\n", name,
           a_alone / 1e3, a_alone / ((double)itA * perA), b_alone / 1e3, b_alone / ((double)itB * perB), a_both / 1e3, b_both / 1e3,
           (a_alone + b_alone) / 1e3);
    hipFree(out); hipFree(gsrc); hipFree(times);
}

int main() {
    setvbuf(stdout, nullptr, _IONBF, 0);
    // iteration counts chosen so that both roles take about 1.5 ms alone
    run<MFMA_F32, PK_ADD>("mfma_f32 | pk_add", 12000, 12000, 8, 64);
    run<MFMA_F32, V_ADD>("mfma_f32 | v_add", 12000, 12000, 8, 64);
    run<MFMA_F32, V_FMA>("mfma_f32 | v_fma", 12000, 12000, 8, 64);
    run<MFMA_F32, PK_FMA>("mfma_f32 | pk_fma", 12000, 12000, 8, 64);
    run<MFMA_F32, ACC_READ>("mfma_f32 | accvgpr_read", 12000, 12000, 8, 64);
    run<MFMA_F32, DS_READ>("mfma_f32 | ds_read_b128", 12000, 6000, 8, 32);
    run<MFMA_F32, DS_WRITE>("mfma_f32 | ds_write_b128", 12000, 6000, 8, 32);
    // (the global_load case faults on ROCm 7.2 -- its address arithmetic is wrong -- and is not needed for the pipe question)
    run<MFMA_F32, MFMA_F32>("mfma_f32 | mfma_f32", 12000, 12000, 8, 8);
    run<MFMA_BF16, PK_ADD>("mfma_bf16 | pk_add", 24000, 12000, 8, 64);
    run<MFMA_BF16, V_ADD>("mfma_bf16 | v_add", 24000, 12000, 8, 64);
    run<MFMA_BF16, PK_FMA>("mfma_bf16 | pk_fma", 24000, 12000, 8, 64);
    run<MFMA_BF16, MFMA_BF16>("mfma_bf16 | mfma_bf16", 24000, 24000, 8, 8);
    run<MFMA_BF16, DS_READ>("mfma_bf16 | ds_read_b128", 24000, 6000, 8, 32);
    run<PK_ADD, PK_ADD>("pk_add | pk_add", 12000, 12000, 64, 64);
    run<V_ADD, V_ADD>("v_add | v_add", 12000, 12000, 64, 64);
    return 0;
}
