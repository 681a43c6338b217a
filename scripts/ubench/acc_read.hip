// Micro-benchmark: issue cost of v_accvgpr_read_b32 (AGPR -> VGPR) against plain VALU moves / packed adds, one wave per SIMD.
// hipcc --offload-arch=gfx950 -O3 -w acc_read.hip -o acc_read.bin && ./acc_read.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int KIND>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void k(float* out, int iters) {
    f32x4 acc[16];
    for (int i = 0; i < 16; ++i) acc[i] = f32x4{(float)i, 1, 2, 3};
    float v[64];
    f32x2 p[16];
    for (int i = 0; i < 64; ++i) v[i] = threadIdx.x + i;
    for (int i = 0; i < 16; ++i) p[i] = f32x2{(float)i, 1.f};
    for (int i = 0; i < 16; ++i) asm volatile("" : "+a"(acc[i]));
    for (int it = 0; it < iters; ++it) {
        if (KIND == 0) {      // 64 reads, consecutive AGPRs
#pragma unroll
            for (int j = 0; j < 64; ++j) asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(v[j]) : "a"(acc[j / 4][j % 4]));
        }
        if (KIND == 1) {      // 64 VGPR moves
#pragma unroll
            for (int j = 0; j < 64; ++j) asm volatile("v_mov_b32 %0, %1" : "=v"(v[j]) : "v"(v[(j + 17) % 64]));
        }
        if (KIND == 2) {      // 64 packed adds
#pragma unroll
            for (int j = 0; j < 64; ++j) asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(p[j % 16]) : "v"(p[(j + 5) % 16]), "v"(p[(j + 9) % 16]));
        }
        if (KIND == 3) {      // the epilogue's pattern: 8 reads then 4 packed adds on the pairs
#pragma unroll
            for (int g = 0; g < 8; ++g) {
#pragma unroll
                for (int j = 0; j < 8; ++j) asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(v[g * 8 + j]) : "a"(acc[(g * 8 + j) / 4][j % 4]));
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    f32x2 a = {v[g * 8 + 2 * j], v[g * 8 + 2 * j + 1]};
                    asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(p[(g * 4 + j) % 16]) : "v"(a), "v"(p[(g * 4 + j + 7) % 16]));
                }
            }
        }
        if (KIND == 4) {      // 64 reads with an s_nop behind each
#pragma unroll
            for (int j = 0; j < 64; ++j) asm volatile("v_accvgpr_read_b32 %0, %1\n\ts_nop 0" : "=v"(v[j]) : "a"(acc[j / 4][j % 4]));
        }
        if (KIND == 5) {      // 64 accvgpr writes
#pragma unroll
            for (int j = 0; j < 64; ++j) asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(acc[j / 4][j % 4]) : "v"(v[j]));
        }
        if (KIND == 6) {      // 64 v_add_f32
#pragma unroll
            for (int j = 0; j < 64; ++j) asm volatile("v_add_f32 %0, %1, %2" : "=v"(v[j]) : "v"(v[(j + 17) % 64]), "v"(v[(j + 31) % 64]));
        }
    }
    float s = 0;
    for (int i = 0; i < 64; ++i) s += v[i];
    for (int i = 0; i < 16; ++i) s += p[i][0] + p[i][1] + acc[i][0];
    if (s == 12345.678f) out[threadIdx.x] = s;
}

template <int KIND>
void run(const char* name, int per) {
    float* out; hipMalloc(&out, 1 << 20);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    k<KIND><<<256, 256>>>(out, 100);
    hipEventRecord(e0);
    k<KIND><<<256, 256>>>(out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-40s %6.2f ns per instruction\n", name, ms * 1e6 / ((double)iters * per));
    hipFree(out);
}

int main() {
    setvbuf(stdout, nullptr, _IONBF, 0);
    run<0>("v_accvgpr_read_b32 x64", 64);
    run<4>("v_accvgpr_read_b32 + s_nop 0", 64);
    run<5>("v_accvgpr_write_b32 x64", 64);
    run<1>("v_mov_b32 x64", 64);
    run<6>("v_add_f32 x64", 64);
    run<2>("v_pk_add_f32 x64", 64);
    run<3>("8 reads + 4 pk_add pattern (96 instr)", 96);
    return 0;
}
