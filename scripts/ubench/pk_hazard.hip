// Do packed fp32 adds with per-half modifiers compute the right thing beside a foreign wave's bf16 MFMAs on the same SIMD?
// Victim kernel: every form below on lane-dependent inputs, results checked against unpacked VALU arithmetic, mismatches counted.
// Neighbour: v_mfma_f32_16x16x32_bf16 back to back on a second stream (or nothing).
// Forms 28-30 (round 6): the op_sel_hi-only patterns that librccl.so's gfx950 kernels (168 x v_pk_fma_f32 op_sel_hi:[0,1,1]) and this
// library's own compiler-packed code contain: the LOW lane reads low halves only.
// hipcc --offload-arch=gfx950 -O3 -w pk_hazard.hip -o pk_hazard.bin && ./pk_hazard.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void spin(int iters, float* sink, int kind) {
    f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    u32x4 ua = {threadIdx.x, 1u, 2u, 3u}, ub = {5u, 6u, threadIdx.x, 8u};
    float a = threadIdx.x, b = 1.5f;
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            if (kind == 1) acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, ua), __builtin_bit_cast(bf16x8, ub), acc[m], 0, 0, 0);
            else acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[m], 0, 0, 0);
        }
    if (acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3] == 12345.678f) sink[threadIdx.x] = 1.f;
}

#define FORMS(X) \
    X(0, "v_pk_add_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,0]") \
    X(1, "v_pk_add_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,1]") \
    X(2, "v_pk_add_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[1,0]") \
    X(3, "v_pk_add_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[1,1]") \
    X(4, "v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[0,0]") \
    X(5, "v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[0,1]") \
    X(6, "v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]") \
    X(7, "v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1]") \
    X(8, "v_pk_add_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[0,0]") \
    X(9, "v_pk_add_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[0,1]") \
    X(10, "v_pk_add_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[1,0]") \
    X(11, "v_pk_add_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[1,1]") \
    X(12, "v_pk_add_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[0,0]") \
    X(13, "v_pk_add_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[0,1]") \
    X(14, "v_pk_add_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,0]") \
    X(15, "v_pk_add_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,1]") \
    X(16, "v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[1,1,1]") \
    X(17, "v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,1] op_sel_hi:[1,1,1]") \
    X(18, "v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,1,1]") \
    X(19, "v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,1,1]") \
    X(20, "v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,1] op_sel_hi:[1,1,1]") \
    X(21, "v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,1] op_sel_hi:[1,1,1]") \
    X(22, "v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,1] op_sel_hi:[1,1,1]") \
    X(23, "v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1]") \
    X(24, "v_pk_mul_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[1,1]") \
    X(25, "v_pk_add_f32 %0, %1, %1 op_sel:[0,1] op_sel_hi:[0,1] neg_hi:[0,1]") \
    X(26, "v_pk_add_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[1,1] neg_lo:[1,0] neg_hi:[0,1]") \
    X(27, "v_pk_add_f32 %0, %1, %2 op_sel_hi:[1,0] neg_lo:[0,1]") \
    X(28, "v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[0,1,1]") \
    X(29, "v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,1,0]") \
    X(30, "v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]")

template <int FORM>
__device__ __forceinline__ f32x2 form(f32x2 a, f32x2 b) {
    const f32x2 c = {a[1] * 0.5f + 1.f, b[0] - 0.25f};      // third source of the fma forms: another register pair
    f32x2 r = {0.f, 0.f};
#define X(N, STR) if (FORM == N) asm volatile(STR : "=v"(r) : "v"(a), "v"(b), "v"(c));
    FORMS(X)
#undef X
    return r;
}

template <int FORM>
__global__ __launch_bounds__(256) void victim(int iters, unsigned* out, unsigned long long* bad) {
    // checksum of the results over many different inputs; compared between a quiet and a busy run by the host
    const unsigned lane = threadIdx.x + blockIdx.x * 256;
    f32x2 a = {(float)(lane % 97) * 0.25f + 1.f, (float)(lane % 89) * 0.5f - 3.f};
    f32x2 b = {(float)(lane % 83) * 0.125f - 2.f, (float)(lane % 79) * 0.75f + 0.5f};
    unsigned h = 0;
    unsigned long long nb = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const f32x2 r = form<FORM>(a, b);
            h = h * 31u + __float_as_uint(r[0]) + 7u * __float_as_uint(r[1]);
            a[0] += 0.5f; a[1] -= 0.25f; b[0] += 0.125f; b[1] += 1.f;
            if (a[0] > 1e3f) { a[0] = 1.f; a[1] = -3.f; b[0] = -2.f; b[1] = 0.5f; }
        }
    }
    out[lane] = h;
    if (nb) atomicAdd(bad, nb);
}

template <int FORM>
void run(const char* name) {
    const int blocks = 2048, iters = 20000;
    unsigned *o0, *o1; unsigned long long* bad; float* sink;
    hipMalloc(&o0, blocks * 256 * 4); hipMalloc(&o1, blocks * 256 * 4); hipMalloc(&bad, 8); hipMalloc(&sink, 4096);
    hipStream_t s1, s2; hipStreamCreate(&s1); hipStreamCreate(&s2);
    unsigned* h0 = (unsigned*)malloc(blocks * 256 * 4); unsigned* h1 = (unsigned*)malloc(blocks * 256 * 4);
    hipMemset(bad, 0, 8);
    victim<FORM><<<blocks, 256, 0, s1>>>(iters, o0, bad);
    hipDeviceSynchronize();
    unsigned long long b0 = 0; hipMemcpy(&b0, bad, 8, hipMemcpyDeviceToHost);
    hipMemcpy(h0, o0, blocks * 256 * 4, hipMemcpyDeviceToHost);
    for (int kind = 1; kind <= 2; ++kind) {
        hipMemset(bad, 0, 8);
        spin<<<8192, 256, 0, s2>>>(40000, sink, kind);
        victim<FORM><<<blocks, 256, 0, s1>>>(iters, o1, bad);
        hipDeviceSynchronize();
        unsigned long long b1 = 0; hipMemcpy(&b1, bad, 8, hipMemcpyDeviceToHost);
        hipMemcpy(h1, o1, blocks * 256 * 4, hipMemcpyDeviceToHost);
        long diff = 0;
        for (int i = 0; i < blocks * 256; ++i) diff += h0[i] != h1[i];
        if (kind == 1 || diff) printf("%-92s beside %s MFMA: %ld of %d lanes differ from the quiet run\n", name, kind == 1 ? "bf16" : "fp32", diff, blocks * 256);
    }
}

int main() {
    setvbuf(stdout, nullptr, _IONBF, 0);
#define X(N, STR) run<N>(STR);
    FORMS(X)
#undef X
    return 0;
}
