// Shared device/host helpers for the gfx950 (MI355X, CDNA4) kernels of the
// Mean-Teacher training-step hot path.  gfx950 only: wave = 64 lanes, 256 CUs
// in 8 XCDs, 160 KiB LDS per CU, fp32-input MFMA (v_mfma_f32_16x16x4_f32).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define MIS_OK 0
#define MIS_ERR_ARG (-1)          // bad pointer / size / stride
#define MIS_ERR_UNSUPPORTED (-2)  // shape the kernel family does not cover
#define MIS_ERR_LAUNCH (-3)       // hipGetLastError() after launch
#define MIS_ERR_WORKSPACE (-4)    // caller workspace too small

#define MIS_NUM_XCD 8
#define MIS_WAVE 64

typedef float f32x4 __attribute__((ext_vector_type(4)));

// ---- "bf16x3" split-precision products (gemm.hip has the full story): an fp32 value is cut EXACTLY into three bf16 pieces by
// truncation (x = h + m + l, 8 + 8 + 8 significant bits, both subtractions exact) and a product of two fp32 values is the sum of
// the six piece products hh + hm + mh + hl + lh + mm on v_mfma_f32_16x16x32_bf16 with fp32 accumulation (ml + lm + ll, below
// 2^-24 |a||b|, are dropped).  A MisBf3 holds one matrix operand of that instruction per piece: 8 contraction elements per lane.
typedef __bf16 mis_bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned mis_u32x4 __attribute__((ext_vector_type(4)));
struct MisBf3 { mis_u32x4 h, m, l; };

__device__ __forceinline__ void mis_bf3_split_pair(float x0, float x1, unsigned& h, unsigned& m, unsigned& l) {
    const unsigned u0 = __float_as_uint(x0), u1 = __float_as_uint(x1);
    h = __builtin_amdgcn_perm(u1, u0, 0x07060302u);                   // {hi16(x1), hi16(x0)}: element 0 in the low half
    const float r0 = x0 - __uint_as_float(u0 & 0xFFFF0000u), r1 = x1 - __uint_as_float(u1 & 0xFFFF0000u);
    const unsigned v0 = __float_as_uint(r0), v1 = __float_as_uint(r1);
    m = __builtin_amdgcn_perm(v1, v0, 0x07060302u);
    const float t0 = r0 - __uint_as_float(v0 & 0xFFFF0000u), t1 = r1 - __uint_as_float(v1 & 0xFFFF0000u);
    l = __builtin_amdgcn_perm(__float_as_uint(t1), __float_as_uint(t0), 0x07060302u);
}

// operand from the lane's 8 contraction elements e0 .. e7
__device__ __forceinline__ MisBf3 mis_bf3_from8(float e0, float e1, float e2, float e3, float e4, float e5, float e6, float e7) {
    MisBf3 r;
    unsigned h, m, l;
    mis_bf3_split_pair(e0, e1, h, m, l); r.h[0] = h; r.m[0] = m; r.l[0] = l;
    mis_bf3_split_pair(e2, e3, h, m, l); r.h[1] = h; r.m[1] = m; r.l[1] = l;
    mis_bf3_split_pair(e4, e5, h, m, l); r.h[2] = h; r.m[2] = m; r.l[2] = l;
    mis_bf3_split_pair(e6, e7, h, m, l); r.h[3] = h; r.m[3] = m; r.l[3] = l;
    return r;
}

__device__ __forceinline__ f32x4 mis_bf3_mfma1(const mis_u32x4& a, const mis_u32x4& b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(mis_bf16x8, a), __builtin_bit_cast(mis_bf16x8, b), c, 0, 0, 0);
}

// c += A . B over the 32 contraction elements of the two operands (smallest piece products first)
__device__ __forceinline__ f32x4 mis_bf3_dot(const MisBf3& a, const MisBf3& b, f32x4 c) {
    c = mis_bf3_mfma1(a.l, b.h, c);
    c = mis_bf3_mfma1(a.h, b.l, c);
    c = mis_bf3_mfma1(a.m, b.m, c);
    c = mis_bf3_mfma1(a.m, b.h, c);
    c = mis_bf3_mfma1(a.h, b.m, c);
    c = mis_bf3_mfma1(a.h, b.h, c);
    return c;
}

// the split-precision mask of mis_gemm_set_split_precision (gemm.hip): bit 0 NT GEMMs, bit 1 TN dW GEMM, bit 2 window attention
int mis_split_precision_mask();

static inline int mis_launch_status() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? MIS_OK : MIS_ERR_LAUNCH;
}

static inline long long mis_cdiv(long long a, long long b) { return (a + b - 1) / b; }

// Opt-in for > 64 KiB of dynamic LDS: a per-function, per-device attribute.  `done` is one bit per device ordinal
// (function-local static of the calling template instantiation); atomic, so any host thread and any device of
// the process may launch first.  Idempotent, never changes results.
#include <atomic>
static inline int mis_set_lds_attr(const void* fn, int lds_bytes, std::atomic<unsigned long long>& done) {
    if (lds_bytes <= 64 * 1024) return MIS_OK;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return MIS_ERR_LAUNCH;
    const unsigned long long bit = 1ull << (dev & 63);
    if (done.load(std::memory_order_acquire) & bit) return MIS_OK;
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes) != hipSuccess)
        return MIS_ERR_LAUNCH;
    done.fetch_or(bit, std::memory_order_release);
    return MIS_OK;
}

// XCD-aware block remap (workgroup b is observed to run on XCD b % 8; each XCD
// has a private L2).  Gives every XCD a contiguous run of logical tiles so
// neighbouring tiles (shared halos / shared weights) hit the same L2.  The
// grid is padded to a multiple of 8; logical ids >= n must exit.  Affects
// speed only, never results.
__device__ __forceinline__ unsigned mis_xcd_remap(unsigned b, unsigned n_padded) {
    const unsigned per = n_padded / MIS_NUM_XCD;
    return (b % MIS_NUM_XCD) * per + (b / MIS_NUM_XCD);
}

// ---- LDS-DMA (global -> LDS without passing through VGPRs) used by the conv kernels ----
namespace mis_dma {

typedef int i32x4 __attribute__((ext_vector_type(4)));

constexpr unsigned OOB = 0x80000000u;   // byte offset beyond every descriptor's num_records (< 2^31)

// raw buffer descriptor (stride 0, range-checked against `bytes`), built from wave-uniform values
__device__ __forceinline__ i32x4 make_rsrc(const void* p, unsigned bytes) {
    const unsigned long long b = (unsigned long long)p;
    i32x4 r;
    r[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)b);
    r[1] = __builtin_amdgcn_readfirstlane((int)(unsigned)(b >> 32)) & 0xffff;
    r[2] = __builtin_amdgcn_readfirstlane((int)bytes);
    r[3] = 0x00020000;
    return r;
}

__device__ __forceinline__ unsigned lds_addr(const float* p) {
    return (unsigned)(uintptr_t)((__attribute__((address_space(3))) const float*)p);
}

// Lane l's dword (16 bytes for _x4) at descriptor byte offset `voff` lands at LDS byte address
// lds_byte + l*4 (l*16); out-of-range offsets deliver zeros.  Issued as asm so that hipcc does not count
// it: with the builtin form hipcc waits vmcnt(0) before the first ds_read that follows (it cannot prove
// the two stage buffers disjoint), which serialises copy and MFMA again.  Completion is waited for
// explicitly (dma_wait) before the barrier that publishes the stage buffer.  M0 is saved/restored in the
// same statement (hipcc owns M0).
__device__ __forceinline__ void dma_dword(unsigned lds_byte, unsigned voff, i32x4 rsrc) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\t"
                 "buffer_load_dword %2, %3, 0 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "s"(lds_byte), "v"(voff), "s"(rsrc) : "memory");
}
__device__ __forceinline__ void dma_dwordx4(unsigned lds_byte, unsigned voff, i32x4 rsrc) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\t"
                 "buffer_load_dwordx4 %2, %3, 0 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "s"(lds_byte), "v"(voff), "s"(rsrc) : "memory");
}
__device__ __forceinline__ void dma_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

}  // namespace mis_dma

// Sum over the LPR (32 or 64) lanes of a row group, result in every lane.  The first four steps are DPP modifiers of the
// add (quad_perm 1032 / 2301, row_half_mirror, row_mirror: vector-ALU rate), the step across the two 16-lane rows is one
// ds_swizzle (xor 16), the one across the wave's halves one ds_bpermute -- __shfl_xor is a ds_bpermute per step on gfx9
// (5 per sum: the fused LayerNorm + head forward issues 6 sums per row and ran at 2.1 TB/s on them).
__device__ __forceinline__ float mis_dpp_f(float v, int ctrl) {
    switch (ctrl) {      // the control word is an immediate
        case 0: return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, false));
        case 1: return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, false));
        case 2: return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, false));
        default: return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, false));
    }
}
template <int LPR>
__device__ __forceinline__ float mis_group_sum(float v) {
    static_assert(LPR == 32 || LPR == 64, "row groups of 32 or 64 lanes");
    v += mis_dpp_f(v, 0);
    v += mis_dpp_f(v, 1);
    v += mis_dpp_f(v, 2);
    v += mis_dpp_f(v, 3);
    v += __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, v), (0x10 << 10) | 0x1F));
    if (LPR == 64) v += __shfl_xor(v, 32, 64);
    return v;
}

// Sum over the 8 lanes of a row group (the first three steps of mis_group_sum), result in every lane.
__device__ __forceinline__ float mis_sum8(float v) {
    v += mis_dpp_f(v, 0);
    v += mis_dpp_f(v, 1);
    v += mis_dpp_f(v, 2);
    return v;
}

// LayerNorm over C = 96 channels + NC <= 4 head dot products for ONE token row held by 8 lanes: lane l8 owns the float4 groups
// l8, l8 + 8, l8 + 16 of the row (12 channels; no idle lane, 3-step reductions -- the 32-lane form of ln_head_fwd_kernel spends
// 6 x 6 cross-lane steps per row on 24 active lanes and was the bound of that kernel, not its bytes).  Shared by
// ln96_head_fwd_kernel (token_ops.hip) and the EP_LNHEAD epilogue of gemm_nt_kernel (gemm.hip): same expression, same bits.
template <int NCM>
__device__ __forceinline__ void mis_ln96_head_row(const float4 (&v)[3], const float4 (&g)[3], const float4 (&bt)[3],
                                                  const float4 (&wv)[NCM][3], float eps, float& mu, float& rs,
                                                  float (&pl)[NCM]) {
    float s = 0.f;
#pragma unroll
    for (int q = 0; q < 3; ++q) s += (v[q].x + v[q].y) + (v[q].z + v[q].w);
    mu = mis_sum8(s) / 96.f;
    float4 a[3];
    float ss = 0.f;
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        a[q] = make_float4(v[q].x - mu, v[q].y - mu, v[q].z - mu, v[q].w - mu);
        ss += (a[q].x * a[q].x + a[q].y * a[q].y) + (a[q].z * a[q].z + a[q].w * a[q].w);
    }
    rs = 1.f / sqrtf(mis_sum8(ss) / 96.f + eps);
    float4 y[3];
#pragma unroll
    for (int q = 0; q < 3; ++q)
        y[q] = make_float4(a[q].x * rs * g[q].x + bt[q].x, a[q].y * rs * g[q].y + bt[q].y, a[q].z * rs * g[q].z + bt[q].z,
                           a[q].w * rs * g[q].w + bt[q].w);
#pragma unroll
    for (int n = 0; n < NCM; ++n) {
        float p = 0.f;
#pragma unroll
        for (int q = 0; q < 3; ++q) p += (y[q].x * wv[n][q].x + y[q].y * wv[n][q].y) + (y[q].z * wv[n][q].z + y[q].w * wv[n][q].w);
        pl[n] = mis_sum8(p);
    }
}

__device__ __forceinline__ float mis_wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

__device__ __forceinline__ double mis_wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// Block-wide sum of NV values per thread (256-thread blocks); result valid in
// thread 0.  `red` must hold 4*NV floats.  Fixed tree => run-to-run deterministic.
template <int NV>
__device__ __forceinline__ void mis_block_sum(float (&v)[NV], float* red) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i] = mis_wave_sum(v[i]);
    __syncthreads();
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i < NV; ++i) red[wave * NV + i] = v[i];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const int nw = blockDim.x >> 6;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            float s = 0.f;
            for (int w = 0; w < nw; ++w) s += red[w * NV + i];
            v[i] = s;
        }
    }
}

// Philox4x32-10 counter RNG (stateless: every kernel re-derives its stream from
// (seed, offset, element index), so dropout masks are recomputed in backward
// instead of being stored).
__device__ __forceinline__ void mis_philox4(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                            uint32_t k0, uint32_t k1, uint32_t out[4]) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        const uint32_t n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        const uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// uniform in [0,1) from 32 random bits (24-bit mantissa path, never returns 1)
__device__ __forceinline__ float mis_u01(uint32_t r) { return (float)(r >> 8) * (1.0f / 16777216.0f); }

// Device-resident step state: lets a captured hipGraph replay with fresh
// dropout masks, learning rate, EMA alpha and consistency weight without
// re-capture or a per-step host->device copy.  Advanced by mis_step_advance().
struct MisStepState {
    uint64_t seed;       // Philox key
    uint64_t offset;     // RNG offset, +1 per step
    int64_t iter_num;    // reference's iter_num (train_mean_teacher_2D.py:198)
    float lr;            // SGD learning rate of this step
    float ema_alpha;     // EMA decay of this step
    float cons_weight;   // consistency weight of this step (0 while gated off)
    float cons_gate;     // 1 if the consistency term is live this step, else 0
};

// Exact-form (erf) GELU, nn.GELU() of the reference (swin_transformer_unet_skip_expand_decoder_sys.py:10,15):
//   gelu(x) = x Phi(x),  gelu'(x) = Phi(x) + x u / sqrt(2 pi),  u = exp(-x^2 / 2).
// Phi(-a) = u(a) P(t), t = 1 / (1 + 0.3 a): the rational-argument form of erfc (Abramowitz & Stegun 7.1.26) with a degree-8
// polynomial fitted here for RELATIVE accuracy over a in [0, 13] (so the negative tail of x Phi(x) keeps its digits):
// |Phi error| <= 2e-7 in fp32 (two ulp of 1), |gelu error| <= 4e-7 at |x| = 13.  2 transcendental + 14 plain VALU
// instructions; libm's erff + expf are ~70, which made the fused GELU epilogues of the NT GEMM VALU-bound (a wave's VALU
// work does not overlap its own MFMAs).
__device__ __forceinline__ void mis_gelu_parts(float x, float& cdf, float& u) {
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3f, fabsf(x), 1.f));
    float p = 3.664988946e-02f;
    p = fmaf(p, t, -1.611761927e-01f);
    p = fmaf(p, t, 2.130432014e-01f);
    p = fmaf(p, t, -5.969779766e-02f);
    p = fmaf(p, t, 1.316696142e-01f);
    p = fmaf(p, t, 9.899286856e-02f);
    p = fmaf(p, t, 1.208983674e-01f);
    p = fmaf(p, t, 1.196200560e-01f);
    u = __builtin_amdgcn_exp2f(x * x * -0.72134752044448170f);      // exp(-x^2 / 2)
    const float e = p * t * u;
    cdf = x < 0.f ? e : 1.f - e;
}
__device__ __forceinline__ float mis_gelu(float x) {
    float cdf, u;
    mis_gelu_parts(x, cdf, u);
    return x * cdf;
}
__device__ __forceinline__ float mis_gelu_grad(float x) {
    float cdf, u;
    mis_gelu_parts(x, cdf, u);
    return fmaf(x * 0.3989422804014327f, u, cdf);
}
