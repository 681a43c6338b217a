// Kernel-2 / stride-2 convolution and transposed convolution of V-Net read straight from the fine volume.
//
// Replaces (reference code/networks/vnet.py): nn.Conv3d(Cin, Cout, 2, stride=2) :73 (DownsamplingConvBlock) and
// nn.ConvTranspose3d(Cin, Cout, 2, stride=2) :100 (UpsamplingDeconvBlock), forward and data gradient, on the two largest
// levels (96^3 <-> 48^3 <-> 24^3).  vnet_ops.hip's form (space_to_depth + the 1x1x1 MFMA kernel) moves the fine tensor three
// times (re-layout read + write, GEMM read); the windows do not overlap, so the GEMM operand can be read in place:
//
//   down  y[co][v]       = b[co] + sum_{ci, tap} W[co][ci*8 + tap] * x[ci][2v + tap]        M = Cout, K = 8 Cin
//   up    y[co][2v + tap] = b[co] + sum_ci        W[ci][co*8 + tap] * x[ci][v]               M = 8 Cout, K = Cin
//
// (tap = dz*4 + dy*2 + dx; the data gradient of `down` is `up` with the same weight array read as [K = Cout][M = 8 Cin], the
// data gradient of `up` is `down` with [M = Cin][K = 8 Cout]: both parameter layouts serve both kernels unchanged.)
//
// v_mfma_f32_16x16x4_f32, one wave per 16 consecutive coarse voxels of a (y, x) plane.  down: the MFMA's contraction of 4
// is (dy, dx) of one (ci, dz): lane (k, n) loads x[ci][2z + dz][2y + dy][2x + dx] -- the 64 lanes cover two 128-byte row
// pieces, every fine element is read exactly once, no LDS staging of the volume.  up: the 16 accumulator rows of a tile are
// (co pair, dz, dy, dx): a lane owns the (dy, dx) quad of one (co, dz) and stores two float2 of two fine rows (128
// contiguous bytes per 16 lanes).  The weights (<= 64 KiB on these levels) sit in LDS, K-major with a row stride of
// M + 16 floats (= 16 mod 32: the two 16-lane halves of an operand read hit disjoint banks).  HBM-bound: 4.5x (down) /
// 2.6x (up) less traffic than the re-layout form.  Deterministic (fixed summation order).
#include "common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct K2Args {
    const float* x; long long x_bs;
    float* y; long long y_bs;
    const float* w;
    const float* bias;
    int N, Do, Ho, Wo;            // coarse geometry
    int tiles;                    // 16-voxel tiles per coarse (y, x) plane
    long long tasks;              // N * Do * tiles
    int accumulate;
};

extern __shared__ __attribute__((aligned(16))) float k2_lds[];

constexpr int NT = 512;

// y [N][COUT][Do][Ho][Wo] = conv_k2s2(x [N][CIN][2Do][2Ho][2Wo]);  w [COUT][CIN * 8]
template <int CIN, int COUT>
__global__ __launch_bounds__(NT) void k2s2_down_kernel(const K2Args a) {
    constexpr int K = CIN * 8, MT = COUT / 16, LDW = COUT + 16;
    float* const lw = k2_lds;                                    // [K][LDW]
    for (int i = threadIdx.x; i < COUT * K; i += NT) {
        const int m = i / K, k = i - m * K;
        lw[k * LDW + m] = a.w[i];
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int kq = lane >> 4, nl = lane & 15;
    const int dy = kq >> 1, dx = kq & 1;
    const int H = 2 * a.Ho, W = 2 * a.Wo;
    const long long HW = (long long)H * W, S = HW * 2 * a.Do, So = (long long)a.Do * a.Ho * a.Wo;
    const int plane = a.Ho * a.Wo;
    for (long long task = (long long)blockIdx.x * (NT / 64) + wave; task < a.tasks; task += (long long)gridDim.x * (NT / 64)) {
        const int tile = (int)(task % a.tiles);
        const long long r = task / a.tiles;
        const int z = (int)(r % a.Do), n = (int)(r / a.Do);
        const int v = tile * 16 + nl;
        const bool valid = v < plane;
        const int vv = valid ? v : 0;
        const int yy = vv / a.Wo, xx = vv - yy * a.Wo;
        const float* __restrict__ xp = a.x + (long long)n * a.x_bs + (2LL * z) * HW + (long long)(2 * yy + dy) * W + 2 * xx + dx;
        f32x4 acc[MT];
#pragma unroll
        for (int t = 0; t < MT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
        for (int c0 = 0; c0 < CIN; c0 += 8) {
            float b[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) b[j] = xp[(long long)(c0 + (j >> 1)) * S + (j & 1) * HW];
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const float bv = valid ? b[j] : 0.f;
                const float* __restrict__ wr = lw + (((c0 + (j >> 1)) * 8 + (j & 1) * 4 + kq) * LDW + nl);
#pragma unroll
                for (int t = 0; t < MT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(wr[t * 16], bv, acc[t], 0, 0, 0);
            }
        }
        if (valid) {
            float* __restrict__ yp = a.y + (long long)n * a.y_bs + (long long)z * plane + v;
#pragma unroll
            for (int t = 0; t < MT; ++t)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int co = t * 16 + kq * 4 + q;
                    float o = acc[t][q] + (a.bias ? a.bias[co] : 0.f);
                    float* p = yp + (long long)co * So;
                    if (a.accumulate) o += *p;
                    *p = o;
                }
        }
    }
}

// y [N][COUT][2Do][2Ho][2Wo] = conv_transpose_k2s2(x [N][CIN][Do][Ho][Wo]);  w [CIN][COUT * 8]
template <int CIN, int COUT>
__global__ __launch_bounds__(NT) void k2s2_up_kernel(const K2Args a) {
    constexpr int M = COUT * 8, MT = M / 16, LDW = M + 16, TC = 4;
    float* const lw = k2_lds;                                    // [CIN][LDW]
    for (int i = threadIdx.x; i < CIN * M; i += NT) {
        const int k = i / M, m = i - k * M;
        lw[k * LDW + m] = a.w[i];
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int kq = lane >> 4, nl = lane & 15;
    const int H = 2 * a.Ho, W = 2 * a.Wo;
    const long long HW = (long long)H * W, S = HW * 2 * a.Do, So = (long long)a.Do * a.Ho * a.Wo;
    const int plane = a.Ho * a.Wo;
    for (long long task = (long long)blockIdx.x * (NT / 64) + wave; task < a.tasks; task += (long long)gridDim.x * (NT / 64)) {
        const int tile = (int)(task % a.tiles);
        const long long r = task / a.tiles;
        const int z = (int)(r % a.Do), n = (int)(r / a.Do);
        const int v = tile * 16 + nl;
        const bool valid = v < plane;
        const int vv = valid ? v : 0;
        const int yy = vv / a.Wo, xx = vv - yy * a.Wo;
        const float* __restrict__ xp = a.x + (long long)n * a.x_bs + (long long)kq * So + (long long)z * plane + vv;
        float b[CIN / 4];
#pragma unroll
        for (int s = 0; s < CIN / 4; ++s) b[s] = xp[(long long)s * 4 * So];
        // accumulator row m = t*16 + kq*4 + q = co*8 + dz*4 + dy*2 + dx:  co = 2t + (kq >> 1), dz = kq & 1, q = (dy, dx)
        const int dz = kq & 1;
        float* __restrict__ yp = a.y + (long long)n * a.y_bs + (long long)(kq >> 1) * S + (long long)(2 * z + dz) * HW +
                                 (long long)(2 * yy) * W + 2 * xx;
#pragma unroll 1
        for (int t0 = 0; t0 < MT; t0 += TC) {      // TC tiles of accumulators at a time (the operand b stays in registers)
            f32x4 acc[TC];
#pragma unroll
            for (int t = 0; t < TC; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < CIN / 4; ++s) {
                const float bv = valid ? b[s] : 0.f;
                const float* __restrict__ wr = lw + ((s * 4 + kq) * LDW + t0 * 16 + nl);
#pragma unroll
                for (int t = 0; t < TC; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(wr[t * 16], bv, acc[t], 0, 0, 0);
            }
            if (valid) {
                float* __restrict__ yt = yp + (long long)(2 * t0) * S;
#pragma unroll
                for (int t = 0; t < TC; ++t) {
                    const float bs = a.bias ? a.bias[2 * (t0 + t) + (kq >> 1)] : 0.f;
                    float2* p0 = reinterpret_cast<float2*>(yt);
                    float2* p1 = reinterpret_cast<float2*>(yt + W);
                    float2 o0 = make_float2(acc[t][0] + bs, acc[t][1] + bs), o1 = make_float2(acc[t][2] + bs, acc[t][3] + bs);
                    if (a.accumulate) {
                        const float2 q0 = *p0, q1 = *p1;
                        o0.x += q0.x; o0.y += q0.y; o1.x += q1.x; o1.y += q1.y;
                    }
                    *p0 = o0;
                    *p1 = o1;
                    yt += 2 * S;
                }
            }
        }
    }
}

template <int CIN, int COUT, bool UP>
int launch_k2(const K2Args& a, hipStream_t stream) {
    static std::atomic<unsigned long long> done{0};
    constexpr int lds = UP ? CIN * (COUT * 8 + 16) * 4 : CIN * 8 * (COUT + 16) * 4;
    long long wgs = mis_cdiv(a.tasks, NT / 64);
    // resident workgroups: the weights are re-read per workgroup, so no more than fill the chip a few times over
    const long long cap = 256LL * (lds > 40 * 1024 ? 1 : 2) * 2;
    if (wgs > cap) wgs = cap;
    if constexpr (UP) {
        if (mis_set_lds_attr(reinterpret_cast<const void*>(&k2s2_up_kernel<CIN, COUT>), lds, done) != MIS_OK) return MIS_ERR_LAUNCH;
        hipLaunchKernelGGL((k2s2_up_kernel<CIN, COUT>), dim3((unsigned)wgs), dim3(NT), lds, stream, a);
    } else {
        if (mis_set_lds_attr(reinterpret_cast<const void*>(&k2s2_down_kernel<CIN, COUT>), lds, done) != MIS_OK) return MIS_ERR_LAUNCH;
        hipLaunchKernelGGL((k2s2_down_kernel<CIN, COUT>), dim3((unsigned)wgs), dim3(NT), lds, stream, a);
    }
    return mis_launch_status();
}

bool k2_shape_ok(int N, int Do, int Ho, int Wo) {
    return N > 0 && Do > 0 && Ho > 0 && Wo > 0 && (long long)Do * Ho * Wo * 8 < (1LL << 31) && (long long)Ho * Wo >= 256;
}

}  // namespace

// 1 when mis_conv_k2s2_down / _up serve (Cin, Cout) on a coarse volume of Do x Ho x Wo (else: mis_space_to_depth2 + the
// 1x1x1 convolution).  The instantiations are V-Net's two largest levels (and their data gradients).
extern "C" int mis_conv_k2s2_eligible(int Cin, int Cout, int Do, int Ho, int Wo, int up) {
    if (!k2_shape_ok(1, Do, Ho, Wo)) return 0;
    if (up) return (Cin == 32 && Cout == 16) || (Cin == 64 && Cout == 32);
    return (Cin == 16 && Cout == 32) || (Cin == 32 && Cout == 64);
}

// y [N][Cout][Do][Ho][Wo] (+)= b + Conv3d(k = 2, stride = 2)(x [N][Cin][2Do][2Ho][2Wo]);  w [Cout][Cin][2][2][2]
// (also: the data gradient of ConvTranspose3d, w = its [Cin_t = Cout here][Cout_t * 8] parameter).  bias may be NULL.
extern "C" int mis_conv_k2s2_down(const float* x, long long x_bs, const float* w, const float* bias, float* y, long long y_bs,
                                  int N, int Cin, int Cout, int Do, int Ho, int Wo, int accumulate, hipStream_t stream) {
    if (!x || !w || !y || !k2_shape_ok(N, Do, Ho, Wo)) return MIS_ERR_ARG;
    if (!mis_conv_k2s2_eligible(Cin, Cout, Do, Ho, Wo, 0)) return MIS_ERR_UNSUPPORTED;
    if (x_bs < (long long)Cin * Do * Ho * Wo * 8 || y_bs < (long long)Cout * Do * Ho * Wo) return MIS_ERR_ARG;
    K2Args a{x, x_bs, y, y_bs, w, bias, N, Do, Ho, Wo, (int)mis_cdiv((long long)Ho * Wo, 16), 0, accumulate};
    a.tasks = (long long)N * Do * a.tiles;
    if (Cin == 16) return launch_k2<16, 32, false>(a, stream);
    return launch_k2<32, 64, false>(a, stream);
}

// y [N][Cout][2Do][2Ho][2Wo] (+)= b + ConvTranspose3d(k = 2, stride = 2)(x [N][Cin][Do][Ho][Wo]);  w [Cin][Cout][2][2][2]
// (also: the data gradient of Conv3d(k2s2), w = its [Cout_c = Cin here][Cin_c * 8] parameter).  y, y_bs 8-byte aligned.
extern "C" int mis_conv_k2s2_up(const float* x, long long x_bs, const float* w, const float* bias, float* y, long long y_bs,
                                int N, int Cin, int Cout, int Do, int Ho, int Wo, int accumulate, hipStream_t stream) {
    if (!x || !w || !y || !k2_shape_ok(N, Do, Ho, Wo)) return MIS_ERR_ARG;
    if (!mis_conv_k2s2_eligible(Cin, Cout, Do, Ho, Wo, 1)) return MIS_ERR_UNSUPPORTED;
    if (((uintptr_t)y & 7) || (y_bs & 1)) return MIS_ERR_UNSUPPORTED;
    if (x_bs < (long long)Cin * Do * Ho * Wo || y_bs < (long long)Cout * Do * Ho * Wo * 8) return MIS_ERR_ARG;
    K2Args a{x, x_bs, y, y_bs, w, bias, N, Do, Ho, Wo, (int)mis_cdiv((long long)Ho * Wo, 16), 0, accumulate};
    a.tasks = (long long)N * Do * a.tiles;
    if (Cin == 32) return launch_k2<32, 16, true>(a, stream);
    return launch_k2<64, 32, true>(a, stream);
}
