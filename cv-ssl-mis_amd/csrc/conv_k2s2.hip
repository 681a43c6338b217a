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
typedef const f32x4 __attribute__((address_space(1)))* gfloat4p;      // global (not flat) loads

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


// ------------------------------------------------------------------------------------------------ weight gradient
// dW[cc][cf*8 + tap] = sum_{n, v} c[n][cc][v] * f[n][cf][2v + tap]      c coarse [N][CC][Do][Ho][Wo], f fine [N][CF][2Do][2Ho][2Wo]
// -- Conv3d(k2s2): c = dy, f = x, dW = the [Cout][Cin*8] parameter; ConvTranspose3d(k2s2): c = x, f = dy, dW = [Cin][Cout*8]:
// both parameter layouts, no space_to_depth view of the fine tensor, no transpose of the result.
// MFMA: M = cc, N = (cf pair, tap), contraction = 4 coarse voxels.  A task is a strip of R coarse rows taken by all 8 waves
// of a workgroup; wave w owns the N tiles of cf in [w*CF/8, (w+1)*CF/8) (its rows of f are read by nobody else; the rows of
// c are shared through L1).  The strip is walked in tiles of 16 voxels = 4 lane groups of 4 consecutive voxels (GPR groups
// side by side in x, 4/GPR rows: 16 x 1 when Wo % 16 == 0, 8 x 2 otherwise): a lane loads a float4 of c and the 8 fine
// floats under its 4 voxels as two float4, of which it keeps its dx parity -- pointer bumps only, no index arithmetic in
// the loop.  Accumulators stay in registers over all strips of the workgroup; one partial dW per workgroup, summed in
// workgroup order by k2s2_wgrad_reduce_kernel (deterministic).
struct K2WgArgs {
    const float* c; long long c_bs;
    const float* f; long long f_bs;
    float* part;                  // [gridDim.x][CC * CF * 8]
    int N, Do, Ho, Wo;
    int sb;                       // strips per task
    int chunks;                   // tasks per (n, z) plane: (Ho / R) / sb
    int tasks;                    // N * Do * chunks
};

// GPR: lane groups side by side in x (4: tiles of 16 x 1 voxels, 2: 8 x 2)
template <int CF, int CC, int XU, int GPR>
__global__ __launch_bounds__(NT) void k2s2_wgrad_kernel(const K2WgArgs a) {
    constexpr int MT = CC / 16, NTW = CF / 16, R = 4 / GPR, XSTEP = 4 * GPR;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int kq = lane >> 4, nl = lane & 15;
    const int cfl = nl >> 3, dz = (nl >> 2) & 1, dy = (nl >> 1) & 1;
    const bool odd = nl & 1;
    const int W = 2 * a.Wo;
    const long long HW = 4LL * a.Ho * a.Wo, S = HW * 2 * a.Do, So = (long long)a.Do * a.Ho * a.Wo;
    const int plane = a.Ho * a.Wo;
    const int kr = kq / GPR, kx = kq % GPR;
    const int xtiles = a.Wo / XSTEP;
    const long long lane_c = (long long)nl * So + kr * a.Wo + kx * 4;
    const long long lane_f = (long long)(wave * NTW * 2 + cfl) * S + (long long)dz * HW + (long long)(2 * kr + dy) * W + kx * 8;
    const int c_next = (R * a.Wo - xtiles * XSTEP) / 4;         // from the end of a strip's x walk to the next strip, in float4
    const int f_next = (2 * R * W - xtiles * 2 * XSTEP) / 4;
    f32x4 acc[MT][NTW];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NTW; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int task = blockIdx.x; task < a.tasks; task += gridDim.x) {
        const int chunk = task % a.chunks, r = task / a.chunks;
        const int z = r % a.Do, n = r / a.Do;
        gfloat4p cp[MT], fp[NTW];
#pragma unroll
        for (int i = 0; i < MT; ++i)
            cp[i] = (gfloat4p)(a.c + (long long)n * a.c_bs + (long long)z * plane + (long long)chunk * a.sb * R * a.Wo + lane_c + (long long)i * 16 * So);
#pragma unroll
        for (int j = 0; j < NTW; ++j)
            fp[j] = (gfloat4p)(a.f + (long long)n * a.f_bs + 2LL * z * HW + (long long)(2 * chunk * a.sb * R) * W + lane_f + (long long)j * 2 * S);
#pragma unroll 1
        for (int s = 0; s < a.sb; ++s) {
#pragma unroll 1
            for (int x0 = 0; x0 < xtiles; x0 += XU) {
                f32x4 av[XU][MT], b0[XU][NTW], b1[XU][NTW];
#pragma unroll
                for (int u = 0; u < XU; ++u) {      // xtiles % XU == 0 (host): no guards, all loads of the step in flight
#pragma unroll
                    for (int i = 0; i < MT; ++i) av[u][i] = cp[i][u * GPR];
#pragma unroll
                    for (int j = 0; j < NTW; ++j) {
                        b0[u][j] = fp[j][u * 2 * GPR];
                        b1[u][j] = fp[j][u * 2 * GPR + 1];
                    }
                }
#pragma unroll
                for (int u = 0; u < XU; ++u) {
                    {
                        float bs[NTW][4];
#pragma unroll
                        for (int j = 0; j < NTW; ++j) {
                            // values in registers first: hipcc otherwise turns the parity select into an indexed re-load
                            // through scratch / LDS
                            float e0 = b0[u][j].x, e1 = b0[u][j].y, e2 = b0[u][j].z, e3 = b0[u][j].w;
                            float e4 = b1[u][j].x, e5 = b1[u][j].y, e6 = b1[u][j].z, e7 = b1[u][j].w;
                            asm volatile("" : "+v"(e0), "+v"(e1), "+v"(e2), "+v"(e3), "+v"(e4), "+v"(e5), "+v"(e6), "+v"(e7));
                            bs[j][0] = odd ? e1 : e0; bs[j][1] = odd ? e3 : e2;
                            bs[j][2] = odd ? e5 : e4; bs[j][3] = odd ? e7 : e6;
                        }
#pragma unroll
                        for (int q = 0; q < 4; ++q)
#pragma unroll
                            for (int i = 0; i < MT; ++i) {
                                const float as = q == 0 ? av[u][i].x : q == 1 ? av[u][i].y : q == 2 ? av[u][i].z : av[u][i].w;
#pragma unroll
                                for (int j = 0; j < NTW; ++j)
                                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(as, bs[j][q], acc[i][j], 0, 0, 0);
                            }
                    }
                }
#pragma unroll
                for (int i = 0; i < MT; ++i) cp[i] += XU * GPR;
#pragma unroll
                for (int j = 0; j < NTW; ++j) fp[j] += XU * 2 * GPR;
            }
#pragma unroll
            for (int i = 0; i < MT; ++i) cp[i] += c_next;
#pragma unroll
            for (int j = 0; j < NTW; ++j) fp[j] += f_next;
        }
    }
    // accumulator row = cc = i*16 + kq*4 + q, column = nl of N tile (wave*NTW + j): dW[cc][(wave*NTW + j)*16 + nl]
    float* __restrict__ out = a.part + (long long)blockIdx.x * (CC * CF * 8);
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NTW; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q)
                out[(i * 16 + kq * 4 + q) * (CF * 8) + (wave * NTW + j) * 16 + nl] = acc[i][j][q];
}

// dw[e] (+)= sum_g part[g][e], fixed order: block = 32 elements x 8 partial lanes (lane gl sums g = gl, gl + 8, ...), then a
// fixed-order LDS tree
__global__ __launch_bounds__(256) void k2s2_wgrad_reduce_kernel(const float* __restrict__ part, int G, int total,
                                                                float* __restrict__ dw, int accumulate) {
    __shared__ float red[256];
    const int el = threadIdx.x & 31, gl = threadIdx.x >> 5;
    const int e = blockIdx.x * 32 + el;
    float s = 0.f;
    if (e < total) {
        int g = gl;
        for (; g + 24 < G; g += 32) {
            const float v0 = part[(long long)g * total + e], v1 = part[(long long)(g + 8) * total + e];
            const float v2 = part[(long long)(g + 16) * total + e], v3 = part[(long long)(g + 24) * total + e];
            s += v0; s += v1; s += v2; s += v3;
        }
        for (; g < G; g += 8) s += part[(long long)g * total + e];
    }
    red[threadIdx.x] = s;
    __syncthreads();
    if (gl == 0 && e < total) {
#pragma unroll
        for (int j = 1; j < 8; ++j) s += red[j * 32 + el];
        dw[e] = accumulate ? dw[e] + s : s;
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

// Weight gradient of Conv3d(k2s2) / ConvTranspose3d(k2s2) from the tensors as they lie (no space_to_depth view):
//   dw[cc][cf*8 + tap] (+)= sum_{n, v} coarse[n][cc][v] * fine[n][cf][2v + tap]
// Conv3d: coarse = dy (CC = Cout), fine = x (CF = Cin), dw = the [Cout][Cin][2][2][2] parameter gradient;
// ConvTranspose3d: coarse = x (CC = Cin), fine = dy (CF = Cout), dw = the [Cin][Cout][2][2][2] parameter gradient.
// (CF, CC) in {(16, 32), (32, 64)} and Wo % 4 == 0 (mis_conv_k2s2_wgrad_eligible).  Deterministic.
// workspace >= mis_conv_k2s2_wgrad_workspace_bytes(CF, CC).
extern "C" int mis_conv_k2s2_wgrad_eligible(int CF, int CC, int Do, int Ho, int Wo) {
    if (!k2_shape_ok(1, Do, Ho, Wo)) return 0;
    if (!(Wo % 16 == 0 || (Wo % 8 == 0 && Ho % 2 == 0))) return 0;
    return (CF == 16 && CC == 32) || (CF == 32 && CC == 64);
}

extern "C" long long mis_conv_k2s2_wgrad_workspace_bytes(int CF, int CC) {
    if (CF <= 0 || CC <= 0) return MIS_ERR_ARG;
    return 512LL * CF * CC * 8 * 4;
}

extern "C" int mis_conv_k2s2_wgrad(const float* coarse, long long c_bs, const float* fine, long long f_bs, float* dw, int N,
                                   int CF, int CC, int Do, int Ho, int Wo, int accumulate, float* workspace,
                                   long long workspace_bytes, hipStream_t stream) {
    if (!coarse || !fine || !dw || !workspace || !k2_shape_ok(N, Do, Ho, Wo)) return MIS_ERR_ARG;
    if (!mis_conv_k2s2_wgrad_eligible(CF, CC, Do, Ho, Wo)) return MIS_ERR_UNSUPPORTED;
    if (((uintptr_t)coarse & 15) || ((uintptr_t)fine & 15) || (c_bs & 3) || (f_bs & 3)) return MIS_ERR_UNSUPPORTED;
    if (c_bs < (long long)CC * Do * Ho * Wo || f_bs < (long long)CF * Do * Ho * Wo * 8) return MIS_ERR_ARG;
    if (workspace_bytes < mis_conv_k2s2_wgrad_workspace_bytes(CF, CC)) return MIS_ERR_WORKSPACE;
    const int gpr = Wo % 16 == 0 ? 4 : 2;
    const int strips = Ho / (4 / gpr);
    // strips per task: a divisor of the plane's strips that leaves >= ~3 tasks per workgroup (the per-task index arithmetic
    // shares the issue pipe with the MFMAs: it has to be rare)
    int sb = 1;
    for (long long want = 1536; want >= 768 && sb == 1; want -= 768)
        for (int d = 1; d <= strips; ++d)
            if (strips % d == 0 && (long long)N * Do * (strips / d) >= want) sb = d;
    const int chunks = strips / sb;
    if ((long long)N * Do * chunks > 0x7fffffffLL) return MIS_ERR_UNSUPPORTED;
    K2WgArgs a{coarse, c_bs, fine, f_bs, workspace, N, Do, Ho, Wo, sb, chunks, N * Do * chunks};
    const int rounds = (a.tasks + 511) / 512;
    const int G = (a.tasks + rounds - 1) / rounds;      // <= 512 workgroups (two per CU), the same number of tasks each
    const int xtiles = Wo / (4 * gpr);
    const int xu = xtiles % 3 == 0 ? 3 : (xtiles % 2 == 0 ? 2 : 1);      // x tiles per step, all their loads in flight together
#define MIS_K2_WG1(CF_, CC_, XU_)                                                                                       \
    if (gpr == 4) hipLaunchKernelGGL((k2s2_wgrad_kernel<CF_, CC_, XU_, 4>), dim3((unsigned)G), dim3(NT), 0, stream, a); \
    else hipLaunchKernelGGL((k2s2_wgrad_kernel<CF_, CC_, XU_, 2>), dim3((unsigned)G), dim3(NT), 0, stream, a)
#define MIS_K2_WG(CF_, CC_)                                                                                             \
    if (xu == 3) { MIS_K2_WG1(CF_, CC_, 3); } else if (xu == 2) { MIS_K2_WG1(CF_, CC_, 2); } else { MIS_K2_WG1(CF_, CC_, 1); }
    if (CF == 16) { MIS_K2_WG(16, 32) } else { MIS_K2_WG(32, 64) }
#undef MIS_K2_WG
#undef MIS_K2_WG1
    int st = mis_launch_status();
    if (st) return st;
    const int total = CC * CF * 8;
    hipLaunchKernelGGL(k2s2_wgrad_reduce_kernel, dim3((total + 31) / 32), dim3(256), 0, stream, workspace, G, total, dw,
                       accumulate);
    return mis_launch_status();
}
