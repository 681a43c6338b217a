// Weight gradient of the FIRST convolution of the 3-D networks: Conv3d(1 -> 16, k = 3, pad = 1) on the full-resolution
// volume (reference code/networks/unet_3D.py:28 conv1 = UnetConv3(in_channels = 1, 16), utils.py:99-107;
// vnet.py:123 block_one; unetr.py encoder1).
//
// With one input channel the generic kernel (conv_wgrad.hip: N = 16 input channels per MFMA) runs 15/16 padding and
// needed 0.82 ms per step for 6 GFLOP.  Here the 27 taps take the place of the input channels:
//   D[co][tap] += sum_voxel dy[co][voxel] * x[voxel + tap]        v_mfma_f32_16x16x4_f32, K = 4 voxels
//   A[i = lane&15][k = lane>>4] = dy[co = i][voxel 4k + s]   -- the lane's own float4 of dy, component s (4 MFMA steps)
//   B[k = lane>>4][j = lane&15] = x[voxel 4k + s + offset(tap = 16 g + j)]   -- one ds_read_b32 of the haloed x tile
// dy (the only large operand, 16 channels) is read exactly once, 16 bytes per lane; x lives in LDS.  A workgroup walks
// tiles of 4 x 8 x 32 voxels (a wave per z plane); per-wave partials, summed in a fixed order by a second kernel.
#include "common.h"

namespace {

// KD = 3: tiles of 4 x 8 x 32 voxels, a wave per z plane; KD = 1 (Conv2d(1 -> 16, 3, pad 1), reference unet.py:37 in_conv of
// the 2-D UNet): tiles of 32 x 32 pixels, a wave per 8 rows, 9 taps = one 16-wide group
template <int KD>
struct WC1 {
    static constexpr int TZ = KD == 3 ? 4 : 1, TY = KD == 3 ? 8 : 32, TX = 32;
    static constexpr int HZ = TZ + KD - 1, HY = TY + 2, HX = TX + 2, HALO = HZ * HY * HX;
    static constexpr int TAPS = 9 * KD, NG = (TAPS + 15) / 16;
};

// NORM form: the conv feeds BatchNorm / InstanceNorm + (Leaky)ReLU and nobody needs the gradient at the conv's input, so
// the gradient at the conv output is only ever read here: it is formed on the load path from the gradient at the
// activation (da), the conv output (y) and the normalisation's backward sums -- mis_norm_act_bwd's apply pass (read da,
// y, write dy: 453 MB each at 96^3) and this kernel's read of dy are gone.
struct Cin1Norm {
    const float* y; long long y_bs;
    const float* mean; const float* rstd; const float* gamma; const float* beta;
    const float2* sums;          // per group (mean of dz, mean of dz*xhat): mis_norm_act_bwd_sums
    float slope;
    int per_sample;
};

struct Cin1Args {
    const float* x; long long x_bs;
    const float* dy; long long dy_bs;
    float* ws;                      // [gridDim.x * 4 waves][2][16 co][16 taps]
    int N, D, H, W;
    int tz, ty, tx, n_tiles;
};

template <int KD, bool NORM>
__global__ __launch_bounds__(256) void wgrad_cin1_kernel(const Cin1Args a, const Cin1Norm nb) {
    using C = WC1<KD>;
    constexpr int TZ = C::TZ, TY = C::TY, TX = C::TX, HY = C::HY, HX = C::HX, HALO = C::HALO, NG = C::NG;
    __shared__ float sx[HALO];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lk = lane >> 4, lj = lane & 15;
    const long long S = (long long)a.D * a.H * a.W;
    // LDS offset of the two taps this lane serves (tap 16 g + j; taps 27 .. 31 repeat tap 26 and are never stored)
    int toff[NG];
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        const int tap = g * 16 + lj < C::TAPS ? g * 16 + lj : C::TAPS - 1;
        toff[g] = ((tap / 9) * HY + (tap / 3) % 3) * HX + tap % 3;
    }
    f32x4 acc[NG];
#pragma unroll
    for (int g = 0; g < NG; ++g) acc[g] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int wz = KD == 3 ? wave : 0, wy = KD == 3 ? 0 : wave * 8;      // this wave's 8 rows of the tile
    for (int t = blockIdx.x; t < a.n_tiles; t += gridDim.x) {
        int r = t;
        const int bx = r % a.tx; r /= a.tx;
        const int by = r % a.ty; r /= a.ty;
        const int bz = r % a.tz; r /= a.tz;
        const int n = r;
        const int z0 = bz * TZ, y0 = by * TY, x0 = bx * TX;
        const float* __restrict__ xn = a.x + (long long)n * a.x_bs;
        __syncthreads();                                   // the previous tile's reads are done
        for (int e = tid; e < HALO; e += 256) {
            const int hz = e / (HY * HX), r2 = e - hz * (HY * HX), hy = r2 / HX, hx = r2 - hy * HX;
            const int gz = z0 + hz - (KD == 3 ? 1 : 0), gy = y0 + hy - 1, gx = x0 + hx - 1;
            const bool ok = (unsigned)gz < (unsigned)a.D && (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W;
            sx[e] = ok ? xn[((long long)gz * a.H + gy) * a.W + gx] : 0.f;
        }
        __syncthreads();
        // this wave: plane z0 + wave; 8 rows x 2 halves of 16 voxels; the lane's dy: channel lj, voxels 4 lk .. 4 lk + 3
        const long long voff = (long long)lj * S + ((long long)(z0 + wz) * a.H + y0 + wy) * a.W + x0 + 4 * lk;
        const float* __restrict__ dyp = a.dy + (long long)n * a.dy_bs + voff;
        const float* __restrict__ yp = NORM ? nb.y + (long long)n * nb.y_bs + voff : nullptr;
        float cm = 0.f, crs = 1.f, cga = 1.f, cbe = 0.f, s1 = 0.f, s2 = 0.f;
        if constexpr (NORM) {      // channel lj of sample n
            const int grp = nb.per_sample ? n * 16 + lj : lj;
            cm = nb.mean[grp]; crs = nb.rstd[grp];
            cga = nb.gamma ? nb.gamma[lj] : 1.f; cbe = nb.beta ? nb.beta[lj] : 0.f;
            const float2 sm = nb.sums[grp];
            s1 = sm.x; s2 = sm.y;
        }
        const float* __restrict__ sb = sx + (wz * HY + wy) * HX + 4 * lk;
#pragma unroll 4
        for (int it = 0; it < 16; ++it) {
            const int row = it >> 1, half = it & 1;
            f32x4 d4 = *reinterpret_cast<const f32x4*>(dyp + (long long)row * a.W + half * 16);
            if constexpr (NORM) {
                const f32x4 y4 = *reinterpret_cast<const f32x4*>(yp + (long long)row * a.W + half * 16);
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const float xh = (y4[s] - cm) * crs;
                    const float dz = xh * cga + cbe > 0.f ? d4[s] : d4[s] * nb.slope;
                    d4[s] = cga * crs * (dz - s1 - xh * s2);
                }
            }
            const float* __restrict__ sp = sb + row * HX + half * 16;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
#pragma unroll
                for (int g = 0; g < NG; ++g)
                    acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(d4[s], sp[toff[g] + s], acc[g], 0, 0, 0);
            }
        }
    }
    // D[row = lk * 4 + r -> co][col = lj -> tap within the group]
    float* __restrict__ out = a.ws + ((long long)blockIdx.x * 4 + wave) * (NG * 256);
#pragma unroll
    for (int g = 0; g < NG; ++g)
#pragma unroll
        for (int r = 0; r < 4; ++r) out[g * 256 + (lk * 4 + r) * 16 + lj] = acc[g][r];
}

// dw[co][tap] (+)= sum of the partials: one workgroup per output (432 of them: with 16 lanes per output and 27 workgroups
// this reduction was a latency chain of 256 dependent 2 KB-strided loads, 86 us), each thread a strided share, combined in
// a fixed order
__global__ __launch_bounds__(256) void wgrad_cin1_reduce_kernel(const float* __restrict__ ws, float* __restrict__ dw,
                                                                int parts, int taps, int ng, int accumulate) {
    __shared__ float red[4];
    const int o = blockIdx.x;
    const int co = o / taps, tap = o - co * taps;
    const float* p = ws + (tap / 16) * 256 + co * 16 + tap % 16;
    float s = 0.f;
    for (int k = threadIdx.x; k < parts; k += 256) s += p[(long long)k * (ng * 256)];
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) s += __shfl_xor(s, d, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        s = (red[0] + red[1]) + (red[2] + red[3]);
        dw[o] = accumulate ? dw[o] + s : s;
    }
}

int grid_for(long long n_tiles) { return (int)(n_tiles < 1024 ? n_tiles : 1024); }

template <int KD>
long long tiles_of(int N, int D, int H, int W) {
    return (long long)N * (D / WC1<KD>::TZ) * (H / WC1<KD>::TY) * (W / WC1<KD>::TX);
}

template <int KD>
int launch_cin1_kd(const float* x, long long x_bs, const float* dy, long long dy_bs, const Cin1Norm* nb, float* dw,
                   float* ws, int N, int D, int H, int W, int accumulate, hipStream_t stream) {
    using C = WC1<KD>;
    Cin1Args a{};
    a.x = x; a.x_bs = x_bs; a.dy = dy; a.dy_bs = dy_bs; a.ws = ws;
    a.N = N; a.D = D; a.H = H; a.W = W;
    a.tz = D / C::TZ; a.ty = H / C::TY; a.tx = W / C::TX;
    const long long n_tiles = tiles_of<KD>(N, D, H, W);
    if (n_tiles > 0x7fffffffLL) return MIS_ERR_ARG;
    a.n_tiles = (int)n_tiles;
    const int grid = grid_for(n_tiles);
    if (nb) hipLaunchKernelGGL((wgrad_cin1_kernel<KD, true>), dim3(grid), dim3(256), 0, stream, a, *nb);
    else hipLaunchKernelGGL((wgrad_cin1_kernel<KD, false>), dim3(grid), dim3(256), 0, stream, a, Cin1Norm{});
    hipLaunchKernelGGL(wgrad_cin1_reduce_kernel, dim3(16 * C::TAPS), dim3(256), 0, stream, ws, dw, grid * 4, C::TAPS, C::NG,
                       accumulate);
    return mis_launch_status();
}

}  // namespace

// 3-D: Conv3d(1 -> 16, 3, pad 1) on [N,1,D,H,W] with D % 4 == 0, H % 8 == 0, W % 32 == 0; 2-D: Conv2d(1 -> 16, 3, pad 1) on
// D == 1 with H % 32 == 0, W % 32 == 0
bool mis_wgrad_cin1_eligible(int N, int Cin, int Cout, int D, int H, int W, int kd, int kh, int kw) {
    if (kh != 3 || kw != 3 || Cin != 1 || Cout != 16) return false;
    if (kd == 3) return D % WC1<3>::TZ == 0 && H % WC1<3>::TY == 0 && W % WC1<3>::TX == 0;
    return kd == 1 && D == 1 && H % WC1<1>::TY == 0 && W % WC1<1>::TX == 0;
}

long long mis_wgrad_cin1_workspace_bytes(int N, int D, int H, int W) {
    return D == 1 ? (long long)grid_for(tiles_of<1>(N, D, H, W)) * 4 * WC1<1>::NG * 256 * 4
                  : (long long)grid_for(tiles_of<3>(N, D, H, W)) * 4 * WC1<3>::NG * 256 * 4;
}

static int launch_cin1(const float* x, long long x_bs, const float* dy, long long dy_bs, const Cin1Norm* nb, float* dw,
                       float* ws, long long ws_bytes, int N, int D, int H, int W, int accumulate, hipStream_t stream) {
    if (((uintptr_t)dy & 15) || dy_bs % 4) return MIS_ERR_UNSUPPORTED;
    if (ws_bytes < mis_wgrad_cin1_workspace_bytes(N, D, H, W)) return MIS_ERR_WORKSPACE;
    return D == 1 ? launch_cin1_kd<1>(x, x_bs, dy, dy_bs, nb, dw, ws, N, D, H, W, accumulate, stream)
                  : launch_cin1_kd<3>(x, x_bs, dy, dy_bs, nb, dw, ws, N, D, H, W, accumulate, stream);
}

int mis_wgrad_cin1(const float* x, long long x_bs, const float* dy, long long dy_bs, float* dw, float* ws,
                   long long ws_bytes, int N, int D, int H, int W, int accumulate, hipStream_t stream) {
    return launch_cin1(x, x_bs, dy, dy_bs, nullptr, dw, ws, ws_bytes, N, D, H, W, accumulate, stream);
}

// Weight gradient of the first layer Conv3d(1 -> 16, k = 3, pad = 1) -- or, with D == 1, Conv2d(1 -> 16, 3, pad 1) --
// straight from the gradient at the activation that follows its BatchNorm / InstanceNorm + (Leaky)ReLU: da [N][16][S]
// (gradient at the activation), y [N][16][S] (the conv's output), sums = mis_norm_act_bwd_sums of the same layer.
// Replaces mis_norm_act_bwd's apply pass + mis_conv_wgrad for a conv whose input needs no gradient.  No dropout on this
// layer's normalisation (unet_3D conv1, vnet block_one).
extern "C" int mis_conv_wgrad_cin1_norm_eligible(int N, int Cout, int D, int H, int W) {
    return mis_wgrad_cin1_eligible(N, 1, Cout, D, H, W, D == 1 ? 1 : 3, 3, 3) ? 1 : 0;
}

extern "C" int mis_conv_wgrad_cin1_norm(const float* x, long long x_bs, const float* da, long long da_bs, const float* y,
                                        long long y_bs, int N, int D, int H, int W, int per_sample, const float* mean,
                                        const float* rstd, const float* gamma, const float* beta, const float* sums,
                                        float slope, float* dw, float* workspace, long long workspace_bytes,
                                        int accumulate, hipStream_t stream) {
    if (!x || !da || !y || !mean || !rstd || !sums || !dw || !workspace || N <= 0) return MIS_ERR_ARG;
    if (!mis_wgrad_cin1_eligible(N, 1, 16, D, H, W, D == 1 ? 1 : 3, 3, 3) || (per_sample && (gamma || beta)))
        return MIS_ERR_UNSUPPORTED;
    if (((uintptr_t)y & 15) || y_bs % 4 || ((uintptr_t)sums & 7)) return MIS_ERR_UNSUPPORTED;
    const Cin1Norm nb{y, y_bs, mean, rstd, gamma, beta, reinterpret_cast<const float2*>(sums), slope, per_sample};
    return launch_cin1(x, x_bs, da, da_bs, &nb, dw, workspace, workspace_bytes, N, D, H, W, accumulate, stream);
}
