// 2x max-pooling and 2x linear up-sampling (forward + backward), 2D and 3D.
//
// Replaces (reference):
//   nn.MaxPool2d(2)                       code/networks/unet.py:56
//   nn.MaxPool3d(kernel_size=(2,2,2))     code/networks/unet_3D.py:35-47
//   nn.Upsample(scale_factor=2, mode='bilinear', align_corners=True)   code/networks/unet.py:74-75
//   nn.Upsample(scale_factor=(2,2,2), mode='trilinear')  (align_corners=False)  code/networks/utils.py:264
//
// All HBM-bound gathers over NCDHW fp32 with an explicit batch stride on every
// tensor, so producers can write straight into (and consumers read straight
// out of) the channel-concatenated skip buffers: torch.cat of the reference
// (unet.py:85, utils.py:276) never materialises.  Backward passes are written
// as gathers (each input element collects from the outputs it fed) so they are
// deterministic and atomics-free.
//
// Launch shape: grid = (plane chunks, depth slice, n*C + c); all index math is 32-bit (a flat
// 64-bit index decoded with 64-bit divisions costs more than the memory traffic of these kernels).
#include "common.h"
#include <stdlib.h>

namespace {

struct PoolArgs {
    const float* x; long long x_bs;
    float* y; long long y_bs;
    unsigned char* idx;  // [N][C][So] argmax position inside the window (z*4 + y*2 + x)
    int N, C, D, H, W, Do, Ho, Wo, pz;  // pz = 2 for 3D pooling, 1 for 2D
};

// one thread per output element; grid (ceil(Ho*Wo/256), Do, N*C)
__global__ __launch_bounds__(256) void maxpool_fwd_kernel(const PoolArgs a) {
    const int pl = blockIdx.x * 256 + threadIdx.x;
    if (pl >= a.Ho * a.Wo) return;
    const int zo = blockIdx.y, nc = blockIdx.z;
    const int n = nc / a.C, c = nc - n * a.C;
    const int yo = pl / a.Wo, xo = pl - yo * a.Wo;
    const long long S = (long long)a.D * a.H * a.W, So = (long long)a.Do * a.Ho * a.Wo;
    const float* __restrict__ xb = a.x + (long long)n * a.x_bs + (long long)c * S;
    float best = -INFINITY;
    int bi = 0;
    for (int dz = 0; dz < a.pz; ++dz)
#pragma unroll
        for (int dy = 0; dy < 2; ++dy) {
            const float2 v = *reinterpret_cast<const float2*>(
                xb + ((long long)(zo * a.pz + dz) * a.H + (yo * 2 + dy)) * a.W + xo * 2);
            // first maximum wins (torch semantics); NaN propagates like torch's ">" || isnan
            if (v.x > best || v.x != v.x) { best = v.x; bi = dz * 4 + dy * 2; }
            if (v.y > best || v.y != v.y) { best = v.y; bi = dz * 4 + dy * 2 + 1; }
        }
    const long long o = ((long long)zo * a.Ho + yo) * a.Wo + xo;
    a.y[(long long)n * a.y_bs + (long long)c * So + o] = best;
    if (a.idx) a.idx[(long long)nc * So + o] = (unsigned char)bi;
}

struct PoolBwdArgs {
    const float* dy; long long dy_bs;
    const unsigned char* idx;
    float* dx; long long dx_bs;
    int N, C, D, H, W, Do, Ho, Wo, pz, accumulate;
};

// one thread per input element PAIR along x (one pooling-window row); grid (ceil(H*W/2/256), D, N*C)
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const PoolBwdArgs a) {
    const int Wh = a.W >> 1;
    const int pl = blockIdx.x * 256 + threadIdx.x;
    if (pl >= a.H * Wh) return;
    const int z = blockIdx.y, nc = blockIdx.z;
    const int n = nc / a.C, c = nc - n * a.C;
    const int y = pl / Wh, xo = pl - y * Wh;
    const long long S = (long long)a.D * a.H * a.W, So = (long long)a.Do * a.Ho * a.Wo;
    const int zo = z / a.pz, yo = y >> 1;
    float2 g = make_float2(0.f, 0.f);
    if (zo < a.Do && yo < a.Ho && xo < a.Wo) {
        const long long o = ((long long)zo * a.Ho + yo) * a.Wo + xo;
        const int bi = a.idx[(long long)nc * So + o];
        const float d = a.dy[(long long)n * a.dy_bs + (long long)c * So + o];
        const int local = (z - zo * a.pz) * 4 + (y & 1) * 2;
        if (bi == local) g.x = d;
        if (bi == local + 1) g.y = d;
    }
    float2* p = reinterpret_cast<float2*>(a.dx + (long long)n * a.dx_bs + (long long)c * S +
                                          ((long long)z * a.H + y) * a.W + xo * 2);
    if (a.accumulate) { const float2 o = *p; g.x += o.x; g.y += o.y; }
    *p = g;
}

// 3-D pooling, 4 outputs per thread (W % 8 == 0, volumes < 2^31 elements).  The one-output-per-thread kernels above
// spend ~50 instructions (64-bit index arithmetic, byte loads) per 8 bytes of dx and are VALU-issue bound on the
// 16-lane SIMDs (measured 2.3 TB/s backward); here a thread owns 4 consecutive pooled outputs = a 2 x 2 x 8 block of
// the fine tensor: float4 loads / stores only, one uchar4 of argmax codes, 32-bit index arithmetic.
// grid = (ceil(Ho*Wo/4 / 256), Do, N*C)
__global__ __launch_bounds__(256) void maxpool3d_fwd4_kernel(const PoolArgs a) {
    const int Wq = a.Wo >> 2;
    const int pl = blockIdx.x * 256 + threadIdx.x;
    if (pl >= a.Ho * Wq) return;
    const int zo = blockIdx.y, nc = blockIdx.z;
    const int n = nc / a.C, c = nc - n * a.C;
    const int yo = pl / Wq, xq = pl - yo * Wq;
    const unsigned S = (unsigned)(a.D * a.H * a.W), So = (unsigned)(a.Do * a.Ho * a.Wo);
    const float* __restrict__ xb = a.x + (long long)n * a.x_bs + (long long)c * S;
    float best[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    unsigned bi[4] = {0, 0, 0, 0};
#pragma unroll
    for (int dz = 0; dz < 2; ++dz)
#pragma unroll
        for (int dy = 0; dy < 2; ++dy) {
            const float* __restrict__ r = xb + (unsigned)(((zo * 2 + dz) * a.H + (yo * 2 + dy)) * a.W + xq * 8);
            const float4 v0 = *reinterpret_cast<const float4*>(r), v1 = *reinterpret_cast<const float4*>(r + 4);
            const float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int dx = 0; dx < 2; ++dx) {
                    const float t = v[2 * j + dx];
                    // first maximum wins (torch semantics); NaN propagates like torch's ">" || isnan
                    if (t > best[j] || t != t) { best[j] = t; bi[j] = dz * 4 + dy * 2 + dx; }
                }
        }
    const unsigned o = (unsigned)((zo * a.Ho + yo) * a.Wo + xq * 4);
    *reinterpret_cast<float4*>(a.y + (long long)n * a.y_bs + (long long)c * So + o) =
        make_float4(best[0], best[1], best[2], best[3]);
    if (a.idx)
        *reinterpret_cast<unsigned*>(a.idx + (long long)nc * So + o) = bi[0] | (bi[1] << 8) | (bi[2] << 16) | (bi[3] << 24);
}

__global__ __launch_bounds__(256) void maxpool3d_bwd4_kernel(const PoolBwdArgs a) {
    const int Wq = a.Wo >> 2;
    const int pl = blockIdx.x * 256 + threadIdx.x;
    if (pl >= a.Ho * Wq) return;
    const int zo = blockIdx.y, nc = blockIdx.z;
    const int n = nc / a.C, c = nc - n * a.C;
    const int yo = pl / Wq, xq = pl - yo * Wq;
    const unsigned S = (unsigned)(a.D * a.H * a.W), So = (unsigned)(a.Do * a.Ho * a.Wo);
    const unsigned o = (unsigned)((zo * a.Ho + yo) * a.Wo + xq * 4);
    const float4 d4 = *reinterpret_cast<const float4*>(a.dy + (long long)n * a.dy_bs + (long long)c * So + o);
    const unsigned code = *reinterpret_cast<const unsigned*>(a.idx + (long long)nc * So + o);
    const float d[4] = {d4.x, d4.y, d4.z, d4.w};
    float* __restrict__ dxb = a.dx + (long long)n * a.dx_bs + (long long)c * S;
#pragma unroll
    for (int dz = 0; dz < 2; ++dz)
#pragma unroll
        for (int dy = 0; dy < 2; ++dy) {
            float g[8];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const unsigned bi = (code >> (8 * j)) & 0xff;
                g[2 * j] = bi == (unsigned)(dz * 4 + dy * 2) ? d[j] : 0.f;
                g[2 * j + 1] = bi == (unsigned)(dz * 4 + dy * 2 + 1) ? d[j] : 0.f;
            }
            float* r = dxb + (unsigned)(((zo * 2 + dz) * a.H + (yo * 2 + dy)) * a.W + xq * 8);
            if (a.accumulate) {
                const float4 o0 = *reinterpret_cast<const float4*>(r), o1 = *reinterpret_cast<const float4*>(r + 4);
                g[0] += o0.x; g[1] += o0.y; g[2] += o0.z; g[3] += o0.w; g[4] += o1.x; g[5] += o1.y; g[6] += o1.z; g[7] += o1.w;
            }
            *reinterpret_cast<float4*>(r) = make_float4(g[0], g[1], g[2], g[3]);
            *reinterpret_cast<float4*>(r + 4) = make_float4(g[4], g[5], g[6], g[7]);
        }
}

// ---- linear 2x up-sampling ----
// source coordinate of output index o (torch's area_pixel_compute_source_index)
__device__ __forceinline__ void src_index(int o, int in, int out, int align, int& i0, int& i1, float& l1) {
    float s;
    if (align) {
        s = out > 1 ? (float)o * ((float)(in - 1) / (float)(out - 1)) : 0.f;
    } else {
        s = ((float)o + 0.5f) * 0.5f - 0.5f;
        if (s < 0.f) s = 0.f;
    }
    i0 = (int)s;
    if (i0 > in - 1) i0 = in - 1;
    i1 = i0 + (i0 < in - 1 ? 1 : 0);
    l1 = s - (float)i0;
}

struct UpArgs {
    const float* x; long long x_bs;
    float* y; long long y_bs;
    int N, C, D, H, W, Do, Ho, Wo, align;
};

// one thread per 2 consecutive outputs along x (float2 store); grid (ceil(Ho*Wo/2/256), Do, N*C).
// The z / y interpolation set-up is shared by the pair.
__global__ __launch_bounds__(256) void upsample_fwd_kernel(const UpArgs a) {
    const int Wh = a.Wo >> 1;
    const int pl = blockIdx.x * 256 + threadIdx.x;
    if (pl >= a.Ho * Wh) return;
    const int zo = blockIdx.y, nc = blockIdx.z;
    const int n = nc / a.C, c = nc - n * a.C;
    const int yo = pl / Wh, xp = pl - yo * Wh;
    const long long S = (long long)a.D * a.H * a.W, So = (long long)a.Do * a.Ho * a.Wo;
    const float* __restrict__ xb = a.x + (long long)n * a.x_bs + (long long)c * S;
    int y0, y1, z0 = 0, z1 = 0;
    float ly, lz = 0.f;
    src_index(yo, a.H, a.Ho, a.align, y0, y1, ly);
    if (a.D > 1) src_index(zo, a.D, a.Do, a.align, z0, z1, lz);
    const float hy = 1.f - ly, hz = 1.f - lz;
    const float* __restrict__ r00 = xb + ((long long)z0 * a.H + y0) * a.W;
    const float* __restrict__ r01 = xb + ((long long)z0 * a.H + y1) * a.W;
    const float* __restrict__ r10 = xb + ((long long)z1 * a.H + y0) * a.W;
    const float* __restrict__ r11 = xb + ((long long)z1 * a.H + y1) * a.W;
    float out[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        int x0, x1;
        float lx;
        src_index(2 * xp + j, a.W, a.Wo, a.align, x0, x1, lx);
        const float hx = 1.f - lx;
        // same association as torch's upsample_{bi,tri}linear kernels
        float v = hz * (hy * (hx * r00[x0] + lx * r00[x1]) + ly * (hx * r01[x0] + lx * r01[x1]));
        if (a.D > 1) v += lz * (hy * (hx * r10[x0] + lx * r10[x1]) + ly * (hx * r11[x0] + lx * r11[x1]));
        out[j] = v;
    }
    float* dst = a.y + (long long)n * a.y_bs + (long long)c * So + ((long long)zo * a.Ho + yo) * a.Wo + 2 * xp;
    *reinterpret_cast<float2*>(dst) = make_float2(out[0], out[1]);
}

__device__ __forceinline__ float pick4(const float* v, int k) {
    return k == 0 ? v[0] : (k == 1 ? v[1] : (k == 2 ? v[2] : v[3]));
}

// 2-D bilinear x2 (align_corners = True, the 2-D UNet decoder): one thread per 2 rows x 4 columns of the output.
// The 8 outputs read at most 3 input rows x 4 input columns (the source coordinate advances by < 0.5 per output), so
// the patch is loaded once (12 loads instead of 32) and every output picks its four neighbours from registers; two
// float4 stores.  Same per-output arithmetic and association as upsample_fwd_kernel.  Wo % 4 == 0, Ho % 2 == 0.
__global__ __launch_bounds__(256) void upsample_bi2_fwd_kernel(const UpArgs a) {
    const int Wq = a.Wo >> 2;
    const int pl = blockIdx.x * 256 + threadIdx.x;
    if (pl >= (a.Ho >> 1) * Wq) return;
    const int nc = blockIdx.z;
    const int n = nc / a.C, c = nc - n * a.C;
    const int yp = pl / Wq, xq = pl - yp * Wq;
    const float* __restrict__ xb = a.x + (long long)n * a.x_bs + (long long)c * a.H * a.W;
    int y0[2], y1[2], x0[4], x1[4];
    float ly[2], lx[4];
#pragma unroll
    for (int i = 0; i < 2; ++i) src_index(2 * yp + i, a.H, a.Ho, 1, y0[i], y1[i], ly[i]);
#pragma unroll
    for (int j = 0; j < 4; ++j) src_index(4 * xq + j, a.W, a.Wo, 1, x0[j], x1[j], lx[j]);
    const int rb = y0[0], cb = x0[0];
    float v[3][4];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        const int rr = rb + r < a.H ? rb + r : a.H - 1;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int cc = cb + q < a.W ? cb + q : a.W - 1;
            v[r][q] = xb[(long long)rr * a.W + cc];
        }
    }
    float* dst = a.y + (long long)n * a.y_bs + (long long)c * a.Ho * a.Wo + (long long)(2 * yp) * a.Wo + 4 * xq;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int r0 = y0[i] - rb, r1 = y1[i] - rb;      // 0..1, 0..2
        float top[4], bot[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            top[q] = r0 == 0 ? v[0][q] : v[1][q];
            bot[q] = r1 == 0 ? v[0][q] : (r1 == 1 ? v[1][q] : v[2][q]);
        }
        float o[4];
        const float hy = 1.f - ly[i];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float hx = 1.f - lx[j];
            const int c0 = x0[j] - cb, c1 = x1[j] - cb;  // 0..2, 0..3
            o[j] = 1.f * (hy * (hx * pick4(top, c0) + lx[j] * pick4(top, c1)) +
                          ly[i] * (hx * pick4(bot, c0) + lx[j] * pick4(bot, c1)));
        }
        *reinterpret_cast<float4*>(dst + (long long)i * a.Wo) = make_float4(o[0], o[1], o[2], o[3]);
    }
}

// The same up-sampling through LDS: a workgroup owns a 32 x 64 tile of the output; the 18 x 34 input cells it reads are
// staged once, and the source index / weight of each of its 64 output columns and 32 output rows is computed once (the
// register form above evaluates them per thread and picks neighbours with select chains: ~300 vector instructions per
// 8 outputs, 2.6 TB/s at 24+24 images of 128^2 -> 256^2 x 16 channels).  A thread writes 8 consecutive outputs of one row
// (two float4).  Same per-output arithmetic and association.  Wo % 4 == 0.
constexpr int BF_TY = 32, BF_TX = 64, BF_RY = BF_TY / 2 + 2, BF_RX = BF_TX / 2 + 2, BF_LD = BF_RX + 1;

__global__ __launch_bounds__(256) void upsample_bi2_fwd_lds_kernel(const UpArgs a) {
    __shared__ float tile[BF_RY * BF_LD];
    __shared__ int tx0[BF_TX], ty0[BF_TY];
    __shared__ float tlx[BF_TX], tly[BF_TY];
    const int oy0 = blockIdx.y * BF_TY, ox0 = blockIdx.x * BF_TX, nc = blockIdx.z;
    const int n = nc / a.C, c = nc - n * a.C;
    const int tid = threadIdx.x;
    if (tid < BF_TX) {
        const int ox = ox0 + tid < a.Wo ? ox0 + tid : a.Wo - 1;
        int i0, i1;
        float l;
        src_index(ox, a.W, a.Wo, 1, i0, i1, l);
        tx0[tid] = i0; tlx[tid] = l;
    } else if (tid < BF_TX + BF_TY) {
        const int k = tid - BF_TX;
        const int oy = oy0 + k < a.Ho ? oy0 + k : a.Ho - 1;
        int i0, i1;
        float l;
        src_index(oy, a.H, a.Ho, 1, i0, i1, l);
        ty0[k] = i0; tly[k] = l;
    }
    __syncthreads();
    const int ry0 = ty0[0], rx0 = tx0[0];
    const float* __restrict__ xb = a.x + (long long)n * a.x_bs + (long long)c * a.H * a.W;
    for (int i = tid; i < BF_RY * BF_RX; i += 256) {
        const int r = i / BF_RX, q = i - r * BF_RX;
        const int rr = ry0 + r < a.H ? ry0 + r : a.H - 1, cc = rx0 + q < a.W ? rx0 + q : a.W - 1;     // clamped: finite values
        tile[r * BF_LD + q] = xb[(long long)rr * a.W + cc];                                         // under zero weights
    }
    __syncthreads();
    const int oyl = tid >> 3, oxl = (tid & 7) * 8;
    const int oy = oy0 + oyl;
    if (oy >= a.Ho || ox0 + oxl >= a.Wo) return;
    const int r0 = ty0[oyl] - ry0;
    const float ly = tly[oyl], hy = 1.f - ly;
    const float* __restrict__ t0 = tile + r0 * BF_LD;
    const float* __restrict__ t1 = t0 + BF_LD;
    float o[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int c0 = tx0[oxl + j] - rx0;
        const float lx = tlx[oxl + j], hx = 1.f - lx;
        o[j] = 1.f * (hy * (hx * t0[c0] + lx * t0[c0 + 1]) + ly * (hx * t1[c0] + lx * t1[c0 + 1]));
    }
    float* dst = a.y + (long long)n * a.y_bs + (long long)c * a.Ho * a.Wo + (long long)oy * a.Wo + ox0 + oxl;
    *reinterpret_cast<float4*>(dst) = make_float4(o[0], o[1], o[2], o[3]);
    if (ox0 + oxl + 4 < a.Wo) *reinterpret_cast<float4*>(dst + 4) = make_float4(o[4], o[5], o[6], o[7]);
}

struct UpBwdArgs {
    const float* dy; long long dy_bs;
    float* dx; long long dx_bs;
    int N, C, D, H, W, Do, Ho, Wo, align, accumulate;
};

// weight with which output index o reads input index i along one axis
__device__ __forceinline__ float axis_w(int o, int i, int in, int out, int align) {
    int i0, i1;
    float l1;
    src_index(o, in, out, align, i0, i1, l1);
    float w = 0.f;
    if (i == i0) w += 1.f - l1;
    if (i == i1) w += l1;
    return w;
}

// candidate outputs that can touch input index i.  align_corners=False (x2): exactly [2i-1, 2i+2];
// align_corners=True: [2i-2, 2i+3] covers every ratio (in-1)/(2in-1).
template <int NC>
__device__ __forceinline__ void axis_candidates(int i, int in, int out, int align, int first, int (&o)[NC],
                                                float (&w)[NC]) {
#pragma unroll
    for (int k = 0; k < NC; ++k) {
        o[k] = 2 * i + first + k;
        w[k] = (o[k] >= 0 && o[k] < out) ? axis_w(o[k], i, in, out, align) : 0.f;
        if (o[k] < 0) o[k] = 0;
        if (o[k] >= out) o[k] = out - 1;
    }
}

// one thread per input element; grid (ceil(H*W/256), D, N*C)
template <int NC>
__global__ __launch_bounds__(256) void upsample_bwd_kernel(const UpBwdArgs a) {
    const int pl = blockIdx.x * 256 + threadIdx.x;
    if (pl >= a.H * a.W) return;
    const int z = blockIdx.y, nc = blockIdx.z;
    const int n = nc / a.C, c = nc - n * a.C;
    const int y = pl / a.W, x = pl - y * a.W;
    const long long S = (long long)a.D * a.H * a.W, So = (long long)a.Do * a.Ho * a.Wo;
    const float* __restrict__ db = a.dy + (long long)n * a.dy_bs + (long long)c * So;
    const int first = NC == 4 ? -1 : -2;
    float wx[NC], wy[NC], wz[NC];
    int ox[NC], oy[NC], oz[NC];
    axis_candidates<NC>(x, a.W, a.Wo, a.align, first, ox, wx);
    axis_candidates<NC>(y, a.H, a.Ho, a.align, first, oy, wy);
    float g = 0.f;
    if (a.D > 1) {
        axis_candidates<NC>(z, a.D, a.Do, a.align, first, oz, wz);
#pragma unroll
        for (int kz = 0; kz < NC; ++kz) {
            float gz = 0.f;
#pragma unroll
            for (int ky = 0; ky < NC; ++ky) {
                const float* __restrict__ row = db + ((long long)oz[kz] * a.Ho + oy[ky]) * a.Wo;
                float r = 0.f;
#pragma unroll
                for (int kx = 0; kx < NC; ++kx) r += wx[kx] * row[ox[kx]];
                gz += wy[ky] * r;
            }
            g += wz[kz] * gz;
        }
    } else {
#pragma unroll
        for (int ky = 0; ky < NC; ++ky) {
            const float* __restrict__ row = db + (long long)oy[ky] * a.Wo;
            float r = 0.f;
#pragma unroll
            for (int kx = 0; kx < NC; ++kx) r += wx[kx] * row[ox[kx]];
            g += wy[ky] * r;
        }
    }
    float* p = a.dx + (long long)n * a.dx_bs + (long long)c * S + ((long long)z * a.H + y) * a.W + x;
    *p = a.accumulate ? *p + g : g;
}

// ---- trilinear x2, align_corners=False (unet_3D's nn.Upsample, reference networks/utils.py:264) ----
// For scale 2 every output 2i reads inputs (i-1, i) with weights (0.25, 0.75) and 2i+1 reads (i, i+1) with
// (0.75, 0.25); at the borders the missing neighbour is the clamped index itself (0.25 + 0.75 on the same
// element = torch's weight 1).  One thread owns an input cell: 27 clamped loads -> its 2x2x2 output block
// (3.4 loads and ~10 flops per output instead of 8 loads and the float source-index arithmetic per output).
// Association as in torch: v = hz*(hy*(hx*a+lx*b) + ly*(hx*c+lx*d)) + lz*(...).
__global__ __launch_bounds__(256) void upsample_tri2_fwd_kernel(const UpArgs a) {
    const int pl = blockIdx.x * 256 + threadIdx.x;
    if (pl >= a.H * a.W) return;
    const int z = blockIdx.y, nc = blockIdx.z;
    const int n = nc / a.C, c = nc - n * a.C;
    const int y = pl / a.W, x = pl - y * a.W;
    const int S = a.D * a.H * a.W;
    const float* __restrict__ xb = a.x + (long long)n * a.x_bs + (long long)c * S;
    const int zi[3] = {z > 0 ? z - 1 : 0, z, z < a.D - 1 ? z + 1 : z};
    const int yi[3] = {y > 0 ? y - 1 : 0, y, y < a.H - 1 ? y + 1 : y};
    const int xi[3] = {x > 0 ? x - 1 : 0, x, x < a.W - 1 ? x + 1 : x};
    float e[3][3][2];   // x-interpolated pairs of the 3x3 (z,y) rows
#pragma unroll
    for (int kz = 0; kz < 3; ++kz)
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const float* __restrict__ r = xb + (zi[kz] * a.H + yi[ky]) * a.W;
            const float v0 = r[xi[0]], v1 = r[xi[1]], v2 = r[xi[2]];
            e[kz][ky][0] = 0.25f * v0 + 0.75f * v1;
            e[kz][ky][1] = 0.75f * v1 + 0.25f * v2;
        }
    const long long So = (long long)a.Do * a.Ho * a.Wo;
    float* __restrict__ yb = a.y + (long long)n * a.y_bs + (long long)c * So;
#pragma unroll
    for (int dz = 0; dz < 2; ++dz)
#pragma unroll
        for (int dy = 0; dy < 2; ++dy) {
            float o[2];
#pragma unroll
            for (int dx = 0; dx < 2; ++dx) {
                // output 2i+d reads planes (i-1+d, i+d) with weights d ? (0.75, 0.25) : (0.25, 0.75)
                const float hy = dy ? 0.75f : 0.25f, ly = 1.f - hy, hz = dz ? 0.75f : 0.25f, lz = 1.f - hz;
                o[dx] = hz * (hy * e[dz][dy][dx] + ly * e[dz][dy + 1][dx]) +
                        lz * (hy * e[dz + 1][dy][dx] + ly * e[dz + 1][dy + 1][dx]);
            }
            float* dst = yb + ((long long)(2 * z + dz) * a.Ho + (2 * y + dy)) * a.Wo + 2 * x;
            *reinterpret_cast<float2*>(dst) = make_float2(o[0], o[1]);
        }
}

// The same arithmetic (bit-identical), marching along z: a thread owns one (y, x) input column segment of ZR cells and
// keeps E(p)[dy][dx] = the (y, x)-interpolated 2 x 2 block of input plane p for three consecutive planes; every new plane
// costs 9 loads and feeds two output planes: out(2z) = 0.25 E(z-1) + 0.75 E(z), out(2z+1) = 0.75 E(z) + 0.25 E(z+1).
// 1.7 loads and ~5 VALU instructions per output instead of 3.4 and ~11 (the one-cell kernel is instruction-bound:
// 3.4 TB/s of the 906 MB it writes at 96^3; this form 4.3).
template <int ZR>
__global__ __launch_bounds__(256) void upsample_tri2_fwd_z_kernel(const UpArgs a) {
    const int pl = blockIdx.x * 256 + threadIdx.x;
    if (pl >= a.H * a.W) return;
    const int z0 = blockIdx.y * ZR, nc = blockIdx.z;
    const int n = nc / a.C, c = nc - n * a.C;
    const int y = pl / a.W, x = pl - y * a.W;
    const int S = a.D * a.H * a.W;
    const float* __restrict__ xb = a.x + (long long)n * a.x_bs + (long long)c * S;
    const int yo[3] = {(y > 0 ? y - 1 : 0) * a.W, y * a.W, (y < a.H - 1 ? y + 1 : y) * a.W};
    const int xi[3] = {x > 0 ? x - 1 : 0, x, x < a.W - 1 ? x + 1 : x};
    auto plane = [&](int p, float (&E)[2][2]) {
        const float* __restrict__ r = xb + (long long)p * a.H * a.W;
        float ex[3][2];
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const float v0 = r[yo[ky] + xi[0]], v1 = r[yo[ky] + xi[1]], v2 = r[yo[ky] + xi[2]];
            ex[ky][0] = 0.25f * v0 + 0.75f * v1;
            ex[ky][1] = 0.75f * v1 + 0.25f * v2;
        }
#pragma unroll
        for (int dy = 0; dy < 2; ++dy)
#pragma unroll
            for (int dx = 0; dx < 2; ++dx) {
                const float hy = dy ? 0.75f : 0.25f, ly = 1.f - hy;
                E[dy][dx] = hy * ex[dy][dx] + ly * ex[dy + 1][dx];
            }
    };
    const long long So = (long long)a.Do * a.Ho * a.Wo;
    float* __restrict__ yb = a.y + (long long)n * a.y_bs + (long long)c * So + (long long)(2 * y) * a.Wo + 2 * x;
    float Ep[2][2], Ec[2][2], En[2][2];
    plane(z0 > 0 ? z0 - 1 : 0, Ep);
    plane(z0, Ec);
    const int z1 = z0 + ZR < a.D ? z0 + ZR : a.D;
#pragma unroll
    for (int zz = 0; zz < ZR; ++zz) {
        const int z = z0 + zz;
        if (z >= z1) break;
        plane(z < a.D - 1 ? z + 1 : z, En);
#pragma unroll
        for (int dy = 0; dy < 2; ++dy) {
            float* d0 = yb + ((long long)(2 * z) * a.Ho + dy) * a.Wo;
            // dz = 0: hz = 0.25 on plane z-1, lz = 0.75 on plane z;  dz = 1: hz = 0.75 on plane z, lz = 0.25 on plane z+1
            *reinterpret_cast<float2*>(d0) = make_float2(0.25f * Ep[dy][0] + 0.75f * Ec[dy][0], 0.25f * Ep[dy][1] + 0.75f * Ec[dy][1]);
            *reinterpret_cast<float2*>(d0 + (long long)a.Ho * a.Wo) =
                make_float2(0.75f * Ec[dy][0] + 0.25f * En[dy][0], 0.75f * Ec[dy][1] + 0.25f * En[dy][1]);
        }
#pragma unroll
        for (int dy = 0; dy < 2; ++dy)
#pragma unroll
            for (int dx = 0; dx < 2; ++dx) { Ep[dy][dx] = Ec[dy][dx]; Ec[dy][dx] = En[dy][dx]; }
    }
}

// Backward of the same: dx[i] collects outputs 2i-1 .. 2i+2 with weights (0.25, 0.75, 0.75, 0.25) (1.0 at the
// two border outputs, 0 outside).  One thread owns an x-pair of input cells and ZR consecutive z: every
// dy plane is reduced over (y, x) once (24 loads for the pair) and feeds two dx planes, i.e. ~30 loads per
// dx element instead of 64.  Gather form: deterministic, no atomics.
template <int ZR>
__global__ __launch_bounds__(256) void upsample_tri2_bwd_kernel(const UpBwdArgs a) {
    const int Wp = (a.W + 1) >> 1;
    const int pl = blockIdx.x * 256 + threadIdx.x;
    if (pl >= a.H * Wp) return;
    const int z0 = blockIdx.y * ZR, nc = blockIdx.z;
    const int n = nc / a.C, c = nc - n * a.C;
    const int y = pl / Wp, x = 2 * (pl - y * Wp);
    const int S = a.D * a.H * a.W;
    const long long So = (long long)a.Do * a.Ho * a.Wo;
    const float* __restrict__ db = a.dy + (long long)n * a.dy_bs + (long long)c * So;
    // candidate outputs along x for the pair (x, x+1): 2x-1 .. 2x+4; along y: 2y-1 .. 2y+2
    int ox[6], oy[4];
    float wx0[6], wx1[6], wy[4];
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        const int o = 2 * x - 1 + k;
        const bool in = o >= 0 && o < a.Wo;
        wx0[k] = (in && k < 4) ? axis_w(o, x, a.W, a.Wo, 0) : 0.f;
        wx1[k] = (in && k >= 2 && x + 1 < a.W) ? axis_w(o, x + 1, a.W, a.Wo, 0) : 0.f;
        ox[k] = o < 0 ? 0 : (o >= a.Wo ? a.Wo - 1 : o);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int o = 2 * y - 1 + k;
        wy[k] = (o >= 0 && o < a.Ho) ? axis_w(o, y, a.H, a.Ho, 0) : 0.f;
        oy[k] = o < 0 ? 0 : (o >= a.Ho ? a.Ho - 1 : o);
    }
    float g0[ZR], g1[ZR];
#pragma unroll
    for (int i = 0; i < ZR; ++i) g0[i] = g1[i] = 0.f;
#pragma unroll
    for (int j = 0; j < 2 * ZR + 2; ++j) {
        const int oz = 2 * z0 - 1 + j;
        if (oz < 0 || oz >= a.Do) continue;
        float p0 = 0.f, p1 = 0.f;   // (y, x)-reduced plane value for x and x+1
#pragma unroll
        for (int ky = 0; ky < 4; ++ky) {
            const float* __restrict__ row = db + ((long long)oz * a.Ho + oy[ky]) * a.Wo;
            float r0 = 0.f, r1 = 0.f;
#pragma unroll
            for (int kx = 0; kx < 6; ++kx) {
                const float v = row[ox[kx]];
                r0 += wx0[kx] * v;
                r1 += wx1[kx] * v;
            }
            p0 += wy[ky] * r0;
            p1 += wy[ky] * r1;
        }
        // plane oz = 2*z0 - 1 + j feeds dx planes z0 + (j-2)/2 .. z0 + j/2 (those inside this thread's run)
#pragma unroll
        for (int i = 0; i < ZR; ++i) {
            const int k = j - 2 * i;   // candidate index of oz for dx plane z0 + i
            if (k >= 0 && k < 4 && z0 + i < a.D) {
                const float wz = axis_w(oz, z0 + i, a.D, a.Do, 0);
                g0[i] += wz * p0;
                g1[i] += wz * p1;
            }
        }
    }
    float* __restrict__ dxb = a.dx + (long long)n * a.dx_bs + (long long)c * S;
#pragma unroll
    for (int i = 0; i < ZR; ++i) {
        if (z0 + i >= a.D) break;
        float* p = dxb + ((z0 + i) * a.H + y) * a.W + x;
        p[0] = a.accumulate ? p[0] + g0[i] : g0[i];
        if (x + 1 < a.W) p[1] = a.accumulate ? p[1] + g1[i] : g1[i];
    }
}

// LDS-tiled form of the same backward (the one the dispatcher picks when rows are float4-aligned).  The gather above
// reads dy with a lane stride of four floats (every load instruction touches 4x the bytes it uses) and was measured
// at 1.5 TB/s on the 96^3 level of unet_3D.  Here a workgroup owns a 4 x 8 x 32 tile of dx: the
// (2*4+2) x (2*8+2) x (2*32+8) region of dy it gathers from is brought into LDS once with coalesced float4 loads
// (z / y indices clamped to the volume = the clamped-neighbour form of the border weights; x handled per lane),
// then every thread reduces an x-pair of cells for two consecutive z from LDS: per (plane, row) one 16-byte read of
// dy[4j .. 4j+3] plus the two neighbours dy[4j-1], dy[4j+4].
//   dx[i] = 0.25 dy[2i-1] + 0.75 dy[2i] + 0.75 dy[2i+1] + 0.25 dy[2i+2]   with dy[-1] := dy[0], dy[2n] := dy[2n-1]
// along each axis (separable).  Gather form, fixed order: deterministic.
template <int ZPT>       // z cells per thread (2: 4 x 8 x 32 tile, 47.5 KB of LDS; 1: 2 x 8 x 32 tile, 28.5 KB)
struct Tri2 {
    static constexpr int TZ = 2 * ZPT, TY = 8, TX = 32;
    static constexpr int RZ = 2 * TZ + 2, RY = 2 * TY + 2, RQ = (2 * TX + 8) / 4, PITCH = RQ * 4;   // 10 x 18 x 72
    static constexpr int LDS_FLOATS = RZ * RY * PITCH;
};

template <int ZPT>
__global__ __launch_bounds__(256) void upsample_tri2_bwd_lds_kernel(const UpBwdArgs a, int tiles_x, int tiles_y,
                                                                    int tiles_z, unsigned n_blocks,
                                                                    unsigned n_blocks_padded) {
    constexpr int TZ = Tri2<ZPT>::TZ, TY = Tri2<ZPT>::TY, TX = Tri2<ZPT>::TX, RZ = Tri2<ZPT>::RZ, RY = Tri2<ZPT>::RY,
                  RQ = Tri2<ZPT>::RQ, PITCH = Tri2<ZPT>::PITCH;
    extern __shared__ __attribute__((aligned(16))) float tri2_lds[];
    // XCD-aware order: neighbouring tiles of one (n, c) volume (shared halo rows, the two x-tiles of a cache line)
    // run on the same XCD, i.e. behind the same L2 -- with the plain 3-D grid every halo line was fetched from HBM
    // once per XCD (measured 2.4 TB/s of useful traffic at ~2.5x that in fetched bytes)
    unsigned t = mis_xcd_remap(blockIdx.x, n_blocks_padded);
    if (t >= n_blocks) return;
    const int tx = t % tiles_x; t /= tiles_x;
    const int ty = t % tiles_y; t /= tiles_y;
    const int tz = t % tiles_z; t /= tiles_z;
    const int nc = (int)t;
    const int n = nc / a.C, c = nc - n * a.C;
    const int z0 = tz * TZ, y0 = ty * TY, x0 = tx * TX;
    const int S = a.D * a.H * a.W;
    const long long So = (long long)a.Do * a.Ho * a.Wo;
    const float* __restrict__ db = a.dy + (long long)n * a.dy_bs + (long long)c * So;
    // ---- stage the dy region: planes 2z0-1 .. 2z0+8, rows 2y0-1 .. 2y0+16 (clamped), floats 2x0-4 .. 2x0+67 ----
    // all loads of a thread are issued before the first LDS write (a load -> store loop serialises on the memory
    // latency).  The kernel is VALU-issue bound (a wave64 instruction takes 4 cycles on the 16-lane SIMDs), so the
    // staging index arithmetic is kept minimal: thread -> (float4 column q, row rr) once, 14 rows further per step.
    constexpr int RPS = 256 / RQ;                               // rows per staging step (14 of 18-float4 rows)
    constexpr int NLD = (RZ * RY + RPS - 1) / RPS;
    const int sq = threadIdx.x % RQ, srr = threadIdx.x / RQ;
    const int sox = 2 * x0 - 4 + 4 * sq;
    const bool scol = threadIdx.x < RPS * RQ && sox >= 0 && sox < a.Wo;
    float4 v[NLD];
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
        const int r = srr + i * RPS;
        const int rz = (r * 57) >> 10, ry = r - rz * RY;        // r / 18 for r < 1024 / 18... (r < 182 here)
        static_assert(RY == 18, "the r / 18 shortcut");
        int oz = 2 * z0 - 1 + rz, oy = 2 * y0 - 1 + ry;
        oz = min(max(oz, 0), a.Do - 1);
        oy = min(max(oy, 0), a.Ho - 1);
        v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (scol && r < RZ * RY) v[i] = *reinterpret_cast<const float4*>(db + (unsigned)((oz * a.Ho + oy) * a.Wo + sox));
    }
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
        const int r = srr + i * RPS;
        if (threadIdx.x < RPS * RQ && r < RZ * RY) *reinterpret_cast<float4*>(&tri2_lds[r * PITCH + 4 * sq]) = v[i];
    }
    __syncthreads();
    const int xl = threadIdx.x & 15, yl = (threadIdx.x >> 4) & 7, zh = threadIdx.x >> 7;
    const int x = x0 + 2 * xl, y = y0 + yl, zb = z0 + ZPT * zh;
    if (x >= a.W || y >= a.H || zb >= a.D) return;
    const bool left_edge = x == 0, right_edge = 2 * x + 4 >= a.Wo;
    // un-normalised weights (1, 3, 3, 1) per axis, one scale by 4^-3 at the end (exact: powers of two)
    float g[2][2] = {{0.f, 0.f}, {0.f, 0.f}};
#pragma unroll
    for (int j = 0; j < 2 * ZPT + 2; ++j) {             // dy planes 2zb-1 .. = region planes 2*ZPT*zh + j
        float p0 = 0.f, p1 = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) {                   // dy rows 2y-1 .. 2y+2 = region rows 2yl + k
            const float* __restrict__ row = &tri2_lds[((2 * ZPT * zh + j) * RY + 2 * yl + k) * PITCH + 4 + 4 * xl];
            const float4 q = *reinterpret_cast<const float4*>(row);
            const float lf = left_edge ? q.x : row[-1];
            const float rt = right_edge ? q.w : row[4];
            const float r0 = fmaf(3.f, q.x + q.y, lf + q.z);
            const float r1 = fmaf(3.f, q.z + q.w, q.y + rt);
            if (k == 0 || k == 3) { p0 += r0; p1 += r1; } else { p0 = fmaf(3.f, r0, p0); p1 = fmaf(3.f, r1, p1); }
        }
        if (j < 4) {
            if (j == 0 || j == 3) { g[0][0] += p0; g[0][1] += p1; }
            else { g[0][0] = fmaf(3.f, p0, g[0][0]); g[0][1] = fmaf(3.f, p1, g[0][1]); }
        }
        if (ZPT == 2 && j >= 2) {
            if (j == 2 || j == 5) { g[1][0] += p0; g[1][1] += p1; }
            else { g[1][0] = fmaf(3.f, p0, g[1][0]); g[1][1] = fmaf(3.f, p1, g[1][1]); }
        }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) { g[i][0] *= 0.015625f; g[i][1] *= 0.015625f; }
    float* __restrict__ dxb = a.dx + (long long)n * a.dx_bs + (long long)c * S;
#pragma unroll
    for (int i = 0; i < ZPT; ++i) {
        if (zb + i >= a.D) break;
        float2* p = reinterpret_cast<float2*>(dxb + ((long long)(zb + i) * a.H + y) * a.W + x);
        float2 o = make_float2(g[i][0], g[i][1]);
        if (a.accumulate) { const float2 old = *p; o.x += old.x; o.y += old.y; }
        *p = o;
    }
}

// Backward of the 2-D bilinear x2 up-sampling with align_corners=True (UNet's decoder, reference unet.py:74-75):
// input index i is read by outputs 2i-2 .. 2i+3 (covers every ratio (in-1)/(2in-1)).  A workgroup owns a 16 x 32
// tile of dx: the (2*16+4) x (2*32+4) region of dy it gathers from is loaded once, coalesced, into LDS (the
// per-thread gather from HBM/L2 ran ~6x off the HBM bound), then each thread reduces an x-pair of cells from LDS
// (6 shared candidate rows x 8 candidate columns).  Gather form: deterministic.
constexpr int BI_TY = 16, BI_TX = 32, BI_RY = 2 * BI_TY + 4, BI_RX = 2 * BI_TX + 4, BI_LD = BI_RX + 1;

// Separable form: the weights with which the candidate outputs 2x-2 .. 2x+5 reach input column x (and 2y-2 .. 2y+3 input row
// y) are computed ONCE per column / row of the tile (one source-index evaluation per thread, tables in LDS), the staged
// dy region is first reduced along x (36 rows x 32 columns, 8 multiply-adds each), then along y (6 each).  The first
// version evaluated 14 source indices per thread and 108 multiply-adds per output pair: ~350 vector instructions per
// pair, 1.7 TB/s at 24+24 images of 256^2 -> 128^2 x 16 channels.  Gather form, fixed order: deterministic.
__global__ __launch_bounds__(256) void upsample_bi2_bwd_kernel(const UpBwdArgs a) {
    __shared__ float s[BI_RY * BI_LD];
    __shared__ float hb[BI_RY * (BI_TX + 1)];
    __shared__ float cw[BI_TX * 8], rw[BI_TY * 6];
    const int nc = blockIdx.z;
    const int n = nc / a.C, c = nc - n * a.C;
    const int y0 = blockIdx.y * BI_TY, x0 = blockIdx.x * BI_TX;
    const int S = a.H * a.W, So = a.Ho * a.Wo;
    const float* __restrict__ db = a.dy + (long long)n * a.dy_bs + (long long)c * So;
    {   // weight of output column 2x - 2 + k for input column x = x0 + (tid >> 3), k = tid & 7 (0 outside the image)
        const int xl = threadIdx.x >> 3, k = threadIdx.x & 7;
        const int x = x0 + xl, o = 2 * x - 2 + k;
        float w = 0.f;
        if (x < a.W && o >= 0 && o < a.Wo) {
            int i0, i1;
            float l1;
            src_index(o, a.W, a.Wo, 1, i0, i1, l1);
            w = (i0 == x ? 1.f - l1 : 0.f) + (i1 == x ? l1 : 0.f);
        }
        cw[threadIdx.x] = w;
        if (threadIdx.x < BI_TY * 6) {
            const int yl = threadIdx.x / 6, ky = threadIdx.x - yl * 6;
            const int y = y0 + yl, oy = 2 * y - 2 + ky;
            float wyv = 0.f;
            if (y < a.H && oy >= 0 && oy < a.Ho) {
                int i0, i1;
                float l1;
                src_index(oy, a.H, a.Ho, 1, i0, i1, l1);
                wyv = (i0 == y ? 1.f - l1 : 0.f) + (i1 == y ? l1 : 0.f);
            }
            rw[threadIdx.x] = wyv;
        }
    }
    if (!(a.dy_bs & 1) && !((uintptr_t)a.dy & 7)) {      // pairs of outputs: 2 x0 - 2 + 2 q is even, Wo is even (8-byte loads)
        for (int e = threadIdx.x; e < BI_RY * (BI_RX / 2); e += 256) {
            const int r = e / (BI_RX / 2), q = e - r * (BI_RX / 2);
            const int oy = 2 * y0 - 2 + r, ox = 2 * x0 - 2 + 2 * q;
            const bool in = oy >= 0 && oy < a.Ho && ox >= 0 && ox < a.Wo;
            const float2 v = in ? *reinterpret_cast<const float2*>(db + oy * a.Wo + ox) : make_float2(0.f, 0.f);
            s[r * BI_LD + 2 * q] = v.x;
            s[r * BI_LD + 2 * q + 1] = v.y;
        }
    } else {
        for (int e = threadIdx.x; e < BI_RY * BI_RX; e += 256) {
            const int r = e / BI_RX, q = e - r * BI_RX;
            const int oy = 2 * y0 - 2 + r, ox = 2 * x0 - 2 + q;
            const bool in = oy >= 0 && oy < a.Ho && ox >= 0 && ox < a.Wo;
            s[r * BI_LD + q] = in ? db[oy * a.Wo + ox] : 0.f;
        }
    }
    __syncthreads();
    {   // along x: hb[r][x] = sum_k cw[x][k] * s[r][2x + k]; thread = column x (its 8 weights in registers), rows r = tid >> 5 + 8 j
        const int xl = threadIdx.x & 31;
        // candidates 2x-2 .. 2x+3 only: 2x+4 and 2x+5 never reach column x (their table entries are 0) and lie outside
        // the staged region for the tile's last column -- 0 * (uninitialised LDS) is a NaN
        float w[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) w[k] = cw[xl * 8 + k];
        for (int r = threadIdx.x >> 5; r < BI_RY; r += 8) {
            const float* __restrict__ row = s + r * BI_LD + 2 * xl;
            float acc = 0.f;
#pragma unroll
            for (int k = 0; k < 6; ++k) acc += w[k] * row[k];
            hb[r * (BI_TX + 1) + xl] = acc;
        }
    }
    __syncthreads();
    const int ly = threadIdx.x >> 4, lx = (threadIdx.x & 15) * 2;
    const int y = y0 + ly, x = x0 + lx;
    if (y >= a.H || x >= a.W) return;
    float g0 = 0.f, g1 = 0.f;
#pragma unroll
    for (int ky = 0; ky < 6; ++ky) {
        const float wv = rw[ly * 6 + ky];
        const float* __restrict__ hr = hb + (2 * ly + ky) * (BI_TX + 1) + lx;
        g0 += wv * hr[0];
        g1 += wv * hr[1];
    }
    float* p = a.dx + (long long)n * a.dx_bs + (long long)c * S + y * a.W + x;
    p[0] = a.accumulate ? p[0] + g0 : g0;
    if (x + 1 < a.W) p[1] = a.accumulate ? p[1] + g1 : g1;
}

bool grid_ok(int planes_y, long long planes_z) { return planes_y <= 65535 && planes_z <= 65535; }

}  // namespace

extern "C" int mis_maxpool2_fwd(const float* x, long long x_bs, float* y, long long y_bs, unsigned char* idx,
                                int N, int C, int D, int H, int W, hipStream_t stream) {
    if (!x || !y || N <= 0 || C <= 0 || D <= 0 || H < 2 || W < 2) return MIS_ERR_ARG;
    if ((W & 1) || (x_bs & 1) || ((uintptr_t)x & 7)) return MIS_ERR_UNSUPPORTED;  // float2 window rows
    PoolArgs a{x, x_bs, y, y_bs, idx, N, C, D, H, W, D > 1 ? D / 2 : 1, H / 2, W / 2, D > 1 ? 2 : 1};
    if (y_bs < (long long)C * a.Do * a.Ho * a.Wo || x_bs < (long long)C * D * H * W) return MIS_ERR_ARG;
    if (!grid_ok(a.Do, (long long)N * C)) return MIS_ERR_UNSUPPORTED;
    const long long S = (long long)D * H * W;
    if (D > 1 && D % 2 == 0 && H % 2 == 0 && W % 8 == 0 && S < (1LL << 31) && !(x_bs & 3) && !(y_bs & 3) &&
        !((uintptr_t)x & 15) && !((uintptr_t)y & 15) && (!idx || !((uintptr_t)idx & 3)))
        hipLaunchKernelGGL(maxpool3d_fwd4_kernel, dim3((a.Ho * (a.Wo / 4) + 255) / 256, a.Do, N * C), dim3(256), 0, stream, a);
    else
        hipLaunchKernelGGL(maxpool_fwd_kernel, dim3((a.Ho * a.Wo + 255) / 256, a.Do, N * C), dim3(256), 0, stream, a);
    return mis_launch_status();
}

extern "C" int mis_maxpool2_bwd(const float* dy, long long dy_bs, const unsigned char* idx, float* dx,
                                long long dx_bs, int N, int C, int D, int H, int W, int accumulate,
                                hipStream_t stream) {
    if (!dy || !idx || !dx || N <= 0 || C <= 0 || D <= 0 || H < 2 || W < 2) return MIS_ERR_ARG;
    if ((W & 1) || (dx_bs & 1) || ((uintptr_t)dx & 7)) return MIS_ERR_UNSUPPORTED;
    PoolBwdArgs a{dy, dy_bs, idx, dx, dx_bs, N, C, D, H, W, D > 1 ? D / 2 : 1, H / 2, W / 2, D > 1 ? 2 : 1,
                  accumulate};
    if (!grid_ok(D, (long long)N * C)) return MIS_ERR_UNSUPPORTED;
    const long long S = (long long)D * H * W;
    if (D > 1 && D % 2 == 0 && H % 2 == 0 && W % 8 == 0 && S < (1LL << 31) && !(dx_bs & 3) && !(dy_bs & 3) &&
        !((uintptr_t)dx & 15) && !((uintptr_t)dy & 15) && !((uintptr_t)idx & 3))
        hipLaunchKernelGGL(maxpool3d_bwd4_kernel, dim3((a.Ho * (a.Wo / 4) + 255) / 256, a.Do, N * C), dim3(256), 0, stream, a);
    else
        hipLaunchKernelGGL(maxpool_bwd_kernel, dim3((H * (W / 2) + 255) / 256, D, N * C), dim3(256), 0, stream, a);
    return mis_launch_status();
}

extern "C" int mis_upsample2_fwd(const float* x, long long x_bs, float* y, long long y_bs, int N, int C, int D,
                                 int H, int W, int align_corners, hipStream_t stream) {
    if (!x || !y || N <= 0 || C <= 0 || D <= 0 || H <= 0 || W <= 0) return MIS_ERR_ARG;
    UpArgs a{x, x_bs, y, y_bs, N, C, D, H, W, D > 1 ? 2 * D : 1, 2 * H, 2 * W, align_corners ? 1 : 0};
    const long long So = (long long)a.Do * a.Ho * a.Wo;
    if (x_bs < (long long)C * D * H * W || y_bs < (long long)C * So) return MIS_ERR_ARG;
    if ((y_bs & 1) || ((uintptr_t)y & 7)) return MIS_ERR_UNSUPPORTED;   // float2 stores (Wo = 2W is even)
    if (!grid_ok(a.Do, (long long)N * C)) return MIS_ERR_UNSUPPORTED;
    // z-marching form, 4 input planes per thread (measured 300 -> 237 us at 48^3 -> 96^3 x 32 channels x 8; 8 planes per
    // thread and an x-pair per thread with float4 stores are slower: 242 / 276 us); MIS_TRI2_FWD_Z=0: the one-cell kernel
    static const int zmarch = getenv("MIS_TRI2_FWD_Z") ? atoi(getenv("MIS_TRI2_FWD_Z")) : 4;
    static const int bi2_lds = getenv("MIS_BI2_FWD_LDS") ? atoi(getenv("MIS_BI2_FWD_LDS")) : 1;
    if (!a.align && D > 1 && So < (1LL << 31) && zmarch >= 8 && D >= 8 && (long long)H * W * N * C >= 256 * 512)
        hipLaunchKernelGGL(upsample_tri2_fwd_z_kernel<8>, dim3((H * W + 255) / 256, (D + 7) / 8, N * C), dim3(256), 0, stream, a);
    else if (!a.align && D > 1 && So < (1LL << 31) && zmarch >= 4 && D >= 4 && (long long)H * W * N * C >= 256 * 256)
        hipLaunchKernelGGL(upsample_tri2_fwd_z_kernel<4>, dim3((H * W + 255) / 256, (D + 3) / 4, N * C), dim3(256), 0, stream, a);
    else if (!a.align && D > 1 && So < (1LL << 31))
        hipLaunchKernelGGL(upsample_tri2_fwd_kernel, dim3((H * W + 255) / 256, D, N * C), dim3(256), 0, stream, a);
    else if (a.align && D == 1 && a.Wo % 4 == 0 && H > 1 && W > 1 && !(y_bs & 3) && !((uintptr_t)y & 15) && bi2_lds &&
             (long long)N * C <= 65535)
        hipLaunchKernelGGL(upsample_bi2_fwd_lds_kernel, dim3((a.Wo + BF_TX - 1) / BF_TX, (a.Ho + BF_TY - 1) / BF_TY, N * C),
                           dim3(256), 0, stream, a);
    else if (a.align && D == 1 && a.Wo % 4 == 0 && H > 1 && W > 1 && !(y_bs & 3) && !((uintptr_t)y & 15))
        hipLaunchKernelGGL(upsample_bi2_fwd_kernel, dim3(((a.Ho / 2) * (a.Wo / 4) + 255) / 256, 1, N * C), dim3(256), 0,
                           stream, a);
    else
        hipLaunchKernelGGL(upsample_fwd_kernel, dim3((a.Ho * (a.Wo / 2) + 255) / 256, a.Do, N * C), dim3(256), 0,
                           stream, a);
    return mis_launch_status();
}

extern "C" int mis_upsample2_bwd(const float* dy, long long dy_bs, float* dx, long long dx_bs, int N, int C,
                                 int D, int H, int W, int align_corners, int accumulate, hipStream_t stream) {
    if (!dy || !dx || N <= 0 || C <= 0 || D <= 0 || H <= 0 || W <= 0) return MIS_ERR_ARG;
    UpBwdArgs a{dy, dy_bs, dx, dx_bs, N, C, D, H, W, D > 1 ? 2 * D : 1, 2 * H, 2 * W, align_corners ? 1 : 0,
                accumulate};
    const long long So = (long long)a.Do * a.Ho * a.Wo;
    if (dx_bs < (long long)C * D * H * W || dy_bs < (long long)C * So) return MIS_ERR_ARG;
    if (!grid_ok(D, (long long)N * C)) return MIS_ERR_UNSUPPORTED;
    const dim3 grid((H * W + 255) / 256, D, N * C);
    if (a.align && D == 1 && So < (1LL << 31)) {
        hipLaunchKernelGGL(upsample_bi2_bwd_kernel, dim3((W + BI_TX - 1) / BI_TX, (H + BI_TY - 1) / BI_TY, N * C),
                           dim3(256), 0, stream, a);
    } else if (a.align) {
        hipLaunchKernelGGL(upsample_bwd_kernel<6>, grid, dim3(256), 0, stream, a);
    } else if (D > 1 && So < (1LL << 31) && W % 2 == 0 && W >= 4 && !(dy_bs & 3) && !((uintptr_t)dy & 15) &&
               !(dx_bs & 1) && !((uintptr_t)dx & 7)) {
        // float4-aligned dy rows (Wo = 2W is a multiple of 4, So a multiple of 8) and float2-aligned dx rows
        static const int zpt = getenv("MIS_TRI2_ZPT") ? atoi(getenv("MIS_TRI2_ZPT")) : 2;
        const int tiles_x = (int)mis_cdiv(W, 32), tiles_y = (int)mis_cdiv(H, 8);
        const int tiles_z = (int)mis_cdiv(D, 2 * zpt);
        const long long nb = (long long)tiles_x * tiles_y * tiles_z * N * C;
        if (nb > 0x7fffffffLL) return MIS_ERR_UNSUPPORTED;
        const unsigned nbp = (unsigned)(mis_cdiv(nb, MIS_NUM_XCD) * MIS_NUM_XCD);
        if (zpt == 1)
            hipLaunchKernelGGL(upsample_tri2_bwd_lds_kernel<1>, dim3(nbp), dim3(256), Tri2<1>::LDS_FLOATS * 4, stream, a,
                               tiles_x, tiles_y, tiles_z, (unsigned)nb, nbp);
        else
            hipLaunchKernelGGL(upsample_tri2_bwd_lds_kernel<2>, dim3(nbp), dim3(256), Tri2<2>::LDS_FLOATS * 4, stream, a,
                               tiles_x, tiles_y, tiles_z, (unsigned)nb, nbp);
    } else if (D > 1 && So < (1LL << 31)) {
        constexpr int ZR = 4;
        hipLaunchKernelGGL(upsample_tri2_bwd_kernel<ZR>, dim3((H * ((W + 1) / 2) + 255) / 256, (D + ZR - 1) / ZR, N * C),
                           dim3(256), 0, stream, a);
    } else {
        hipLaunchKernelGGL(upsample_bwd_kernel<4>, grid, dim3(256), 0, stream, a);
    }
    return mis_launch_status();
}
