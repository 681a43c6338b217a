// V-Net specific data movement: the stride-2 / kernel-2 (de)convolutions and the additive skips.
//
// Replaces (reference code/networks/vnet.py):
//   nn.Conv3d(Cin, Cout, 2, stride=2)             :73   DownsamplingConvBlock
//   nn.ConvTranspose3d(Cin, Cout, 2, stride=2)    :100  UpsamplingDeconvBlock
//   x_up = x_up + skip                            :210-222
//
// A kernel-2 / stride-2 (de)convolution has non-overlapping windows, so it is exactly a 1x1x1
// convolution on a space-to-depth view: down = space_to_depth (C -> 8C channels at half resolution)
// followed by the MFMA 1x1x1 kernel of conv_fwd.hip with the weight viewed as [Cout][8*Cin]; up = the
// 1x1x1 kernel producing 8*Cout channels followed by depth_to_space (+ bias).  Only the pure data
// movement lives here (HBM-bound, explicit batch strides, float2 along x).
#include "common.h"

namespace {

struct S2DArgs {
    const float* src; long long src_bs;
    float* dst; long long dst_bs;
    const float* bias;      // depth-to-space only: added per fine channel (ConvTranspose3d bias)
    int N, C, D, H, W;      // FINE geometry: C channels at D x H x W (all even)
    int to_depth;           // 1: fine -> coarse (8C channels at D/2 x H/2 x W/2); 0: coarse -> fine
    int accumulate;
};

// one thread per fine x-pair; grid (ceil(H*W/2 / 256), D, N*C).  coarse channel = c*8 + kz*4 + ky*2 + kx
__global__ __launch_bounds__(256) void s2d_kernel(const S2DArgs a) {
    const int Wh = a.W >> 1;
    const int pl = blockIdx.x * 256 + threadIdx.x;
    if (pl >= a.H * Wh) return;
    const int z = blockIdx.y, nc = blockIdx.z;
    const int n = nc / a.C, c = nc - n * a.C;
    const int y = pl / Wh, xc = pl - y * Wh;
    const long long S = (long long)a.D * a.H * a.W, Sc = S >> 3;
    const int Hc = a.H >> 1;
    const int zc = z >> 1, yc = y >> 1, kz = z & 1, ky = y & 1;
    const long long fine = (long long)c * S + ((long long)z * a.H + y) * a.W + 2 * xc;
    const long long coarse0 = (long long)(c * 8 + kz * 4 + ky * 2) * Sc + ((long long)zc * Hc + yc) * Wh + xc;
    if (a.to_depth) {
        const float2 v = *reinterpret_cast<const float2*>(a.src + (long long)n * a.src_bs + fine);
        float* d = a.dst + (long long)n * a.dst_bs + coarse0;
        if (a.accumulate) { d[0] += v.x; d[Sc] += v.y; } else { d[0] = v.x; d[Sc] = v.y; }
    } else {
        const float* s = a.src + (long long)n * a.src_bs + coarse0;
        const float b = a.bias ? a.bias[c] : 0.f;
        float2 v = make_float2(s[0] + b, s[Sc] + b);
        float2* d = reinterpret_cast<float2*>(a.dst + (long long)n * a.dst_bs + fine);
        if (a.accumulate) { const float2 o = *d; v.x += o.x; v.y += o.y; }
        *d = v;
    }
}

// W % 8 == 0, volumes < 2^31 elements: a thread owns a 2 x 2 x 8 block of the fine tensor = 4 consecutive coarse voxels of
// the 8 coarse channels of c: float4 loads / stores only, 32-bit index arithmetic (the kernel above spends ~40 instructions
// per 8 bytes and is VALU-issue bound: 4 TB/s on V-Net's re-layouts).  grid = (ceil(Hc*Wc/4 / 256), Dc, N*C)
__global__ __launch_bounds__(256) void s2d8_kernel(const S2DArgs a) {
    const int Hc = a.H >> 1, Wc = a.W >> 1, Wq = Wc >> 2;
    const int pl = blockIdx.x * 256 + threadIdx.x;
    if (pl >= Hc * Wq) return;
    const int zc = blockIdx.y, nc = blockIdx.z;
    const int n = nc / a.C, c = nc - n * a.C;
    const int yc = pl / Wq, xq = pl - yc * Wq;
    const unsigned S = (unsigned)(a.D * a.H * a.W), Sc = S >> 3;
    const unsigned co = (unsigned)((zc * Hc + yc) * Wc + xq * 4);
    const float* __restrict__ src = a.src + (long long)n * a.src_bs;
    float* __restrict__ dst = a.dst + (long long)n * a.dst_bs;
    const float b = (!a.to_depth && a.bias) ? a.bias[c] : 0.f;
#pragma unroll
    for (int kz = 0; kz < 2; ++kz)
#pragma unroll
        for (int ky = 0; ky < 2; ++ky) {
            const unsigned f = (unsigned)c * S + (unsigned)(((zc * 2 + kz) * a.H + (yc * 2 + ky)) * a.W + xq * 8);
            const unsigned c0 = (unsigned)(c * 8 + kz * 4 + ky * 2) * Sc + co;      // kx = 0; kx = 1: + Sc
            if (a.to_depth) {
                const float4 v0 = *reinterpret_cast<const float4*>(src + f), v1 = *reinterpret_cast<const float4*>(src + f + 4);
                float4 e = make_float4(v0.x, v0.z, v1.x, v1.z), o = make_float4(v0.y, v0.w, v1.y, v1.w);
                float4* d0 = reinterpret_cast<float4*>(dst + c0);
                float4* d1 = reinterpret_cast<float4*>(dst + c0 + Sc);
                if (a.accumulate) {
                    const float4 p = *d0, q = *d1;
                    e.x += p.x; e.y += p.y; e.z += p.z; e.w += p.w; o.x += q.x; o.y += q.y; o.z += q.z; o.w += q.w;
                }
                *d0 = e; *d1 = o;
            } else {
                const float4 e = *reinterpret_cast<const float4*>(src + c0), o = *reinterpret_cast<const float4*>(src + c0 + Sc);
                float4 v0 = make_float4(e.x + b, o.x + b, e.y + b, o.y + b), v1 = make_float4(e.z + b, o.z + b, e.w + b, o.w + b);
                float4* d0 = reinterpret_cast<float4*>(dst + f);
                if (a.accumulate) {
                    const float4 p = d0[0], q = d0[1];
                    v0.x += p.x; v0.y += p.y; v0.z += p.z; v0.w += p.w; v1.x += q.x; v1.y += q.y; v1.z += q.z; v1.w += q.w;
                }
                d0[0] = v0; d0[1] = v1;
            }
        }
}

// 2-D twin (nn.ConvTranspose2d(k=2, s=2) of the 2-D UNet's UpBlock(bilinear=False), reference networks/unet.py:76-78): fine
// [N][C][H][W] <-> coarse [N][4C][H/2][W/2], coarse channel = c*4 + ky*2 + kx.  One thread per fine x-pair (float2);
// grid (ceil(H*W/2 / 256), 1, N*C).
__global__ __launch_bounds__(256) void s2d2d_kernel(const S2DArgs a) {
    const int Wh = a.W >> 1, Hc = a.H >> 1;
    const int pl = blockIdx.x * 256 + threadIdx.x;
    if (pl >= a.H * Wh) return;
    const int nc = blockIdx.z;
    const int n = nc / a.C, c = nc - n * a.C;
    const int y = pl / Wh, xc = pl - y * Wh;
    const long long S = (long long)a.H * a.W, Sc = S >> 2;
    const long long fine = (long long)c * S + (long long)y * a.W + 2 * xc;
    const long long coarse0 = (long long)(c * 4 + (y & 1) * 2) * Sc + (long long)(y >> 1) * Wh + xc;
    (void)Hc;
    if (a.to_depth) {
        const float2 v = *reinterpret_cast<const float2*>(a.src + (long long)n * a.src_bs + fine);
        float* d = a.dst + (long long)n * a.dst_bs + coarse0;
        if (a.accumulate) { d[0] += v.x; d[Sc] += v.y; } else { d[0] = v.x; d[Sc] = v.y; }
    } else {
        const float* s = a.src + (long long)n * a.src_bs + coarse0;
        const float b = a.bias ? a.bias[c] : 0.f;
        float2 v = make_float2(s[0] + b, s[Sc] + b);
        float2* d = reinterpret_cast<float2*>(a.dst + (long long)n * a.dst_bs + fine);
        if (a.accumulate) { const float2 o = *d; v.x += o.x; v.y += o.y; }
        *d = v;
    }
}

// out = a (+ b); dense (C, S), batch strides free; S % 4 == 0
__global__ __launch_bounds__(256) void add_kernel(const float* __restrict__ a, long long a_bs,
                                                  const float* __restrict__ b, long long b_bs,
                                                  float* __restrict__ out, long long o_bs, long long CS4) {
    const int n = blockIdx.y;
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < CS4; i += (long long)gridDim.x * 256) {
        float4 v = *reinterpret_cast<const float4*>(a + (long long)n * a_bs + i * 4);
        if (b) {
            const float4 w = *reinterpret_cast<const float4*>(b + (long long)n * b_bs + i * 4);
            v.x += w.x; v.y += w.y; v.z += w.z; v.w += w.w;
        }
        *reinterpret_cast<float4*>(out + (long long)n * o_bs + i * 4) = v;
    }
}

}  // namespace

// to_depth = 1: src = fine [N][C][D][H][W] -> dst = coarse [N][8C][D/2][H/2][W/2];
// to_depth = 0: src = coarse -> dst = fine (+ bias[C]).  N, C, D, H, W always describe the FINE tensor.
extern "C" int mis_space_to_depth2(const float* src, long long src_bs, float* dst, long long dst_bs,
                                   const float* bias, int N, int C, int D, int H, int W, int to_depth,
                                   int accumulate, hipStream_t stream) {
    if (!src || !dst || N <= 0 || C <= 0 || D <= 0 || H <= 0 || W <= 0) return MIS_ERR_ARG;
    if ((D | H | W) & 1) return MIS_ERR_UNSUPPORTED;
    const float* fine = to_depth ? src : dst;
    const long long fine_bs = to_depth ? src_bs : dst_bs;
    if ((fine_bs & 1) || ((uintptr_t)fine & 7)) return MIS_ERR_UNSUPPORTED;
    if ((long long)N * C > 65535 || D > 65535) return MIS_ERR_UNSUPPORTED;
    S2DArgs a{src, src_bs, dst, dst_bs, bias, N, C, D, H, W, to_depth, accumulate};
    const long long Sall = (long long)C * D * H * W;
    if (W % 8 == 0 && Sall < (1LL << 31) && !(src_bs & 3) && !(dst_bs & 3) && !((uintptr_t)src & 15) && !((uintptr_t)dst & 15)) {
        hipLaunchKernelGGL(s2d8_kernel, dim3(((H / 2) * (W / 8) + 255) / 256, D / 2, N * C), dim3(256), 0, stream, a);
        return mis_launch_status();
    }
    hipLaunchKernelGGL(s2d_kernel, dim3((H * (W / 2) + 255) / 256, D, N * C), dim3(256), 0, stream, a);
    return mis_launch_status();
}

// 2-D: to_depth = 1: fine [N][C][H][W] -> coarse [N][4C][H/2][W/2]; to_depth = 0: coarse -> fine (+ bias[C]).
extern "C" int mis_space_to_depth2d(const float* src, long long src_bs, float* dst, long long dst_bs, const float* bias,
                                    int N, int C, int H, int W, int to_depth, int accumulate, hipStream_t stream) {
    if (!src || !dst || N <= 0 || C <= 0 || H <= 0 || W <= 0) return MIS_ERR_ARG;
    if ((H | W) & 1) return MIS_ERR_UNSUPPORTED;
    const float* fine = to_depth ? src : dst;
    const long long fine_bs = to_depth ? src_bs : dst_bs;
    if ((fine_bs & 1) || ((uintptr_t)fine & 7)) return MIS_ERR_UNSUPPORTED;
    if ((long long)N * C > 65535) return MIS_ERR_UNSUPPORTED;
    S2DArgs a{src, src_bs, dst, dst_bs, bias, N, C, 1, H, W, to_depth, accumulate};
    hipLaunchKernelGGL(s2d2d_kernel, dim3((H * (W / 2) + 255) / 256, 1, N * C), dim3(256), 0, stream, a);
    return mis_launch_status();
}

// out[n][c][s] = a[n][c][s] (+ b[n][c][s] when b != NULL)
extern "C" int mis_add(const float* a, long long a_bs, const float* b, long long b_bs, float* out, long long o_bs,
                       int N, int C, long long S, hipStream_t stream) {
    if (!a || !out || N <= 0 || C <= 0 || S <= 0) return MIS_ERR_ARG;
    const long long CS = (long long)C * S;
    if (CS % 4 || a_bs % 4 || o_bs % 4 || (b && b_bs % 4) || ((uintptr_t)a & 15) || ((uintptr_t)out & 15) ||
        (b && ((uintptr_t)b & 15)))
        return MIS_ERR_UNSUPPORTED;
    long long bx = mis_cdiv(CS / 4, 256);
    if (bx > 2048) bx = 2048;
    hipLaunchKernelGGL(add_kernel, dim3((unsigned)bx, N), dim3(256), 0, stream, a, a_bs, b, b_bs, out, o_bs, CS / 4);
    return mis_launch_status();
}
