// Fused normalisation + activation + dropout (forward and backward) for the
// conv blocks of the Mean-Teacher step.
//
// Replaces (reference, train mode):
//   2D  nn.BatchNorm2d -> nn.LeakyReLU(0.01) -> nn.Dropout(p)   code/networks/unet.py:38-40,42-43
//   3D  nn.InstanceNorm3d(affine=False) -> nn.ReLU [-> nn.Dropout(0.3)]
//                                                   code/networks/utils.py:105-106,108-109; unet_3D.py:61-62,85,90
// Semantics: biased variance for normalisation, eps inside
// the sqrt, BatchNorm running stats updated with momentum and the unbiased
// variance, inverted dropout.
//
// HBM-bound.  Statistics are a two-stage fixed-order tree (float4 loads, wave
// shuffles, double in the last stage) -> deterministic, no atomics.  The apply
// pass fuses scale/shift, activation and the Philox dropout mask in one
// read+write; the mask is never stored (backward regenerates it from
// (seed, offset, salt, element index)).
//
// A "group" is what one mean/variance is taken over: per channel over (N, S)
// for BatchNorm (per_sample = 0), per (n, c) over S for InstanceNorm (per_sample = 1).
#include "common.h"
#include <type_traits>

namespace {

struct Geo {
    int N, C, per_sample;
    long long S;      // D*H*W, multiple of 4
    long long x_bs;   // batch stride of x (elements)
    int P;            // splits of one (group, chunk)
    int nchunks;      // chunks per group: N (batch norm) or 1 (instance norm)
    int G;            // groups
    int cg;           // per_sample only: channels sharing one (mean, rstd): 1 = InstanceNorm, C/16 = GroupNorm(16)
};

__host__ __device__ inline int pick_P(long long S) {
    long long p = (S + 16383) / 16384;
    if (p < 1) p = 1;
    if (p > 32) p = 32;
    return (int)p;
}

Geo make_geo(int N, int C, long long S, long long x_bs, int per_sample, int cg = 1) {
    Geo g;
    g.N = N; g.C = C; g.S = S; g.x_bs = x_bs; g.per_sample = per_sample; g.cg = cg;
    g.P = pick_P(S);
    g.nchunks = per_sample ? 1 : N;
    g.G = per_sample ? N * C : C;
    return g;
}

// residual branch of y = act(norm(x) + r) (MONAI UnetResBlock: conv-IN (+ shortcut) -> LeakyReLU): r read on the load path
// of all three passes, its gradient dr (+)= dz written by the backward apply
// post != 0: y = act(norm(x)) + r instead (V-Net's x_up + skip, vnet.py:210-222): the activation's sign does not involve r
struct ResArgs {
    const float* r; long long r_bs;
    float* dr; long long dr_bs; int dr_acc;
    int post;
};

struct DropCfg {
    float p;                   // drop probability (0 = off)
    unsigned salt;             // per-layer stream id
    const MisStepState* st;    // device seed/offset (required when p > 0 and mask == nullptr)
    const float* mask;         // optional explicit scale mask, contiguous [N][C][S] (parity tests)
};

// scale factors (0 or 1/(1-p)) for the 4 elements starting at logical index idx (multiple of 4)
__device__ __forceinline__ void drop_scale4(const DropCfg& d, unsigned long long idx, unsigned chan, float s[4]) {
    if (d.mask) {
        const float4 m = *reinterpret_cast<const float4*>(d.mask + idx);
        s[0] = m.x; s[1] = m.y; s[2] = m.z; s[3] = m.w;
        return;
    }
    const unsigned long long seed = d.st->seed, off = d.st->offset;
    const float keep = 1.f / (1.f - d.p);
    uint32_t r[4];
    if (d.salt >> 31) {
        // channel mode (nn.Dropout3d, reference vnet.py:177): one Bernoulli draw per (n, c) feature map
        mis_philox4((uint32_t)chan, 0x3D0D3D0Du, d.salt, (uint32_t)off, (uint32_t)seed,
                    (uint32_t)(seed >> 32) ^ (uint32_t)(off >> 32), r);
        const float v = mis_u01(r[0]) >= d.p ? keep : 0.f;
        s[0] = s[1] = s[2] = s[3] = v;
        return;
    }
    const unsigned long long u = idx >> 2;
    mis_philox4((uint32_t)u, (uint32_t)(u >> 32), d.salt, (uint32_t)off, (uint32_t)seed,
                (uint32_t)(seed >> 32) ^ (uint32_t)(off >> 32), r);
#pragma unroll
    for (int i = 0; i < 4; ++i) s[i] = mis_u01(r[i]) >= d.p ? keep : 0.f;
}

// Optional second source of the incoming gradient: the activation also feeds a 2x max-pool (reference unet_3D.py:35-47
// conv_k -> maxpool_k, unet.py:56 DownBlock) whose backward scatters dpool[cell] to the argmax element of each window.
// Instead of a separate read-modify-write pass over the full-resolution gradient (mis_maxpool2_bwd with accumulate), the
// two passes of the normalisation backward add that term on their load path: da_total = da + (idx[cell] == local ? dpool : 0).
struct PoolGrad {
    const float* dp; long long dp_bs;      // dpool [N][C][So]; null: no pooled consumer
    const unsigned char* idx;              // argmax codes dz*4 + dy*2 + dx, [N*C][So]
    int H, W, pz;                          // fine geometry; pz = 2 (3-D) or 1 (2-D: D == 1)
    long long So;                          // pooled elements per channel
};

// the pool's contribution to the 4 gradient elements at linear index e (multiple of 4; W % 4 == 0: one row) of (n, c)
__device__ __forceinline__ void pool_grad4(const PoolGrad& pg, int n, int c, int C, unsigned e, float (&g)[4]) {
    const unsigned HW = (unsigned)(pg.H * pg.W);
    const unsigned z = e / HW, r = e - z * HW;
    const unsigned y = r / (unsigned)pg.W, x0 = r - y * (unsigned)pg.W;
    const unsigned zo = z / (unsigned)pg.pz;
    const unsigned o = (zo * ((unsigned)pg.H >> 1) + (y >> 1)) * ((unsigned)pg.W >> 1) + (x0 >> 1);     // even
    const unsigned local = (z - zo * (unsigned)pg.pz) * 4u + (y & 1u) * 2u;
    const float2 d = *reinterpret_cast<const float2*>(pg.dp + (long long)n * pg.dp_bs + (long long)c * pg.So + o);
    const unsigned code = *reinterpret_cast<const unsigned short*>(pg.idx + ((long long)n * C + c) * pg.So + o);
    const unsigned c0 = code & 0xffu, c1 = code >> 8;
    g[0] = c0 == local ? d.x : 0.f;
    g[1] = c0 == local + 1u ? d.x : 0.f;
    g[2] = c1 == local ? d.y : 0.f;
    g[3] = c1 == local + 1u ? d.y : 0.f;
}

// block-wide sum of two doubles (256 threads), result in thread 0; fixed tree
__device__ __forceinline__ void block_sum2_d(double (&v)[2], double* red) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    v[0] = mis_wave_sum_d(v[0]); v[1] = mis_wave_sum_d(v[1]);
    __syncthreads();
    if (lane == 0) { red[wave * 2] = v[0]; red[wave * 2 + 1] = v[1]; }
    __syncthreads();
    if (threadIdx.x == 0) {
        v[0] = ((red[0] + red[2]) + red[4]) + red[6];
        v[1] = ((red[1] + red[3]) + red[5]) + red[7];
    }
}

// ---------------- statistics ----------------
// grid = (P, nchunks, G); partial[(g*nchunks + k)*P + p] = (sum, sumsq)
// The stand-alone pass (layers whose producer is not a conv with fused statistics, and every GroupNorm) accumulates in
// double, as torch's CPU kernels do (at::acc_type<float>): the pass is HBM-bound, the fp64 adds are free
__global__ __launch_bounds__(256) void stats_partial_kernel(const float* __restrict__ x, Geo g,
                                                            float2* __restrict__ part) {
    __shared__ double red[8];
    const int p = blockIdx.x, k = blockIdx.y, grp = blockIdx.z;
    const int n = g.per_sample ? grp / g.C : k;
    const int c = g.per_sample ? grp % g.C : grp;
    const float* __restrict__ base = x + (long long)n * g.x_bs + (long long)c * g.S;
    const long long units = g.S >> 2;
    const long long per = (units + g.P - 1) / g.P;
    const long long u0 = p * per, u1 = (u0 + per < units) ? u0 + per : units;
    double v[2] = {0.0, 0.0};
    for (long long u = u0 + threadIdx.x; u < u1; u += 256) {
        const float4 q = *reinterpret_cast<const float4*>(base + u * 4);
        v[0] += ((double)q.x + (double)q.y) + ((double)q.z + (double)q.w);
        v[1] += ((double)q.x * q.x + (double)q.y * q.y) + ((double)q.z * q.z + (double)q.w * q.w);
    }
    block_sum2_d(v, red);
    if (threadIdx.x == 0) part[((long long)grp * g.nchunks + k) * g.P + p] = make_float2((float)v[0], (float)v[1]);
}

// one 256-thread block per group (the fused conv statistics leave thousands of partials per channel)
__global__ __launch_bounds__(256) void stats_final_kernel(const float2* __restrict__ part, Geo g, float eps,
                                                          float* __restrict__ mean, float* __restrict__ rstd,
                                                          float* running_mean, float* running_var,
                                                          long long* num_batches, float momentum, int tpg) {
    // tpg = threads per group: 64 (four groups per block) or 256 (one group per block)
    __shared__ double red[8];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int grp = tpg == 256 ? (int)blockIdx.x : (int)blockIdx.x * 4 + wave;
    const int t = tpg == 256 ? (int)threadIdx.x : lane;
    const int np = g.nchunks * g.P;
    double s = 0.0, ss = 0.0;
    if (grp < g.G) {
        const float2* __restrict__ pg = part + (long long)grp * np;
        int i = t;
        for (; i + 3 * tpg < np; i += 4 * tpg) {   // four independent loads in flight
            const float2 q0 = pg[i], q1 = pg[i + tpg], q2 = pg[i + 2 * tpg], q3 = pg[i + 3 * tpg];
            s += q0.x; ss += q0.y; s += q1.x; ss += q1.y; s += q2.x; ss += q2.y; s += q3.x; ss += q3.y;
        }
        for (; i < np; i += tpg) {
            const float2 q = pg[i];
            s += q.x; ss += q.y;
        }
    }
    s = mis_wave_sum_d(s); ss = mis_wave_sum_d(ss);
    if (tpg == 256) {
        if (lane == 0) { red[wave] = s; red[4 + wave] = ss; }
        __syncthreads();
        s = ((red[0] + red[1]) + red[2]) + red[3];
        ss = ((red[4] + red[5]) + red[6]) + red[7];
    }
    if (grp < g.G && t == 0) {
        const double E = (double)g.nchunks * (double)g.S;
        const double m = s / E;
        double var = ss / E - m * m;
        if (var < 0.0) var = 0.0;
        mean[grp] = (float)m;
        rstd[grp] = (float)(1.0 / sqrt(var + (double)eps));
        if (running_mean && !g.per_sample) {
            const double unb = E > 1.0 ? var * E / (E - 1.0) : var;
            running_mean[grp] = (float)((1.0 - momentum) * running_mean[grp] + momentum * m);
            running_var[grp] = (float)((1.0 - momentum) * running_var[grp] + momentum * unb);
            if (num_batches && grp == 0) *num_batches += 1;
        }
    }
}

void launch_stats_final(const float2* part, const Geo& g, float eps, float* mean, float* rstd, float* running_mean,
                        float* running_var, long long* num_batches, float momentum, hipStream_t stream) {
    const int tpg = g.nchunks * g.P > 256 ? 256 : 64;
    hipLaunchKernelGGL(stats_final_kernel, dim3(tpg == 256 ? g.G : (g.G + 3) / 4), dim3(256), 0, stream, part, g, eps,
                       mean, rstd, running_mean, running_var, num_batches, momentum, tpg);
}

// one wave per channel: plain sum of the partials' .x
__global__ __launch_bounds__(256) void channel_sum_final_kernel(const float2* __restrict__ part, Geo g,
                                                                float* __restrict__ out, int accumulate) {
    const int grp = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (grp >= g.G) return;
    const int np = g.nchunks * g.P;
    double s = 0.0;
    for (int i = lane; i < np; i += 64) s += part[(long long)grp * np + i].x;
    s = mis_wave_sum_d(s);
    if (lane == 0) out[grp] = accumulate ? out[grp] + (float)s : (float)s;
}

__global__ void running_to_stats_kernel(const float* __restrict__ rm, const float* __restrict__ rv, float eps,
                                        float* __restrict__ mean, float* __restrict__ rstd, int C) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < C) { mean[c] = rm[c]; rstd[c] = 1.f / sqrtf(rv[c] + eps); }
}

// ---------------- forward apply ----------------
// grid = (ceil(S/4 / (256*U)), C, N)
constexpr int APPLY_U = 4;

template <bool RES>
__global__ __launch_bounds__(256) void apply_fwd_kernel(const float* __restrict__ x, Geo g,
                                                        const float* __restrict__ mean,
                                                        const float* __restrict__ rstd,
                                                        const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, float slope, DropCfg d,
                                                        float* __restrict__ y, long long y_bs, ResArgs ra) {
    const int c = blockIdx.y, n = blockIdx.z;
    const int grp = g.per_sample ? (n * g.C + c) / g.cg : c;
    const float sc = (gamma ? gamma[c] : 1.f) * rstd[grp];
    const float sh = (beta ? beta[c] : 0.f) - mean[grp] * sc;
    const float* __restrict__ xb = x + (long long)n * g.x_bs + (long long)c * g.S;
    float* __restrict__ yb = y + (long long)n * y_bs + (long long)c * g.S;
    const unsigned long long lbase = ((unsigned long long)n * g.C + c) * g.S;
    const long long units = g.S >> 2;
    const bool drop = d.p > 0.f;
#pragma unroll
    for (int i = 0; i < APPLY_U; ++i) {
        const long long u = ((long long)blockIdx.x * APPLY_U + i) * 256 + threadIdx.x;
        if (u >= units) break;
        const float4 q = *reinterpret_cast<const float4*>(xb + u * 4);
        float v[4] = {q.x * sc + sh, q.y * sc + sh, q.z * sc + sh, q.w * sc + sh};
        float4 rq = make_float4(0.f, 0.f, 0.f, 0.f);
        if constexpr (RES) {
            rq = *reinterpret_cast<const float4*>(ra.r + (long long)n * ra.r_bs + (long long)c * g.S + u * 4);
            if (!ra.post) { v[0] += rq.x; v[1] += rq.y; v[2] += rq.z; v[3] += rq.w; }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = v[j] > 0.f ? v[j] : v[j] * slope;
        if constexpr (RES) {
            if (ra.post) { v[0] += rq.x; v[1] += rq.y; v[2] += rq.z; v[3] += rq.w; }
        }
        if (drop) {
            float s[4];
            drop_scale4(d, lbase + u * 4, (unsigned)(n * g.C + c), s);
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] *= s[j];
        }
        *reinterpret_cast<float4*>(yb + u * 4) = make_float4(v[0], v[1], v[2], v[3]);
    }
}

// Forward apply for an activation that feeds a 2x max-pool (reference unet_3D.py:35-47 conv_k -> maxpool_k, unet.py:56): the
// thread that writes a 2 (z) x 2 (y) x 8 (x) block of the activation (1 x 2 x 8 in 2-D) also takes its four window maxima
// and argmax codes (mis_maxpool2_fwd's: dz*4 + dy*2 + dx, first maximum wins, NaN propagates) -- the pooling pass's read
// of the full-resolution activation is gone.  grid = (ceil(Ho*Wo/4 / 256), Do, N*C); W % 8 == 0.
template <int PZ>
__global__ __launch_bounds__(256) void apply_fwd_pool_kernel(const float* __restrict__ x, Geo g,
                                                             const float* __restrict__ mean,
                                                             const float* __restrict__ rstd,
                                                             const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, float slope, DropCfg d,
                                                             float* __restrict__ y, long long y_bs,
                                                             float* __restrict__ pooled, long long p_bs,
                                                             unsigned char* __restrict__ idx, int H, int W) {
    const int Ho = H >> 1, Wo = W >> 1, Wq = Wo >> 2;
    const int pl = blockIdx.x * 256 + threadIdx.x;
    if (pl >= Ho * Wq) return;
    const int zo = blockIdx.y, nc = blockIdx.z;
    const int n = nc / g.C, c = nc - n * g.C;
    const int yo = pl / Wq, xq = pl - yo * Wq;
    const int grp = g.per_sample ? (n * g.C + c) / g.cg : c;
    const float sc = (gamma ? gamma[c] : 1.f) * rstd[grp];
    const float sh = (beta ? beta[c] : 0.f) - mean[grp] * sc;
    const float* __restrict__ xb = x + (long long)n * g.x_bs + (long long)c * g.S;
    float* __restrict__ yb = y + (long long)n * y_bs + (long long)c * g.S;
    const unsigned long long lbase = ((unsigned long long)n * g.C + c) * g.S;
    const bool drop = d.p > 0.f;
    float best[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    unsigned bi[4] = {0, 0, 0, 0};
#pragma unroll
    for (int dz = 0; dz < PZ; ++dz)
#pragma unroll
        for (int dy = 0; dy < 2; ++dy) {
            const unsigned e = (unsigned)(((zo * PZ + dz) * H + (yo * 2 + dy)) * W + xq * 8);
            float v[8];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const float4 q = *reinterpret_cast<const float4*>(xb + e + 4 * h);
                float t[4] = {q.x * sc + sh, q.y * sc + sh, q.z * sc + sh, q.w * sc + sh};
#pragma unroll
                for (int j = 0; j < 4; ++j) t[j] = t[j] > 0.f ? t[j] : t[j] * slope;
                if (drop) {
                    float s4[4];
                    drop_scale4(d, lbase + e + 4 * h, (unsigned)(n * g.C + c), s4);
#pragma unroll
                    for (int j = 0; j < 4; ++j) t[j] *= s4[j];
                }
                *reinterpret_cast<float4*>(yb + e + 4 * h) = make_float4(t[0], t[1], t[2], t[3]);
#pragma unroll
                for (int j = 0; j < 4; ++j) v[4 * h + j] = t[j];
            }
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int dx = 0; dx < 2; ++dx) {
                    const float t = v[2 * j + dx];
                    if (t > best[j] || t != t) { best[j] = t; bi[j] = dz * 4 + dy * 2 + dx; }
                }
        }
    const long long So = (long long)gridDim.y * Ho * Wo;
    const unsigned o = (unsigned)((zo * Ho + yo) * Wo + xq * 4);
    *reinterpret_cast<float4*>(pooled + (long long)n * p_bs + (long long)c * So + o) =
        make_float4(best[0], best[1], best[2], best[3]);
    *reinterpret_cast<unsigned*>(idx + (long long)nc * So + o) = bi[0] | (bi[1] << 8) | (bi[2] << 16) | (bi[3] << 24);
}

// ---------------- backward ----------------
// dz = da * dropscale * (z > 0 ? 1 : slope),  z = xhat*gamma + beta,  xhat = (x-mean)*rstd
// partial sums per group: s1 = sum dz, s2 = sum dz*xhat
// DACC: per-thread accumulation in double (GroupNorm: the reference's CPU kernels accumulate in double there, and its
// gradient noise is what the parity gates are measured against)
template <bool DACC, bool RES = false>
__global__ __launch_bounds__(256) void bwd_partial_kernel(const float* __restrict__ x, Geo g,
                                                          const float* __restrict__ da, long long da_bs,
                                                          const float* __restrict__ mean,
                                                          const float* __restrict__ rstd,
                                                          const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, float slope, DropCfg d,
                                                          float2* __restrict__ part, PoolGrad pg, ResArgs ra) {
    __shared__ double red[8];
    const int p = blockIdx.x, k = blockIdx.y, grp = blockIdx.z;
    const int n = g.per_sample ? grp / g.C : k;
    const int c = g.per_sample ? grp % g.C : grp;
    const int sg = g.per_sample ? grp / g.cg : grp;   // statistics group of this (n, c)
    const float m = mean[sg], rs = rstd[sg];
    const float ga = gamma ? gamma[c] : 1.f, be = beta ? beta[c] : 0.f;
    const float* __restrict__ xb = x + (long long)n * g.x_bs + (long long)c * g.S;
    const float* __restrict__ db = da ? da + (long long)n * da_bs + (long long)c * g.S : nullptr;
    const unsigned long long lbase = ((unsigned long long)n * g.C + c) * g.S;
    const long long units = g.S >> 2;
    const long long per = (units + g.P - 1) / g.P;
    const long long u0 = p * per, u1 = (u0 + per < units) ? u0 + per : units;
    const bool drop = d.p > 0.f;
    typedef typename std::conditional<DACC, double, float>::type acc_t;
    acc_t v[2] = {0, 0};
    for (long long u = u0 + threadIdx.x; u < u1; u += 256) {
        const float4 q = *reinterpret_cast<const float4*>(xb + u * 4);
        const float4 gq = db ? *reinterpret_cast<const float4*>(db + u * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
        const float xs[4] = {q.x, q.y, q.z, q.w};
        float gs[4] = {gq.x, gq.y, gq.z, gq.w};
        float rv[4] = {0.f, 0.f, 0.f, 0.f};
        if constexpr (RES) {
            const float4 rq = *reinterpret_cast<const float4*>(ra.r + (long long)n * ra.r_bs + (long long)c * g.S + u * 4);
            rv[0] = rq.x; rv[1] = rq.y; rv[2] = rq.z; rv[3] = rq.w;
        }
        if (pg.dp) {
            float pgr[4];
            pool_grad4(pg, n, c, g.C, (unsigned)(u * 4), pgr);
#pragma unroll
            for (int j = 0; j < 4; ++j) gs[j] += pgr[j];
        }
        if (drop) {
            float s[4];
            drop_scale4(d, lbase + u * 4, (unsigned)(n * g.C + c), s);
#pragma unroll
            for (int j = 0; j < 4; ++j) gs[j] *= s[j];
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float xh = (xs[j] - m) * rs;
            const float z = xh * ga + be + rv[j];
            const float dz = z > 0.f ? gs[j] : gs[j] * slope;
            v[0] += dz;
            v[1] += (acc_t)dz * (acc_t)xh;
        }
    }
    if constexpr (DACC) {
        block_sum2_d(v, red);
    } else {
        mis_block_sum<2>(v, reinterpret_cast<float*>(red));
    }
    if (threadIdx.x == 0) part[((long long)grp * g.nchunks + k) * g.P + p] = make_float2((float)v[0], (float)v[1]);
}

// one wave per group: sums[g] = (s1/E, s2/E); affine grads for batch norm
__global__ __launch_bounds__(256) void bwd_final_kernel(const float2* __restrict__ part, Geo g,
                                                        float2* __restrict__ sums, float* dgamma, float* dbeta,
                                                        int accumulate) {
    const int grp = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (grp >= g.G) return;
    const int np = g.nchunks * g.P;
    double s1 = 0.0, s2 = 0.0;
    for (int i = lane; i < np; i += 64) {
        const float2 q = part[(long long)grp * np + i];
        s1 += q.x; s2 += q.y;
    }
    s1 = mis_wave_sum_d(s1); s2 = mis_wave_sum_d(s2);
    if (lane == 0) {
        const double E = (double)g.nchunks * (double)g.S;
        sums[grp] = make_float2((float)(s1 / E), (float)(s2 / E));
        if (dgamma && !g.per_sample) {
            dgamma[grp] = accumulate ? dgamma[grp] + (float)s2 : (float)s2;
            dbeta[grp] = accumulate ? dbeta[grp] + (float)s1 : (float)s1;
        }
    }
}

// Second stage from the partials a Winograd data-gradient launch left (mis_conv3d_wino_dgrad_norm): per (n, c) and tile
// (sum dz, sum dz * x) with the RAW x; sum dz * xhat = rstd * (t2 - mean * s1).  InstanceNorm without affine: one wave
// per (n, c), sums[g] = (s1 / S, s2 / S).
__global__ __launch_bounds__(256) void bwd_final_raw_kernel(const float2* __restrict__ part, int G, int tiles, long long S,
                                                            const float* __restrict__ mean,
                                                            const float* __restrict__ rstd, float2* __restrict__ sums) {
    const int grp = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (grp >= G) return;
    double s1 = 0.0, t2 = 0.0;
    for (int i = lane; i < tiles; i += 64) {
        const float2 q = part[(long long)grp * tiles + i];
        s1 += q.x; t2 += q.y;
    }
    s1 = mis_wave_sum_d(s1); t2 = mis_wave_sum_d(t2);
    if (lane == 0) {
        const double E = (double)S;
        const double s2 = (double)rstd[grp] * (t2 - (double)mean[grp] * s1);
        sums[grp] = make_float2((float)(s1 / E), (float)(s2 / E));
    }
}

// GroupNorm backward, second stage.  part[(n*C + c)*P + p] = (sum dz, sum dz*xhat) of one channel of one sample
// (dz = gradient at the affine output).  With dxhat = gamma_c * dz:
//   sums[n*G + grp] = (sum_{c in grp} gamma_c * a_nc, sum gamma_c * b_nc) / (cg * S)      one wave per (n, grp)
__global__ __launch_bounds__(256) void gn_bwd_group_kernel(const float2* __restrict__ part, Geo g,
                                                           const float* __restrict__ gamma,
                                                           float2* __restrict__ sums) {
    const int G = g.C / g.cg;
    const int sg = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (sg >= g.N * G) return;
    const int n = sg / G, c0 = (sg - n * G) * g.cg;
    const int np = g.cg * g.P;
    const float2* __restrict__ pg = part + ((long long)n * g.C + c0) * g.P;
    double s1 = 0.0, s2 = 0.0;
    for (int i = lane; i < np; i += 64) {
        const float ga = gamma ? gamma[c0 + i / g.P] : 1.f;
        const float2 q = pg[i];
        s1 += (double)ga * q.x; s2 += (double)ga * q.y;
    }
    s1 = mis_wave_sum_d(s1); s2 = mis_wave_sum_d(s2);
    if (lane == 0) {
        const double E = (double)g.cg * (double)g.S;
        sums[sg] = make_float2((float)(s1 / E), (float)(s2 / E));
    }
}

//   dgamma[c] = sum_n b_nc, dbeta[c] = sum_n a_nc                                          one wave per channel
__global__ __launch_bounds__(256) void gn_bwd_affine_kernel(const float2* __restrict__ part, Geo g, float* dgamma,
                                                            float* dbeta, int accumulate) {
    const int c = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (c >= g.C) return;
    double s1 = 0.0, s2 = 0.0;
    for (int i = lane; i < g.N * g.P; i += 64) {
        const int n = i / g.P, p = i - n * g.P;
        const float2 q = part[((long long)n * g.C + c) * g.P + p];
        s1 += q.x; s2 += q.y;
    }
    s1 = mis_wave_sum_d(s1); s2 = mis_wave_sum_d(s2);
    if (lane == 0) {
        dgamma[c] = accumulate ? dgamma[c] + (float)s2 : (float)s2;
        dbeta[c] = accumulate ? dbeta[c] + (float)s1 : (float)s1;
    }
}

// kind 0: BatchNorm / InstanceNorm (sums = means of dz, dz*xhat; gamma is constant over the group and factors out)
// kind 1: GroupNorm (sums = means of gamma*dz, gamma*dz*xhat over the channel group)
// kind 2: no normalisation (mean = 0, rstd = 1, gamma = null): dx = dz
template <bool RES>
__global__ __launch_bounds__(256) void apply_bwd_kernel(const float* __restrict__ x, Geo g,
                                                        const float* __restrict__ da, long long da_bs,
                                                        const float* __restrict__ mean,
                                                        const float* __restrict__ rstd,
                                                        const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, float slope, DropCfg d,
                                                        const float2* __restrict__ sums, float* __restrict__ dx,
                                                        long long dx_bs, int kind, PoolGrad pg, ResArgs ra) {
    const int c = blockIdx.y, n = blockIdx.z;
    const int grp = g.per_sample ? (n * g.C + c) / g.cg : c;
    const float m = mean[grp], rs = rstd[grp];
    const float ga = gamma ? gamma[c] : 1.f, be = beta ? beta[c] : 0.f;
    const float2 sm = kind == 2 ? make_float2(0.f, 0.f) : sums[grp];
    const float k = ga * rs;
    const float* __restrict__ xb = x + (long long)n * g.x_bs + (long long)c * g.S;
    const float* __restrict__ db = da ? da + (long long)n * da_bs + (long long)c * g.S : nullptr;
    float* __restrict__ ob = dx + (long long)n * dx_bs + (long long)c * g.S;
    const unsigned long long lbase = ((unsigned long long)n * g.C + c) * g.S;
    const long long units = g.S >> 2;
    const bool drop = d.p > 0.f;
#pragma unroll
    for (int i = 0; i < APPLY_U; ++i) {
        const long long u = ((long long)blockIdx.x * APPLY_U + i) * 256 + threadIdx.x;
        if (u >= units) break;
        const float4 q = *reinterpret_cast<const float4*>(xb + u * 4);
        const float4 gq = db ? *reinterpret_cast<const float4*>(db + u * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
        const float xs[4] = {q.x, q.y, q.z, q.w};
        float gs[4] = {gq.x, gq.y, gq.z, gq.w};
        if (pg.dp) {
            float pgr[4];
            pool_grad4(pg, n, c, g.C, (unsigned)(u * 4), pgr);
#pragma unroll
            for (int j = 0; j < 4; ++j) gs[j] += pgr[j];
        }
        if (drop) {
            float s[4];
            drop_scale4(d, lbase + u * 4, (unsigned)(n * g.C + c), s);
#pragma unroll
            for (int j = 0; j < 4; ++j) gs[j] *= s[j];
        }
        float o[4], rv[4] = {0.f, 0.f, 0.f, 0.f}, dzs[4];
        if constexpr (RES) {
            if (!ra.post) {
                const float4 rq = *reinterpret_cast<const float4*>(ra.r + (long long)n * ra.r_bs + (long long)c * g.S + u * 4);
                rv[0] = rq.x; rv[1] = rq.y; rv[2] = rq.z; rv[3] = rq.w;
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float xh = (xs[j] - m) * rs;
            const float z = xh * ga + be + rv[j];
            const float dz = z > 0.f ? gs[j] : gs[j] * slope;
            dzs[j] = (RES && ra.post) ? gs[j] : dz;
            o[j] = kind == 1 ? rs * (ga * dz - sm.x - xh * sm.y) : k * (dz - sm.x - xh * sm.y);
        }
        *reinterpret_cast<float4*>(ob + u * 4) = make_float4(o[0], o[1], o[2], o[3]);
        if constexpr (RES) {
            float* const dp = ra.dr + (long long)n * ra.dr_bs + (long long)c * g.S + u * 4;
            float4 w = make_float4(dzs[0], dzs[1], dzs[2], dzs[3]);
            if (ra.dr_acc) {
                const float4 old = *reinterpret_cast<const float4*>(dp);
                w.x += old.x; w.y += old.y; w.z += old.z; w.w += old.w;
            }
            *reinterpret_cast<float4*>(dp) = w;
        }
    }
}

bool aligned16(const void* p) { return ((uintptr_t)p & 15) == 0; }

// ---------------- volumes whose voxel count is not a multiple of 4 (3 x 3 x 3: the deepest level of SwinUNETR at 96^3) ----
// The float4 kernels above want S % 4 == 0.  These scalar forms serve the few-voxel levels: one 256-thread workgroup per
// statistics group (channel for BatchNorm: N * S elements; (n, c) for InstanceNorm: S elements), sums in double.
// No dropout, no GroupNorm, no fused pooling on this path.
__device__ __forceinline__ void small_group(const Geo& g, int grp, int& n0, int& nn, int& c) {
    if (g.per_sample) { n0 = grp / g.C; nn = 1; c = grp % g.C; } else { n0 = 0; nn = g.N; c = grp; }
}

__global__ __launch_bounds__(256) void stats_small_kernel(const float* __restrict__ x, Geo g, float eps,
                                                          float* __restrict__ mean, float* __restrict__ rstd,
                                                          float* running_mean, float* running_var,
                                                          long long* num_batches, float momentum) {
    __shared__ double red[8];
    int n0, nn, c;
    small_group(g, blockIdx.x, n0, nn, c);
    double v[2] = {0.0, 0.0};
    for (long long i = threadIdx.x; i < (long long)nn * g.S; i += 256) {
        const float q = x[(long long)(n0 + i / g.S) * g.x_bs + (long long)c * g.S + i % g.S];
        v[0] += q; v[1] += (double)q * q;
    }
    block_sum2_d(v, red);
    if (threadIdx.x == 0) {
        const double E = (double)nn * (double)g.S, m = v[0] / E;
        double var = v[1] / E - m * m;
        if (var < 0.0) var = 0.0;
        mean[blockIdx.x] = (float)m;
        rstd[blockIdx.x] = (float)(1.0 / sqrt(var + (double)eps));
        if (running_mean && !g.per_sample) {
            const double unb = E > 1.0 ? var * E / (E - 1.0) : var;
            running_mean[c] = (float)((1.0 - momentum) * running_mean[c] + momentum * m);
            running_var[c] = (float)((1.0 - momentum) * running_var[c] + momentum * unb);
            if (num_batches && c == 0) *num_batches += 1;
        }
    }
}

__global__ __launch_bounds__(256) void apply_fwd_small_kernel(const float* __restrict__ x, Geo g,
                                                              const float* __restrict__ mean,
                                                              const float* __restrict__ rstd,
                                                              const float* __restrict__ gamma,
                                                              const float* __restrict__ beta, float slope,
                                                              float* __restrict__ y, long long y_bs) {
    const long long total = (long long)g.N * g.C * g.S;
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long s = i % g.S;
        const int c = (int)((i / g.S) % g.C), n = (int)(i / (g.S * g.C));
        const int grp = g.per_sample ? n * g.C + c : c;
        const float sc = (gamma ? gamma[c] : 1.f) * rstd[grp];
        const float sh = (beta ? beta[c] : 0.f) - mean[grp] * sc;
        const float v = x[(long long)n * g.x_bs + (long long)c * g.S + s] * sc + sh;
        y[(long long)n * y_bs + (long long)c * g.S + s] = v > 0.f ? v : v * slope;
    }
}

// backward of one statistics group in one workgroup: sums (double), then dx; kind as apply_bwd_kernel (0 norm, 2 none)
__global__ __launch_bounds__(256) void bwd_small_kernel(const float* __restrict__ x, Geo g,
                                                        const float* __restrict__ da, long long da_bs,
                                                        const float* __restrict__ mean, const float* __restrict__ rstd,
                                                        const float* __restrict__ gamma, const float* __restrict__ beta,
                                                        float slope, float* __restrict__ dx, long long dx_bs,
                                                        float* dgamma, float* dbeta, int accumulate, int kind) {
    __shared__ double red[8];
    __shared__ float sm[2];
    int n0, nn, c;
    small_group(g, blockIdx.x, n0, nn, c);
    const float m = kind == 2 ? 0.f : mean[blockIdx.x], rs = kind == 2 ? 1.f : rstd[blockIdx.x];      // kind 2: mean / rstd are [C] of (0, 1)
    const float ga = gamma ? gamma[c] : 1.f, be = beta ? beta[c] : 0.f;
    const long long cnt = (long long)nn * g.S;
    double v[2] = {0.0, 0.0};
    if (kind != 2) {
        for (long long i = threadIdx.x; i < cnt; i += 256) {
            const long long o = (long long)c * g.S + i % g.S;
            const int n = n0 + (int)(i / g.S);
            const float xh = (x[(long long)n * g.x_bs + o] - m) * rs;
            const float gq = da[(long long)n * da_bs + o];
            const float dz = (xh * ga + be) > 0.f ? gq : gq * slope;
            v[0] += dz; v[1] += (double)dz * xh;
        }
        block_sum2_d(v, red);
        if (threadIdx.x == 0) {
            sm[0] = (float)(v[0] / (double)cnt); sm[1] = (float)(v[1] / (double)cnt);
            if (dgamma && !g.per_sample) {
                dgamma[c] = accumulate ? dgamma[c] + (float)v[1] : (float)v[1];
                dbeta[c] = accumulate ? dbeta[c] + (float)v[0] : (float)v[0];
            }
        }
        __syncthreads();
    }
    const float k = ga * rs;
    for (long long i = threadIdx.x; i < cnt; i += 256) {
        const long long o = (long long)c * g.S + i % g.S;
        const int n = n0 + (int)(i / g.S);
        const float xh = (x[(long long)n * g.x_bs + o] - m) * rs;
        const float gq = da[(long long)n * da_bs + o];
        const float dz = (xh * ga + be) > 0.f ? gq : gq * slope;
        dx[(long long)n * dx_bs + o] = kind == 2 ? dz : k * (dz - sm[0] - xh * sm[1]);
    }
}

// geometry check of the scalar path
int check_geo_small(const void* x, int N, int C, long long S, long long x_bs) {
    if (!x || N <= 0 || C <= 0 || S <= 0) return MIS_ERR_ARG;
    if (x_bs < (long long)C * S) return MIS_ERR_ARG;
    if ((long long)N * C > 0x7fffffffLL / 4) return MIS_ERR_UNSUPPORTED;
    return MIS_OK;
}

int check_geo(const void* x, int N, int C, long long S, long long x_bs) {
    if (!x || N <= 0 || C <= 0 || S <= 0) return MIS_ERR_ARG;
    if (S % 4 != 0 || x_bs % 4 != 0 || !aligned16(x)) return MIS_ERR_UNSUPPORTED;
    if (x_bs < (long long)C * S) return MIS_ERR_ARG;
    if (C > 65535 || N > 65535) return MIS_ERR_UNSUPPORTED;
    return MIS_OK;
}

}  // namespace

// bytes of scratch needed by mis_norm_stats / mis_norm_act_bwd for this geometry
extern "C" long long mis_norm_workspace_bytes(int N, int C, long long S, int per_sample) {
    if (N <= 0 || C <= 0 || S <= 0) return MIS_ERR_ARG;
    const Geo g = make_geo(N, C, S, (long long)C * S, per_sample);
    // partials + per-group (s1/E, s2/E)
    return ((long long)g.G * g.nchunks * g.P + g.G) * (long long)sizeof(float2);
}

extern "C" int mis_norm_stats(const float* x, long long x_bs, int N, int C, long long S, int per_sample, float eps,
                              float* mean, float* rstd, float* running_mean, float* running_var,
                              long long* num_batches_tracked, float momentum, void* workspace,
                              long long workspace_bytes, hipStream_t stream) {
    if (S % 4 != 0) {      // few-voxel volumes: scalar kernels
        int st = check_geo_small(x, N, C, S, x_bs);
        if (st) return st;
        if (!mean || !rstd) return MIS_ERR_ARG;
        const Geo g = make_geo(N, C, S, x_bs, per_sample);
        hipLaunchKernelGGL(stats_small_kernel, dim3(g.G), dim3(256), 0, stream, x, g, eps, mean, rstd, running_mean,
                           running_var, num_batches_tracked, momentum);
        return mis_launch_status();
    }
    int st = check_geo(x, N, C, S, x_bs);
    if (st) return st;
    if (!mean || !rstd || !workspace) return MIS_ERR_ARG;
    const Geo g = make_geo(N, C, S, x_bs, per_sample);
    if (workspace_bytes < mis_norm_workspace_bytes(N, C, S, per_sample)) return MIS_ERR_WORKSPACE;
    float2* part = reinterpret_cast<float2*>(workspace);
    hipLaunchKernelGGL(stats_partial_kernel, dim3(g.P, g.nchunks, g.G), dim3(256), 0, stream, x, g, part);
    launch_stats_final(part, g, eps, mean, rstd, running_mean, running_var, num_batches_tracked, momentum, stream);
    return mis_launch_status();
}

// Same result from per-tile partial statistics written by the producing convolution (mis_conv_fwd_stats):
// part[group][i] = (sum, sumsq), i < nparts; BatchNorm groups = channels with nparts = N*T, InstanceNorm groups =
// (n, c) with nparts = T.  No pass over the activation at all.
extern "C" int mis_norm_stats_finalize(const float* part, int N, int C, long long S, int tiles, int per_sample,
                                       float eps, float* mean, float* rstd, float* running_mean, float* running_var,
                                       long long* num_batches_tracked, float momentum, hipStream_t stream) {
    if (!part || !mean || !rstd || N <= 0 || C <= 0 || S <= 0 || tiles <= 0) return MIS_ERR_ARG;
    Geo g{};
    g.N = N; g.C = C; g.per_sample = per_sample ? 1 : 0; g.S = S; g.x_bs = 0;
    g.P = tiles; g.nchunks = per_sample ? 1 : N; g.G = per_sample ? N * C : C;
    launch_stats_final(reinterpret_cast<const float2*>(part), g, eps, mean, rstd, running_mean, running_var,
                       num_batches_tracked, momentum, stream);
    return mis_launch_status();
}

// Conv bias gradient: out[c] (+)= sum over (N, S) of x[n][c][:]  (autograd of the bias add in
// nn.Conv2d/3d; only needed for convs that are NOT followed by a normalisation)
extern "C" int mis_channel_sum(const float* x, long long x_bs, int N, int C, long long S, float* out,
                               int accumulate, void* workspace, long long workspace_bytes,
                               hipStream_t stream) {
    int st = check_geo(x, N, C, S, x_bs);
    if (st) return st;
    if (!out || !workspace) return MIS_ERR_ARG;
    const Geo g = make_geo(N, C, S, x_bs, 0);
    if (workspace_bytes < mis_norm_workspace_bytes(N, C, S, 0)) return MIS_ERR_WORKSPACE;
    float2* part = reinterpret_cast<float2*>(workspace);
    hipLaunchKernelGGL(stats_partial_kernel, dim3(g.P, g.nchunks, g.G), dim3(256), 0, stream, x, g, part);
    hipLaunchKernelGGL(channel_sum_final_kernel, dim3((g.G + 3) / 4), dim3(256), 0, stream, part, g, out,
                       accumulate);
    return mis_launch_status();
}

// eval-mode BatchNorm: (mean, rstd) from the running buffers
extern "C" int mis_norm_stats_from_running(const float* running_mean, const float* running_var, float eps,
                                           float* mean, float* rstd, int C, hipStream_t stream) {
    if (!running_mean || !running_var || !mean || !rstd || C <= 0) return MIS_ERR_ARG;
    hipLaunchKernelGGL(running_to_stats_kernel, dim3((C + 255) / 256), dim3(256), 0, stream, running_mean,
                       running_var, eps, mean, rstd, C);
    return mis_launch_status();
}

// Generalised form: ``cg`` channels share one (mean, rstd) when per_sample != 0 (mean / rstd hold N * C/cg entries):
// cg = 1 is InstanceNorm, cg = C/16 is nn.GroupNorm(16, C) (reference code/networks/vnet.py:19-20) with its per-channel
// affine in gamma / beta.  Statistics of a GroupNorm come from mis_norm_stats on the same memory viewed as
// [N, C/cg, cg*S] with per_sample = 1 (the channels of a group are contiguous in NCDHW), or from the conv epilogue's
// per-(n, c, tile) partials through mis_norm_stats_finalize(part, N, C/cg, cg*S, cg*tiles, 1, ...).
extern "C" int mis_norm_act_fwd_g(const float* x, long long x_bs, float* y, long long y_bs, int N, int C,
                                  long long S, int per_sample, int cg, const float* mean, const float* rstd,
                                  const float* gamma, const float* beta, float slope, float drop_p,
                                  unsigned drop_salt, const MisStepState* state, const float* drop_mask,
                                  hipStream_t stream) {
    if (S % 4 != 0) {      // few-voxel volumes: scalar kernel (no dropout, no GroupNorm)
        int st = check_geo_small(x, N, C, S, x_bs);
        if (st) return st;
        if (!y || !mean || !rstd || y_bs < (long long)C * S) return MIS_ERR_ARG;
        if (cg != 1 || drop_p != 0.f) return MIS_ERR_UNSUPPORTED;
        const Geo g = make_geo(N, C, S, x_bs, per_sample, 1);
        long long b = mis_cdiv((long long)N * C * S, 256);
        hipLaunchKernelGGL(apply_fwd_small_kernel, dim3((unsigned)(b > 4096 ? 4096 : b)), dim3(256), 0, stream, x, g, mean,
                           rstd, gamma, beta, slope, y, y_bs);
        return mis_launch_status();
    }
    int st = check_geo(x, N, C, S, x_bs);
    if (st) return st;
    if (cg < 1 || C % cg != 0 || (cg > 1 && !per_sample)) return MIS_ERR_ARG;
    if (!y || !mean || !rstd || y_bs % 4 != 0 || !aligned16(y) || y_bs < (long long)C * S) return MIS_ERR_ARG;
    if (drop_p < 0.f || drop_p >= 1.f) return MIS_ERR_ARG;
    if (drop_p > 0.f && !state && !drop_mask) return MIS_ERR_ARG;
    const Geo g = make_geo(N, C, S, x_bs, per_sample, cg);
    DropCfg d{drop_p, drop_salt, state, drop_mask};
    const long long units = S >> 2;
    const unsigned gx = (unsigned)mis_cdiv(units, 256 * APPLY_U);
    hipLaunchKernelGGL(apply_fwd_kernel<false>, dim3(gx, C, N), dim3(256), 0, stream, x, g, mean, rstd, gamma, beta,
                       slope, d, y, y_bs, ResArgs{});
    return mis_launch_status();
}

extern "C" int mis_norm_act_fwd(const float* x, long long x_bs, float* y, long long y_bs, int N, int C,
                                long long S, int per_sample, const float* mean, const float* rstd,
                                const float* gamma, const float* beta, float slope, float drop_p,
                                unsigned drop_salt, const MisStepState* state, const float* drop_mask,
                                hipStream_t stream) {
    return mis_norm_act_fwd_g(x, x_bs, y, y_bs, N, C, S, per_sample, 1, mean, rstd, gamma, beta, slope, drop_p,
                              drop_salt, state, drop_mask, stream);
}

// mis_norm_act_fwd_g + mis_maxpool2_fwd in one pass: y = drop(act(norm(x))) [N][C][D][H][W], pooled = its 2x max-pool
// [N][C][D/2 (D > 1)][H/2][W/2] with the argmax codes idx [N*C][So].  W % 8 == 0, H even, D even or 1.
extern "C" int mis_norm_act_fwd_pool(const float* x, long long x_bs, float* y, long long y_bs, float* pooled,
                                     long long p_bs, unsigned char* idx, int N, int C, int D, int H, int W,
                                     int per_sample, int cg, const float* mean, const float* rstd, const float* gamma,
                                     const float* beta, float slope, float drop_p, unsigned drop_salt,
                                     const MisStepState* state, const float* drop_mask, hipStream_t stream) {
    if (D <= 0 || H <= 0 || W <= 0) return MIS_ERR_ARG;
    const long long S = (long long)D * H * W;
    int st = check_geo(x, N, C, S, x_bs);
    if (st) return st;
    if (cg < 1 || C % cg != 0 || (cg > 1 && !per_sample)) return MIS_ERR_ARG;
    if (!y || !pooled || !idx || !mean || !rstd || y_bs < (long long)C * S) return MIS_ERR_ARG;
    if (W % 8 || H % 2 || (D > 1 && D % 2) || S >= (1LL << 31)) return MIS_ERR_UNSUPPORTED;
    const int pz = D > 1 ? 2 : 1, Do = D / pz;
    const long long So = (long long)Do * (H / 2) * (W / 2);
    if (p_bs < (long long)C * So) return MIS_ERR_ARG;
    if (y_bs % 4 || p_bs % 4 || !aligned16(y) || !aligned16(pooled) || ((uintptr_t)idx & 3)) return MIS_ERR_UNSUPPORTED;
    if (drop_p < 0.f || drop_p >= 1.f || (drop_p > 0.f && !state && !drop_mask)) return MIS_ERR_ARG;
    if (Do > 65535 || (long long)N * C > 65535) return MIS_ERR_UNSUPPORTED;
    const Geo g = make_geo(N, C, S, x_bs, per_sample, cg);
    const DropCfg d{drop_p, drop_salt, state, drop_mask};
    const dim3 grid((unsigned)(((H / 2) * (W / 8) + 255) / 256), Do, N * C);
    if (pz == 2)
        hipLaunchKernelGGL(apply_fwd_pool_kernel<2>, grid, dim3(256), 0, stream, x, g, mean, rstd, gamma, beta, slope, d, y,
                           y_bs, pooled, p_bs, idx, H, W);
    else
        hipLaunchKernelGGL(apply_fwd_pool_kernel<1>, grid, dim3(256), 0, stream, x, g, mean, rstd, gamma, beta, slope, d, y,
                           y_bs, pooled, p_bs, idx, H, W);
    return mis_launch_status();
}

// Backward of mis_norm_act_fwd_g.  ``no_norm`` != 0: the layer is activation (+ dropout) only -- the reference's
// normalization='none' blocks -- mean must hold zeros, rstd ones, gamma / beta null; no reduction runs.
// Workspace: mis_norm_workspace_bytes(N, C, S, per_sample) (covers every cg).
namespace {
int norm_act_bwd_impl(const float* x, long long x_bs, const float* da, long long da_bs, float* dx, long long dx_bs,
                      int N, int C, long long S, int per_sample, int cg, int no_norm, const float* mean,
                      const float* rstd, const float* gamma, const float* beta, float slope, float drop_p,
                      unsigned drop_salt, const MisStepState* state, const float* drop_mask, float* dgamma,
                      float* dbeta, int accumulate_affine, void* workspace, long long workspace_bytes,
                      const PoolGrad& pg, hipStream_t stream, const ResArgs& ra = ResArgs{}) {
    if (S % 4 != 0) {
        if (ra.r) return MIS_ERR_UNSUPPORTED;      // few-voxel volumes: one workgroup per statistics group (no dropout, GroupNorm, pooling)
        int st = check_geo_small(x, N, C, S, x_bs);
        if (st) return st;
        if (!da || !dx || !mean || !rstd || da_bs < (long long)C * S || dx_bs < (long long)C * S) return MIS_ERR_ARG;
        if (cg != 1 || drop_p != 0.f || pg.dp || (per_sample && (gamma || beta))) return MIS_ERR_UNSUPPORTED;
        // no normalisation: every (n, c) is its own group of the element-wise pass
        const Geo g = make_geo(N, C, S, x_bs, no_norm ? 1 : per_sample, 1);
        hipLaunchKernelGGL(bwd_small_kernel, dim3(g.G), dim3(256), 0, stream, x, g, da, da_bs, mean, rstd, gamma, beta,
                           slope, dx, dx_bs, dgamma, dbeta, accumulate_affine, no_norm ? 2 : 0);
        return mis_launch_status();
    }
    int st = check_geo(x, N, C, S, x_bs);
    if (st) return st;
    if (cg < 1 || C % cg != 0 || (cg > 1 && !per_sample) || (no_norm && (gamma || beta))) return MIS_ERR_ARG;
    if ((!da && !pg.dp) || !dx || !mean || !rstd || !workspace) return MIS_ERR_ARG;
    if (dx_bs % 4 != 0 || !aligned16(dx) || (da && (da_bs % 4 != 0 || !aligned16(da)))) return MIS_ERR_UNSUPPORTED;
    if ((da && da_bs < (long long)C * S) || dx_bs < (long long)C * S) return MIS_ERR_ARG;
    if (drop_p < 0.f || drop_p >= 1.f) return MIS_ERR_ARG;
    if (drop_p > 0.f && !state && !drop_mask) return MIS_ERR_ARG;
    if (workspace_bytes < mis_norm_workspace_bytes(N, C, S, per_sample)) return MIS_ERR_WORKSPACE;
    const Geo g = make_geo(N, C, S, x_bs, per_sample, cg);
    DropCfg d{drop_p, drop_salt, state, drop_mask};
    float2* part = reinterpret_cast<float2*>(workspace);
    float2* sums = part + (long long)g.G * g.nchunks * g.P;
    const bool gn = per_sample && (cg > 1 || gamma);    // per-channel affine inside a per-sample group
    if (!no_norm) {
        if (cg > 1)
            hipLaunchKernelGGL(bwd_partial_kernel<true>, dim3(g.P, g.nchunks, g.G), dim3(256), 0, stream, x, g, da, da_bs,
                               mean, rstd, gamma, beta, slope, d, part, pg, ResArgs{});
        else if (ra.r && !ra.post)
            hipLaunchKernelGGL((bwd_partial_kernel<false, true>), dim3(g.P, g.nchunks, g.G), dim3(256), 0, stream, x, g, da,
                               da_bs, mean, rstd, gamma, beta, slope, d, part, pg, ra);
        else
            hipLaunchKernelGGL(bwd_partial_kernel<false>, dim3(g.P, g.nchunks, g.G), dim3(256), 0, stream, x, g, da, da_bs,
                               mean, rstd, gamma, beta, slope, d, part, pg, ResArgs{});
        if (gn) {
            hipLaunchKernelGGL(gn_bwd_group_kernel, dim3((N * (C / cg) + 3) / 4), dim3(256), 0, stream, part, g, gamma,
                               sums);
            if (dgamma && dbeta)
                hipLaunchKernelGGL(gn_bwd_affine_kernel, dim3((C + 3) / 4), dim3(256), 0, stream, part, g, dgamma,
                                   dbeta, accumulate_affine);
        } else {
            hipLaunchKernelGGL(bwd_final_kernel, dim3((g.G + 3) / 4), dim3(256), 0, stream, part, g, sums, dgamma,
                               dbeta, accumulate_affine);
        }
    }
    const long long units = S >> 2;
    const unsigned gx = (unsigned)mis_cdiv(units, 256 * APPLY_U);
    if (ra.r)
        hipLaunchKernelGGL(apply_bwd_kernel<true>, dim3(gx, C, N), dim3(256), 0, stream, x, g, da, da_bs, mean, rstd, gamma,
                           beta, slope, d, sums, dx, dx_bs, no_norm ? 2 : (gn ? 1 : 0), pg, ra);
    else
        hipLaunchKernelGGL(apply_bwd_kernel<false>, dim3(gx, C, N), dim3(256), 0, stream, x, g, da, da_bs, mean, rstd, gamma,
                           beta, slope, d, sums, dx, dx_bs, no_norm ? 2 : (gn ? 1 : 0), pg, ResArgs{});
    return mis_launch_status();
}
}  // namespace

extern "C" int mis_norm_act_bwd_g(const float* x, long long x_bs, const float* da, long long da_bs, float* dx,
                                  long long dx_bs, int N, int C, long long S, int per_sample, int cg, int no_norm,
                                  const float* mean, const float* rstd, const float* gamma, const float* beta,
                                  float slope, float drop_p, unsigned drop_salt, const MisStepState* state,
                                  const float* drop_mask, float* dgamma, float* dbeta, int accumulate_affine,
                                  void* workspace, long long workspace_bytes, hipStream_t stream) {
    if (!da) return MIS_ERR_ARG;
    return norm_act_bwd_impl(x, x_bs, da, da_bs, dx, dx_bs, N, C, S, per_sample, cg, no_norm, mean, rstd, gamma, beta,
                             slope, drop_p, drop_salt, state, drop_mask, dgamma, dbeta, accumulate_affine, workspace,
                             workspace_bytes, PoolGrad{}, stream);
}

// mis_norm_act_bwd_g for an activation [N][C][D][H][W] that ALSO feeds a 2x max-pool (mis_maxpool2_fwd's idx codes):
// the gradient arriving at the activation is da (the other consumers: skip connection; NULL: none) plus the pool's
// backward of dpool [N][C][D/2 (D > 1)][H/2][W/2], formed on the load path of both passes -- mis_maxpool2_bwd's
// read-modify-write of the full-resolution gradient is gone.  W % 4 == 0, H even, D even or 1.
extern "C" int mis_norm_act_bwd_pool(const float* x, long long x_bs, const float* da, long long da_bs,
                                     const float* dpool, long long dp_bs, const unsigned char* idx, float* dx,
                                     long long dx_bs, int N, int C, int D, int H, int W, int per_sample, int cg,
                                     const float* mean, const float* rstd, const float* gamma, const float* beta,
                                     float slope, float drop_p, unsigned drop_salt, const MisStepState* state,
                                     const float* drop_mask, float* dgamma, float* dbeta, int accumulate_affine,
                                     void* workspace, long long workspace_bytes, hipStream_t stream) {
    if (!dpool || !idx || D <= 0 || H <= 0 || W <= 0) return MIS_ERR_ARG;
    if (W % 4 || H % 2 || (D > 1 && D % 2)) return MIS_ERR_UNSUPPORTED;
    const long long S = (long long)D * H * W;
    if (S >= (1LL << 31)) return MIS_ERR_UNSUPPORTED;
    const int pz = D > 1 ? 2 : 1;
    const long long So = (long long)(D / pz) * (H / 2) * (W / 2);
    if (dp_bs < (long long)C * So || dp_bs % 2 || ((uintptr_t)dpool & 7) || ((uintptr_t)idx & 1)) return MIS_ERR_UNSUPPORTED;
    const PoolGrad pg{dpool, dp_bs, idx, H, W, pz, So};
    return norm_act_bwd_impl(x, x_bs, da, da_bs, dx, dx_bs, N, C, S, per_sample, cg, 0, mean, rstd, gamma, beta, slope,
                             drop_p, drop_salt, state, drop_mask, dgamma, dbeta, accumulate_affine, workspace,
                             workspace_bytes, pg, stream);
}

// The reduction half of mis_norm_act_bwd alone (no dropout): sums[group] = (mean of dz, mean of dz*xhat) as two floats per
// group (C groups, or N*C with per_sample), plus dgamma / dbeta for BatchNorm.  For consumers that form the gradient at
// the normalisation's input themselves (mis_conv_wgrad_cin1_norm).  Workspace: mis_norm_workspace_bytes.
extern "C" int mis_norm_act_bwd_sums(const float* x, long long x_bs, const float* da, long long da_bs, int N, int C,
                                     long long S, int per_sample, const float* mean, const float* rstd,
                                     const float* gamma, const float* beta, float slope, float* sums, float* dgamma,
                                     float* dbeta, int accumulate_affine, void* workspace, long long workspace_bytes,
                                     hipStream_t stream) {
    int st = check_geo(x, N, C, S, x_bs);
    if (st) return st;
    if (!da || !mean || !rstd || !sums || !workspace || (per_sample && (gamma || beta))) return MIS_ERR_ARG;
    if (da_bs % 4 != 0 || !aligned16(da) || ((uintptr_t)sums & 7)) return MIS_ERR_UNSUPPORTED;
    if (da_bs < (long long)C * S) return MIS_ERR_ARG;
    if (workspace_bytes < mis_norm_workspace_bytes(N, C, S, per_sample)) return MIS_ERR_WORKSPACE;
    const Geo g = make_geo(N, C, S, x_bs, per_sample);
    const DropCfg d{0.f, 0u, nullptr, nullptr};
    float2* part = reinterpret_cast<float2*>(workspace);
    hipLaunchKernelGGL(bwd_partial_kernel<false>, dim3(g.P, g.nchunks, g.G), dim3(256), 0, stream, x, g, da, da_bs, mean, rstd,
                       gamma, beta, slope, d, part, PoolGrad{}, ResArgs{});
    hipLaunchKernelGGL(bwd_final_kernel, dim3((g.G + 3) / 4), dim3(256), 0, stream, part, g,
                       reinterpret_cast<float2*>(sums), dgamma, dbeta, accumulate_affine);
    return mis_launch_status();
}

// The backward of InstanceNorm (no affine) + (Leaky)ReLU whose first stage ran inside the data-gradient launch that
// produced da (mis_conv3d_wino_dgrad_norm): part[(n * C + c) * tiles + t] = (sum dz, sum dz * x).  Writes
// sums[n * C + c] = (mean dz, mean dz * xhat) and, when dx != NULL, the gradient at the normalisation's input (dx NULL:
// the consumer forms it itself from the sums -- mis_conv_wgrad_cin1_norm).  No dropout on this path.
extern "C" int mis_norm_act_bwd_tiles(const float* x, long long x_bs, const float* da, long long da_bs, float* dx,
                                      long long dx_bs, int N, int C, long long S, const float* mean, const float* rstd,
                                      float slope, const float* part, int tiles, float* sums, hipStream_t stream) {
    int st = check_geo(x, N, C, S, x_bs);
    if (st) return st;
    if (!da || !mean || !rstd || !part || !sums || tiles <= 0) return MIS_ERR_ARG;
    if (da_bs % 4 != 0 || !aligned16(da) || da_bs < (long long)C * S || ((uintptr_t)sums & 7) || ((uintptr_t)part & 7))
        return MIS_ERR_UNSUPPORTED;
    if (dx && (dx_bs % 4 != 0 || !aligned16(dx) || dx_bs < (long long)C * S)) return MIS_ERR_UNSUPPORTED;
    const Geo g = make_geo(N, C, S, x_bs, 1);
    hipLaunchKernelGGL(bwd_final_raw_kernel, dim3((g.G + 3) / 4), dim3(256), 0, stream,
                       reinterpret_cast<const float2*>(part), g.G, tiles, S, mean, rstd, reinterpret_cast<float2*>(sums));
    if (dx) {
        const DropCfg d{0.f, 0u, nullptr, nullptr};
        const unsigned gx = (unsigned)mis_cdiv(S >> 2, 256 * APPLY_U);
        hipLaunchKernelGGL(apply_bwd_kernel<false>, dim3(gx, C, N), dim3(256), 0, stream, x, g, da, da_bs, mean, rstd,
                           (const float*)nullptr, (const float*)nullptr, slope, d, reinterpret_cast<const float2*>(sums),
                           dx, dx_bs, 0, PoolGrad{}, ResArgs{});
    }
    return mis_launch_status();
}

// y = act(norm(x) + res): the tail of MONAI's UnetResBlock (UNETR / SwinUNETR: conv - IN - lrelu - conv - IN, + shortcut,
// lrelu; reference code/networks/unetr.py's UnetrBasicBlock(res_block=True) / net_factory_3d.py:37-38) in one pass
// instead of normalise, add, activate.  add_after_act != 0: y = act(norm(x)) + res -- V-Net's x_up + skip (vnet.py:210-222).  BatchNorm / InstanceNorm with optional affine; no dropout, no GroupNorm; S % 4 == 0.
extern "C" int mis_norm_res_act_fwd(const float* x, long long x_bs, const float* res, long long res_bs, float* y,
                                    long long y_bs, int N, int C, long long S, int per_sample, const float* mean,
                                    const float* rstd, const float* gamma, const float* beta, float slope,
                                    int add_after_act, hipStream_t stream) {
    int st = check_geo(x, N, C, S, x_bs);
    if (st) return st;
    if (!res || !y || !mean || !rstd || y_bs < (long long)C * S || res_bs < (long long)C * S) return MIS_ERR_ARG;
    if (y_bs % 4 != 0 || !aligned16(y) || res_bs % 4 != 0 || !aligned16(res)) return MIS_ERR_UNSUPPORTED;
    const Geo g = make_geo(N, C, S, x_bs, per_sample, 1);
    const DropCfg d{0.f, 0u, nullptr, nullptr};
    const unsigned gx = (unsigned)mis_cdiv(S >> 2, 256 * APPLY_U);
    hipLaunchKernelGGL(apply_fwd_kernel<true>, dim3(gx, C, N), dim3(256), 0, stream, x, g, mean, rstd, gamma, beta, slope,
                       d, y, y_bs, ResArgs{res, res_bs, nullptr, 0, 0, add_after_act});
    return mis_launch_status();
}

// Backward of mis_norm_res_act_fwd: dz = dy * act'(norm(x) + res);  dres (+)= dz;  dx = the normalisation's backward of
// dz.  accumulate_dres != 0 adds to dres (the shortcut has other consumers whose gradient is already there).
// Workspace: mis_norm_workspace_bytes(N, C, S, per_sample).
extern "C" int mis_norm_res_act_bwd(const float* x, long long x_bs, const float* res, long long res_bs, const float* dy,
                                    long long dy_bs, float* dx, long long dx_bs, float* dres, long long dres_bs,
                                    int accumulate_dres, int N, int C, long long S, int per_sample, const float* mean,
                                    const float* rstd, const float* gamma, const float* beta, float slope,
                                    float* dgamma, float* dbeta, int accumulate_affine, int add_after_act,
                                    void* workspace, long long workspace_bytes, hipStream_t stream) {
    if (!dy || !res || !dres || S % 4 != 0) return !dy || !res || !dres ? MIS_ERR_ARG : MIS_ERR_UNSUPPORTED;
    if (res_bs % 4 != 0 || !aligned16(res) || dres_bs % 4 != 0 || !aligned16(dres)) return MIS_ERR_UNSUPPORTED;
    if (res_bs < (long long)C * S || dres_bs < (long long)C * S) return MIS_ERR_ARG;
    if (per_sample && (gamma || beta)) return MIS_ERR_UNSUPPORTED;
    return norm_act_bwd_impl(x, x_bs, dy, dy_bs, dx, dx_bs, N, C, S, per_sample, 1, 0, mean, rstd, gamma, beta, slope, 0.f,
                             0u, nullptr, nullptr, dgamma, dbeta, accumulate_affine, workspace, workspace_bytes, PoolGrad{},
                             stream, ResArgs{res, res_bs, dres, dres_bs, accumulate_dres, add_after_act});
}

extern "C" int mis_norm_act_bwd(const float* x, long long x_bs, const float* da, long long da_bs, float* dx,
                                long long dx_bs, int N, int C, long long S, int per_sample, const float* mean,
                                const float* rstd, const float* gamma, const float* beta, float slope,
                                float drop_p, unsigned drop_salt, const MisStepState* state,
                                const float* drop_mask, float* dgamma, float* dbeta, int accumulate_affine,
                                void* workspace, long long workspace_bytes, hipStream_t stream) {
    return mis_norm_act_bwd_g(x, x_bs, da, da_bs, dx, dx_bs, N, C, S, per_sample, 1, 0, mean, rstd, gamma, beta,
                              slope, drop_p, drop_salt, state, drop_mask, dgamma, dbeta, accumulate_affine, workspace,
                              workspace_bytes, stream);
}

// =====================================================================================================================
// Last block of the 3-D networks: normalisation + (Leaky)ReLU + dropout, then the 1x1x1 classifier (reference
// code/networks/unet_3D.py: up_concat1 -> dropout2 -> final = nn.Conv3d(16, n_classes, 1); vnet.py:180-181 block_nine ->
// Dropout3d -> out_conv; unetr.py out = UnetOutBlock).  The activation z = drop(act(norm(y))) has C = 16 channels on the
// full-resolution volume and exactly one consumer with 2 .. 4 output channels.  Un-fused it costs: apply (read y, write
// z), head (read z), and in backward head-dgrad (write dz), head-wgrad (read z), partial sums (read dz, y), apply (read
// dz, y, write dy) -- 11 passes over a 16-channel volume.  Here z and dz are never stored: a thread takes 4 voxels of all
// C channels, forward reads y once and writes the logits; backward recomputes z and dz = W^T dlogits from y and the
// K-channel dlogits in both the partial-sum pass (which also yields the classifier's dW and db) and the apply pass:
// 4 passes + 3 over the K-channel logits.
// =====================================================================================================================
namespace {

struct HeadArgs {
    const float* x; long long x_bs;       // y = conv output before the normalisation, [N][C][S]
    int N, K, per_sample;
    long long S;
    const float* mean; const float* rstd; const float* gamma; const float* beta;
    float slope;
    const float* w; const float* b;       // classifier [K][C], [K] (or null)
    float* logits; long long l_bs;        // forward: [N][K][S]
    const float* dl; long long dl_bs;     // backward: dlogits
    float* dx; long long dx_bs;           // backward: gradient at y
    float2* part;                         // backward: (sum dz, sum dz*xhat) in bwd_final_kernel's layout
    const float2* sums;                   // ... and its result
    float* hpart;                         // backward: classifier partials [N*P][K*C + K]
    int P;
};

// Dropout of the 4 voxels of unit u in all C channels.  Philox mode (MASK = false): the C draws in a rolled loop (inlined C
// times into the unrolled channel loops the generator is 160 KB of code), kept as one bit per element -- all ones and
// keep = 1 when dropout is off, so the channel loops carry no branch.  MASK = true (explicit scale mask: parity tests)
// reads the scale.
template <int C, bool MASK>
__device__ __forceinline__ unsigned long long head_drop_bits(const DropCfg& d, int n, long long S, long long u) {
    unsigned long long bits = ~0ull;
    if (!MASK && d.p > 0.f) {
        bits = 0;
#pragma unroll 1
        for (int c = 0; c < C; ++c) {
            float s[4];
            drop_scale4(d, ((unsigned long long)n * C + c) * S + u * 4, (unsigned)(n * C + c), s);
#pragma unroll
            for (int j = 0; j < 4; ++j) bits |= (unsigned long long)(s[j] != 0.f) << (4 * c + j);
        }
    }
    return bits;
}
template <int C, bool MASK>
__device__ __forceinline__ void head_drop_scale(const DropCfg& d, unsigned long long bits, float keep, int n, long long S,
                                                long long u, int c, float (&s)[4]) {
    if constexpr (MASK) {
        const float4 m = *reinterpret_cast<const float4*>(d.mask + ((unsigned long long)n * C + c) * S + u * 4);
        s[0] = m.x; s[1] = m.y; s[2] = m.z; s[3] = m.w;
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) s[j] = ((bits >> (4 * c + j)) & 1ull) ? keep : 0.f;
    }
}

// per-channel scale / shift (and, backward, rstd-normalised form) of sample n into LDS
template <int C>
__device__ __forceinline__ void head_coeffs(const HeadArgs& a, int n, float* sc, float* sh) {
    if (threadIdx.x < C) {
        const int c = threadIdx.x, grp = a.per_sample ? n * C + c : c;
        const float s = (a.gamma ? a.gamma[c] : 1.f) * a.rstd[grp];
        sc[c] = s;
        sh[c] = (a.beta ? a.beta[c] : 0.f) - a.mean[grp] * s;
    }
}

// grid = (ceil(S/4 / 256), N)
template <int C, int K, bool MASK>
__global__ __launch_bounds__(256) void head_fwd_fused_kernel(const HeadArgs a, DropCfg d) {
    __shared__ float sc[C], sh[C], sw[K * C];
    const int n = blockIdx.y;
    head_coeffs<C>(a, n, sc, sh);
    for (int i = threadIdx.x; i < K * C; i += 256) sw[i] = a.w[i];
    __syncthreads();
    const long long units = a.S >> 2, u = (long long)blockIdx.x * 256 + threadIdx.x;
    if (u >= units) return;
    const float* __restrict__ xb = a.x + (long long)n * a.x_bs + u * 4;
    float4 q[C];
#pragma unroll
    for (int c = 0; c < C; ++c) q[c] = *reinterpret_cast<const float4*>(xb + (long long)c * a.S);
    const float keep = d.p > 0.f ? 1.f / (1.f - d.p) : 1.f;
    const unsigned long long bits = head_drop_bits<C, MASK>(d, n, a.S, u);
    float acc[K][4];
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const float bk = a.b ? a.b[k] : 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[k][j] = bk;
    }
#pragma unroll
    for (int c = 0; c < C; ++c) {
        float v[4] = {q[c].x * sc[c] + sh[c], q[c].y * sc[c] + sh[c], q[c].z * sc[c] + sh[c], q[c].w * sc[c] + sh[c]};
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = v[j] > 0.f ? v[j] : v[j] * a.slope;
        {
            float s[4];
            head_drop_scale<C, MASK>(d, bits, keep, n, a.S, u, c, s);
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] *= s[j];
        }
#pragma unroll
        for (int k = 0; k < K; ++k)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[k][j] = fmaf(sw[k * C + c], v[j], acc[k][j]);
    }
    float* __restrict__ lb = a.logits + (long long)n * a.l_bs + u * 4;
#pragma unroll
    for (int k = 0; k < K; ++k)
        *reinterpret_cast<float4*>(lb + (long long)k * a.S) = make_float4(acc[k][0], acc[k][1], acc[k][2], acc[k][3]);
}

// grid = (P, N): block (p, n) walks its share of the sample; per channel (sum dz, sum dz*xhat) into `part` (the layout
// bwd_final_kernel reads), the classifier's sum dl[k]*z[c] and sum dl[k] into `hpart`
// 2C + KC + K accumulators per thread; the channel loop is fenced every 4 channels so that at most 4 float4 loads are
// hoisted (128 registers, four waves per SIMD: with all 16 in flight the kernel took 512 registers, one wave per SIMD,
// and ran at 1.4 TB/s)
template <int C, int K, bool MASK>
__global__ __launch_bounds__(256, K <= 2 ? 3 : 2) void head_bwd_partial_kernel(const HeadArgs a, DropCfg d) {
    __shared__ float sc[C], sh[C], sw[K * C], red[4 * (2 * C + K * C + K)];
    const int p = blockIdx.x, n = blockIdx.y;
    head_coeffs<C>(a, n, sc, sh);
    for (int i = threadIdx.x; i < K * C; i += 256) sw[i] = a.w[i];
    __syncthreads();
    const long long units = a.S >> 2;
    const long long per = (units + a.P - 1) / a.P;
    const long long u0 = p * per, u1 = (u0 + per < units) ? u0 + per : units;
    const float keep = d.p > 0.f ? 1.f / (1.f - d.p) : 1.f;
    constexpr int NV = 2 * C + K * C + K;
    float v[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i] = 0.f;
    for (long long u = u0 + threadIdx.x; u < u1; u += 256) {
        float4 g4[K];
#pragma unroll
        for (int k = 0; k < K; ++k) {
            g4[k] = *reinterpret_cast<const float4*>(a.dl + (long long)n * a.dl_bs + (long long)k * a.S + u * 4);
            v[2 * C + K * C + k] += (g4[k].x + g4[k].y) + (g4[k].z + g4[k].w);
        }
        const float* __restrict__ xb = a.x + (long long)n * a.x_bs + u * 4;
        const unsigned long long bits = head_drop_bits<C, MASK>(d, n, a.S, u);
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const float4 q = *reinterpret_cast<const float4*>(xb + (long long)c * a.S);
            const float xs[4] = {q.x, q.y, q.z, q.w};
            float s[4];
            head_drop_scale<C, MASK>(d, bits, keep, n, a.S, u, c, s);
            // gamma * rstd = sc, so xhat = (x*sc + sh - beta) / gamma is avoided: xhat from mean / rstd directly
            const int grp = a.per_sample ? n * C + c : c;
            const float m = a.mean[grp], rs = a.rstd[grp];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float xh = (xs[j] - m) * rs;
                const float z0 = xs[j] * sc[c] + sh[c];
                const float act = (z0 > 0.f ? z0 : z0 * a.slope) * s[j];          // z as the classifier saw it
                float dzd = 0.f;
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    const float gk = (&g4[k].x)[j];
                    dzd = fmaf(sw[k * C + c], gk, dzd);
                    v[2 * C + k * C + c] = fmaf(gk, act, v[2 * C + k * C + c]);
                }
                const float ds = dzd * s[j];
                const float dz = z0 > 0.f ? ds : ds * a.slope;
                v[2 * c] += dz;
                v[2 * c + 1] = fmaf(dz, xh, v[2 * c + 1]);
            }
            if (c % 4 == 3) __builtin_amdgcn_sched_barrier(0);
        }
    }
    // block sum: wave shuffles, then one thread per value adds the four waves' rows and stores its value
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i] = mis_wave_sum(v[i]);
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i < NV; ++i) red[wave * NV + i] = v[i];
    }
    __syncthreads();
    if (threadIdx.x < NV) {
        const int i = threadIdx.x;
        const float t = (red[i] + red[NV + i]) + (red[2 * NV + i] + red[3 * NV + i]);
        if (i < 2 * C) {
            const int c = i >> 1;
            const int nchunks = a.per_sample ? 1 : a.N;
            const long long grp = a.per_sample ? (long long)n * C + c : c;
            const int k = a.per_sample ? 0 : n;
            reinterpret_cast<float*>(a.part)[((grp * nchunks + k) * a.P + p) * 2 + (i & 1)] = t;
        } else {
            a.hpart[((long long)n * a.P + p) * (K * C + K) + (i - 2 * C)] = t;
        }
    }
}

// dw[k][c] (+)= sum of the partial rows, db[k] likewise: one wave per output, fixed order
__global__ __launch_bounds__(64) void head_final_kernel(const float* __restrict__ hpart, int rows, int KC, int K,
                                                        float* __restrict__ dw, float* __restrict__ db, int accumulate) {
    const int o = blockIdx.x, lane = threadIdx.x;
    double s = 0.0;
    for (int r = lane; r < rows; r += 64) s += hpart[(long long)r * (KC + K) + o];
    s = mis_wave_sum_d(s);
    if (lane == 0) {
        if (o < KC) dw[o] = accumulate ? dw[o] + (float)s : (float)s;
        else if (db) db[o - KC] = accumulate ? db[o - KC] + (float)s : (float)s;
    }
}

// grid = (ceil(S/4 / 256), N): dx = gamma*rstd * (dz - mean(dz) - xhat * mean(dz*xhat))
template <int C, int K, bool MASK>
__global__ __launch_bounds__(256) void head_bwd_apply_kernel(const HeadArgs a, DropCfg d) {
    __shared__ float sc[C], sh[C], sw[K * C];
    const int n = blockIdx.y;
    head_coeffs<C>(a, n, sc, sh);
    for (int i = threadIdx.x; i < K * C; i += 256) sw[i] = a.w[i];
    __syncthreads();
    const long long units = a.S >> 2, u = (long long)blockIdx.x * 256 + threadIdx.x;
    if (u >= units) return;
    float4 g4[K];
#pragma unroll
    for (int k = 0; k < K; ++k)
        g4[k] = *reinterpret_cast<const float4*>(a.dl + (long long)n * a.dl_bs + (long long)k * a.S + u * 4);
    const float* __restrict__ xb = a.x + (long long)n * a.x_bs + u * 4;
    float* __restrict__ ob = a.dx + (long long)n * a.dx_bs + u * 4;
    const float keep = d.p > 0.f ? 1.f / (1.f - d.p) : 1.f;
    const unsigned long long bits = head_drop_bits<C, MASK>(d, n, a.S, u);
#pragma unroll
    for (int c = 0; c < C; ++c) {
        const float4 q = *reinterpret_cast<const float4*>(xb + (long long)c * a.S);
        const float xs[4] = {q.x, q.y, q.z, q.w};
        float s[4];
        head_drop_scale<C, MASK>(d, bits, keep, n, a.S, u, c, s);
        const int grp = a.per_sample ? n * C + c : c;
        const float m = a.mean[grp], rs = a.rstd[grp];
        const float2 sm = a.sums[grp];
        float o[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float dzd = 0.f;
#pragma unroll
            for (int k = 0; k < K; ++k) dzd = fmaf(sw[k * C + c], (&g4[k].x)[j], dzd);
            const float xh = (xs[j] - m) * rs;
            const float z0 = xs[j] * sc[c] + sh[c];
            const float ds = dzd * s[j];
            const float dz = z0 > 0.f ? ds : ds * a.slope;
            o[j] = sc[c] * (dz - sm.x - xh * sm.y);
        }
        *reinterpret_cast<float4*>(ob + (long long)c * a.S) = make_float4(o[0], o[1], o[2], o[3]);
    }
}

template <int C>
int head_launch_fwd(const HeadArgs& a, const DropCfg& d, hipStream_t stream) {
    const dim3 grid((unsigned)mis_cdiv(a.S >> 2, 256), a.N);
    // K in 2 .. 4 (mis_norm_head_eligible)
#define MIS_HEAD_LAUNCH(KK) do { if (d.mask) hipLaunchKernelGGL((head_fwd_fused_kernel<C, KK, true>), grid, dim3(256), 0, stream, a, d); \
        else hipLaunchKernelGGL((head_fwd_fused_kernel<C, KK, false>), grid, dim3(256), 0, stream, a, d); } while (0)
    if (a.K == 2) MIS_HEAD_LAUNCH(2); else if (a.K == 3) MIS_HEAD_LAUNCH(3); else MIS_HEAD_LAUNCH(4);
#undef MIS_HEAD_LAUNCH
    return mis_launch_status();
}

template <int C>
void head_launch_partial(const HeadArgs& a, const DropCfg& d, hipStream_t stream) {
    const dim3 grid(a.P, a.N);
    // K in 2 .. 4 (mis_norm_head_eligible)
#define MIS_HEAD_LAUNCH(KK) do { if (d.mask) hipLaunchKernelGGL((head_bwd_partial_kernel<C, KK, true>), grid, dim3(256), 0, stream, a, d); \
        else hipLaunchKernelGGL((head_bwd_partial_kernel<C, KK, false>), grid, dim3(256), 0, stream, a, d); } while (0)
    if (a.K == 2) MIS_HEAD_LAUNCH(2); else if (a.K == 3) MIS_HEAD_LAUNCH(3); else MIS_HEAD_LAUNCH(4);
#undef MIS_HEAD_LAUNCH
}

template <int C>
void head_launch_apply(const HeadArgs& a, const DropCfg& d, hipStream_t stream) {
    const dim3 grid((unsigned)mis_cdiv(a.S >> 2, 256), a.N);
    // K in 2 .. 4 (mis_norm_head_eligible)
#define MIS_HEAD_LAUNCH(KK) do { if (d.mask) hipLaunchKernelGGL((head_bwd_apply_kernel<C, KK, true>), grid, dim3(256), 0, stream, a, d); \
        else hipLaunchKernelGGL((head_bwd_apply_kernel<C, KK, false>), grid, dim3(256), 0, stream, a, d); } while (0)
    if (a.K == 2) MIS_HEAD_LAUNCH(2); else if (a.K == 3) MIS_HEAD_LAUNCH(3); else MIS_HEAD_LAUNCH(4);
#undef MIS_HEAD_LAUNCH
}

}  // namespace

// partial blocks per sample = HEAD_PMUL x the plain normalisation's (the partial-sum kernel holds 66 accumulators per
// thread: three workgroups per CU, 768 blocks for 8 volumes = one round)
constexpr int HEAD_PMUL = 3;

// The fused form covers C = 16 channels and 2 .. 4 classes (the binary 3-D tasks -- BraTS whole tumour, LA, Pancreas -- and
// the 3- / 4-class label sets), BatchNorm / InstanceNorm statistics (one group per channel).  The partial-sum kernel
// holds 2C + KC + K accumulators per thread: three workgroups per CU for K = 2 (66 accumulators), two for K = 3, 4
// (83 / 100: at three per CU they spill 2.3 KB of scratch per thread).
extern "C" int mis_norm_head_eligible(int C, int K) { return C == 16 && K >= 2 && K <= 4; }

extern "C" long long mis_norm_head_workspace_bytes(int N, int C, long long S, int per_sample, int K) {
    if (N <= 0 || C <= 0 || S <= 0 || K <= 0) return MIS_ERR_ARG;
    const long long nb = mis_norm_workspace_bytes(N, C, S, per_sample);
    return nb * HEAD_PMUL + (long long)N * pick_P(S) * HEAD_PMUL * (K * C + K) * 4;
}

// logits[N][K][S] = W . drop(act(norm(x))) + b  (mis_norm_act_fwd followed by a 1x1x1 mis_conv_fwd, without the
// activation in between).  mean / rstd: per channel (per_sample = 0) or per (n, c).
extern "C" int mis_norm_head_fwd(const float* x, long long x_bs, int N, int C, long long S, int per_sample,
                                 const float* mean, const float* rstd, const float* gamma, const float* beta,
                                 float slope, float drop_p, unsigned drop_salt, const MisStepState* state,
                                 const float* drop_mask, const float* w, const float* b, int K, float* logits,
                                 long long l_bs, hipStream_t stream) {
    int st = check_geo(x, N, C, S, x_bs);
    if (st) return st;
    if (!mis_norm_head_eligible(C, K) || (per_sample && (gamma || beta))) return MIS_ERR_UNSUPPORTED;
    if (!mean || !rstd || !w || !logits || l_bs < (long long)K * S) return MIS_ERR_ARG;
    if (l_bs % 4 != 0 || !aligned16(logits)) return MIS_ERR_UNSUPPORTED;
    if (drop_p < 0.f || drop_p >= 1.f || (drop_p > 0.f && !state && !drop_mask)) return MIS_ERR_ARG;
    HeadArgs a{};
    a.x = x; a.x_bs = x_bs; a.N = N; a.K = K; a.per_sample = per_sample; a.S = S;
    a.mean = mean; a.rstd = rstd; a.gamma = gamma; a.beta = beta; a.slope = slope; a.w = w; a.b = b;
    a.logits = logits; a.l_bs = l_bs;
    const DropCfg d{drop_p, drop_salt, state, drop_mask};
    return head_launch_fwd<16>(a, d, stream);
}

// Backward of mis_norm_head_fwd from dlogits: dx (gradient at x), dgamma / dbeta (BatchNorm), dw[K][C] and db[K]
// (null: none) of the classifier.  accumulate_* as in mis_norm_act_bwd / mis_conv_wgrad.
extern "C" int mis_norm_head_bwd(const float* x, long long x_bs, const float* dlogits, long long dl_bs, float* dx,
                                 long long dx_bs, int N, int C, long long S, int per_sample, const float* mean,
                                 const float* rstd, const float* gamma, const float* beta, float slope, float drop_p,
                                 unsigned drop_salt, const MisStepState* state, const float* drop_mask, const float* w,
                                 int K, float* dgamma, float* dbeta, int accumulate_affine, float* dw, float* db,
                                 int accumulate_w, void* workspace, long long workspace_bytes, hipStream_t stream) {
    int st = check_geo(x, N, C, S, x_bs);
    if (st) return st;
    if (!mis_norm_head_eligible(C, K) || (per_sample && (gamma || beta))) return MIS_ERR_UNSUPPORTED;
    if (!dlogits || !dx || !mean || !rstd || !w || !dw || !workspace) return MIS_ERR_ARG;
    if (dl_bs % 4 != 0 || dx_bs % 4 != 0 || !aligned16(dlogits) || !aligned16(dx)) return MIS_ERR_UNSUPPORTED;
    if (dl_bs < (long long)K * S || dx_bs < (long long)C * S) return MIS_ERR_ARG;
    if (drop_p < 0.f || drop_p >= 1.f || (drop_p > 0.f && !state && !drop_mask)) return MIS_ERR_ARG;
    if (workspace_bytes < mis_norm_head_workspace_bytes(N, C, S, per_sample, K)) return MIS_ERR_WORKSPACE;
    Geo g = make_geo(N, C, S, x_bs, per_sample);
    g.P *= HEAD_PMUL;
    float2* part = reinterpret_cast<float2*>(workspace);
    float2* sums = part + (long long)g.G * g.nchunks * g.P;
    HeadArgs a{};
    a.x = x; a.x_bs = x_bs; a.N = N; a.K = K; a.per_sample = per_sample; a.S = S;
    a.mean = mean; a.rstd = rstd; a.gamma = gamma; a.beta = beta; a.slope = slope; a.w = w;
    a.dl = dlogits; a.dl_bs = dl_bs; a.dx = dx; a.dx_bs = dx_bs;
    a.part = part; a.sums = sums; a.hpart = reinterpret_cast<float*>(sums + g.G); a.P = g.P;
    const DropCfg d{drop_p, drop_salt, state, drop_mask};
    head_launch_partial<16>(a, d, stream);
    hipLaunchKernelGGL(bwd_final_kernel, dim3((g.G + 3) / 4), dim3(256), 0, stream, part, g, sums,
                       per_sample ? nullptr : dgamma, per_sample ? nullptr : dbeta, accumulate_affine);
    hipLaunchKernelGGL(head_final_kernel, dim3(K * C + K), dim3(64), 0, stream, a.hpart, N * g.P, K * C, K, dw, db,
                       accumulate_w);
    head_launch_apply<16>(a, d, stream);
    return mis_launch_status();
}
