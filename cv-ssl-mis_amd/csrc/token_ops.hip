// Token-major (rows x C, explicit row stride) HBM-bound operators of SwinUnet.
//
// Replaces (reference code/networks/swin_transformer_unet_skip_expand_decoder_sys.py):
//   nn.LayerNorm                                  :204,211,323,365,393,716-717
//   nn.GELU (exact erf form)                      :10,15
//   x = shortcut + drop_path(branch)  (timm DropPath, per-sample)   :285-286
//   PatchMerging 2x2 gather                       :336-344
//   PatchExpand / FinalPatchExpand_X4 'b h w (p1 p2 c) -> b (h p1) (w p2) c'   :377-380,405-408
//   PatchEmbed.proj as im2col (+ the 1->3 channel repeat of vision_transformer.py:49-50)   :573-588
//   bias gradients of nn.Linear (column sums)
//   up_x4: permute to NCHW + 1x1 output conv without bias        :775-786
//
// Rows may live inside wider buffers (row stride `ld` > C), so the decoder's torch.cat([x, skip], -1)
// (:767) is realised by writing the two halves into one buffer.  Reductions use fixed-order trees.
#include "common.h"
#include <string.h>

namespace {

// ------------------------------------------------------------------ LayerNorm
// one wave per row; C % 4 == 0
__global__ __launch_bounds__(256) void ln_fwd_kernel(const float* __restrict__ x, long long ldx,
                                                     float* __restrict__ y, long long ldy,
                                                     const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, float* __restrict__ mean,
                                                     float* __restrict__ rstd, long long M, int C, float eps) {
    const long long row = blockIdx.x * 4LL + (threadIdx.x >> 6);
    if (row >= M) return;
    const int lane = threadIdx.x & 63;
    const float* __restrict__ xr = x + row * ldx;
    float s = 0.f;
    for (int c = lane * 4; c < C; c += 256) {
        const float4 v = *reinterpret_cast<const float4*>(xr + c);
        s += (v.x + v.y) + (v.z + v.w);
    }
    s = mis_wave_sum(s);
    const float m = s / (float)C;
    float ss = 0.f;
    for (int c = lane * 4; c < C; c += 256) {
        const float4 v = *reinterpret_cast<const float4*>(xr + c);
        const float a = v.x - m, b = v.y - m, cc = v.z - m, d = v.w - m;
        ss += (a * a + b * b) + (cc * cc + d * d);
    }
    ss = mis_wave_sum(ss);
    const float rs = 1.f / sqrtf(ss / (float)C + eps);
    if (lane == 0) { mean[row] = m; rstd[row] = rs; }
    float* __restrict__ yr = y + row * ldy;
    for (int c = lane * 4; c < C; c += 256) {
        const float4 v = *reinterpret_cast<const float4*>(xr + c);
        const float4 g = *reinterpret_cast<const float4*>(gamma + c);
        const float4 b = *reinterpret_cast<const float4*>(beta + c);
        *reinterpret_cast<float4*>(yr + c) = make_float4((v.x - m) * rs * g.x + b.x, (v.y - m) * rs * g.y + b.y,
                                                         (v.z - m) * rs * g.z + b.z, (v.w - m) * rs * g.w + b.w);
    }
}

template <int LPR>
__device__ __forceinline__ float group_sum(float v) { return mis_group_sum<LPR>(v); }      // common.h

// Register-resident rows: LPR lanes per row (32 -> two rows per wave for C <= 128), NV float4 per lane,
// so x is read from HBM/L2 once and the three passes run on registers.  C <= LPR*4*NV.
// RU rows of a row group are loaded before any of them is reduced (a thread otherwise has one 16-byte load in flight and
// the kernel runs at 45 % of the HBM rate on the 96-channel stages).
template <int LPR, int NV, int RU>
__global__ __launch_bounds__(256) void ln_fwd_reg_kernel(const float* __restrict__ x, long long ldx,
                                                         float* __restrict__ y, long long ldy,
                                                         const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, float* __restrict__ mean,
                                                         float* __restrict__ rstd, long long M, int C, float eps) {
    constexpr int RPI = 256 / LPR;
    const int lane = threadIdx.x % LPR, rg = threadIdx.x / LPR;
    const long long base = (long long)blockIdx.x * (RPI * RU) + rg;
    float4 v[RU][NV], g[NV], b[NV];
#pragma unroll
    for (int u = 0; u < RU; ++u) {
        const long long row = base + (long long)u * RPI;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = (lane + i * LPR) * 4;
            v[u][i] = (row < M && c < C) ? *reinterpret_cast<const float4*>(x + row * ldx + c) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = (lane + i * LPR) * 4;
        g[i] = c < C ? *reinterpret_cast<const float4*>(gamma + c) : make_float4(0.f, 0.f, 0.f, 0.f);
        b[i] = c < C ? *reinterpret_cast<const float4*>(beta + c) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int u = 0; u < RU; ++u) {
        const long long row = base + (long long)u * RPI;
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) s += (v[u][i].x + v[u][i].y) + (v[u][i].z + v[u][i].w);
        const float m = group_sum<LPR>(s) / (float)C;
        float ss = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = (lane + i * LPR) * 4;
            const float a = v[u][i].x - m, bb = v[u][i].y - m, cc = v[u][i].z - m, d = v[u][i].w - m;
            ss += c < C ? (a * a + bb * bb) + (cc * cc + d * d) : 0.f;
        }
        const float rs = 1.f / sqrtf(group_sum<LPR>(ss) / (float)C + eps);
        if (row >= M) continue;          // after the lane reductions: every lane of the group takes part in them
        if (lane == 0) { mean[row] = m; rstd[row] = rs; }
        float* __restrict__ yr = y + row * ldy;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = (lane + i * LPR) * 4;
            if (c < C)
                *reinterpret_cast<float4*>(yr + c) =
                    make_float4((v[u][i].x - m) * rs * g[i].x + b[i].x, (v[u][i].y - m) * rs * g[i].y + b[i].y,
                                (v[u][i].z - m) * rs * g[i].z + b[i].z, (v[u][i].w - m) * rs * g[i].w + b[i].w);
        }
    }
}

// Backward of the residual add that produced the LayerNorm's input, in the same pass (x = a + s_b * y, reference
// ...sys.py:276, :281: `x = shortcut + self.drop_path(x)` followed by `self.norm2(x)` / the next block's norm1): with
// total = gin + LayerNorm'(dy) -- gin = what the input's other readers already left in its gradient -- the kernel writes
// dx (+)= total (the shortcut's gradient) and d2 = s_b * total (the branch's), and the input's own gradient buffer is not
// touched again.  d2 == nullptr: plain LayerNorm backward.
struct LnRes {
    const float* gin; long long ldgin;
    float* d2; long long ldd2;
    const float* rowscale; long long rps;
};

template <int LPR, int NV>
__global__ __launch_bounds__(256) void ln_bwd_dx_reg_kernel(const float* __restrict__ x, long long ldx,
                                                            const float* __restrict__ dy, long long lddy,
                                                            float* __restrict__ dx, long long lddx,
                                                            const float* __restrict__ gamma,
                                                            const float* __restrict__ mean,
                                                            const float* __restrict__ rstd, long long M, int C,
                                                            int accumulate, long long rows_per_slab,
                                                            float2* __restrict__ part, const LnRes rz) {
    // One block per slab of rows; a lane owns the same columns in every row, so the affine-gradient column
    // sums (sum dy*xhat, sum dy) ride along in registers and are written as one partial row per slab
    // (part == nullptr: dx only).  rows_per_slab % (256/LPR) == 0.
    constexpr int RPI = 256 / LPR;
    __shared__ float4 red[256 * NV];
    const int lane = threadIdx.x % LPR, rg = threadIdx.x / LPR;
    const long long r0 = blockIdx.x * rows_per_slab;
    long long r1 = r0 + rows_per_slab;
    if (r1 > M) r1 = M;
    float4 g[NV], dg[NV], db[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = (lane + i * LPR) * 4;
        g[i] = c < C ? *reinterpret_cast<const float4*>(gamma + c) : make_float4(0.f, 0.f, 0.f, 0.f);
        dg[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        db[i] = dg[i];
    }
    constexpr int RU = 1;      // (two rows in flight measured slower here: 4.3 against 4.5 TB/s at 150 528 x 96)
    for (long long row0 = r0 + rg; row0 < r1; row0 += RPI * RU) {
        float4 vv[RU][NV], dd[RU][NV];
        float mm[RU], rr[RU];
#pragma unroll
        for (int u = 0; u < RU; ++u) {
            const long long row = row0 + (long long)u * RPI < r1 ? row0 + (long long)u * RPI : row0;   // dead slot: the live row again
            mm[u] = mean[row]; rr[u] = rstd[row];
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int c = (lane + i * LPR) * 4;
                vv[u][i] = make_float4(mm[u], mm[u], mm[u], mm[u]);
                dd[u][i] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (c < C) {
                    vv[u][i] = *reinterpret_cast<const float4*>(x + row * ldx + c);
                    dd[u][i] = *reinterpret_cast<const float4*>(dy + row * lddy + c);
                }
            }
        }
#pragma unroll
        for (int u = 0; u < RU; ++u) {
            const long long row = row0 + (long long)u * RPI;
            if (row >= r1) break;                  // uniform within the row group
            const float m = mm[u], rs = rr[u];
            float4 xc[NV], a[NV];   // centred x, dy*gamma
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const float4 v = vv[u][i], d = dd[u][i];
                xc[i] = make_float4(v.x - m, v.y - m, v.z - m, v.w - m);
                a[i] = make_float4(d.x * g[i].x, d.y * g[i].y, d.z * g[i].z, d.w * g[i].w);
                dg[i].x += d.x * xc[i].x * rs; dg[i].y += d.y * xc[i].y * rs;
                dg[i].z += d.z * xc[i].z * rs; dg[i].w += d.w * xc[i].w * rs;
                db[i].x += d.x; db[i].y += d.y; db[i].z += d.z; db[i].w += d.w;
                s1 += (a[i].x + a[i].y) + (a[i].z + a[i].w);
                s2 += (a[i].x * xc[i].x + a[i].y * xc[i].y) + (a[i].z * xc[i].z + a[i].w * xc[i].w);
            }
            s1 = group_sum<LPR>(s1) / (float)C;
            s2 = group_sum<LPR>(s2) * rs / (float)C;
            float* __restrict__ or_ = dx + row * lddx;
            const float sb = (rz.d2 && rz.rowscale) ? rz.rowscale[row / rz.rps] : 1.f;
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int c = (lane + i * LPR) * 4;
                if (c < C) {
                    float4 o = make_float4(rs * (a[i].x - s1 - xc[i].x * rs * s2), rs * (a[i].y - s1 - xc[i].y * rs * s2),
                                           rs * (a[i].z - s1 - xc[i].z * rs * s2), rs * (a[i].w - s1 - xc[i].w * rs * s2));
                    if (rz.d2) {
                        if (rz.gin) {
                            const float4 q = *reinterpret_cast<const float4*>(rz.gin + row * rz.ldgin + c);
                            o.x += q.x; o.y += q.y; o.z += q.z; o.w += q.w;
                        }
                        *reinterpret_cast<float4*>(rz.d2 + row * rz.ldd2 + c) = make_float4(sb * o.x, sb * o.y, sb * o.z, sb * o.w);
                    }
                    if (accumulate) {
                        const float4 p = *reinterpret_cast<const float4*>(or_ + c);
                        o.x += p.x; o.y += p.y; o.z += p.z; o.w += p.w;
                    }
                    *reinterpret_cast<float4*>(or_ + c) = o;
                }
            }
        }
    }
    if (!part) return;
    // fixed-order sum over the RPI row groups, first the dgamma partials then the dbeta ones
    float4 sg[NV], sb[NV];
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < NV; ++i) red[(rg * NV + i) * LPR + lane] = pass ? db[i] : dg[i];
        __syncthreads();
        if (rg == 0) {
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                float4 t = red[i * LPR + lane];
#pragma unroll
                for (int k = 1; k < RPI; ++k) {
                    const float4 q = red[(k * NV + i) * LPR + lane];
                    t.x += q.x; t.y += q.y; t.z += q.z; t.w += q.w;
                }
                if (pass) sb[i] = t; else sg[i] = t;
            }
        }
    }
    if (rg == 0) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = (lane + i * LPR) * 4;
            if (c < C) {
                float4* o = reinterpret_cast<float4*>(part + (long long)blockIdx.x * C + c);
                o[0] = make_float4(sg[i].x, sb[i].x, sg[i].y, sb[i].y);
                o[1] = make_float4(sg[i].z, sb[i].z, sg[i].w, sb[i].w);
            }
        }
    }
}

// dx (+)= rstd * (dy*g - mean(dy*g) - xhat * mean(dy*g*xhat)); one wave per row
__global__ __launch_bounds__(256) void ln_bwd_dx_kernel(const float* __restrict__ x, long long ldx,
                                                        const float* __restrict__ dy, long long lddy,
                                                        float* __restrict__ dx, long long lddx,
                                                        const float* __restrict__ gamma,
                                                        const float* __restrict__ mean,
                                                        const float* __restrict__ rstd, long long M, int C,
                                                        int accumulate) {
    const long long row = blockIdx.x * 4LL + (threadIdx.x >> 6);
    if (row >= M) return;
    const int lane = threadIdx.x & 63;
    const float* __restrict__ xr = x + row * ldx;
    const float* __restrict__ gr = dy + row * lddy;
    const float m = mean[row], rs = rstd[row];
    float s1 = 0.f, s2 = 0.f;
    for (int c = lane * 4; c < C; c += 256) {
        const float4 v = *reinterpret_cast<const float4*>(xr + c);
        const float4 d = *reinterpret_cast<const float4*>(gr + c);
        const float4 g = *reinterpret_cast<const float4*>(gamma + c);
        const float a0 = d.x * g.x, a1 = d.y * g.y, a2 = d.z * g.z, a3 = d.w * g.w;
        s1 += (a0 + a1) + (a2 + a3);
        s2 += (a0 * (v.x - m) + a1 * (v.y - m)) + (a2 * (v.z - m) + a3 * (v.w - m));
    }
    s1 = mis_wave_sum(s1) / (float)C;
    s2 = mis_wave_sum(s2) * rs / (float)C;
    float* __restrict__ or_ = dx + row * lddx;
    for (int c = lane * 4; c < C; c += 256) {
        const float4 v = *reinterpret_cast<const float4*>(xr + c);
        const float4 d = *reinterpret_cast<const float4*>(gr + c);
        const float4 g = *reinterpret_cast<const float4*>(gamma + c);
        float4 o = make_float4(rs * (d.x * g.x - s1 - (v.x - m) * rs * s2), rs * (d.y * g.y - s1 - (v.y - m) * rs * s2),
                               rs * (d.z * g.z - s1 - (v.z - m) * rs * s2), rs * (d.w * g.w - s1 - (v.w - m) * rs * s2));
        if (accumulate) {
            const float4 p = *reinterpret_cast<const float4*>(or_ + c);
            o.x += p.x; o.y += p.y; o.z += p.z; o.w += p.w;
        }
        *reinterpret_cast<float4*>(or_ + c) = o;
    }
}

// column partial sums over a slab of rows: mode 0: (sum dy*xhat, sum dy) for LayerNorm affine grads;
// mode 1: (sum x, -) plain column sums (bias gradient).  grid = (ceil(C/64), slabs); block = 64 cols x 4 row-lanes
__global__ __launch_bounds__(256) void col_partial_kernel(const float* __restrict__ x, long long ldx,
                                                          const float* __restrict__ dy, long long lddy,
                                                          const float* __restrict__ mean,
                                                          const float* __restrict__ rstd, long long M, int C,
                                                          long long rows_per_slab, int mode,
                                                          float2* __restrict__ part) {
    // block = 32 float4 column lanes (128 columns) x 8 row lanes; fixed-order LDS tree over the row lanes
    __shared__ float4 ra[256], rb[256];
    const int cl = threadIdx.x & 31, rl = threadIdx.x >> 5;
    const int col = blockIdx.x * 128 + cl * 4;
    const long long r0 = blockIdx.y * rows_per_slab;
    long long r1 = r0 + rows_per_slab;
    if (r1 > M) r1 = M;
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = make_float4(0.f, 0.f, 0.f, 0.f);
    if (col < C) {
        long long r = r0 + rl;
        if (mode == 1)
            for (; r + 24 < r1; r += 32) {   // plain column sums: four rows in flight
                const float4 v0 = *reinterpret_cast<const float4*>(x + r * ldx + col);
                const float4 v1 = *reinterpret_cast<const float4*>(x + (r + 8) * ldx + col);
                const float4 v2 = *reinterpret_cast<const float4*>(x + (r + 16) * ldx + col);
                const float4 v3 = *reinterpret_cast<const float4*>(x + (r + 24) * ldx + col);
                a.x += v0.x; a.y += v0.y; a.z += v0.z; a.w += v0.w;
                a.x += v1.x; a.y += v1.y; a.z += v1.z; a.w += v1.w;
                a.x += v2.x; a.y += v2.y; a.z += v2.z; a.w += v2.w;
                a.x += v3.x; a.y += v3.y; a.z += v3.z; a.w += v3.w;
            }
        for (; r < r1; r += 8) {
            const float4 xv = *reinterpret_cast<const float4*>(x + r * ldx + col);
            if (mode == 0) {
                const float4 d = *reinterpret_cast<const float4*>(dy + r * lddy + col);
                const float m = mean[r], rs = rstd[r];
                a.x += d.x * (xv.x - m) * rs; a.y += d.y * (xv.y - m) * rs;
                a.z += d.z * (xv.z - m) * rs; a.w += d.w * (xv.w - m) * rs;
                b.x += d.x; b.y += d.y; b.z += d.z; b.w += d.w;
            } else {
                a.x += xv.x; a.y += xv.y; a.z += xv.z; a.w += xv.w;
            }
        }
    }
    ra[threadIdx.x] = a;
    rb[threadIdx.x] = b;
    __syncthreads();
    if (rl == 0 && col < C) {
        float4 sa = ra[cl], sb = rb[cl];
#pragma unroll
        for (int k = 1; k < 8; ++k) {
            const float4 pa = ra[k * 32 + cl], pb = rb[k * 32 + cl];
            sa.x += pa.x; sa.y += pa.y; sa.z += pa.z; sa.w += pa.w;
            sb.x += pb.x; sb.y += pb.y; sb.z += pb.z; sb.w += pb.w;
        }
        float2* o = part + (long long)blockIdx.y * C + col;
        o[0] = make_float2(sa.x, sb.x); o[1] = make_float2(sa.y, sb.y);
        o[2] = make_float2(sa.z, sb.z); o[3] = make_float2(sa.w, sb.w);
    }
}

__global__ __launch_bounds__(1024) void col_final_kernel(const float2* __restrict__ part, int slabs, int C,
                                                         float* out_a, float* out_b, int accumulate) {
    // block = 32 columns x 32 slab lanes (coalesced 256-byte reads of the partial rows), double accumulators
    __shared__ double ra[1024], rb[1024];
    const int cl = threadIdx.x & 31, sl = threadIdx.x >> 5;
    const int col = blockIdx.x * 32 + cl;
    double a = 0.0, b = 0.0;
    if (col < C) {
        int s = sl;
        for (; s + 224 < slabs; s += 256) {   // eight independent loads in flight (the kernel is latency-bound)
            float2 p[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) p[j] = part[(long long)(s + 32 * j) * C + col];
#pragma unroll
            for (int j = 0; j < 8; ++j) { a += p[j].x; b += p[j].y; }
        }
        for (; s + 96 < slabs; s += 128) {   // four
            const float2 p0 = part[(long long)s * C + col], p1 = part[(long long)(s + 32) * C + col];
            const float2 p2 = part[(long long)(s + 64) * C + col], p3 = part[(long long)(s + 96) * C + col];
            a += p0.x; b += p0.y; a += p1.x; b += p1.y; a += p2.x; b += p2.y; a += p3.x; b += p3.y;
        }
        for (; s < slabs; s += 32) {
            const float2 p = part[(long long)s * C + col];
            a += p.x; b += p.y;
        }
    }
    ra[threadIdx.x] = a;
    rb[threadIdx.x] = b;
    __syncthreads();
    if (sl == 0 && col < C) {
#pragma unroll
        for (int k = 1; k < 32; ++k) { a += ra[k * 32 + cl]; b += rb[k * 32 + cl]; }
        if (out_a) out_a[col] = accumulate ? out_a[col] + (float)a : (float)a;
        if (out_b) out_b[col] = accumulate ? out_b[col] + (float)b : (float)b;
    }
}

// Every finishing column sum of a backward pass in ONE launch (round 6): the affine gradients of the LayerNorms (float2 partial
// rows per slab of tokens), the weight / bias gradients of the split-K dW GEMMs (one partial matrix per k-slice) and the
// relative-position-bias gradients of the window attentions (one partial table per (image, window)) are all "out[c] (+)= sum over
// slabs of part[slab][c]" -- ~130 launches of 5 .. 9 us per SwinUnet step that nothing downstream of the backward reads.  The
// token plans queue a job per reduction and run the queue as one launch per flush (mis_hip/plan.py::flush_deferred).  A job owns
// ceil(C / 32) consecutive workgroups (`first` = prefix sum); the workgroup is col_final_kernel's: 32 columns x 32 slab lanes,
// double accumulators, fixed order -- a job's result does not depend on what else is in the batch.
struct ColsumJob {
    const void* part;          // float [slabs][stride] (pairs == 0) or float2 [slabs][stride] (pairs == 1)
    float* out_a; float* out_b;
    long long stride;          // elements between two slabs
    int slabs, C, pairs, accumulate;
    int first, blocks;
    int wide, pad;             // wide: few slabs x many columns (the k-slices of a weight gradient), see below
};

// Two workgroup shapes (1024 threads): "tall" jobs (LayerNorm / bias-table partials: 10^2 .. 10^3 slabs of 10^2 .. 10^3 columns)
// take 32 columns x 32 slab lanes like col_final_kernel; "wide" jobs (split-K partials of a weight gradient: 2 .. 85 slices of
// 10^4 .. 10^6 columns) take 128 float4 column groups x 8 slice lanes in gemm_reduce_kernel's own summation order (fp32) -- a
// tall workgroup would keep KS of its 32 slab lanes busy and read 128-byte pieces (first version: one batch of a SwinUnet step
// took 3 ms).  Fixed order; tall jobs accumulate in double like col_final_kernel.
constexpr int COLSUM_WIDE_COLS = 512;

__global__ __launch_bounds__(1024) void colsum_batch_kernel(const ColsumJob* __restrict__ jobs, int n) {
    __shared__ ColsumJob job;
    __shared__ double ra[1024], rb[1024];
    if (threadIdx.x == 0) {
        int lo = 0, hi = n - 1;
        const int t = (int)blockIdx.x;
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (jobs[mid].first <= t) lo = mid; else hi = mid - 1;
        }
        job = jobs[lo];
    }
    __syncthreads();
    const int slabs = job.slabs;
    const long long ld = job.stride;
    if (job.wide) {
        // thread = (column group cg of 4 floats, slice lane kl of 8): lane kl sums slices kl, kl + 8, ... in order, then lane 0
        // adds lanes 1 .. 7 in order -- gemm_reduce_kernel's sums exactly (fp32, same order): the batched weight gradients are
        // bit-identical to the per-op launches
        const int cg = threadIdx.x & 127, kl = threadIdx.x >> 7;
        const int col = ((int)blockIdx.x - job.first) * COLSUM_WIDE_COLS + cg * 4;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        if (col < job.C) {
            const float* __restrict__ part = reinterpret_cast<const float*>(job.part) + col;
            int k = kl;
            for (; k + 24 < slabs; k += 32) {
                float4 p[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) p[j] = *reinterpret_cast<const float4*>(part + (long long)(k + 8 * j) * ld);
#pragma unroll
                for (int j = 0; j < 4; ++j) { acc.x += p[j].x; acc.y += p[j].y; acc.z += p[j].z; acc.w += p[j].w; }
            }
            for (; k < slabs; k += 8) {
                const float4 p = *reinterpret_cast<const float4*>(part + (long long)k * ld);
                acc.x += p.x; acc.y += p.y; acc.z += p.z; acc.w += p.w;
            }
        }
        __shared__ float4 wred[7][128];
        if (kl) wred[kl - 1][cg] = acc;
        __syncthreads();
        if (kl == 0 && col < job.C) {
#pragma unroll
            for (int j = 0; j < 7; ++j) { const float4 q = wred[j][cg]; acc.x += q.x; acc.y += q.y; acc.z += q.z; acc.w += q.w; }
            float4* o = reinterpret_cast<float4*>(job.out_a + col);
            if (job.accumulate) { const float4 q = *o; acc.x = q.x + acc.x; acc.y = q.y + acc.y; acc.z = q.z + acc.z; acc.w = q.w + acc.w; }
            *o = acc;
        }
        return;
    }
    const int cl = threadIdx.x & 31, sl = threadIdx.x >> 5;
    const int col = ((int)blockIdx.x - job.first) * 32 + cl;
    double a = 0.0, b = 0.0;
    if (col < job.C) {
        if (job.pairs) {
            const float2* __restrict__ part = reinterpret_cast<const float2*>(job.part);
            int s = sl;
            for (; s + 224 < slabs; s += 256) {
                float2 p[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) p[j] = part[(long long)(s + 32 * j) * ld + col];
#pragma unroll
                for (int j = 0; j < 8; ++j) { a += p[j].x; b += p[j].y; }
            }
            for (; s < slabs; s += 32) {
                const float2 p = part[(long long)s * ld + col];
                a += p.x; b += p.y;
            }
        } else {
            const float* __restrict__ part = reinterpret_cast<const float*>(job.part);
            int s = sl;
            for (; s + 224 < slabs; s += 256) {
                float p[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) p[j] = part[(long long)(s + 32 * j) * ld + col];
#pragma unroll
                for (int j = 0; j < 8; ++j) a += p[j];
            }
            for (; s < slabs; s += 32) a += part[(long long)s * ld + col];
        }
    }
    ra[threadIdx.x] = a;
    rb[threadIdx.x] = b;
    __syncthreads();
    if (sl == 0 && col < job.C) {
#pragma unroll
        for (int k = 1; k < 32; ++k) { a += ra[k * 32 + cl]; b += rb[k * 32 + cl]; }
        if (job.out_a) job.out_a[col] = job.accumulate ? job.out_a[col] + (float)a : (float)a;
        if (job.out_b) job.out_b[col] = job.accumulate ? job.out_b[col] + (float)b : (float)b;
    }
}

// ------------------------------------------------------------------ GELU (erf form; mis_gelu in common.h)
__global__ __launch_bounds__(256) void gelu_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                   float* __restrict__ out, long long n4, int backward) {
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
        const float4 v = reinterpret_cast<const float4*>(x)[i];
        const float xs[4] = {v.x, v.y, v.z, v.w};
        float o[4];
        if (!backward) {
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = mis_gelu(xs[j]);
        } else {
            const float4 g = reinterpret_cast<const float4*>(dy)[i];
            const float gs[4] = {g.x, g.y, g.z, g.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = gs[j] * mis_gelu_grad(xs[j]);
        }
        reinterpret_cast<float4*>(out)[i] = make_float4(o[0], o[1], o[2], o[3]);
    }
}

// ------------------------------------------------------------------ residual + DropPath
__device__ __forceinline__ float droppath_scale(float p, unsigned salt, const MisStepState* st, int sample) {
    if (p <= 0.f) return 1.f;
    uint32_t r[4];
    const unsigned long long seed = st->seed, off = st->offset;
    mis_philox4((uint32_t)sample, 0x9E3779B9u, salt, (uint32_t)off, (uint32_t)seed,
                (uint32_t)(seed >> 32) ^ (uint32_t)(off >> 32), r);
    return mis_u01(r[0]) >= p ? 1.f / (1.f - p) : 0.f;
}

// table[site][b] = DropPath scale of sample b at residual site `site` (the values residual_kernel derives on the fly):
// one launch per forward gives the GEMM epilogues that fuse the residual add their per-row scales
__global__ __launch_bounds__(256) void droppath_table_kernel(float* __restrict__ table, const float* __restrict__ p,
                                                             const unsigned* __restrict__ salt, int nsites, int B,
                                                             const MisStepState* st) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= nsites * B) return;
    const int site = i / B, b = i - site * B;
    table[i] = droppath_scale(p[site], salt[site], st, b);
}

// forward:  out[row] = a[row] + s_b * y[row]           (rows of sample b = row / rows_per_sample)
// backward: d_a[row] = dout[row] ; d_y[row] = s_b * dout[row]   (a == dout, y == nullptr)
__global__ __launch_bounds__(256) void residual_kernel(const float* __restrict__ a, long long lda,
                                                       const float* __restrict__ y, long long ldy,
                                                       float* __restrict__ out, long long ldo,
                                                       float* __restrict__ out2, long long ldo2, long long M, int C,
                                                       long long rows_per_sample, float p, unsigned salt,
                                                       const MisStepState* st, const float* __restrict__ scale_override,
                                                       int backward) {
    const int c4 = C >> 2;
    const long long total = M * c4;
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long row = i / c4;
        const int c = (int)(i - row * c4) * 4;
        const int b = (int)(row / rows_per_sample);
        const float s = scale_override ? scale_override[b] : droppath_scale(p, salt, st, b);
        const float4 va = *reinterpret_cast<const float4*>(a + row * lda + c);
        if (!backward) {
            const float4 vy = *reinterpret_cast<const float4*>(y + row * ldy + c);
            *reinterpret_cast<float4*>(out + row * ldo + c) =
                make_float4(va.x + s * vy.x, va.y + s * vy.y, va.z + s * vy.z, va.w + s * vy.w);
        } else {
            if (out) {   // gradient of the shortcut: assign (backward == 1) or accumulate (backward == 2)
                float4 o = va;
                if (backward == 2) {
                    const float4 p0 = *reinterpret_cast<const float4*>(out + row * ldo + c);
                    o.x += p0.x; o.y += p0.y; o.z += p0.z; o.w += p0.w;
                }
                *reinterpret_cast<float4*>(out + row * ldo + c) = o;
            }
            *reinterpret_cast<float4*>(out2 + row * ldo2 + c) = make_float4(s * va.x, s * va.y, s * va.z, s * va.w);
        }
    }
}

// ------------------------------------------------------------------ token re-arrangements
// mode 0: PatchMerging gather   y[b][i][j][q*C + c] = x[b][2i + (q&1)][2j + (q>>1)][c]   (x: HxW tokens of C)
// mode 1: PatchExpand shuffle   y[b][h*P+p1][w*P+p2][c] = x[b][h][w][(p1*P+p2)*C + c]    (x: HxW tokens of P*P*C)
// inverse = 1 swaps the roles (the backward pass: every map is a bijection).
struct RearrArgs {
    const float* src; long long lds;
    float* dst; long long ldd;
    int B, H, W, C, P, mode, inverse;
};

__global__ __launch_bounds__(256) void rearrange_kernel(const RearrArgs a) {
    // enumerate the elements of the "fine" side in float4 units: fine = H*W*... tokens x C channels
    const int c4 = a.C >> 2;
    const int P = a.mode == 0 ? 2 : a.P;
    // mode 0: fine tokens = (H, W) with C ch (input side);  coarse tokens = (H/2, W/2) with 4C ch
    // mode 1: fine tokens = (H*P, W*P) with C ch (output side); coarse tokens = (H, W) with P*P*C ch
    const int FH = a.mode == 0 ? a.H : a.H * P, FW = a.mode == 0 ? a.W : a.W * P;
    const long long total = (long long)a.B * FH * FW * c4;
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int c = (int)(i % c4) * 4;
        long long t = i / c4;
        const int fw = (int)(t % FW); t /= FW;
        const int fh = (int)(t % FH);
        const int b = (int)(t / FH);
        const int ch = fh / P, cw = fw / P, p1 = fh - ch * P, p2 = fw - cw * P;
        const long long fine_row = ((long long)b * FH + fh) * FW + fw;
        const long long coarse_row = ((long long)b * (FH / P) + ch) * (FW / P) + cw;
        const int blk = a.mode == 0 ? (p2 * 2 + p1) : (p1 * P + p2);   // merge: q = dx*2 + dy
        const long long coarse_off = coarse_row * (a.mode == 0 ? (a.inverse ? a.lds : a.ldd) : (a.inverse ? a.ldd : a.lds)) +
                                     (long long)blk * a.C + c;
        const long long fine_off = fine_row * (a.mode == 0 ? (a.inverse ? a.ldd : a.lds) : (a.inverse ? a.lds : a.ldd)) + c;
        // forward: mode 0 reads fine writes coarse; mode 1 reads coarse writes fine.  inverse flips.
        const bool read_fine = (a.mode == 0) != (a.inverse != 0);
        const float4 v = *reinterpret_cast<const float4*>(a.src + (read_fine ? fine_off : coarse_off));
        *reinterpret_cast<float4*>(a.dst + (read_fine ? coarse_off : fine_off)) = v;
    }
}

// PatchEmbed im2col: out[b*Hp*Wp + ph*Wp + pw][c*16 + ky*4 + kx] = x[b][c][4ph+ky][4pw+kx]; src_chans == 1: the
// single input channel feeds every c (the 1 -> 3 channel repeat of vision_transformer.py:49-50)
__global__ __launch_bounds__(256) void patch_im2col_kernel(const float* __restrict__ x, long long x_bs,
                                                           float* __restrict__ out, int B, int H, int W,
                                                           int in_chans, int src_chans) {
    const int Hp = H >> 2, Wp = W >> 2;
    const long long total = (long long)B * Hp * Wp * 4;   // one thread per (patch, ky): 4 floats
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int ky = (int)(i & 3);
        long long t = i >> 2;
        const int pw = (int)(t % Wp); t /= Wp;
        const int ph = (int)(t % Hp);
        const int b = (int)(t / Hp);
        const float* __restrict__ px = x + (long long)b * x_bs + (long long)(4 * ph + ky) * W + 4 * pw;
        float4 v = *reinterpret_cast<const float4*>(px);
        float* o = out + (((long long)b * Hp + ph) * Wp + pw) * (in_chans * 16) + ky * 4;
        for (int c = 0; c < in_chans; ++c) {
            if (src_chans > 1 && c > 0) v = *reinterpret_cast<const float4*>(px + (long long)c * H * W);
            *reinterpret_cast<float4*>(o + c * 16) = v;
        }
    }
}

// ------------------------------------------------------------------ UNETR patch embedding helpers
// MONAI PatchEmbeddingBlock, pos_embed = "perceptron" (reference code/networks/unetr.py:88-99 builds ViT with it):
//   Rearrange("b c (h p1) (w p2) (d p3) -> b (h w d) (p1 p2 p3 c)") -> Linear -> + position_embeddings[1, L, hidden]
// out[(b*L + (h*Wp + w)*Dp + d)][((p1*P + p2)*P + p3)*C + c] = x[b][c][h*P + p1][w*P + p2][d*P + p3];  C == 1 here
// (the reference builds UNETR with in_channels = 1): 4 consecutive p3 = one float4.
__global__ __launch_bounds__(256) void patch3d_im2col_kernel(const float* __restrict__ x, long long x_bs,
                                                             float* __restrict__ out, int B, int H, int W, int D, int P) {
    const int Hp = H / P, Wp = W / P, Dp = D / P, P4 = P >> 2;
    const long long total = (long long)B * Hp * Wp * Dp * P * P * P4;
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        long long t = i;
        const int q3 = (int)(t % P4); t /= P4;
        const int p2 = (int)(t % P); t /= P;
        const int p1 = (int)(t % P); t /= P;
        const int d = (int)(t % Dp); t /= Dp;
        const int w = (int)(t % Wp); t /= Wp;
        const int h = (int)(t % Hp);
        const int b = (int)(t / Hp);
        const float4 v = *reinterpret_cast<const float4*>(
            x + (long long)b * x_bs + ((long long)(h * P + p1) * W + (w * P + p2)) * D + d * P + q3 * 4);
        float* o = out + ((((long long)b * Hp + h) * Wp + w) * Dp + d) * ((long long)P * P * P) + ((p1 * P + p2) * P + q3 * 4);
        *reinterpret_cast<float4*>(o) = v;
    }
}

// out[row][c] = x[row][c] + pos[row % L][c]
__global__ __launch_bounds__(256) void add_rowcycle_kernel(const float* __restrict__ x, long long ldx,
                                                           const float* __restrict__ pos, float* __restrict__ out,
                                                           long long ldo, long long M, int C, int L) {
    const int c4 = C >> 2;
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < M * c4; i += (long long)gridDim.x * 256) {
        const long long row = i / c4;
        const int c = (int)(i - row * c4) * 4;
        const float4 a = *reinterpret_cast<const float4*>(x + row * ldx + c);
        const float4 p = *reinterpret_cast<const float4*>(pos + (row % L) * C + c);
        *reinterpret_cast<float4*>(out + row * ldo + c) = make_float4(a.x + p.x, a.y + p.y, a.z + p.z, a.w + p.w);
    }
}

// dpos[l][c] = sum_b dy[b*L + l][c]     (fixed order over b: deterministic)
__global__ __launch_bounds__(256) void sum_rowcycle_kernel(const float* __restrict__ dy, long long ld,
                                                           float* __restrict__ dpos, long long M, int C, int L) {
    const int c4 = C >> 2;
    const long long i = blockIdx.x * 256LL + threadIdx.x;
    if (i >= (long long)L * c4) return;
    const int l = (int)(i / c4), c = (int)(i - (long long)l * c4) * 4;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    for (long long row = l; row < M; row += L) {
        const float4 v = *reinterpret_cast<const float4*>(dy + row * ld + c);
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    *reinterpret_cast<float4*>(dpos + (long long)l * C + c) = s;
}

// ------------------------------------------------------------------ output head (token-major -> NCHW logits)
// logits[b][n][pix] = sum_k x[b*S + pix][k] * w[n][k]      (K % 4 == 0, NC <= 8)
template <int NC>
__global__ __launch_bounds__(256) void head_fwd_kernel(const float* __restrict__ x, long long ldx,
                                                       const float* __restrict__ w, float* __restrict__ y,
                                                       long long y_bs, int B, long long S, int K) {
    extern __shared__ float sw[];
    for (int i = threadIdx.x; i < NC * K; i += 256) sw[i] = w[i];
    __syncthreads();
    const long long total = (long long)B * S;
    for (long long t = blockIdx.x * 256LL + threadIdx.x; t < total; t += (long long)gridDim.x * 256) {
        const float* __restrict__ xr = x + t * ldx;
        float acc[NC];
#pragma unroll
        for (int n = 0; n < NC; ++n) acc[n] = 0.f;
        for (int k = 0; k < K; k += 4) {
            const float4 v = *reinterpret_cast<const float4*>(xr + k);
#pragma unroll
            for (int n = 0; n < NC; ++n)
                acc[n] += (v.x * sw[n * K + k] + v.y * sw[n * K + k + 1]) + (v.z * sw[n * K + k + 2] + v.w * sw[n * K + k + 3]);
        }
        const int b = (int)(t / S);
        const long long pix = t - (long long)b * S;
#pragma unroll
        for (int n = 0; n < NC; ++n) y[(long long)b * y_bs + (long long)n * S + pix] = acc[n];
    }
}

// dx[t][k] = sum_n dlogits[b][n][pix] * w[n][k];  partial dW[n][k] per block (fixed order).
// Thread = (float4 column quad kq = tid & 31, token sub-row tr = tid >> 5): a row of x is read as coalesced
// float4s, the NC logits gradients of the token are the same for its 32 lanes, and every thread keeps its own
// NC x 4 slice of dW in registers over the block's token range -- no cross-lane reduction inside the loop (the first
// version did 4*NC wave reductions per token group and took 1.3 ms on config 4); the 8 sub-rows are combined
// through LDS at the end in a fixed order.
template <int NC>
__global__ __launch_bounds__(256) void head_bwd_kernel(const float* __restrict__ x, long long ldx,
                                                       const float* __restrict__ w, const float* __restrict__ dy,
                                                       long long dy_bs, float* __restrict__ dx, long long lddx,
                                                       float* __restrict__ part, int B, long long S, int K) {
    extern __shared__ float sm[];          // [8][NC*K]
    const int kq = threadIdx.x & 31, tr = threadIdx.x >> 5;
    const bool kact = kq * 4 < K;
    const long long total = (long long)B * S;
    const long long per = (total + gridDim.x - 1) / gridDim.x;
    const long long t_begin = blockIdx.x * per, t_end = t_begin + per < total ? t_begin + per : total;
    float4 wv[NC], acc[NC];
#pragma unroll
    for (int n = 0; n < NC; ++n) {
        wv[n] = kact ? *reinterpret_cast<const float4*>(w + (long long)n * K + kq * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
        acc[n] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    for (long long t = t_begin + tr; t < t_end; t += 8) {
        const int b = (int)(t / S);
        const long long pix = t - (long long)b * S;
        float g[NC];
#pragma unroll
        for (int n = 0; n < NC; ++n) g[n] = dy[(long long)b * dy_bs + (long long)n * S + pix];
        if (kact) {
            const float4 v = *reinterpret_cast<const float4*>(x + t * ldx + kq * 4);
            float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int n = 0; n < NC; ++n) {
                o.x += g[n] * wv[n].x; o.y += g[n] * wv[n].y; o.z += g[n] * wv[n].z; o.w += g[n] * wv[n].w;
                acc[n].x += g[n] * v.x; acc[n].y += g[n] * v.y; acc[n].z += g[n] * v.z; acc[n].w += g[n] * v.w;
            }
            *reinterpret_cast<float4*>(dx + t * lddx + kq * 4) = o;
        }
    }
    if (kact) {
#pragma unroll
        for (int n = 0; n < NC; ++n) *reinterpret_cast<float4*>(sm + (tr * NC + n) * K + kq * 4) = acc[n];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < NC * K; i += 256) {
        float s = 0.f;
#pragma unroll
        for (int r = 0; r < 8; ++r) s += sm[r * NC * K + i];
        part[(long long)blockIdx.x * NC * K + i] = s;
    }
}

__global__ __launch_bounds__(1024) void head_dw_final_kernel(const float* __restrict__ part, int blocks, int n,
                                                             float* __restrict__ dw, int accumulate) {
    // block = 32 outputs x 32 partial lanes (coalesced 128-byte reads of the partial rows), double accumulators, fixed-order
    // LDS tree: the fused LayerNorm + head backward leaves ~2 000 partial rows -- one thread per output summed them serially
    // in 0.67 ms
    __shared__ double red[1024];
    const int cl = threadIdx.x & 31, sl = threadIdx.x >> 5;
    const int i = blockIdx.x * 32 + cl;
    double s = 0.0;
    if (i < n) {
        int b = sl;
        for (; b + 96 < blocks; b += 128) {
            const float p0 = part[(long long)b * n + i], p1 = part[(long long)(b + 32) * n + i];
            const float p2 = part[(long long)(b + 64) * n + i], p3 = part[(long long)(b + 96) * n + i];
            s += p0; s += p1; s += p2; s += p3;
        }
        for (; b < blocks; b += 32) s += part[(long long)b * n + i];
    }
    red[threadIdx.x] = s;
    __syncthreads();
    if (sl == 0 && i < n) {
#pragma unroll
        for (int k = 1; k < 32; ++k) s += red[k * 32 + cl];
        dw[i] = accumulate ? dw[i] + (float)s : (float)s;
    }
}

// ------------------------------------------------------------------ LayerNorm + output head in one pass
// The last two layers of SwinUnet: FinalPatchExpand_X4.norm = nn.LayerNorm(C = 96) on the 16x expanded token tensor, then
// self.output = nn.Conv2d(C, NC, 1, bias=False) (reference swin_transformer_unet_skip_expand_decoder_sys.py:390-409, :671,
// :749-752).  The normalised tensor (925 MB at 24+24 images of 224^2) has one reader; here it never exists:
//   forward   x -> (mean, rstd) and logits[b][n][pix] = sum_c w[n][c] * (xhat_c gamma_c + beta_c)        one read of x
//   backward  dlogits, x -> dx (LayerNorm backward of dy_c = sum_n dlogits_n w[n][c]), and the column sums that make
//             dgamma, dbeta, dw -- one read of x, one write of dx (un-fused: 3 reads + 2 writes of a 925 MB tensor more)
// 32 lanes per row (C <= 128), a float4 of columns per lane, 8 rows per 256 threads.
template <int NC>
__global__ __launch_bounds__(256) void ln_head_fwd_kernel(const float* __restrict__ x, long long ldx,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta,
                                                          const float* __restrict__ w, float* __restrict__ mean,
                                                          float* __restrict__ rstd, float* __restrict__ y, long long y_bs,
                                                          long long M, long long S, int C, float eps) {
    const int lane = threadIdx.x & 31, rg = threadIdx.x >> 5;
    const int c = lane * 4;
    const bool act = c < C;
    float4 g = make_float4(0.f, 0.f, 0.f, 0.f), bt = g, wv[NC];
    if (act) { g = *reinterpret_cast<const float4*>(gamma + c); bt = *reinterpret_cast<const float4*>(beta + c); }
#pragma unroll
    for (int n = 0; n < NC; ++n) wv[n] = act ? *reinterpret_cast<const float4*>(w + (long long)n * C + c) : make_float4(0.f, 0.f, 0.f, 0.f);
    constexpr int RU = 4;      // rows of a row group in flight
    __shared__ float sl[NC][8 * RU];
    for (long long row0 = blockIdx.x * (8LL * RU) + rg; row0 - rg < M; row0 += (long long)gridDim.x * 8 * RU) {
        float4 vv[RU];
#pragma unroll
        for (int u = 0; u < RU; ++u) {
            const long long row = row0 + 8LL * u;
            vv[u] = (act && row < M) ? *reinterpret_cast<const float4*>(x + row * ldx + c) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int u = 0; u < RU; ++u) {
            const long long row = row0 + 8LL * u;
            const float4 v = vv[u];
            const float m = group_sum<32>((v.x + v.y) + (v.z + v.w)) / (float)C;
            const float a0 = v.x - m, a1 = v.y - m, a2 = v.z - m, a3 = v.w - m;
            const float ss = act ? (a0 * a0 + a1 * a1) + (a2 * a2 + a3 * a3) : 0.f;
            const float rs = 1.f / sqrtf(group_sum<32>(ss) / (float)C + eps);
            const float y0 = a0 * rs * g.x + bt.x, y1 = a1 * rs * g.y + bt.y, y2 = a2 * rs * g.z + bt.z, y3 = a3 * rs * g.w + bt.w;
            float p[NC];
#pragma unroll
            for (int n = 0; n < NC; ++n) p[n] = group_sum<32>((y0 * wv[n].x + y1 * wv[n].y) + (y2 * wv[n].z + y3 * wv[n].w));
            if (lane == 0) {
                if (row < M) { mean[row] = m; rstd[row] = rs; }
#pragma unroll
                for (int n = 0; n < NC; ++n) sl[n][u * 8 + rg] = p[n];
            }
        }
        // the block's 32 rows are consecutive pixels: one 128-byte piece per class instead of 4-byte stores from lane 0
        __syncthreads();
        if (threadIdx.x < NC * 32) {
            const int n = threadIdx.x >> 5, j = threadIdx.x & 31;
            const long long row = row0 - rg + (j >> 3) * 8LL + (j & 7);      // slot u*8 + rg  <->  row row0' + 8u + rg
            if (row < M) {
                const long long b = row / S, pix = row - b * S;
                y[b * y_bs + (long long)n * S + pix] = sl[n][j];
            }
        }
        __syncthreads();
    }
}

// C = 96 (SwinUnet's tail): 8 lanes per token row, three float4 per lane (mis_ln96_head_row, common.h) -- no idle lanes, 3-step
// reductions; 32 rows per pass, RU passes in flight.  The logits of a pass are 32 consecutive pixels per class: staged through LDS
// and stored as 128-byte pieces.
template <int NC>
__global__ __launch_bounds__(256) void ln96_head_fwd_kernel(const float* __restrict__ x, long long ldx,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta,
                                                            const float* __restrict__ w, float* __restrict__ mean,
                                                            float* __restrict__ rstd, float* __restrict__ y, long long y_bs,
                                                            long long M, long long S, float eps) {
    constexpr int RU = 2, NCM = 4;
    const int l8 = threadIdx.x & 7, rg = threadIdx.x >> 3;
    float4 g[3], bt[3], wv[NCM][3];
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        const int c4 = (l8 + 8 * q) * 4;
        g[q] = *reinterpret_cast<const float4*>(gamma + c4);
        bt[q] = *reinterpret_cast<const float4*>(beta + c4);
#pragma unroll
        for (int n = 0; n < NCM; ++n) wv[n][q] = n < NC ? *reinterpret_cast<const float4*>(w + n * 96 + c4) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    __shared__ float sl[NC][32 * RU];
    for (long long base = blockIdx.x * (32LL * RU); base < M; base += (long long)gridDim.x * 32 * RU) {
        float4 vv[RU][3];
#pragma unroll
        for (int u = 0; u < RU; ++u) {
            const long long row = base + 32 * u + rg < M ? base + 32 * u + rg : M - 1;
#pragma unroll
            for (int q = 0; q < 3; ++q) vv[u][q] = *reinterpret_cast<const float4*>(x + row * ldx + (l8 + 8 * q) * 4);
        }
#pragma unroll
        for (int u = 0; u < RU; ++u) {
            const long long row = base + 32 * u + rg;
            float mu, rs, pl[NCM];
            mis_ln96_head_row<NCM>(vv[u], g, bt, wv, eps, mu, rs, pl);
            if (l8 == 0) {
                if (row < M) { mean[row] = mu; rstd[row] = rs; }
#pragma unroll
                for (int n = 0; n < NC; ++n) sl[n][u * 32 + rg] = pl[n];
            }
        }
        __syncthreads();
        if (threadIdx.x < NC * 32 * RU) {
            const int n = threadIdx.x / (32 * RU), j = threadIdx.x % (32 * RU);
            const long long row = base + j;
            if (row < M) {
                const long long b = row / S, pix = row - b * S;
                y[b * y_bs + (long long)n * S + pix] = sl[n][j];
            }
        }
        __syncthreads();
    }
}

// one block per slab of rows; partial column sums per slab: pln[slab][c] = (sum dy xhat, sum dy), pw[slab][n][c] = sum dl_n y_c
template <int NC>
__global__ __launch_bounds__(256) void ln_head_bwd_kernel(const float* __restrict__ x, long long ldx,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta,
                                                          const float* __restrict__ w, const float* __restrict__ mean,
                                                          const float* __restrict__ rstd, const float* __restrict__ dl,
                                                          long long dl_bs, float* __restrict__ dx, long long lddx,
                                                          int accumulate, long long M, long long S, int C,
                                                          long long rows_per_slab, float2* __restrict__ pln,
                                                          float* __restrict__ pw, int ex_P, int ex_H, int ex_W) {
    // ex_P > 0: row r of x is token (b, h P + p1, w P + p2) of the pixel-shuffled grid of FinalPatchExpand_X4 ('b h w (p1 p2 c) ->
    // b (h p1) (w p2) c'); its gradient row goes straight to the expand Linear's output layout dx[(b, h, w)][(p1 P + p2) C + c]
    // (row stride lddx = P P C): the un-shuffle pass over the largest tensor of the network is gone
    __shared__ float4 red[256];
    const int lane = threadIdx.x & 31, rg = threadIdx.x >> 5;
    const int c = lane * 4;
    const bool act = c < C;
    float4 g = make_float4(0.f, 0.f, 0.f, 0.f), bt = g, wv[NC], aw[NC];
    if (act) { g = *reinterpret_cast<const float4*>(gamma + c); bt = *reinterpret_cast<const float4*>(beta + c); }
#pragma unroll
    for (int n = 0; n < NC; ++n) {
        wv[n] = act ? *reinterpret_cast<const float4*>(w + (long long)n * C + c) : make_float4(0.f, 0.f, 0.f, 0.f);
        aw[n] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    float4 dg = make_float4(0.f, 0.f, 0.f, 0.f), db = dg;
    const long long r0 = blockIdx.x * rows_per_slab;
    long long r1 = r0 + rows_per_slab;
    if (r1 > M) r1 = M;
    constexpr int RU = 2;      // rows of a row group in flight (loads first, then the reductions)
    for (long long row0 = r0 + rg; row0 < r1; row0 += 8 * RU) {
        float d[RU][NC], mm[RU], rr[RU];
        float4 vv[RU];
#pragma unroll
        for (int u = 0; u < RU; ++u) {
            const long long row = row0 + 8 * u < r1 ? row0 + 8 * u : row0;      // a dead slot re-reads the live row
            const long long b = row / S, pix = row - b * S;
#pragma unroll
            for (int n = 0; n < NC; ++n) d[u][n] = dl[b * dl_bs + (long long)n * S + pix];
            mm[u] = mean[row]; rr[u] = rstd[row];
            vv[u] = act ? *reinterpret_cast<const float4*>(x + row * ldx + c) : make_float4(mm[u], mm[u], mm[u], mm[u]);
        }
#pragma unroll
        for (int u = 0; u < RU; ++u) {
            const long long row = row0 + 8 * u;
            if (row >= r1) break;          // uniform within the row group (all 32 lanes share the row)
            const float m = mm[u], rs = rr[u];
            const float4 v = vv[u];
            const float h0 = (v.x - m) * rs, h1 = (v.y - m) * rs, h2 = (v.z - m) * rs, h3 = (v.w - m) * rs;      // xhat
            const float y0 = h0 * g.x + bt.x, y1 = h1 * g.y + bt.y, y2 = h2 * g.z + bt.z, y3 = h3 * g.w + bt.w;
            float e0 = 0.f, e1 = 0.f, e2 = 0.f, e3 = 0.f;                                                     // dy
#pragma unroll
            for (int n = 0; n < NC; ++n) {
                const float dn = d[u][n];
                e0 += dn * wv[n].x; e1 += dn * wv[n].y; e2 += dn * wv[n].z; e3 += dn * wv[n].w;
                aw[n].x += dn * y0; aw[n].y += dn * y1; aw[n].z += dn * y2; aw[n].w += dn * y3;
            }
            dg.x += e0 * h0; dg.y += e1 * h1; dg.z += e2 * h2; dg.w += e3 * h3;
            db.x += e0; db.y += e1; db.z += e2; db.w += e3;
            const float q0 = e0 * g.x, q1 = e1 * g.y, q2 = e2 * g.z, q3 = e3 * g.w;
            const float s1 = group_sum<32>((q0 + q1) + (q2 + q3)) / (float)C;
            const float s2 = group_sum<32>((q0 * h0 + q1 * h1) + (q2 * h2 + q3 * h3)) / (float)C;
            if (act) {
                float4 o = make_float4(rs * (q0 - s1 - h0 * s2), rs * (q1 - s1 - h1 * s2), rs * (q2 - s1 - h2 * s2),
                                       rs * (q3 - s1 - h3 * s2));
                float* op = dx + row * lddx + c;
                if (ex_P) {
                    const long long bimg = row / S;                      // S = H P W P < 2^31 (checked by the launcher)
                    const int WP = ex_W * ex_P, pix = (int)(row - bimg * S), yy = pix / WP, xx = pix - yy * WP;
                    const int h = yy / ex_P, w_ = xx / ex_P;
                    const int p1 = yy - h * ex_P, p2 = xx - w_ * ex_P;
                    op = dx + ((bimg * ex_H + h) * ex_W + w_) * lddx + (p1 * ex_P + p2) * C + c;
                }
                if (accumulate) {
                    const float4 pv = *reinterpret_cast<const float4*>(op);
                    o.x += pv.x; o.y += pv.y; o.z += pv.z; o.w += pv.w;
                }
                *reinterpret_cast<float4*>(op) = o;
            }
        }
    }
    // the 8 row groups of the block, fixed order, one quantity at a time through LDS
    auto combine = [&](float4 val) -> float4 {
        __syncthreads();
        red[threadIdx.x] = val;
        __syncthreads();
        float4 s = red[lane];
#pragma unroll
        for (int k = 1; k < 8; ++k) {
            const float4 pv = red[k * 32 + lane];
            s.x += pv.x; s.y += pv.y; s.z += pv.z; s.w += pv.w;
        }
        return s;
    };
    const float4 sg = combine(dg), sb = combine(db);
    if (rg == 0 && act) {
        float2* o = pln + (long long)blockIdx.x * C + c;
        o[0] = make_float2(sg.x, sb.x); o[1] = make_float2(sg.y, sb.y);
        o[2] = make_float2(sg.z, sb.z); o[3] = make_float2(sg.w, sb.w);
    }
#pragma unroll
    for (int n = 0; n < NC; ++n) {
        const float4 sw_ = combine(aw[n]);
        if (rg == 0 && act) *reinterpret_cast<float4*>(pw + ((long long)blockIdx.x * NC + n) * C + c) = sw_;
    }
}

// ln_head_bwd_kernel for C = 96 with 8 lanes per token row and three float4 per lane (as ln96_head_fwd_kernel): two 3-step
// reductions per row instead of two 6-step ones on 24 of 32 lanes, 32 rows per pass.  Same partial layout (one row per slab).
template <int NC>
__global__ __launch_bounds__(256) void ln96_head_bwd_kernel(const float* __restrict__ x, long long ldx,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta,
                                                            const float* __restrict__ w, const float* __restrict__ mean,
                                                            const float* __restrict__ rstd, const float* __restrict__ dl,
                                                            long long dl_bs, float* __restrict__ dx, long long lddx,
                                                            int accumulate, long long M, long long S, long long rows_per_slab,
                                                            float2* __restrict__ pln, float* __restrict__ pw, int ex_P, int ex_H,
                                                            int ex_W) {
    constexpr int C = 96;
    __shared__ float4 red[256];
    const int l8 = threadIdx.x & 7, rg = threadIdx.x >> 3;
    float4 g[3], bt[3], wv[NC][3], aw[NC][3], dg[3], db[3];
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        const int c4 = (l8 + 8 * q) * 4;
        g[q] = *reinterpret_cast<const float4*>(gamma + c4);
        bt[q] = *reinterpret_cast<const float4*>(beta + c4);
        dg[q] = make_float4(0.f, 0.f, 0.f, 0.f); db[q] = dg[q];
#pragma unroll
        for (int n = 0; n < NC; ++n) {
            wv[n][q] = *reinterpret_cast<const float4*>(w + n * C + c4);
            aw[n][q] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    const long long r0 = blockIdx.x * rows_per_slab;
    long long r1 = r0 + rows_per_slab;
    if (r1 > M) r1 = M;
    for (long long row = r0 + rg; row < r1; row += 32) {
        const long long b = row / S, pix = row - b * S;
        float d[NC];
#pragma unroll
        for (int n = 0; n < NC; ++n) d[n] = dl[b * dl_bs + (long long)n * S + pix];
        const float m = mean[row], rs = rstd[row];
        float4 v[3];
#pragma unroll
        for (int q = 0; q < 3; ++q) v[q] = *reinterpret_cast<const float4*>(x + row * ldx + (l8 + 8 * q) * 4);
        float4 h[3], qv[3];
        float t1 = 0.f, t2 = 0.f;
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            h[q] = make_float4((v[q].x - m) * rs, (v[q].y - m) * rs, (v[q].z - m) * rs, (v[q].w - m) * rs);          // xhat
            const float4 y = make_float4(h[q].x * g[q].x + bt[q].x, h[q].y * g[q].y + bt[q].y, h[q].z * g[q].z + bt[q].z,
                                         h[q].w * g[q].w + bt[q].w);
            float4 e = make_float4(0.f, 0.f, 0.f, 0.f);                                                             // dy
#pragma unroll
            for (int n = 0; n < NC; ++n) {
                const float dn = d[n];
                e.x += dn * wv[n][q].x; e.y += dn * wv[n][q].y; e.z += dn * wv[n][q].z; e.w += dn * wv[n][q].w;
                aw[n][q].x += dn * y.x; aw[n][q].y += dn * y.y; aw[n][q].z += dn * y.z; aw[n][q].w += dn * y.w;
            }
            dg[q].x += e.x * h[q].x; dg[q].y += e.y * h[q].y; dg[q].z += e.z * h[q].z; dg[q].w += e.w * h[q].w;
            db[q].x += e.x; db[q].y += e.y; db[q].z += e.z; db[q].w += e.w;
            qv[q] = make_float4(e.x * g[q].x, e.y * g[q].y, e.z * g[q].z, e.w * g[q].w);
            t1 += (qv[q].x + qv[q].y) + (qv[q].z + qv[q].w);
            t2 += (qv[q].x * h[q].x + qv[q].y * h[q].y) + (qv[q].z * h[q].z + qv[q].w * h[q].w);
        }
        const float s1 = mis_sum8(t1) / (float)C, s2 = mis_sum8(t2) / (float)C;
        float* orow = dx + row * lddx;
        if (ex_P) {
            const int WP = ex_W * ex_P, pixi = (int)pix, yy = pixi / WP, xx = pixi - yy * WP;
            const int hh = yy / ex_P, w_ = xx / ex_P;
            const int p1 = yy - hh * ex_P, p2 = xx - w_ * ex_P;
            orow = dx + ((b * ex_H + hh) * ex_W + w_) * lddx + (p1 * ex_P + p2) * C;
        }
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            float4 o = make_float4(rs * (qv[q].x - s1 - h[q].x * s2), rs * (qv[q].y - s1 - h[q].y * s2),
                                   rs * (qv[q].z - s1 - h[q].z * s2), rs * (qv[q].w - s1 - h[q].w * s2));
            float* op = orow + (l8 + 8 * q) * 4;
            if (accumulate) {
                const float4 pv = *reinterpret_cast<const float4*>(op);
                o.x += pv.x; o.y += pv.y; o.z += pv.z; o.w += pv.w;
            }
            *reinterpret_cast<float4*>(op) = o;
        }
    }
    // the 32 row groups of the block, fixed order, one float4 quantity at a time through LDS
    auto combine = [&](float4 val) -> float4 {
        __syncthreads();
        red[threadIdx.x] = val;
        __syncthreads();
        float4 sres = red[l8];
#pragma unroll 8
        for (int k = 1; k < 32; ++k) {
            const float4 pv = red[k * 8 + l8];
            sres.x += pv.x; sres.y += pv.y; sres.z += pv.z; sres.w += pv.w;
        }
        return sres;
    };
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        const float4 sg = combine(dg[q]), sb = combine(db[q]);
        if (rg == 0) {
            float2* o = pln + (long long)blockIdx.x * C + (l8 + 8 * q) * 4;
            o[0] = make_float2(sg.x, sb.x); o[1] = make_float2(sg.y, sb.y);
            o[2] = make_float2(sg.z, sb.z); o[3] = make_float2(sg.w, sb.w);
        }
#pragma unroll
        for (int n = 0; n < NC; ++n) {
            const float4 sw_ = combine(aw[n][q]);
            if (rg == 0) *reinterpret_cast<float4*>(pw + ((long long)blockIdx.x * NC + n) * C + (l8 + 8 * q) * 4) = sw_;
        }
    }
}

// out[c][r] = in[r][c]   (weight transpose for the Linear input-gradient GEMM; 32x32 LDS tiles)
__global__ __launch_bounds__(256) void transpose_kernel(const float* __restrict__ in, long long ldi,
                                                        float* __restrict__ out, long long ldo, int rows, int cols) {
    __shared__ float tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
    const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
#pragma unroll
    for (int i = 0; i < 32; i += 8) {
        const int r = r0 + ty + i, c = c0 + tx;
        tile[ty + i][tx] = (r < rows && c < cols) ? in[(long long)r * ldi + c] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 32; i += 8) {
        const int c = c0 + ty + i, r = r0 + tx;
        if (c < cols && r < rows) out[(long long)c * ldo + r] = tile[tx][ty + i];
    }
}

// every Linear weight of a network in one launch: a device table of jobs ordered by `first` (prefix sum of their 32x32
// tiles); a workgroup finds the job of its tile by binary search (66 separate launches of ~5 us per SwinUnet backward before)
struct TransposeJob {
    const float* in; float* out;
    int rows, cols, tiles_c, first;
};

__global__ __launch_bounds__(256) void transpose_batch_kernel(const TransposeJob* __restrict__ jobs, int n) {
    __shared__ float tile[32][33];
    __shared__ TransposeJob job;
    __shared__ int local;
    if (threadIdx.x == 0) {
        int lo = 0, hi = n - 1;
        const int t = (int)blockIdx.x;
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (jobs[mid].first <= t) lo = mid; else hi = mid - 1;
        }
        job = jobs[lo];
        local = t - jobs[lo].first;
    }
    __syncthreads();
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int r0 = (local / job.tiles_c) * 32, c0 = (local % job.tiles_c) * 32;
    const int rows = job.rows, cols = job.cols;
#pragma unroll
    for (int i = 0; i < 32; i += 8) {
        const int r = r0 + ty + i, c = c0 + tx;
        tile[ty + i][tx] = (r < rows && c < cols) ? job.in[(long long)r * cols + c] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 32; i += 8) {
        const int c = c0 + ty + i, r = r0 + tx;
        if (c < cols && r < rows) job.out[(long long)c * rows + r] = tile[tx][ty + i];
    }
}

unsigned sgrid(long long units) {
    long long b = mis_cdiv(units, 256);
    if (b > 4096) b = 4096;
    if (b < 1) b = 1;
    return (unsigned)b;
}

bool a16(const void* p) { return ((uintptr_t)p & 15) == 0; }

// rows per slab of the column reductions (and of the LayerNorm backward blocks): a multiple of 8 giving
// about 2048 slabs, at least 8 rows
long long col_slab_rows(long long M) {
    long long r = mis_cdiv(M, 2048);
    r = (r + 7) / 8 * 8;
    return r < 8 ? 8 : r;
}
#define COL_SLAB_ROWS col_slab_rows(M)
constexpr int HEAD_BLOCKS = 512;

}  // namespace

extern "C" int mis_transpose(const float* in, long long ldi, float* out, long long ldo, int rows, int cols,
                             hipStream_t stream) {
    if (!in || !out || rows <= 0 || cols <= 0 || ldi < cols || ldo < rows) return MIS_ERR_ARG;
    hipLaunchKernelGGL(transpose_kernel, dim3((cols + 31) / 32, (rows + 31) / 32), dim3(256), 0, stream, in, ldi, out,
                       ldo, rows, cols);
    return mis_launch_status();
}

// Batched form: `jobs` = device array of n job records built by mis_transpose_job (dense row-major matrices:
// out[cols][rows] = in[rows][cols]), `tiles` = the sum of their 32x32 tiles.
extern "C" long long mis_transpose_job_bytes(void) { return (long long)sizeof(TransposeJob); }

// fills one record on the HOST; returns the job's number of tiles (the next job's `first` = first + that), < 0: error
extern "C" long long mis_transpose_job(void* job, const float* in, float* out, int rows, int cols, long long first) {
    if (!job || !in || !out || rows <= 0 || cols <= 0 || first < 0) return MIS_ERR_ARG;
    TransposeJob j;
    j.in = in; j.out = out; j.rows = rows; j.cols = cols; j.tiles_c = (cols + 31) / 32; j.first = (int)first;
    const long long tiles = (long long)j.tiles_c * ((rows + 31) / 32);
    if (first + tiles > 0x7fffffffLL) return MIS_ERR_ARG;
    memcpy(job, &j, sizeof(j));
    return tiles;
}

extern "C" int mis_transpose_batch(const void* jobs, int n, long long tiles, hipStream_t stream) {
    if (!jobs || n <= 0 || tiles <= 0 || tiles > 0x7fffffffLL) return MIS_ERR_ARG;
    hipLaunchKernelGGL(transpose_batch_kernel, dim3((unsigned)tiles), dim3(256), 0, stream,
                       reinterpret_cast<const TransposeJob*>(jobs), n);
    return mis_launch_status();
}

extern "C" int mis_layernorm_fwd(const float* x, long long ldx, float* y, long long ldy, const float* gamma,
                                 const float* beta, float* mean, float* rstd, long long M, int C, float eps,
                                 hipStream_t stream) {
    if (!x || !y || !gamma || !beta || !mean || !rstd || M <= 0 || C <= 0) return MIS_ERR_ARG;
    if (C % 4 || ldx % 4 || ldy % 4 || !a16(x) || !a16(y) || !a16(gamma) || !a16(beta)) return MIS_ERR_UNSUPPORTED;
#define MIS_LN_FWD(LPR, NV, RU)                                                                                             \
    hipLaunchKernelGGL((ln_fwd_reg_kernel<LPR, NV, RU>), dim3((unsigned)mis_cdiv(M, (256 / LPR) * RU)), dim3(256), 0, stream, \
                       x, ldx, y, ldy, gamma, beta, mean, rstd, M, C, eps)
    if (C <= 128) MIS_LN_FWD(32, 1, 4);
    else if (C <= 256) MIS_LN_FWD(64, 1, 4);
    else if (C <= 512) MIS_LN_FWD(64, 2, 2);
    else if (C <= 768) MIS_LN_FWD(64, 3, 2);
    else if (C <= 1024) MIS_LN_FWD(64, 4, 1);
    else if (C <= 1536) MIS_LN_FWD(64, 6, 1);
    else
        hipLaunchKernelGGL(ln_fwd_kernel, dim3((unsigned)mis_cdiv(M, 4)), dim3(256), 0, stream, x, ldx, y, ldy,
                           gamma, beta, mean, rstd, M, C, eps);
#undef MIS_LN_FWD
    return mis_launch_status();
}

extern "C" long long mis_colreduce_workspace_bytes(long long M, int C) {
    if (M <= 0 || C <= 0) return MIS_ERR_ARG;
    return mis_cdiv(M, COL_SLAB_ROWS) * C * (long long)sizeof(float2);
}

// The data-gradient half of mis_layernorm_bwd: dx (+)= ..., and (affine) the per-slab partials of dgamma / dbeta into
// `workspace`.  mis_layernorm_bwd_final turns the partials into dgamma / dbeta -- on any stream ordered behind this one: the
// token plans run it beside the data-gradient chain (nothing downstream reads the affine gradients).
static int layernorm_bwd_parts(const float* x, long long ldx, const float* dy, long long lddy, float* dx, long long lddx,
                               const float* gamma, const float* mean, const float* rstd, long long M, int C,
                               int accumulate_dx, bool affine, void* workspace, long long workspace_bytes,
                               hipStream_t stream, const LnRes rz = LnRes{nullptr, 0, nullptr, 0, nullptr, 1}) {
    if (!x || !dy || !dx || !gamma || !mean || !rstd || !workspace || M <= 0 || C <= 0) return MIS_ERR_ARG;
    if (C % 4 || ldx % 4 || lddy % 4 || lddx % 4 || !a16(x) || !a16(dy) || !a16(dx) || !a16(gamma))
        return MIS_ERR_UNSUPPORTED;
    if (workspace_bytes < mis_colreduce_workspace_bytes(M, C)) return MIS_ERR_WORKSPACE;
    float2* part = reinterpret_cast<float2*>(workspace);
    const int slabs = (int)mis_cdiv(M, COL_SLAB_ROWS);
    // C <= 1536: one pass, the slab kernel writes dx and the per-slab affine partials together
#define MIS_LN_BWD(LPR, NV)                                                                                     \
    hipLaunchKernelGGL((ln_bwd_dx_reg_kernel<LPR, NV>), dim3(slabs), dim3(256), 0, stream, x, ldx, dy, lddy, dx, \
                       lddx, gamma, mean, rstd, M, C, accumulate_dx, (long long)COL_SLAB_ROWS,                  \
                       affine ? part : (float2*)nullptr, rz)
    if (C <= 128) MIS_LN_BWD(32, 1);
    else if (C <= 256) MIS_LN_BWD(64, 1);
    else if (C <= 512) MIS_LN_BWD(64, 2);
    else if (C <= 768) MIS_LN_BWD(64, 3);
    else if (C <= 1024) MIS_LN_BWD(64, 4);
    else if (C <= 1536) MIS_LN_BWD(64, 6);
    else {
        if (rz.d2) return MIS_ERR_UNSUPPORTED;
        if (affine)
            hipLaunchKernelGGL(col_partial_kernel, dim3((C + 127) / 128, slabs), dim3(256), 0, stream, x, ldx, dy, lddy,
                               mean, rstd, M, C, (long long)COL_SLAB_ROWS, 0, part);
        hipLaunchKernelGGL(ln_bwd_dx_kernel, dim3((unsigned)mis_cdiv(M, 4)), dim3(256), 0, stream, x, ldx, dy, lddy,
                           dx, lddx, gamma, mean, rstd, M, C, accumulate_dx);
    }
#undef MIS_LN_BWD
    return mis_launch_status();
}

extern "C" int mis_layernorm_bwd_parts(const float* x, long long ldx, const float* dy, long long lddy, float* dx,
                                       long long lddx, const float* gamma, const float* mean, const float* rstd,
                                       long long M, int C, int accumulate_dx, void* workspace, long long workspace_bytes,
                                       hipStream_t stream) {
    return layernorm_bwd_parts(x, ldx, dy, lddy, dx, lddx, gamma, mean, rstd, M, C, accumulate_dx, true, workspace,
                               workspace_bytes, stream);
}

// mis_layernorm_bwd_parts + the backward of the residual add x = shortcut + s_b * branch that produced the LayerNorm's input
// (struct LnRes above): d_shortcut (+)= total, d_branch = rowscale[row / rows_per_scale] * total with total = gin + LayerNorm'(dy);
// gin (may be NULL) = the gradient the input's other readers left, rowscale NULL = 1 (no DropPath).  C <= 1536.
extern "C" int mis_layernorm_bwd_residual_parts(const float* x, long long ldx, const float* dy, long long lddy, const float* gin,
                                                long long ldgin, float* d_shortcut, long long ldds, int accumulate_shortcut,
                                                float* d_branch, long long lddb, const float* rowscale,
                                                long long rows_per_scale, const float* gamma, const float* mean,
                                                const float* rstd, long long M, int C, void* workspace,
                                                long long workspace_bytes, hipStream_t stream) {
    if (!d_branch || !d_shortcut || (rowscale && rows_per_scale <= 0)) return MIS_ERR_ARG;
    if (lddb % 4 || !a16(d_branch) || (gin && (ldgin % 4 || !a16(gin)))) return MIS_ERR_UNSUPPORTED;
    const LnRes rz{gin, ldgin, d_branch, lddb, rowscale, rowscale ? rows_per_scale : 1};
    return layernorm_bwd_parts(x, ldx, dy, lddy, d_shortcut, ldds, gamma, mean, rstd, M, C, accumulate_shortcut, true, workspace,
                               workspace_bytes, stream, rz);
}

extern "C" int mis_layernorm_bwd_final(const void* workspace, long long workspace_bytes, long long M, int C, float* dgamma,
                                       float* dbeta, int accumulate_affine, hipStream_t stream) {
    if (!workspace || M <= 0 || C <= 0 || (!dgamma && !dbeta)) return MIS_ERR_ARG;
    if (workspace_bytes < mis_colreduce_workspace_bytes(M, C)) return MIS_ERR_WORKSPACE;
    hipLaunchKernelGGL(col_final_kernel, dim3((C + 31) / 32), dim3(1024), 0, stream, reinterpret_cast<const float2*>(workspace),
                       (int)mis_cdiv(M, COL_SLAB_ROWS), C, dgamma, dbeta, accumulate_affine);
    return mis_launch_status();
}

extern "C" int mis_layernorm_bwd(const float* x, long long ldx, const float* dy, long long lddy, float* dx,
                                 long long lddx, const float* gamma, const float* mean, const float* rstd,
                                 float* dgamma, float* dbeta, long long M, int C, int accumulate_dx,
                                 int accumulate_affine, void* workspace, long long workspace_bytes,
                                 hipStream_t stream) {
    const bool affine = dgamma || dbeta;
    const int st = layernorm_bwd_parts(x, ldx, dy, lddy, dx, lddx, gamma, mean, rstd, M, C, accumulate_dx, affine, workspace,
                                       workspace_bytes, stream);
    if (st != MIS_OK || !affine) return st;
    return mis_layernorm_bwd_final(workspace, workspace_bytes, M, C, dgamma, dbeta, accumulate_affine, stream);
}

// ---- batched finishing sums (colsum_batch_kernel) ----
extern "C" long long mis_colsum_job_bytes(void) { return (long long)sizeof(ColsumJob); }

// number of partial rows mis_layernorm_bwd_parts / _residual_parts leave in their workspace (float2 [slabs][C])
extern "C" long long mis_colreduce_slabs(long long M) { return M > 0 ? mis_cdiv(M, COL_SLAB_ROWS) : MIS_ERR_ARG; }

// fills one job of a batch: out_a[c] (+)= sum_s part[s][c] (pairs: .x -> out_a, .y -> out_b; either may be NULL);
// returns the number of workgroups the job owns (the caller's running prefix sum is the next job's `first`)
extern "C" long long mis_colsum_job(void* job, const void* part, long long stride, long long slabs, int C, int pairs,
                                    float* out_a, float* out_b, int accumulate, long long first) {
    if (!job || !part || slabs <= 0 || C <= 0 || stride < C || (!out_a && !out_b) || (pairs != 1 && out_b) || pairs < 0 || pairs > 2)
        return MIS_ERR_ARG;
    // pairs == 2: the split-K partials of a GEMM, summed in gemm_reduce_kernel's order (the wide form: float4 columns)
    const bool wide = pairs == 2;
    if (wide && (C % 4 || stride % 4 || ((uintptr_t)part & 15) || ((uintptr_t)out_a & 15) || out_b)) return MIS_ERR_UNSUPPORTED;
    if (wide) pairs = 0;
    const long long blocks = wide ? mis_cdiv(C, COLSUM_WIDE_COLS) : mis_cdiv(C, 32);
    if (slabs > 0x7fffffffLL || first + blocks > 0x7fffffffLL) return MIS_ERR_UNSUPPORTED;
    ColsumJob j;
    memset(&j, 0, sizeof(j));
    j.part = part; j.out_a = out_a; j.out_b = out_b; j.stride = stride;
    j.slabs = (int)slabs; j.C = C; j.pairs = pairs ? 1 : 0; j.accumulate = accumulate ? 1 : 0;
    j.first = (int)first; j.blocks = (int)blocks; j.wide = wide ? 1 : 0;
    memcpy(job, &j, sizeof(j));
    return blocks;
}

// jobs: device array of n jobs ordered by `first`; blocks = the sum of their workgroup counts
extern "C" int mis_colsum_batch(const void* jobs, int n, long long blocks, hipStream_t stream) {
    if (!jobs || n <= 0 || blocks <= 0 || blocks > 0x7fffffffLL) return MIS_ERR_ARG;
    hipLaunchKernelGGL(colsum_batch_kernel, dim3((unsigned)blocks), dim3(1024), 0, stream,
                       reinterpret_cast<const ColsumJob*>(jobs), n);
    return mis_launch_status();
}

// out[c] (+)= sum_rows x[row][c]   (nn.Linear bias gradient)
extern "C" int mis_colsum(const float* x, long long ldx, long long M, int C, float* out, int accumulate,
                          void* workspace, long long workspace_bytes, hipStream_t stream) {
    if (!x || !out || !workspace || M <= 0 || C <= 0) return MIS_ERR_ARG;
    if (workspace_bytes < mis_colreduce_workspace_bytes(M, C)) return MIS_ERR_WORKSPACE;
    float2* part = reinterpret_cast<float2*>(workspace);
    const int slabs = (int)mis_cdiv(M, COL_SLAB_ROWS);
    hipLaunchKernelGGL(col_partial_kernel, dim3((C + 127) / 128, slabs), dim3(256), 0, stream, x, ldx, nullptr, 0,
                       nullptr, nullptr, M, C, (long long)COL_SLAB_ROWS, 1, part);
    hipLaunchKernelGGL(col_final_kernel, dim3((C + 31) / 32), dim3(1024), 0, stream, part, slabs, C, out, nullptr,
                       accumulate);
    return mis_launch_status();
}

// backward == 0: out = gelu(x); backward == 1: out = dy * gelu'(x).  n % 4 == 0, dense buffers.
extern "C" int mis_gelu(const float* x, const float* dy, float* out, long long n, int backward, hipStream_t stream) {
    if (!x || !out || n <= 0 || (backward && !dy)) return MIS_ERR_ARG;
    if (n % 4 || !a16(x) || !a16(out) || (dy && !a16(dy))) return MIS_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(gelu_kernel, dim3(sgrid(n >> 2)), dim3(256), 0, stream, x, dy, out, n >> 2, backward);
    return mis_launch_status();
}

// forward: out = a + s_b*y.  backward (y == NULL): out (may be NULL) = a, out2 = s_b * a, with a = d(out).
// s_b = DropPath scale of sample b = row / rows_per_sample: Philox(state, salt, b) or scale_override[b].
extern "C" int mis_residual_droppath(const float* a, long long lda, const float* y, long long ldy, float* out,
                                     long long ldo, float* out2, long long ldo2, long long M, int C,
                                     long long rows_per_sample, float drop_p, unsigned salt,
                                     const MisStepState* state, const float* scale_override, int backward,
                                     hipStream_t stream) {
    if (!a || M <= 0 || C <= 0 || rows_per_sample <= 0) return MIS_ERR_ARG;
    if (!backward && (!y || !out)) return MIS_ERR_ARG;
    if (backward && !out2) return MIS_ERR_ARG;
    if (drop_p < 0.f || drop_p >= 1.f || (drop_p > 0.f && !state && !scale_override)) return MIS_ERR_ARG;
    if (C % 4 || lda % 4 || ldy % 4 || ldo % 4 || ldo2 % 4) return MIS_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(residual_kernel, dim3(sgrid(M * (C >> 2))), dim3(256), 0, stream, a, lda, y, ldy, out, ldo, out2,
                       ldo2, M, C, rows_per_sample, drop_p, salt, state, scale_override, backward);
    return mis_launch_status();
}

// table[site*B + b] = DropPath scale (0 or 1/(1-p[site])) of sample b at site `site`, from Philox(state, salt[site], b):
// exactly what mis_residual_droppath applies.  p / salt: device arrays of nsites entries; p[site] == 0 gives 1.
extern "C" int mis_droppath_table(float* table, const float* p, const unsigned* salt, int nsites, int B,
                                  const MisStepState* state, hipStream_t stream) {
    if (!table || !p || !salt || nsites <= 0 || B <= 0 || !state) return MIS_ERR_ARG;
    hipLaunchKernelGGL(droppath_table_kernel, dim3((nsites * B + 255) / 256), dim3(256), 0, stream, table, p, salt, nsites,
                       B, state);
    return mis_launch_status();
}

// mode 0 PatchMerging gather (src: B x H x W tokens of C -> dst: B x H/2 x W/2 tokens of 4C);
// mode 1 PatchExpand shuffle (src: B x H x W tokens of P*P*C -> dst: B x HP x WP tokens of C);
// inverse = 1: the backward scatter (src/dst roles swapped, same H, W, C, P arguments).
extern "C" int mis_token_rearrange(const float* src, long long lds, float* dst, long long ldd, int B, int H, int W,
                                   int C, int P, int mode, int inverse, hipStream_t stream) {
    if (!src || !dst || B <= 0 || H <= 0 || W <= 0 || C <= 0) return MIS_ERR_ARG;
    if (C % 4 || lds % 4 || ldd % 4 || !a16(src) || !a16(dst)) return MIS_ERR_UNSUPPORTED;
    if (mode == 0 && ((H | W) & 1)) return MIS_ERR_ARG;
    if (mode == 1 && P <= 0) return MIS_ERR_ARG;
    RearrArgs a{src, lds, dst, ldd, B, H, W, C, P, mode, inverse};
    const int PP = mode == 0 ? 1 : P;
    hipLaunchKernelGGL(rearrange_kernel, dim3(sgrid((long long)B * H * PP * W * PP * (C >> 2))), dim3(256), 0, stream, a);
    return mis_launch_status();
}

extern "C" int mis_patch_im2col_c(const float* x, long long x_bs, float* out, int B, int H, int W, int in_chans,
                                  int src_chans, hipStream_t stream) {
    if (!x || !out || B <= 0 || H <= 0 || W <= 0 || in_chans <= 0) return MIS_ERR_ARG;
    if (src_chans != 1 && src_chans != in_chans) return MIS_ERR_ARG;
    if ((H | W) & 3 || x_bs % 4 || !a16(x) || !a16(out)) return MIS_ERR_UNSUPPORTED;
    if (x_bs < (long long)src_chans * H * W) return MIS_ERR_ARG;
    hipLaunchKernelGGL(patch_im2col_kernel, dim3(sgrid((long long)B * (H / 4) * (W / 4) * 4)), dim3(256), 0, stream, x,
                       x_bs, out, B, H, W, in_chans, src_chans);
    return mis_launch_status();
}

extern "C" int mis_patch_im2col(const float* x, long long x_bs, float* out, int B, int H, int W, int in_chans,
                                hipStream_t stream) {
    return mis_patch_im2col_c(x, x_bs, out, B, H, W, in_chans, 1, stream);
}

extern "C" int mis_patch3d_im2col(const float* x, long long x_bs, float* out, int B, int H, int W, int D, int P,
                                  hipStream_t stream) {
    if (!x || !out || B <= 0 || H <= 0 || W <= 0 || D <= 0 || P <= 0) return MIS_ERR_ARG;
    if (H % P || W % P || D % P || P % 4 || D % 4 || x_bs % 4 || !a16(x) || !a16(out)) return MIS_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(patch3d_im2col_kernel, dim3(sgrid((long long)B * H * W * D / 4)), dim3(256), 0, stream, x, x_bs,
                       out, B, H, W, D, P);
    return mis_launch_status();
}

extern "C" int mis_add_rowcycle(const float* x, long long ldx, const float* pos, float* out, long long ldo,
                                long long M, int C, int L, hipStream_t stream) {
    if (!x || !pos || !out || M <= 0 || C <= 0 || L <= 0) return MIS_ERR_ARG;
    if (C % 4 || ldx % 4 || ldo % 4 || !a16(x) || !a16(pos) || !a16(out)) return MIS_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(add_rowcycle_kernel, dim3(sgrid(M * (C >> 2))), dim3(256), 0, stream, x, ldx, pos, out, ldo, M,
                       C, L);
    return mis_launch_status();
}

extern "C" int mis_sum_rowcycle(const float* dy, long long ld, float* dpos, long long M, int C, int L,
                                hipStream_t stream) {
    if (!dy || !dpos || M <= 0 || C <= 0 || L <= 0 || M % L) return MIS_ERR_ARG;
    if (C % 4 || ld % 4 || !a16(dy) || !a16(dpos)) return MIS_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(sum_rowcycle_kernel, dim3((unsigned)mis_cdiv((long long)L * (C >> 2), 256)), dim3(256), 0, stream,
                       dy, ld, dpos, M, C, L);
    return mis_launch_status();
}

extern "C" int mis_head_fwd(const float* x, long long ldx, const float* w, float* logits, long long y_bs, int B,
                            long long S, int K, int NC, hipStream_t stream) {
    if (!x || !w || !logits || B <= 0 || S <= 0 || K <= 0) return MIS_ERR_ARG;
    if (K % 4 || ldx % 4 || !a16(x)) return MIS_ERR_UNSUPPORTED;
    const unsigned grid = sgrid((long long)B * S);
    const size_t sh = (size_t)NC * K * 4;
    switch (NC) {
        case 2: hipLaunchKernelGGL(head_fwd_kernel<2>, dim3(grid), dim3(256), sh, stream, x, ldx, w, logits, y_bs, B, S, K); break;
        case 3: hipLaunchKernelGGL(head_fwd_kernel<3>, dim3(grid), dim3(256), sh, stream, x, ldx, w, logits, y_bs, B, S, K); break;
        case 4: hipLaunchKernelGGL(head_fwd_kernel<4>, dim3(grid), dim3(256), sh, stream, x, ldx, w, logits, y_bs, B, S, K); break;
        default: return MIS_ERR_UNSUPPORTED;
    }
    return mis_launch_status();
}

extern "C" long long mis_head_workspace_bytes(int K, int NC) {
    if (K <= 0 || NC <= 0) return MIS_ERR_ARG;
    return (long long)HEAD_BLOCKS * NC * K * 4;
}

extern "C" int mis_head_bwd(const float* x, long long ldx, const float* w, const float* dlogits, long long dy_bs,
                            float* dx, long long lddx, float* dw, int accumulate_dw, int B, long long S, int K, int NC,
                            void* workspace, long long workspace_bytes, hipStream_t stream) {
    if (!x || !w || !dlogits || !dx || !dw || !workspace || B <= 0 || S <= 0 || K <= 0) return MIS_ERR_ARG;
    if (K % 4 || ldx % 4 || lddx % 4 || !a16(x) || !a16(dx)) return MIS_ERR_UNSUPPORTED;
    if (workspace_bytes < mis_head_workspace_bytes(K, NC)) return MIS_ERR_WORKSPACE;
    float* part = reinterpret_cast<float*>(workspace);
    if (K > 128) return MIS_ERR_UNSUPPORTED;   // 32 float4 column lanes
    const size_t sh = (size_t)8 * NC * K * 4;
    switch (NC) {
        case 2: hipLaunchKernelGGL(head_bwd_kernel<2>, dim3(HEAD_BLOCKS), dim3(256), sh, stream, x, ldx, w, dlogits, dy_bs, dx, lddx, part, B, S, K); break;
        case 3: hipLaunchKernelGGL(head_bwd_kernel<3>, dim3(HEAD_BLOCKS), dim3(256), sh, stream, x, ldx, w, dlogits, dy_bs, dx, lddx, part, B, S, K); break;
        case 4: hipLaunchKernelGGL(head_bwd_kernel<4>, dim3(HEAD_BLOCKS), dim3(256), sh, stream, x, ldx, w, dlogits, dy_bs, dx, lddx, part, B, S, K); break;
        default: return MIS_ERR_UNSUPPORTED;
    }
    hipLaunchKernelGGL(head_dw_final_kernel, dim3((NC * K + 31) / 32), dim3(1024), 0, stream, part, HEAD_BLOCKS, NC * K,
                       dw, accumulate_dw);
    return mis_launch_status();
}

// LayerNorm(C) + the bias-free 1x1 output convolution in one pass (the tail of SwinUnet: FinalPatchExpand_X4.norm + output,
// reference swin_transformer_unet_skip_expand_decoder_sys.py:390-409, :671, :749-752).  x [B*S][C] token-major (row stride
// ldx), w [NC][C], logits [B][NC][S] (batch stride y_bs).  C % 4 == 0, C <= 128, NC in 2..4.  mean / rstd [B*S] are kept
// for the backward.
extern "C" int mis_ln_head_fwd(const float* x, long long ldx, const float* gamma, const float* beta, const float* w,
                               float* mean, float* rstd, float* logits, long long y_bs, int B, long long S, int C, int NC,
                               float eps, hipStream_t stream) {
    if (!x || !gamma || !beta || !w || !mean || !rstd || !logits || B <= 0 || S <= 0 || C <= 0) return MIS_ERR_ARG;
    if (C % 4 || C > 128 || ldx % 4 || !a16(x) || !a16(gamma) || !a16(beta) || !a16(w)) return MIS_ERR_UNSUPPORTED;
    const long long M = (long long)B * S;
    static const bool l8 = !(getenv("MIS_LN96_LANES8") && getenv("MIS_LN96_LANES8")[0] == '0');
    if (C == 96 && l8 && NC >= 2 && NC <= 4) {
        long long blocks8 = mis_cdiv(M, 64);
        if (blocks8 > 8192) blocks8 = 8192;
#define MIS_LNH_F8(N_) hipLaunchKernelGGL(ln96_head_fwd_kernel<N_>, dim3((unsigned)blocks8), dim3(256), 0, stream, x, ldx, gamma, \
                                          beta, w, mean, rstd, logits, y_bs, M, S, eps)
        if (NC == 2) MIS_LNH_F8(2); else if (NC == 3) MIS_LNH_F8(3); else MIS_LNH_F8(4);
#undef MIS_LNH_F8
        return mis_launch_status();
    }
    long long blocks = mis_cdiv(M, 32);
    if (blocks > 8192) blocks = 8192;
#define MIS_LNH_F(N_) hipLaunchKernelGGL(ln_head_fwd_kernel<N_>, dim3((unsigned)blocks), dim3(256), 0, stream, x, ldx, gamma, \
                                         beta, w, mean, rstd, logits, y_bs, M, S, C, eps)
    switch (NC) {
        case 2: MIS_LNH_F(2); break;
        case 3: MIS_LNH_F(3); break;
        case 4: MIS_LNH_F(4); break;
        default: return MIS_ERR_UNSUPPORTED;
    }
#undef MIS_LNH_F
    return mis_launch_status();
}

extern "C" long long mis_ln_head_workspace_bytes(long long M, int C, int NC) {
    if (M <= 0 || C <= 0 || NC <= 0) return MIS_ERR_ARG;
    return mis_cdiv(M, COL_SLAB_ROWS) * C * (long long)(sizeof(float2) + NC * sizeof(float));
}

// Backward of mis_ln_head_fwd: dx [B*S][C] (+)= LayerNorm backward of dy = dlogits . w;  dgamma, dbeta [C] and dw [NC][C] (+)=
// their column sums (accumulate_params).  Deterministic (per-slab partials, fixed-order sums).
static int ln_head_bwd_impl(const float* x, long long ldx, const float* gamma, const float* beta, const float* w,
                            const float* mean, const float* rstd, const float* dlogits, long long dl_bs, float* dx,
                            long long lddx, int accumulate_dx, float* dgamma, float* dbeta, float* dw,
                            int accumulate_params, int B, long long S, int C, int NC, void* workspace,
                            long long workspace_bytes, int ex_P, int ex_H, int ex_W, hipStream_t stream);

extern "C" int mis_ln_head_bwd(const float* x, long long ldx, const float* gamma, const float* beta, const float* w,
                               const float* mean, const float* rstd, const float* dlogits, long long dl_bs, float* dx,
                               long long lddx, int accumulate_dx, float* dgamma, float* dbeta, float* dw,
                               int accumulate_params, int B, long long S, int C, int NC, void* workspace,
                               long long workspace_bytes, hipStream_t stream) {
    return ln_head_bwd_impl(x, ldx, gamma, beta, w, mean, rstd, dlogits, dl_bs, dx, lddx, accumulate_dx, dgamma, dbeta, dw,
                            accumulate_params, B, S, C, NC, workspace, workspace_bytes, 0, 0, 0, stream);
}

// ... with the gradient rows stored through the inverse pixel shuffle of FinalPatchExpand_X4: x rows are the tokens of the
// (H P) x (W P) grid (S = H P W P per image), dx is the expand Linear's output gradient [B H W][P P C] (lddx >= P P C)
extern "C" int mis_ln_head_bwd_unshuffle(const float* x, long long ldx, const float* gamma, const float* beta, const float* w,
                                         const float* mean, const float* rstd, const float* dlogits, long long dl_bs, float* dx,
                                         long long lddx, int accumulate_dx, float* dgamma, float* dbeta, float* dw,
                                         int accumulate_params, int B, int H, int W, int P, int C, int NC, void* workspace,
                                         long long workspace_bytes, hipStream_t stream) {
    if (H <= 0 || W <= 0 || P <= 0 || lddx < (long long)P * P * C || (long long)H * P * W * P > 0x7fffffffLL) return MIS_ERR_ARG;
    return ln_head_bwd_impl(x, ldx, gamma, beta, w, mean, rstd, dlogits, dl_bs, dx, lddx, accumulate_dx, dgamma, dbeta, dw,
                            accumulate_params, B, (long long)H * P * W * P, C, NC, workspace, workspace_bytes, P, H, W, stream);
}

static int ln_head_bwd_impl(const float* x, long long ldx, const float* gamma, const float* beta, const float* w,
                            const float* mean, const float* rstd, const float* dlogits, long long dl_bs, float* dx,
                            long long lddx, int accumulate_dx, float* dgamma, float* dbeta, float* dw,
                            int accumulate_params, int B, long long S, int C, int NC, void* workspace,
                            long long workspace_bytes, int ex_P, int ex_H, int ex_W, hipStream_t stream) {
    if (!x || !gamma || !beta || !w || !mean || !rstd || !dlogits || !dx || !dgamma || !dbeta || !dw || !workspace ||
        B <= 0 || S <= 0 || C <= 0)
        return MIS_ERR_ARG;
    if (C % 4 || C > 128 || ldx % 4 || lddx % 4 || !a16(x) || !a16(dx) || !a16(gamma) || !a16(beta) || !a16(w))
        return MIS_ERR_UNSUPPORTED;
    const long long M = (long long)B * S;
    if (workspace_bytes < mis_ln_head_workspace_bytes(M, C, NC)) return MIS_ERR_WORKSPACE;
    const int slabs = (int)mis_cdiv(M, COL_SLAB_ROWS);
    float2* pln = reinterpret_cast<float2*>(workspace);
    float* pw = reinterpret_cast<float*>(pln + (long long)slabs * C);
    static const bool l8 = !(getenv("MIS_LN96_LANES8") && getenv("MIS_LN96_LANES8")[0] == '0');
#define MIS_LNH_B8(N_) hipLaunchKernelGGL(ln96_head_bwd_kernel<N_>, dim3(slabs), dim3(256), 0, stream, x, ldx, gamma, beta, w, mean, \
                                          rstd, dlogits, dl_bs, dx, lddx, accumulate_dx, M, S, (long long)COL_SLAB_ROWS, pln, pw, ex_P, ex_H, ex_W)
#define MIS_LNH_B(N_) hipLaunchKernelGGL(ln_head_bwd_kernel<N_>, dim3(slabs), dim3(256), 0, stream, x, ldx, gamma, beta, w, mean, \
                                         rstd, dlogits, dl_bs, dx, lddx, accumulate_dx, M, S, C, (long long)COL_SLAB_ROWS, pln, pw, ex_P, ex_H, ex_W)
    if (C == 96 && l8) {
        switch (NC) {
            case 2: MIS_LNH_B8(2); break;
            case 3: MIS_LNH_B8(3); break;
            case 4: MIS_LNH_B8(4); break;
            default: return MIS_ERR_UNSUPPORTED;
        }
    } else
    switch (NC) {
        case 2: MIS_LNH_B(2); break;
        case 3: MIS_LNH_B(3); break;
        case 4: MIS_LNH_B(4); break;
        default: return MIS_ERR_UNSUPPORTED;
    }
#undef MIS_LNH_B
    hipLaunchKernelGGL(col_final_kernel, dim3((C + 31) / 32), dim3(1024), 0, stream, pln, slabs, C, dgamma, dbeta,
                       accumulate_params);
    hipLaunchKernelGGL(head_dw_final_kernel, dim3((NC * C + 31) / 32), dim3(1024), 0, stream, pw, slabs, NC * C, dw,
                       accumulate_params);
    return mis_launch_status();
}
