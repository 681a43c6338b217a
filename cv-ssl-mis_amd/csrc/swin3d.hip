// 3-D shifted-window attention and patch merging of SwinUNETR's encoder.
//
// Replaces, for the network net_factory_3d('swinunetr') builds (reference code/networks/net_factory_3d.py:7,37-38:
// monai.networks.nets.SwinUNETR(img_size=(64,64,64), in_channels, out_channels, feature_size=48)), the arithmetic of
// MONAI's SwinTransformerBlock.forward_part1 (pad -> torch.roll -> window_partition -> WindowAttention -> window_reverse
// -> roll back -> un-pad) and PatchMerging ("merging", the v0.9 slice order).  MONAI is an un-vendored dependency of the
// reference and absent from the build image: PARITY UNPINNED, the published algorithm is restated in oracle/swinunetr.py
// and these kernels are tested against it.
//
// Layouts: activations token-major [B][D][H][W][C] fp32 (C % 4 == 0); windows [B * nW][n][C] with the window order
// (b, wz, wy, wx) and the token order (tz, ty, tx) of window_partition; qkv [B * nW * n][3 * C] with columns ordered
// (q | k | v, head, 16 dims) as WindowAttention's reshape(b, n, 3, heads, c / heads).
//
// Attention runs on the vector pipe: one 384-thread workgroup per (window, head), one query (forward, dQ) or key (dK, dV)
// row per thread, the other operand of that head in LDS (n <= 343 rows of 16 floats, read as broadcasts), relative
// position bias of the head (2197 floats) and the window's shift-region ids in LDS.  head_dim is 16 (feature_size 48 with
// 3 / 6 / 12 / 24 heads): a 343 x 343 x 16 product per workgroup is 7.5 MFLOP -- the whole encoder's attention is two
// orders of magnitude below the convolutions of the decoder.  Deterministic: no atomics; the bias-table gradient is one
// partial table per (window, head) -- accumulated per wave in LDS inside the dQ kernel -- summed in a fixed order.
#include "common.h"

namespace {

constexpr int HD = 16;            // channels per head
constexpr int WSZ = 7;            // window of the relative-position index (also for clipped windows: index[:n, :n])
constexpr int NMAX = WSZ * WSZ * WSZ;
constexpr int TBL = (2 * WSZ - 1) * (2 * WSZ - 1) * (2 * WSZ - 1);
constexpr int NT = 384;

struct WinGeo {
    int B, D, H, W, C;            // token volume
    int wd, wh, ww;               // window
    int sd, sh, sw;               // cyclic shift (0: none)
    int Dp, Hp, Wp;               // padded to multiples of the window
    int nwd, nwh, nww;            // windows per axis
};

// windows[bw][t][:] = tokens[b][(wz * wd + tz + sd) mod Dp][...][:]  (0 where that position is padding);  inverse:
// tokens[b][z][y][x][:] = windows[...] of the slot that position maps to.  One thread = 4 channels of one slot / token.
__global__ __launch_bounds__(256) void win3d_gather_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                                           WinGeo g, int inverse) {
    const int c4 = g.C >> 2;
    const long long total = inverse ? (long long)g.B * g.D * g.H * g.W * c4
                                    : (long long)g.B * g.Dp * g.Hp * g.Wp * c4;
    const int n = g.wd * g.wh * g.ww, nW = g.nwd * g.nwh * g.nww;
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int c = (int)(i % c4);
        long long r = i / c4;
        if (!inverse) {
            const int t = (int)(r % n); r /= n;
            const int w = (int)(r % nW);
            const int b = (int)(r / nW);
            const int tx = t % g.ww, ty = (t / g.ww) % g.wh, tz = t / (g.ww * g.wh);
            const int wx = w % g.nww, wy = (w / g.nww) % g.nwh, wz = w / (g.nww * g.nwh);
            const int z = (wz * g.wd + tz + g.sd) % g.Dp, y = (wy * g.wh + ty + g.sh) % g.Hp,
                      x = (wx * g.ww + tx + g.sw) % g.Wp;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (z < g.D && y < g.H && x < g.W)
                v = reinterpret_cast<const float4*>(src)[((((long long)b * g.D + z) * g.H + y) * g.W + x) * c4 + c];
            reinterpret_cast<float4*>(dst)[i] = v;
        } else {
            const int x = (int)(r % g.W); r /= g.W;
            const int y = (int)(r % g.H); r /= g.H;
            const int z = (int)(r % g.D);
            const int b = (int)(r / g.D);
            const int iz = (z - g.sd + g.Dp) % g.Dp, iy = (y - g.sh + g.Hp) % g.Hp, ix = (x - g.sw + g.Wp) % g.Wp;
            const int w = ((iz / g.wd) * g.nwh + iy / g.wh) * g.nww + ix / g.ww;
            const int t = ((iz % g.wd) * g.wh + iy % g.wh) * g.ww + ix % g.ww;
            reinterpret_cast<float4*>(dst)[i] =
                reinterpret_cast<const float4*>(src)[(((long long)b * nW + w) * n + t) * c4 + c];
        }
    }
}

// PatchMerging ("merging"): slot s of the 8C-wide row of coarse voxel (z, y, x) = fine voxel (2z + oz_s, 2y + oy_s,
// 2x + ox_s); MONAI's v0.9 order reads (0,1,0) and (0,0,1) twice and never (1,1,0), (0,1,1).
__constant__ int MERGE_OFF[8][3] = {{0, 0, 0}, {1, 0, 0}, {0, 1, 0}, {0, 0, 1}, {1, 0, 1}, {0, 1, 0}, {0, 0, 1}, {1, 1, 1}};

__global__ __launch_bounds__(256) void merge3d_kernel(const float* __restrict__ src, float* __restrict__ dst, int B,
                                                      int D, int H, int W, int C, int inverse) {
    const int c4 = C >> 2, Do = D >> 1, Ho = H >> 1, Wo = W >> 1;
    if (!inverse) {       // dst [B][Do][Ho][Wo][8C]
        const long long total = (long long)B * Do * Ho * Wo * 8 * c4;
        for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
            const int c = (int)(i % c4);
            long long r = i / c4;
            const int s = (int)(r % 8); r /= 8;
            const int x = (int)(r % Wo); r /= Wo;
            const int y = (int)(r % Ho); r /= Ho;
            const int z = (int)(r % Do);
            const int b = (int)(r / Do);
            const long long f = (((long long)b * D + 2 * z + MERGE_OFF[s][0]) * H + 2 * y + MERGE_OFF[s][1]) * W + 2 * x +
                                MERGE_OFF[s][2];
            reinterpret_cast<float4*>(dst)[i] = reinterpret_cast<const float4*>(src)[f * c4 + c];
        }
    } else {              // src = gradient of the merged rows, dst = gradient of the fine tokens (every element written)
        const long long total = (long long)B * D * H * W * c4;
        for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
            const int c = (int)(i % c4);
            long long r = i / c4;
            const int x = (int)(r % W); r /= W;
            const int y = (int)(r % H); r /= H;
            const int z = (int)(r % D);
            const int b = (int)(r / D);
            const long long row = (((long long)b * Do + (z >> 1)) * Ho + (y >> 1)) * Wo + (x >> 1);
            float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                if (MERGE_OFF[s][0] == (z & 1) && MERGE_OFF[s][1] == (y & 1) && MERGE_OFF[s][2] == (x & 1)) {
                    const float4 v = reinterpret_cast<const float4*>(src)[(row * 8 + s) * c4 + c];
                    a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
                }
            }
            reinterpret_cast<float4*>(dst)[i] = a;
        }
    }
}

// ---------------------------------------------------------------------------------------------------- attention
struct AttnArgs {
    const float* qkv; long long ldq;      // [BW * n][3 * C]
    float* out; long long ldo;            // forward: [BW * n][C]
    float* stats;                         // [BW * nH][n][2]: row max, row sum of exp
    const float* table;                   // [TBL][nH]
    const int* region;                    // [nW][n] shift-region ids, or null (no mask)
    const float* dout; const float* o;    // backward: gradient and value of `out`
    float* dqkv; long long lddq;
    float* delta;                         // [BW * nH][n]
    float* tpart;                         // [BW * nH][TBL] partial bias-table gradients
    int BW, nW, n, nH;
    float scale;
};

__device__ __forceinline__ void load16(const float* __restrict__ p, float (&v)[HD]) {
#pragma unroll
    for (int i = 0; i < HD / 4; ++i) {
        const float4 q = reinterpret_cast<const float4*>(p)[i];
        v[4 * i] = q.x; v[4 * i + 1] = q.y; v[4 * i + 2] = q.z; v[4 * i + 3] = q.w;
    }
}
__device__ __forceinline__ float dot16(const float (&a)[HD], const float* __restrict__ b) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < HD / 4; ++i) {
        const float4 q = reinterpret_cast<const float4*>(b)[i];
        s = fmaf(a[4 * i], q.x, s); s = fmaf(a[4 * i + 1], q.y, s); s = fmaf(a[4 * i + 2], q.z, s); s = fmaf(a[4 * i + 3], q.w, s);
    }
    return s;
}
// packed fp32 (v_pk_fma_f32: two lanes of arithmetic per instruction -- these kernels are vector-ALU bound)
typedef float f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void load16p(const float* __restrict__ p, f2 (&v)[HD / 2]) {
#pragma unroll
    for (int i = 0; i < HD / 4; ++i) {
        const float4 q = reinterpret_cast<const float4*>(p)[i];
        v[2 * i] = f2{q.x, q.y}; v[2 * i + 1] = f2{q.z, q.w};
    }
}
__device__ __forceinline__ float dot16p(const f2 (&a)[HD / 2], const float* __restrict__ b) {
    f2 s = {0.f, 0.f};
#pragma unroll
    for (int i = 0; i < HD / 4; ++i) {
        const float4 q = reinterpret_cast<const float4*>(b)[i];
        s = a[2 * i] * f2{q.x, q.y} + s;
        s = a[2 * i + 1] * f2{q.z, q.w} + s;
    }
    return s.x + s.y;
}
// acc += e * b[0..15]
__device__ __forceinline__ void axpy16p(f2 (&acc)[HD / 2], float e, const float* __restrict__ b) {
    const f2 e2 = {e, e};
#pragma unroll
    for (int i = 0; i < HD / 4; ++i) {
        const float4 q = reinterpret_cast<const float4*>(b)[i];
        acc[2 * i] = e2 * f2{q.x, q.y} + acc[2 * i];
        acc[2 * i + 1] = e2 * f2{q.z, q.w} + acc[2 * i + 1];
    }
}
__device__ __forceinline__ void store16p(float* __restrict__ p, const f2 (&v)[HD / 2], float scale) {
#pragma unroll
    for (int i = 0; i < HD / 4; ++i)
        reinterpret_cast<float4*>(p)[i] = make_float4(v[2 * i].x * scale, v[2 * i].y * scale, v[2 * i + 1].x * scale, v[2 * i + 1].y * scale);
}
// relative-position index of (query t, key j): the [:n, :n] block of the full 7^3 window's index matrix
__device__ __forceinline__ int rel_index(int t, int j) {
    const int tz = t / 49, ty = (t / 7) % 7, tx = t % 7, jz = j / 49, jy = (j / 7) % 7, jx = j % 7;
    return (tz - jz + 6) * 169 + (ty - jy + 6) * 13 + (tx - jx + 6);
}

// stage `cols` (0 = q, 1 = k, 2 = v of head h; or a [rows][C] matrix with ld) of window bw into LDS rows of 16 floats
__device__ __forceinline__ void stage_rows(float* __restrict__ lds, const float* __restrict__ base, long long ld, int n) {
    for (int i = threadIdx.x; i < n * (HD / 4); i += NT) {
        const int r = i / (HD / 4), q = i % (HD / 4);
        reinterpret_cast<float4*>(lds)[i] = reinterpret_cast<const float4*>(base + (long long)r * ld)[q];
    }
}

extern __shared__ __attribute__((aligned(16))) float swin3d_lds[];

// forward: one query row per thread
__global__ __launch_bounds__(NT) void attn3d_fwd_kernel(const AttnArgs a) {
    float* const sk = swin3d_lds;                      // [n][16]
    float* const sv = sk + NMAX * HD;
    float* const sb = sv + NMAX * HD;                  // [TBL] bias of this head
    int* const sr = reinterpret_cast<int*>(sb + TBL);  // [n] region ids
    const int bw = blockIdx.x / a.nH, h = blockIdx.x % a.nH, n = a.n, C = a.nH * HD;
    const float* __restrict__ qb = a.qkv + (long long)bw * n * a.ldq + h * HD;
    stage_rows(sk, qb + C, a.ldq, n);
    stage_rows(sv, qb + 2 * C, a.ldq, n);
    for (int i = threadIdx.x; i < TBL; i += NT) sb[i] = a.table[(long long)i * a.nH + h];
    if (a.region) for (int i = threadIdx.x; i < n; i += NT) sr[i] = a.region[(long long)(bw % a.nW) * n + i];
    __syncthreads();
    const int t = threadIdx.x;
    if (t >= n) return;
    f2 q[HD / 2];
    load16p(qb + (long long)t * a.ldq, q);
#pragma unroll
    for (int i = 0; i < HD / 2; ++i) q[i] *= a.scale;
    const int rt = a.region ? sr[t] : 0;
    // one pass, running maximum: a new maximum rescales the sums (rare after the first keys; the branch is wave-divergent
    // only then) -- the scores are formed once instead of twice
    float m = -INFINITY, l = 0.f;
    f2 acc[HD / 2];
#pragma unroll
    for (int i = 0; i < HD / 2; ++i) acc[i] = f2{0.f, 0.f};
    for (int j = 0; j < n; ++j) {
        float s = dot16p(q, sk + j * HD) + sb[rel_index(t, j)];
        if (a.region && sr[j] != rt) s -= 100.f;
        if (s > m) {
            const float c = __expf(m - s);
            l *= c;
#pragma unroll
            for (int i = 0; i < HD / 2; ++i) acc[i] *= c;
            m = s;
        }
        const float e = __expf(s - m);
        l += e;
        axpy16p(acc, e, sv + j * HD);
    }
    store16p(a.out + ((long long)bw * n + t) * a.ldo + h * HD, acc, 1.f / l);
    reinterpret_cast<float2*>(a.stats)[((long long)blockIdx.x) * n + t] = make_float2(m, l);
}

// backward, queries: delta_t = dO_t . O_t, dQ_t = scale * sum_j dS_tj K_j,  dS = P (dP - delta); and the bias-table
// gradient of this (window, head): tpart[r] = sum of dS_tj over the pairs with relative index r.  At a given j the lanes
// of a wave (distinct t) hit distinct table entries, so every wave adds into its OWN LDS copy of the table with plain
// read-modify-writes in program order (deterministic); the six copies are summed in a fixed order at the end.
__global__ __launch_bounds__(NT) void attn3d_bwd_q_kernel(const AttnArgs a) {
    float* const sk = swin3d_lds;
    float* const sv = sk + NMAX * HD;
    float* const sb = sv + NMAX * HD;
    int* const sr = reinterpret_cast<int*>(sb + TBL);
    float* const stab = reinterpret_cast<float*>(sr + NMAX);       // [NT / 64][TBL]
    for (int i = threadIdx.x; i < (NT / 64) * TBL; i += NT) stab[i] = 0.f;
    const int bw = blockIdx.x / a.nH, h = blockIdx.x % a.nH, n = a.n, C = a.nH * HD;
    const float* __restrict__ qb = a.qkv + (long long)bw * n * a.ldq + h * HD;
    stage_rows(sk, qb + C, a.ldq, n);
    stage_rows(sv, qb + 2 * C, a.ldq, n);
    for (int i = threadIdx.x; i < TBL; i += NT) sb[i] = a.table[(long long)i * a.nH + h];
    if (a.region) for (int i = threadIdx.x; i < n; i += NT) sr[i] = a.region[(long long)(bw % a.nW) * n + i];
    __syncthreads();
    const int t = threadIdx.x;
    if (t < n) {
        f2 q[HD / 2], dO[HD / 2], o[HD / 2];
        load16p(qb + (long long)t * a.ldq, q);
        load16p(a.dout + ((long long)bw * n + t) * a.ldo + h * HD, dO);
        load16p(a.o + ((long long)bw * n + t) * a.ldo + h * HD, o);
        f2 d2 = {0.f, 0.f};
#pragma unroll
        for (int i = 0; i < HD / 2; ++i) { q[i] *= a.scale; d2 = dO[i] * o[i] + d2; }
        const float delta = d2.x + d2.y;
        const float2 st = reinterpret_cast<const float2*>(a.stats)[(long long)blockIdx.x * n + t];
        const float inv = 1.f / st.y;
        const int rt = a.region ? sr[t] : 0;
        float* const mytab = stab + (threadIdx.x >> 6) * TBL;
        f2 dq[HD / 2];
#pragma unroll
        for (int i = 0; i < HD / 2; ++i) dq[i] = f2{0.f, 0.f};
        for (int j = 0; j < n; ++j) {
            const int r = rel_index(t, j);
            float s = dot16p(q, sk + j * HD) + sb[r];
            if (a.region && sr[j] != rt) s -= 100.f;
            const float p = __expf(s - st.x) * inv;
            const float ds = p * (dot16p(dO, sv + j * HD) - delta);
            mytab[r] += ds;
            axpy16p(dq, ds, sk + j * HD);
        }
        store16p(a.dqkv + ((long long)bw * n + t) * a.lddq + h * HD, dq, a.scale);
        a.delta[(long long)blockIdx.x * n + t] = delta;
    }
    __syncthreads();
    for (int r = threadIdx.x; r < TBL; r += NT) {
        float acc = 0.f;
#pragma unroll
        for (int w = 0; w < NT / 64; ++w) acc += stab[w * TBL + r];
        a.tpart[(long long)blockIdx.x * TBL + r] = acc;
    }
}

// backward, keys: dK_j = scale * sum_t dS_tj Q_t,  dV_j = sum_t P_tj dO_t    (one key row per thread; Q, dO in LDS)
__global__ __launch_bounds__(NT) void attn3d_bwd_kv_kernel(const AttnArgs a) {
    float* const sq = swin3d_lds;                         // [n][16] q * scale
    float* const sdo = sq + NMAX * HD;
    float* const sb = sdo + NMAX * HD;
    int* const sr = reinterpret_cast<int*>(sb + TBL);
    float* const sm = reinterpret_cast<float*>(sr + NMAX); // [n] row max
    float* const sl = sm + NMAX;                          // [n] 1 / row sum
    float* const sdl = sl + NMAX;                         // [n] delta
    const int bw = blockIdx.x / a.nH, h = blockIdx.x % a.nH, n = a.n, C = a.nH * HD;
    const float* __restrict__ qb = a.qkv + (long long)bw * n * a.ldq + h * HD;
    stage_rows(sq, qb, a.ldq, n);
    stage_rows(sdo, a.dout + (long long)bw * n * a.ldo + h * HD, a.ldo, n);
    for (int i = threadIdx.x; i < TBL; i += NT) sb[i] = a.table[(long long)i * a.nH + h];
    for (int i = threadIdx.x; i < n; i += NT) {
        if (a.region) sr[i] = a.region[(long long)(bw % a.nW) * n + i];
        const float2 st = reinterpret_cast<const float2*>(a.stats)[(long long)blockIdx.x * n + i];
        sm[i] = st.x; sl[i] = 1.f / st.y;
        sdl[i] = a.delta[(long long)blockIdx.x * n + i];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < n * HD; i += NT) sq[i] *= a.scale;
    __syncthreads();
    const int j = threadIdx.x;
    if (j >= n) return;
    f2 k[HD / 2], v[HD / 2], dk[HD / 2], dv[HD / 2];
    load16p(qb + C + (long long)j * a.ldq, k);
    load16p(qb + 2 * C + (long long)j * a.ldq, v);
#pragma unroll
    for (int i = 0; i < HD / 2; ++i) { dk[i] = f2{0.f, 0.f}; dv[i] = f2{0.f, 0.f}; }
    const int rj = a.region ? sr[j] : 0;
    for (int t = 0; t < n; ++t) {
        float s = dot16p(k, sq + t * HD) + sb[rel_index(t, j)];
        if (a.region && sr[t] != rj) s -= 100.f;
        const float p = __expf(s - sm[t]) * sl[t];
        const float ds = p * (dot16p(v, sdo + t * HD) - sdl[t]);
        axpy16p(dk, ds, sq + t * HD);
        axpy16p(dv, p, sdo + t * HD);
    }
    // sq holds q * scale: dK = sum dS * (scale q) already carries the scale
    float* __restrict__ db = a.dqkv + ((long long)bw * n + j) * a.lddq + h * HD;
    store16p(db + C, dk, 1.f);
    store16p(db + 2 * C, dv, 1.f);
}

// dtable[r][h] (+)= sum over the windows of tpart[(bw, h)][r], fixed order, in double
__global__ __launch_bounds__(256) void attn3d_table_reduce_kernel(const float* __restrict__ tpart, float* __restrict__ dtable,
                                                                  int BW, int nH, int accumulate) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= TBL * nH) return;
    const int r = i / nH, h = i % nH;
    double s = 0.0;
    for (int bw = 0; bw < BW; ++bw) s += tpart[((long long)bw * nH + h) * TBL + r];
    dtable[i] = accumulate ? dtable[i] + (float)s : (float)s;
}

int fill_geo(WinGeo& g, int B, int D, int H, int W, int C, int wd, int wh, int ww, int sd, int sh, int sw) {
    if (B <= 0 || D <= 0 || H <= 0 || W <= 0 || C <= 0 || C % 4 || wd <= 0 || wh <= 0 || ww <= 0) return MIS_ERR_ARG;
    if (sd < 0 || sh < 0 || sw < 0 || sd >= wd || sh >= wh || sw >= ww) return MIS_ERR_ARG;
    g.B = B; g.D = D; g.H = H; g.W = W; g.C = C; g.wd = wd; g.wh = wh; g.ww = ww; g.sd = sd; g.sh = sh; g.sw = sw;
    g.nwd = (D + wd - 1) / wd; g.nwh = (H + wh - 1) / wh; g.nww = (W + ww - 1) / ww;
    g.Dp = g.nwd * wd; g.Hp = g.nwh * wh; g.Wp = g.nww * ww;
    return MIS_OK;
}

unsigned grid_for(long long total) {
    long long b = mis_cdiv(total, 256);
    return (unsigned)(b > 8192 ? 8192 : (b < 1 ? 1 : b));
}

constexpr int FWD_LDS = (2 * NMAX * HD + TBL + NMAX) * 4;
constexpr int KV_LDS = (2 * NMAX * HD + TBL + 4 * NMAX) * 4;
constexpr int BQ_LDS = FWD_LDS + (NT / 64) * TBL * 4;

int check_attn(const float* qkv, long long ldq, int BW, int nW, int n, int nH) {
    if (!qkv || BW <= 0 || nW <= 0 || n <= 0 || nH <= 0 || BW % nW) return MIS_ERR_ARG;
    if (n > NMAX || ldq < 3LL * nH * HD || ldq % 4 || ((uintptr_t)qkv & 15)) return MIS_ERR_UNSUPPORTED;
    return MIS_OK;
}

}  // namespace

// windows [B * nW][n][C] <- tokens [B][D][H][W][C] (inverse != 0: tokens <- windows), zero padding to multiples of the
// window, cyclic shift (sd, sh, sw) as torch.roll(x, (-sd, -sh, -sw)).  The inverse is also the gradient of the forward
// and vice versa (padding slots receive zero).
extern "C" int mis_win3d_gather(const float* src, float* dst, int B, int D, int H, int W, int C, int wd, int wh, int ww,
                                int sd, int sh, int sw, int inverse, hipStream_t stream) {
    WinGeo g;
    int st = fill_geo(g, B, D, H, W, C, wd, wh, ww, sd, sh, sw);
    if (st) return st;
    if (!src || !dst || ((uintptr_t)src & 15) || ((uintptr_t)dst & 15)) return MIS_ERR_ARG;
    const long long total = (inverse ? (long long)B * D * H * W : (long long)B * g.Dp * g.Hp * g.Wp) * (C / 4);
    hipLaunchKernelGGL(win3d_gather_kernel, dim3(grid_for(total)), dim3(256), 0, stream, src, dst, g, inverse);
    return mis_launch_status();
}

// windows of the padded volume: B * ceil(D / wd) * ceil(H / wh) * ceil(W / ww)
extern "C" long long mis_win3d_windows(int B, int D, int H, int W, int wd, int wh, int ww) {
    if (B <= 0 || D <= 0 || H <= 0 || W <= 0 || wd <= 0 || wh <= 0 || ww <= 0) return MIS_ERR_ARG;
    return (long long)B * ((D + wd - 1) / wd) * ((H + wh - 1) / wh) * ((W + ww - 1) / ww);
}

// merged [B][D/2][H/2][W/2][8C] <- tokens [B][D][H][W][C] in MONAI's v0.9 "merging" slot order; inverse != 0: the
// gradient of the tokens (every element written) from the gradient of the merged rows.  Even D, H, W.
extern "C" int mis_merge3d(const float* src, float* dst, int B, int D, int H, int W, int C, int inverse,
                           hipStream_t stream) {
    if (!src || !dst || B <= 0 || D <= 0 || H <= 0 || W <= 0 || C <= 0) return MIS_ERR_ARG;
    if (C % 4 || D % 2 || H % 2 || W % 2 || ((uintptr_t)src & 15) || ((uintptr_t)dst & 15)) return MIS_ERR_UNSUPPORTED;
    const long long total = inverse ? (long long)B * D * H * W * (C / 4) : (long long)B * (D / 2) * (H / 2) * (W / 2) * 2 * C;
    hipLaunchKernelGGL(merge3d_kernel, dim3(grid_for(total)), dim3(256), 0, stream, src, dst, B, D, H, W, C, inverse);
    return mis_launch_status();
}

// out[BW * n][nH * 16] = softmax(q k^T / 4 + bias + mask) v per (window, head); stats [BW * nH][n][2] kept for the
// backward.  table [2197][nH]; region [nW][n] int32 or NULL (un-shifted blocks: no mask; padded tokens are ordinary keys)
extern "C" int mis_win3d_attn_fwd(const float* qkv, long long ldq, float* out, long long ldo, float* stats,
                                  const float* table, const int* region, int BW, int nW, int n, int nH,
                                  hipStream_t stream) {
    int st = check_attn(qkv, ldq, BW, nW, n, nH);
    if (st) return st;
    if (!out || !stats || !table || ldo < (long long)nH * HD || ldo % 4 || ((uintptr_t)out & 15) || ((uintptr_t)stats & 7))
        return MIS_ERR_ARG;
    AttnArgs a{};
    a.qkv = qkv; a.ldq = ldq; a.out = out; a.ldo = ldo; a.stats = stats; a.table = table; a.region = region;
    a.BW = BW; a.nW = nW; a.n = n; a.nH = nH; a.scale = 0.25f;      // 16^-0.5
    static std::atomic<unsigned long long> done{0};
    if (mis_set_lds_attr(reinterpret_cast<const void*>(&attn3d_fwd_kernel), FWD_LDS, done) != MIS_OK) return MIS_ERR_LAUNCH;
    hipLaunchKernelGGL(attn3d_fwd_kernel, dim3(BW * nH), dim3(NT), FWD_LDS, stream, a);
    return mis_launch_status();
}

extern "C" long long mis_win3d_attn_workspace_bytes(int BW, int n, int nH) {
    if (BW <= 0 || n <= 0 || nH <= 0) return MIS_ERR_ARG;
    return ((long long)BW * nH * n + (long long)BW * nH * TBL) * 4;
}

// dqkv [BW * n][3 * nH * 16] (every element written) and dtable [2197][nH] (+)= from dout; `out`, `stats` from the forward
extern "C" int mis_win3d_attn_bwd(const float* qkv, long long ldq, const float* out, const float* dout, long long ldo,
                                  float* dqkv, long long lddq, const float* stats, const float* table, const int* region,
                                  float* dtable, int accumulate_table, int BW, int nW, int n, int nH, void* workspace,
                                  long long workspace_bytes, hipStream_t stream) {
    int st = check_attn(qkv, ldq, BW, nW, n, nH);
    if (st) return st;
    if (!out || !dout || !dqkv || !stats || !table || !dtable || !workspace) return MIS_ERR_ARG;
    if (ldo < (long long)nH * HD || ldo % 4 || lddq < 3LL * nH * HD || lddq % 4 || ((uintptr_t)dout & 15) ||
        ((uintptr_t)out & 15) || ((uintptr_t)dqkv & 15))
        return MIS_ERR_UNSUPPORTED;
    if (workspace_bytes < mis_win3d_attn_workspace_bytes(BW, n, nH)) return MIS_ERR_WORKSPACE;
    AttnArgs a{};
    a.qkv = qkv; a.ldq = ldq; a.o = out; a.dout = dout; a.ldo = ldo; a.dqkv = dqkv; a.lddq = lddq;
    a.stats = const_cast<float*>(stats); a.table = table; a.region = region;
    a.delta = reinterpret_cast<float*>(workspace); a.tpart = a.delta + (long long)BW * nH * n;
    a.BW = BW; a.nW = nW; a.n = n; a.nH = nH; a.scale = 0.25f;
    static std::atomic<unsigned long long> d0{0}, d1{0};
    if (mis_set_lds_attr(reinterpret_cast<const void*>(&attn3d_bwd_q_kernel), BQ_LDS, d0) != MIS_OK ||
        mis_set_lds_attr(reinterpret_cast<const void*>(&attn3d_bwd_kv_kernel), KV_LDS, d1) != MIS_OK)
        return MIS_ERR_LAUNCH;
    hipLaunchKernelGGL(attn3d_bwd_q_kernel, dim3(BW * nH), dim3(NT), BQ_LDS, stream, a);       // dQ, delta, table partials
    hipLaunchKernelGGL(attn3d_bwd_kv_kernel, dim3(BW * nH), dim3(NT), KV_LDS, stream, a);
    hipLaunchKernelGGL(attn3d_table_reduce_kernel, dim3((TBL * nH + 255) / 256), dim3(256), 0, stream, a.tpart, dtable, BW,
                       nH, accumulate_table);
    return mis_launch_status();
}
