// Full (un-windowed) multi-head self-attention core of the UNETR encoder.
//
// Replaces the attention arithmetic of MONAI's SABlock as the reference's UNETR builds it
// (reference code/networks/unetr.py:88-99: ViT(hidden_size=768, num_heads=12, ...) over the 6 x 6 x 6 = 216 patch
// tokens of a 96^3 volume; code/networks/net_factory_3d.py:23-36):
//     q, k, v = split(qkv)           qkv [B*N][3*nH*64], columns ordered (q|k|v, head, dim)
//     out = softmax(q k^T * scale) v
// MONAI is not part of the reference repository (un-vendored dependency): parity of this block is pinned to a torch
// restatement of the published algorithm only (oracle/unetr.py, "parity unpinned").
//
// N <= 256 tokens, head_dim 64, fp32 on the vector pipe (the whole encoder's attention is 14 GFLOP per step
// against 3.4 TFLOP of convolutions: HBM / latency matter here, not MFMA).  One 512-thread workgroup per
// (sample, head, chunk of query rows) -- B * nH alone is 48 .. 96 workgroups for 256 CUs, so the rows are split until
// the launch holds >= 512 of them; K and V of that head live in LDS (rows padded to 68 floats: 16-byte aligned, conflict-free
// ds_read_b128 across consecutive rows).  A wave owns query rows i = wave, wave + 8, ... of its chunk:
//   scores   lane j (4 passes of 64 keys): s_j = sum_d q_i[d] K[j][d]   q_i[d] broadcast by v_readlane (no LDS)
//   softmax  wave reductions; row statistics (max, sum) are kept for the backward
//   output   lane d: o[d] = sum_j p_j V[j][d]                           p_j broadcast by v_readlane
// Backward = the same row walk twice (flash-attention split, deterministic, no atomics):
//   pass Q (rows = queries, K/V in LDS):  P from the kept statistics, dP = dO V^T, delta_i = sum_j P dP,
//                                         dS = P (dP - delta), dQ_i = scale * sum_j dS_ij K_j
//   pass K (rows = keys, scaled Q / dO in LDS): the same P and dS by columns,
//                                         dK_j = sum_i dS_ij (scale q_i),  dV_j = sum_i P_ij dO_i
#include "common.h"

namespace {

constexpr int HD = 64, LDR = 68, MAXN = 256, PASSES = MAXN / 64, NT = 512, NW = NT / 64;

struct FullAttnArgs {
    const float* qkv; long long ldq;     // [B*N][3*nH*64]
    float* out; long long ldo;           // [B*N][nH*64]
    float* stats;                        // [B*nH][N][2] (row max, row sum of exp)
    int B, N, nH, rows;                  // rows: query rows per workgroup (blockIdx.y = chunk)
    float scale;
};

struct FullAttnBwdArgs {
    const float* qkv; long long ldq;
    const float* dout; long long ldo;
    float* dqkv; long long lddq;
    const float* stats;                  // from the forward
    float* delta;                        // [B*nH][N] workspace (written by pass Q, read by pass K)
    int B, N, nH, rows;
    float scale;
};

// dot of this lane's LDS row (64 floats at `row`) with a vector held one element per lane (broadcast by readlane)
__device__ __forceinline__ float dot_bcast(const float* __restrict__ row, float vec_lane) {
    float acc = 0.f;
#pragma unroll
    for (int d4 = 0; d4 < HD / 4; ++d4) {
        const float4 k = *reinterpret_cast<const float4*>(row + d4 * 4);
        acc = fmaf(__builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, vec_lane), d4 * 4 + 0)), k.x, acc);
        acc = fmaf(__builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, vec_lane), d4 * 4 + 1)), k.y, acc);
        acc = fmaf(__builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, vec_lane), d4 * 4 + 2)), k.z, acc);
        acc = fmaf(__builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, vec_lane), d4 * 4 + 3)), k.w, acc);
    }
    return acc;
}

// lane d: sum over the rows j < N of w_j * M[j][d], the weights w held one per lane in PASSES registers.
// Groups of 8 rows: the 8 ds_reads are issued together (a rolled loop pays one LDS round trip per row).
__device__ __forceinline__ float wsum_rows(const float* __restrict__ M, const float (&w)[PASSES], int N, int lane) {
    float acc0 = 0.f, acc1 = 0.f;
#pragma unroll
    for (int p = 0; p < PASSES; ++p) {
        if (p * 64 >= N) break;                          // uniform
        const int cnt = N - p * 64 < 64 ? N - p * 64 : 64;
        const float* __restrict__ Mp = M + p * 64 * LDR + lane;
        const int wbits = __builtin_bit_cast(int, w[p]);
        int jj = 0;
        for (; jj + 8 <= cnt; jj += 8) {
            float m[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) m[u] = Mp[(jj + u) * LDR];
#pragma unroll
            for (int u = 0; u < 8; u += 2) {
                acc0 = fmaf(__builtin_bit_cast(float, __builtin_amdgcn_readlane(wbits, jj + u)), m[u], acc0);
                acc1 = fmaf(__builtin_bit_cast(float, __builtin_amdgcn_readlane(wbits, jj + u + 1)), m[u + 1], acc1);
            }
        }
        for (; jj < cnt; ++jj)
            acc0 = fmaf(__builtin_bit_cast(float, __builtin_amdgcn_readlane(wbits, jj)), Mp[jj * LDR], acc0);
    }
    return acc0 + acc1;
}

// stage rows [N][64] of one head (column offset `col`) into LDS [N][LDR], optionally scaled
__device__ __forceinline__ void stage_rows(const float* __restrict__ base, long long ld, int col, int N, float scale,
                                           float* __restrict__ dst) {
    for (int e = threadIdx.x; e < N * (HD / 4); e += NT) {
        const int r = e / (HD / 4), q = e - r * (HD / 4);
        float4 v = *reinterpret_cast<const float4*>(base + (long long)r * ld + col + q * 4);
        v.x *= scale; v.y *= scale; v.z *= scale; v.w *= scale;
        *reinterpret_cast<float4*>(dst + r * LDR + q * 4) = v;
    }
}

extern __shared__ __attribute__((aligned(16))) float mis_fattn_lds[];

__global__ __launch_bounds__(NT) void full_attn_fwd_kernel(const FullAttnArgs a) {
    float* const sk = mis_fattn_lds;
    float* const sv = sk + a.N * LDR;
    const int bh = blockIdx.x, b = bh / a.nH, h = bh - b * a.nH;
    const int C = a.nH * HD, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float* __restrict__ base = a.qkv + (long long)b * a.N * a.ldq;
    stage_rows(base, a.ldq, C + h * HD, a.N, 1.f, sk);
    stage_rows(base, a.ldq, 2 * C + h * HD, a.N, 1.f, sv);
    __syncthreads();
    const int i0 = blockIdx.y * a.rows, i1 = i0 + a.rows < a.N ? i0 + a.rows : a.N;
    for (int i = i0 + wave; i < i1; i += NW) {
        const float q = base[(long long)i * a.ldq + h * HD + lane] * a.scale;
        float s[PASSES], mx = -INFINITY;
#pragma unroll
        for (int p = 0; p < PASSES; ++p) {
            const int j = p * 64 + lane;
            s[p] = -INFINITY;
            if (p * 64 < a.N) {                          // uniform
                const float v = dot_bcast(sk + (j < a.N ? j : 0) * LDR, q);
                if (j < a.N) s[p] = v;
            }
            mx = fmaxf(mx, s[p]);
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
        float sum = 0.f;
#pragma unroll
        for (int p = 0; p < PASSES; ++p) { s[p] = expf(s[p] - mx); sum += s[p]; }     // exp(-inf) = 0 for j >= N
        sum = mis_wave_sum(sum);
        const float inv = 1.f / sum;
#pragma unroll
        for (int p = 0; p < PASSES; ++p) s[p] *= inv;
        a.out[((long long)b * a.N + i) * a.ldo + h * HD + lane] = wsum_rows(sv, s, a.N, lane);
        if (lane == 0) {
            float2* st = reinterpret_cast<float2*>(a.stats) + (long long)bh * a.N + i;
            *st = make_float2(mx, sum);
        }
    }
}

// pass Q (KEYS == false): rows = queries, LDS = K, V; writes dQ and delta.
// pass K (KEYS == true):  rows = keys,    LDS = scale*Q, dO; writes dK and dV.
template <bool KEYS>
__global__ __launch_bounds__(NT) void full_attn_bwd_kernel(const FullAttnBwdArgs a) {
    float* const s0 = mis_fattn_lds;               // K      | scale * Q
    float* const s1 = s0 + a.N * LDR;              // V      | dO
    const int bh = blockIdx.x, b = bh / a.nH, h = bh - b * a.nH;
    const int C = a.nH * HD, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float* __restrict__ base = a.qkv + (long long)b * a.N * a.ldq;
    const float* __restrict__ dob = a.dout + (long long)b * a.N * a.ldo;
    const float2* __restrict__ st = reinterpret_cast<const float2*>(a.stats) + (long long)bh * a.N;
    float* __restrict__ delta = a.delta + (long long)bh * a.N;
    if (!KEYS) {
        stage_rows(base, a.ldq, C + h * HD, a.N, 1.f, s0);
        stage_rows(base, a.ldq, 2 * C + h * HD, a.N, 1.f, s1);
    } else {
        stage_rows(base, a.ldq, h * HD, a.N, a.scale, s0);
        stage_rows(dob, a.ldo, h * HD, a.N, 1.f, s1);
    }
    __syncthreads();
    const int i0 = blockIdx.y * a.rows, i1 = i0 + a.rows < a.N ? i0 + a.rows : a.N;
    for (int i = i0 + wave; i < i1; i += NW) {
        // this row's two vectors, one element per lane
        float u, w;
        if (!KEYS) {
            u = base[(long long)i * a.ldq + h * HD + lane] * a.scale;         // scaled q_i
            w = dob[(long long)i * a.ldo + h * HD + lane];                    // dO_i
        } else {
            u = base[(long long)i * a.ldq + C + h * HD + lane];               // k_j
            w = base[(long long)i * a.ldq + 2 * C + h * HD + lane];           // v_j
        }
        float2 rst = make_float2(0.f, 1.f);
        float rdelta = 0.f;
        if (!KEYS) rst = st[i];
        float p[PASSES], ds[PASSES];
        float dsum = 0.f;
#pragma unroll
        for (int ps = 0; ps < PASSES; ++ps) {
            const int j = ps * 64 + lane;
            p[ps] = ds[ps] = 0.f;
            if (ps * 64 < a.N) {                         // uniform
                const int jr = j < a.N ? j : 0;
                const float sc = dot_bcast(s0 + jr * LDR, u);                 // q_i . k_j * scale (either orientation)
                const float dp = dot_bcast(s1 + jr * LDR, w);                 // dO_i . v_j
                if (KEYS) rst = st[jr];                                       // statistics of query row j
                if (j < a.N) {
                    p[ps] = expf(sc - rst.x) / rst.y;
                    ds[ps] = dp;                                              // dP for now
                    dsum += p[ps] * dp;
                }
            }
        }
        if (!KEYS) {
            dsum = mis_wave_sum(dsum);                                        // delta_i
            if (lane == 0) delta[i] = dsum;
            rdelta = dsum;
        }
#pragma unroll
        for (int ps = 0; ps < PASSES; ++ps) {
            if (KEYS) {                                                       // per-lane query row: its own delta
                const int j = ps * 64 + lane;
                const float dl = (ps * 64 < a.N && j < a.N) ? delta[j] : 0.f;
                ds[ps] = p[ps] * (ds[ps] - dl);
            } else {
                ds[ps] = p[ps] * (ds[ps] - rdelta);
            }
        }
        float* __restrict__ dq = a.dqkv + ((long long)b * a.N + i) * a.lddq + h * HD + lane;
        if (!KEYS) {
            dq[0] = a.scale * wsum_rows(s0, ds, a.N, lane);                   // dQ_i = scale * sum_j dS_ij K_j
        } else {
            dq[C] = wsum_rows(s0, ds, a.N, lane);                             // dK_j = sum_i dS_ij (scale q_i)
            dq[2 * C] = wsum_rows(s1, p, a.N, lane);                          // dV_j = sum_i P_ij dO_i
        }
    }
}

int check(const void* qkv, long long ldq, int B, int N, int nH) {
    if (!qkv || B <= 0 || N <= 0 || nH <= 0) return MIS_ERR_ARG;
    if (N > MAXN || ldq % 4 || ((uintptr_t)qkv & 15)) return MIS_ERR_UNSUPPORTED;
    if (ldq < 3LL * nH * HD) return MIS_ERR_ARG;
    return MIS_OK;
}

// rows per workgroup: enough chunks that the launch is >= 2 workgroups per CU (B * nH alone is 48 .. 96 here)
int rows_per_wg(int B, int N, int nH) {
    int chunks = (512 + B * nH - 1) / (B * nH);
    if (chunks < 1) chunks = 1;
    int rows = (N + chunks - 1) / chunks;
    rows = (rows + NW - 1) / NW * NW;
    return rows;
}

template <int ID, class K>       // ID: one attribute record per kernel (the two backward kernels share a type)
int set_lds(K kernel, int bytes) {
    static std::atomic<unsigned long long> done{0};
    return mis_set_lds_attr(reinterpret_cast<const void*>(kernel), bytes, done);
}

}  // namespace

// stats: B*nH*N*2 floats (kept for the backward)
extern "C" int mis_full_attention_fwd(const float* qkv, long long ldq, float* out, long long ldo, float* stats, int B,
                                      int N, int nH, float scale, hipStream_t stream) {
    int st = check(qkv, ldq, B, N, nH);
    if (st) return st;
    if (!out || !stats || ldo < (long long)nH * HD || ((uintptr_t)stats & 7)) return MIS_ERR_ARG;
    FullAttnArgs a{qkv, ldq, out, ldo, stats, B, N, nH, rows_per_wg(B, N, nH), scale};
    const int lds = 2 * N * LDR * 4;
    if (set_lds<0>(full_attn_fwd_kernel, lds) != MIS_OK) return MIS_ERR_LAUNCH;
    hipLaunchKernelGGL(full_attn_fwd_kernel, dim3(B * nH, (N + a.rows - 1) / a.rows), dim3(NT), lds, stream, a);
    return mis_launch_status();
}

extern "C" long long mis_full_attention_workspace_bytes(int B, int N, int nH) {
    if (B <= 0 || N <= 0 || nH <= 0) return MIS_ERR_ARG;
    return (long long)B * nH * N * 4;
}

// dqkv [B*N][3*nH*64] (every element written); workspace: mis_full_attention_workspace_bytes
extern "C" int mis_full_attention_bwd(const float* qkv, long long ldq, const float* dout, long long ldo, float* dqkv,
                                      long long lddq, const float* stats, int B, int N, int nH, float scale,
                                      void* workspace, long long workspace_bytes, hipStream_t stream) {
    int st = check(qkv, ldq, B, N, nH);
    if (st) return st;
    if (!dout || !dqkv || !stats || !workspace || ldo < (long long)nH * HD || lddq < 3LL * nH * HD) return MIS_ERR_ARG;
    if (ldo % 4 || ((uintptr_t)dout & 15)) return MIS_ERR_UNSUPPORTED;
    if (workspace_bytes < mis_full_attention_workspace_bytes(B, N, nH)) return MIS_ERR_WORKSPACE;
    FullAttnBwdArgs a{qkv, ldq, dout, ldo, dqkv, lddq, stats, reinterpret_cast<float*>(workspace), B, N, nH,
                      rows_per_wg(B, N, nH), scale};
    const int lds = 2 * N * LDR * 4;
    if (set_lds<1>(full_attn_bwd_kernel<false>, lds) != MIS_OK || set_lds<2>(full_attn_bwd_kernel<true>, lds) != MIS_OK)
        return MIS_ERR_LAUNCH;
    const dim3 grid(B * nH, (N + a.rows - 1) / a.rows);
    hipLaunchKernelGGL(full_attn_bwd_kernel<false>, grid, dim3(NT), lds, stream, a);
    hipLaunchKernelGGL(full_attn_bwd_kernel<true>, grid, dim3(NT), lds, stream, a);
    return mis_launch_status();
}
