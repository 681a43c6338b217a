// fp32 GEMM on the gfx950 matrix pipe for the token-major (B, L, C) layers of SwinUnet.
//
// Replaces nn.Linear forward / backward (reference code/networks/
// swin_transformer_unet_skip_expand_decoder_sys.py: qkv/proj :107,109; Mlp fc1/fc2 :14,16;
// PatchMerging.reduction :320; PatchExpand.expand :361-362; FinalPatchExpand_X4.expand :390;
// concat_back_dim :690-691) and the im2col'ed PatchEmbed conv (:573-574):
//
//   trans = 0 ("NT"):  C[M,N] (+)= A[M,K] . B[N,K]^T (+ bias[N])      forward  (B = weight)
//                                                                     dX       (B = weight^T, packed once per step)
//   trans = 1 ("TN"):  C[M,N] (+)= A[K,M]^T . B[K,N]                  dW = dY^T . X   (contraction over tokens)
//
// v_mfma_f32_16x16x4_f32 (exact fp32 fmaf chain; there is no TF32 on gfx950), 256-thread workgroups of 2 x 2 waves,
// BK = 32.  Tiles: NT 128 x {128, 96}, TN {128, 96}^2 -- every channel width of Swin-T is a multiple of 96
// (96 ... 1536), and with 128-wide tiles the N = 96 / 192 / 288 GEMMs (half of the FLOPs) would spend 25-44 % of
// their MFMAs on padding.
//
// NT: operand tiles arrive by LDS-DMA (buffer_load_dwordx4 ... lds, common.h) into a double-buffered stage: the
// copy of k-step s+1 overlaps the MFMAs of k-step s, one barrier per k-step.  The DMA writes LDS lane-linearly, so
// padding a row is impossible; instead the tile is stored [row][32] with the k index XOR-swizzled by
// 4*((row>>1)&7) -- applied on the SOURCE address of each lane's 16 bytes and again on the operand reads -- which
// makes the 8-byte operand reads (the 4 k-lane groups take k = 2g, 2g+1 of an 8-wide slab: one LDS read feeds two
// MFMAs) hit 64 distinct banks per 32-lane group.  Out-of-range rows / k are the descriptor's range check (zeros).
// The epilogue is buffer stores (range check drops rows beyond M).
//
// TN (dW, contraction over ~10^5 tokens, M x N only a few tiles): operands are contraction-major rows, row stride
// = tile + 16 (= 16 mod 32), staged through registers into a single buffer (these launches are split-K with short
// per-workgroup loops and profit more from 4 resident workgroups per CU than from a double-buffered DMA stage:
// 9.5 ms/step vs 13.1 ms/step on config 4), partial tiles in a caller workspace and a fixed-order reduction
// (deterministic).  Workgroups are ordered XCD-aware over one linear (k-slice, tile) index.
#include <atomic>
#include <cstring>
#include "common.h"

namespace {

using namespace mis_dma;

constexpr int BM = 128, BK = 32;

struct GemmArgs {
    const float* A; long long lda;
    const float* B; long long ldb;
    float* C; long long ldc;
    const float* bias;
    float* ws;            // split-K partials [KS][M][N] (row stride N)
    int M, N, K, KS, kchunk, accumulate;
    int tiles_n, tiles_m;
    unsigned n_blocks, n_blocks_padded;   // over (k-slice, tile)
    // NT only, optional: store C through the PatchExpand pixel shuffle 'b h w (p1 p2 c) -> b (h p1) (w p2) c'
    // (row m = (b, h, w) of an ex_H x ex_W token grid, column n = (p1, p2, c)); ex_P == 0: plain row-major C
    int ex_P, ex_H, ex_W, ex_c;
    int vec4;   // NT: C / bias (/ E1 / C2) are 16-byte aligned with row strides and N multiples of 4: LDS-staged float4 epilogue
    // NT only, optional fused epilogue (mis_gemm_ex): EP_GELU_FWD  C = v, C2 = gelu(v)          (v = acc + bias)
    //                                                 EP_GELU_BWD  C = v * gelu'(E1)             (E1 = pre-activation)
    //                                                 EP_RESIDUAL  C = E1 + rowscale[m / rps] * v (E1 = shortcut)
    int ep;
    const float* E1; long long lde1;
    float* C2; long long ldc2;
    const float* rowscale; long long rps;
    // TN only, optional (mis_gemm_dw): dbias[m] (+)= sum_k A[k][m] -- the bias gradient of nn.Linear rides on the weight
    // gradient's read of dy.  Split-K: per-slice sums in wsb [KS][M] (behind the M x N partials), summed by the reduction
    float* dbias; float* wsb; int dbias_acc;
    // NT only, optional (mis_gemm_nt_split): B as the three bf16 piece planes mis_gemm_split_b wrote (PREC = 2 kernels; the
    // fp32 B is not read).  b3_plane = bytes of one plane (N x K3 x 2), K3 = K rounded up to the k-step
    const void* B3; unsigned b3_plane; int K3;
    int bm_force;          // host only: rows per tile chosen by the caller (0: nt_tile_m)
    // EP_LNHEAD (mis_gemm_expand_ln_head): LayerNorm over the c = 96 columns of a tile row (= one token of the pixel-shuffled
    // tensor) + the bias-free output head, in the epilogue; C may be NULL (the expanded tokens are not kept)
    const float* ln_g; const float* ln_b; const float* head_w;
    float* ln_mean; float* ln_rstd; float* logits; long long logits_bs;
    int head_nc; float ln_eps;
};

enum { EP_NONE = 0, EP_GELU_FWD = 1, EP_GELU_BWD = 2, EP_RESIDUAL = 3, EP_LNHEAD = 4, EP_RESIDUAL_LN = 5 };

// same functions as token_ops.hip::gelu_kernel (common.h)
__device__ __forceinline__ float gelu_f(float x) { return mis_gelu(x); }
__device__ __forceinline__ float gelu_grad_f(float x) { return mis_gelu_grad(x); }

extern __shared__ __attribute__((aligned(16))) float mis_gemm_lds[];

#ifndef MIS_GEMM_DBG_CT
#define MIS_GEMM_DBG_CT 0
#endif
// ablation builds of the NT kernel (timing only, results wrong): 1 no global stores in the float4 epilogue, 2 no DMA in
// the k-loop, 4 no MFMAs (fp32 form), 8 no epilogue at all, 16 / 32 no B / A split, 64 no MFMAs in the pre-split bf16x3 form
constexpr int GDBG = MIS_GEMM_DBG_CT;

// ------------------------------------------------------------------------------------------------ split-precision products
// PREC = 1 ("bf16x3", MIS_GEMM_BF3=1): an fp32 value is cut EXACTLY into three bf16 pieces by truncation, x = h + m + l (8 + 8 + 8
// significant bits; the two subtractions are exact), and a product of two such values is the sum of the six piece products
// hh + hm + mh + hl + lh + mm on v_mfma_f32_16x16x32_bf16 with fp32 accumulation -- what is dropped (ml + lm + ll) is below
// 2^-24 of |a||b|, i.e. below the rounding of the fp32 fmaf chain it replaces (tests: error against float64 no larger than the
// fp32 kernel's).  One K = 32 block costs 6 instructions of 18 cycles instead of 8 of 32 (scripts/ubench/pipe_share.hip:
// 7.5 ns against 13.8 ns per instruction), and the 32-bit integer / fp32 ops of the split overlap a co-resident wave's bf16
// MFMAs (v_add beside mfma_bf16: 2175 us together against 2967 serial), which packed fp32 ops and the fp32 MFMA do not.
typedef mis_u32x4 u32x4;
__device__ __forceinline__ void bf3_split_pair(float x0, float x1, unsigned& h, unsigned& m, unsigned& l) {
    mis_bf3_split_pair(x0, x1, h, m, l);
}
__device__ __forceinline__ f32x4 bf3_mfma(const u32x4& a, const u32x4& b, f32x4 c) { return mis_bf3_mfma1(a, b, c); }

// B operand of the NT form, split ahead of the GEMM (weights: once per step instead of once per tile and k-step).  Three planes
// (h, m, l) of bf16 [N][K3], K3 = K rounded up to 32 (zeros beyond K); inside each block of 32 contraction elements the order is
// the one the kernels' lanes consume: position lk * 8 + 2 s + e holds element 8 s + 2 lk + e (lane group lk = lane >> 4 takes the
// float2 at 8 s + 2 lk of the fp32 A rows, s = 0 .. 3), so a lane's 8 elements of a plane are one 16-byte group.  The pieces are
// mis_bf3_from8's: the PREC = 2 kernels give the same bits as PREC = 1.
struct SplitJob {
    const float* B; long long ldb; void* B3;
    int N, K, K3, first;       // first: prefix sum of the jobs' units (one unit = 256 (row, block, lane group) triples)
    int natural, pad;          // natural: lane group lk's 16-byte group holds elements 8 lk .. 8 lk + 7 (gemm_nt_rega_kernel)
};

__device__ __forceinline__ void split_unit(const SplitJob& j, int unit) {
    const int kb_n = j.K3 / 32;
    const long long t = (long long)unit * 256 + threadIdx.x;          // (n, kb, lk)
    const long long total = (long long)j.N * kb_n * 4;
    if (t >= total) return;
    const int lk = (int)(t & 3);
    const long long r = t >> 2;
    const int kb = (int)(r % kb_n);
    const long long n = r / kb_n;
    float e[8];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const int k = j.natural ? kb * 32 + 8 * lk + 2 * s : kb * 32 + 8 * s + 2 * lk;
        float2 v = make_float2(0.f, 0.f);
        if (k < j.K) v = *reinterpret_cast<const float2*>(j.B + n * j.ldb + k);      // K % 4 == 0, ldb % 4 == 0: pairs are whole
        e[2 * s] = v.x; e[2 * s + 1] = v.y;
    }
    const MisBf3 o = mis_bf3_from8(e[0], e[1], e[2], e[3], e[4], e[5], e[6], e[7]);
    const long long plane = (long long)j.N * j.K3 * 2;
    char* const base = reinterpret_cast<char*>(j.B3) + (n * j.K3 + kb * 32 + lk * 8) * 2;
    *reinterpret_cast<u32x4*>(base) = o.h;
    *reinterpret_cast<u32x4*>(base + plane) = o.m;
    *reinterpret_cast<u32x4*>(base + 2 * plane) = o.l;
}

__global__ __launch_bounds__(256) void gemm_split_kernel(const SplitJob j) { split_unit(j, (int)blockIdx.x); }

// every Linear weight of a network in one launch: a device table of jobs ordered by `first`, binary search per workgroup
__global__ __launch_bounds__(256) void gemm_split_batch_kernel(const SplitJob* __restrict__ jobs, int n) {
    __shared__ SplitJob job;
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
    split_unit(job, (int)blockIdx.x - job.first);
}

// Split-precision mask: bit 0 the NT GEMMs (forward, dX), bit 1 the register-only TN GEMM (dW), bit 2 the window-attention
// products (attention_impl.inc) as bf16x3 products; 0 = fp32 MFMA everywhere.  Default 7 (MIS_GEMM_BF3 overrides at load;
// mis_gemm_set_split_precision at run time: bench.py times both).
std::atomic<int>& gemm_bf3_state() {
    static std::atomic<int> m{[] { const char* e = getenv("MIS_GEMM_BF3"); return e ? atoi(e) & 7 : 7; }()};
    return m;
}
bool gemm_bf3() { return gemm_bf3_state().load(std::memory_order_relaxed) & 1; }
bool gemm_bf3_tn() { return gemm_bf3_state().load(std::memory_order_relaxed) & 2; }

// ------------------------------------------------------------------------------------------------ NT
// NW waves per workgroup: 4 (2 x 2 wave tiles of BMT/2 x BN/2) or 8 (4 x 2 wave tiles of BMT/4 x BN/2: the same tile, stages and
// LDS, twice the waves per SIMD -- the operand stage is what limits residency (52 KB: three workgroups per CU), the kernel needs
// ~75 registers, and what the k-loop lacks is waves to hide its per-k-step DMA wait and barrier behind: profiles/r06_gemm_nt_pmc.txt)
template <int BMT, int BN, int PREC = 0, int NW = 4>
struct NtCfg {
    static constexpr int WROWS = BMT / (NW / 2);          // rows of a wave tile
    static constexpr int MI = WROWS / 16;                 // 16-row MFMA tiles per wave
    static constexpr int NJ = BN / 32;                    // 16-column MFMA tiles per wave
    // PREC = 2: the B stage holds three bf16 planes [BN][32] (64-byte rows) instead of fp32 [BN][32]
    static constexpr int A_FLOATS = BMT * BK, B_FLOATS = PREC == 2 ? BN * BK * 3 / 2 : BN * BK;
    static constexpr int RB3 = BN / 16, PB3 = 3 * RB3;    // PREC = 2: 16-row x 64-byte DMA pieces per plane / per stage
    static constexpr int STAGE = A_FLOATS + B_FLOATS;
    // epilogue: the accumulator tile goes through LDS (row stride BN + 4: the four 4-row lane groups land 16 banks
    // apart) so that C -- and the operands of the fused epilogues -- move as float4 rows instead of 64-byte dword pieces
    static constexpr int LDC_T = BN + 4;
    static constexpr int CT_FLOATS = BMT * LDC_T;
    // (a ring of THREE stages -- two in flight per workgroup, two workgroups per CU -- was measured in round 6 and is slower: 1335
    // -> 1514 us over the 20 shapes of a step, profiles/r06_gemm_nt_stages.txt; residency hides more than prefetch depth)
    static constexpr int LDS_FLOATS = 2 * STAGE > CT_FLOATS ? 2 * STAGE : CT_FLOATS;
    static constexpr int LDS_BYTES = LDS_FLOATS * 4;      // double-buffered operand stage, re-used by the epilogue
    static constexpr int PA = BMT / 8, PB = BN / 8;       // 8-row DMA pieces of the A / B tile
    static_assert(NW == 4 || NW == 8, "2 x 2 or 4 x 2 waves");
    static_assert(BMT % 32 == 0 && BN % 32 == 0 && PA % NW == 0 && (PREC == 2 || PB % NW == 0), "pieces split evenly over the waves");
    static_assert(WROWS % 16 == 0 && (BMT * (BN / 4)) % (NW * 64) == 0, "wave tiles / epilogue rows");
};

// EP: compile-time epilogue (EP_NONE / EP_GELU_FWD / EP_GELU_BWD / EP_RESIDUAL).  The fused epilogues are separate
// instantiations: compiled into the plain kernel they cost it 41 registers and one resident workgroup per CU
// (measured: every Linear of the step slowed down, 38.5 -> 41.7 ms).
template <int BMT, int BN, int EP, int PREC = 0, int NW = 4>
__global__ __launch_bounds__(NW * 64) void gemm_nt_kernel(const GemmArgs a) {
    using G = NtCfg<BMT, BN, PREC, NW>;
    static_assert(NW == 4 || EP != EP_LNHEAD, "the fused tail's row passes assume 256 threads");
    float* const lds = mis_gemm_lds;

    const unsigned L = mis_xcd_remap(blockIdx.x, a.n_blocks_padded);
    if (L >= a.n_blocks) return;
    // (k-slice, tile): split-K only for the few-tile shapes of the deep stages (M = 1176..4704 tokens, K up to 3072)
    const unsigned tiles = (unsigned)a.tiles_n * (unsigned)a.tiles_m;
    const int kz = L / tiles;
    const unsigned T = L - kz * tiles;
    const int tn = T % a.tiles_n, tm = T / a.tiles_n;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lk = lane >> 4, lj = lane & 15;
    const int wm = (wave >> 1) * G::WROWS, wn = (wave & 1) * (BN / 2);
    const int m0 = tm * BMT, n0 = tn * BN;
    const int kbeg = kz * a.kchunk;
    const int kend = kbeg + a.kchunk < a.K ? kbeg + a.kchunk : a.K;
    const unsigned lds0 = lds_addr(lds);
    const i32x4 rA = make_rsrc(a.A, (unsigned)((long long)(a.M - 1) * a.lda + a.K) * 4u);
    const i32x4 rB = make_rsrc(a.B, (unsigned)((long long)(a.N - 1) * a.ldb + a.K) * 4u);

    // per-lane DMA source offsets (bytes, without the k-step term): piece p = 8 rows x 32 k (64 lanes x 16 B);
    // lane -> row p*8 + (lane>>3), LDS k-slot (lane&7)*4 holds source k = slot ^ swz(row).  (row>>1)&7 only depends
    // on lane>>4 and the piece parity, and a wave's pieces w, w+4, ... share their parity: one ksrc per lane.
    unsigned voA[G::PA / NW], voB[(G::PB + NW - 1) / NW];
    const int ksrc = ((lane & 7) * 4) ^ ((((wave * 8 + (lane >> 3)) >> 1) & 7) * 4);
#pragma unroll
    for (int i = 0; i < G::PA / NW; ++i) {
        const int row = (wave + NW * i) * 8 + (lane >> 3);
        voA[i] = m0 + row < a.M ? (unsigned)((long long)(m0 + row) * a.lda + ksrc) * 4u : OOB;
    }
    if constexpr (PREC != 2) {
#pragma unroll
        for (int i = 0; i < G::PB / NW; ++i) {
            const int row = (wave + NW * i) * 8 + (lane >> 3);
            voB[i] = n0 + row < a.N ? (unsigned)((long long)(n0 + row) * a.ldb + ksrc) * 4u : OOB;
        }
    }
    // PREC = 2: piece q = (plane q / RB3, rows (q % RB3) * 16 ..+16) x 64 bytes of the k-step: lane -> row lane >> 2, 16-byte
    // group lane & 3 (= the 8 contraction elements of lane group lk, in the order mis_gemm_split_b stored them); the 1 KiB a
    // DMA instruction writes is 16 whole rows of the plane, and a wave's operand read is one contiguous KiB again: no swizzle
    constexpr int NB3 = (G::PB3 + NW - 1) / NW;
    unsigned voB3[NB3];
    const i32x4 rB3 = PREC == 2 ? make_rsrc(a.B3, 3u * a.b3_plane) : rB;
    if constexpr (PREC == 2) {
#pragma unroll
        for (int i = 0; i < NB3; ++i) {
            const int q = wave + NW * i, plane = q / G::RB3, row = (q % G::RB3) * 16 + (lane >> 2);
            voB3[i] = (q < G::PB3 && n0 + row < a.N)
                          ? (unsigned)plane * a.b3_plane + (unsigned)(((long long)(n0 + row) * a.K3 + (lane & 3) * 8) * 2)
                          : OOB;
        }
    }

    auto stage = [&](int buf, int k0) {
        const unsigned st = lds0 + (unsigned)buf * (G::STAGE * 4);
        const bool kout = k0 + BK > kend && k0 + ksrc >= kend;   // only the last k-step can be partial
        const unsigned kb = (unsigned)k0 * 4u;
#pragma unroll
        for (int i = 0; i < G::PA / NW; ++i) dma_dwordx4(st + (unsigned)((wave + NW * i) * 256) * 4u, kout ? OOB : voA[i] + kb, rA);
        if constexpr (PREC == 2) {
#pragma unroll
            for (int i = 0; i < NB3; ++i)
                if (wave + NW * i < G::PB3)      // uniform; the planes are zero beyond K: no k test
                    dma_dwordx4(st + (unsigned)(G::A_FLOATS * 4 + (wave + NW * i) * 1024), voB3[i] + (unsigned)k0 * 2u, rB3);
        } else {
#pragma unroll
            for (int i = 0; i < G::PB / NW; ++i)
                dma_dwordx4(st + (unsigned)(G::A_FLOATS + (wave + NW * i) * 256) * 4u, kout ? OOB : voB[i] + kb, rB);
        }
    };

    f32x4 acc[G::MI][G::NJ];
#pragma unroll
    for (int i = 0; i < G::MI; ++i)
#pragma unroll
        for (int j = 0; j < G::NJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int swz = ((lj >> 1) & 7) * 4;   // read-side swizzle (rows wm + i*16 + lj, wn + j*16 + lj: only lj matters)

    auto compute = [&](const float* st) {
        const float2* __restrict__ sA2 = reinterpret_cast<const float2*>(st);
        const float2* __restrict__ sB2 = reinterpret_cast<const float2*>(st + G::A_FLOATS);
#pragma unroll
        for (int s = 0; s < BK / 8; ++s) {
            float2 af[G::MI], bf[G::NJ];
            const int kk = (s * 8 + 2 * lk) ^ swz;
#pragma unroll
            for (int i = 0; i < G::MI; ++i) af[i] = sA2[((wm + i * 16 + lj) * BK + kk) >> 1];
#pragma unroll
            for (int j = 0; j < G::NJ; ++j) bf[j] = sB2[((wn + j * 16 + lj) * BK + kk) >> 1];
#pragma unroll
            for (int i = 0; i < G::MI; ++i)
#pragma unroll
                for (int j = 0; j < G::NJ; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i].x, bf[j].x, acc[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < G::MI; ++i)
#pragma unroll
                for (int j = 0; j < G::NJ; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i].y, bf[j].y, acc[i][j], 0, 0, 0);
        }
    };

    // bf16x3 form of the same k-step (PREC = 1): the lane's 8 operand floats of a row -- the four float2 reads above, k = 8 s +
    // 2 lk + {0, 1} -- ARE the 8 contraction elements of a v_mfma_f32_16x16x32_bf16 operand (any assignment of the 32 k to
    // (lane group, slot) works as long as A and B share it): same LDS reads, 6 instructions per K = 32 block instead of 8
    auto compute_bf3 = [&](const float* st) {
        const float2* __restrict__ sA2 = reinterpret_cast<const float2*>(st);
        const float2* __restrict__ sB2 = reinterpret_cast<const float2*>(st + G::A_FLOATS);
        u32x4 ah[G::MI], am[G::MI], al[G::MI];
#pragma unroll
        for (int i = 0; i < G::MI; ++i)
#pragma unroll
            for (int s = 0; s < BK / 8; ++s) {
                const float2 v = sA2[((wm + i * 16 + lj) * BK + ((s * 8 + 2 * lk) ^ swz)) >> 1];
                unsigned h, m, l;
                if constexpr (GDBG & 32) { h = __float_as_uint(v.x); m = __float_as_uint(v.y); l = h; }   // timing only: no A split
                else bf3_split_pair(v.x, v.y, h, m, l);
                ah[i][s] = h; am[i][s] = m; al[i][s] = l;
            }
#pragma unroll
        for (int j = 0; j < G::NJ; ++j) {
            u32x4 bh, bm, bl;
#pragma unroll
            for (int s = 0; s < BK / 8; ++s) {
                const float2 v = sB2[((wn + j * 16 + lj) * BK + ((s * 8 + 2 * lk) ^ swz)) >> 1];
                unsigned h, m, l;
                if constexpr (GDBG & 16) { h = __float_as_uint(v.x); m = __float_as_uint(v.y); l = h; }   // timing only: no B split
                else bf3_split_pair(v.x, v.y, h, m, l);
                bh[s] = h; bm[s] = m; bl[s] = l;
            }
#pragma unroll
            for (int i = 0; i < G::MI; ++i) {
                f32x4 c = acc[i][j];
                c = bf3_mfma(al[i], bh, c);
                c = bf3_mfma(ah[i], bl, c);
                c = bf3_mfma(am[i], bm, c);
                c = bf3_mfma(am[i], bh, c);
                c = bf3_mfma(ah[i], bm, c);
                c = bf3_mfma(ah[i], bh, c);
                acc[i][j] = c;
            }
        }
    };

    // PREC = 2: B arrives split (mis_gemm_split_b): three 16-byte LDS reads per 16-column tile and no VALU work for B -- the
    // split of the B fragments was two thirds of the k-loop's vector instructions (264 per k-step of a 64 x 128 tile against
    // 48 MFMAs of 16 cycles; ablation without it: the 20 SwinUnet shapes 1461 -> 1231 us)
    auto compute_b3 = [&](const float* st) {
        const float2* __restrict__ sA2 = reinterpret_cast<const float2*>(st);
        const u32x4* __restrict__ sB3 = reinterpret_cast<const u32x4*>(st + G::A_FLOATS);
        u32x4 ah[G::MI], am[G::MI], al[G::MI];
#pragma unroll
        for (int i = 0; i < G::MI; ++i)
#pragma unroll
            for (int s = 0; s < BK / 8; ++s) {
                const float2 v = sA2[((wm + i * 16 + lj) * BK + ((s * 8 + 2 * lk) ^ swz)) >> 1];
                unsigned h, m, l;
                if constexpr (GDBG & 32) { h = __float_as_uint(v.x); m = __float_as_uint(v.y); l = h; }   // timing only: no A split
                else bf3_split_pair(v.x, v.y, h, m, l);
                ah[i][s] = h; am[i][s] = m; al[i][s] = l;
            }
#pragma unroll
        for (int j = 0; j < G::NJ; ++j) {
            const int r = (wn + j * 16 + lj) * 4 + lk;
            const u32x4 bh = sB3[r], bm = sB3[BN * 4 + r], bl = sB3[2 * BN * 4 + r];
#pragma unroll
            for (int i = 0; i < G::MI; ++i) {
                f32x4 c = acc[i][j];
                if constexpr (GDBG & 64) {      // timing only: no MFMA (one integer op keeps every operand read alive)
                    c[0] += __uint_as_float((al[i][0] ^ am[i][1] ^ ah[i][2] ^ bh[0] ^ bm[1] ^ bl[2]) & 0x007fffffu);
                    acc[i][j] = c;
                    continue;
                }
                c = bf3_mfma(al[i], bh, c);
                c = bf3_mfma(ah[i], bl, c);
                c = bf3_mfma(am[i], bm, c);
                c = bf3_mfma(am[i], bh, c);
                c = bf3_mfma(ah[i], bm, c);
                c = bf3_mfma(ah[i], bh, c);
                acc[i][j] = c;
            }
        }
    };

    // ---- software pipeline over k-steps: DMA(s+1) || MFMA(s) ----
    stage(0, kbeg);
    dma_wait();
    __syncthreads();
    int s = 0;
    for (int k0 = kbeg; k0 < kend; k0 += BK, ++s) {
        if (k0 + BK < kend && !(GDBG & 2)) stage((s + 1) & 1, k0 + BK);
        if constexpr (PREC == 2) compute_b3(lds + (s & 1) * G::STAGE);
        else if constexpr (PREC == 1) compute_bf3(lds + (s & 1) * G::STAGE);
        else if constexpr (!(GDBG & 4)) compute(lds + (s & 1) * G::STAGE);
        dma_wait();
        __syncthreads();   // k-step s+1 landed in the other buffer; everyone is done reading this one
    }

    // ---- epilogue: D row = lk*4 + r -> m, col = lj -> n ----
    // Fast path (N a multiple of 16, C addressable with 32 bits): buffer stores whose range check drops the rows
    // beyond M, the 16-column groups beyond N skipped by a uniform branch -- one v_add + one store per value instead
    // of 64-bit address arithmetic and two predicates (K is only 96..384 for most of these GEMMs, so the epilogue is a
    // large share of a tile).
    if constexpr (GDBG & 8) {
        if (acc[0][0][0] == 1.2345f) a.C[tid] = acc[0][0][1];
        return;
    }
    if (a.KS > 1) {   // split-K slice: raw partial into the workspace, bias / accumulate happen in gemm_reduce_kernel
        float* __restrict__ out = a.ws + (long long)kz * a.M * a.N;
#pragma unroll
        for (int i = 0; i < G::MI; ++i)
#pragma unroll
            for (int j = 0; j < G::NJ; ++j) {
                const int n = n0 + wn + j * 16 + lj;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int m = m0 + wm + i * 16 + lk * 4 + r;
                    if (m < a.M && n < a.N) out[(long long)m * a.N + n] = acc[i][j][r];
                }
            }
        return;
    }
    if constexpr (EP == EP_LNHEAD) {
        // FinalPatchExpand_X4 + its LayerNorm + the output head (reference ...sys.py:401-409, :671, :749-752) on the tile: BN = c =
        // 96, so the tile's columns are one (p1, p2) and a tile row is one token of the pixel-shuffled tensor.  Rows through LDS;
        // the row arithmetic is ln96_head_fwd_kernel's (token_ops.hip): same bits.
        static_assert(BN == 96, "a tile row is one expanded token");
        float* const ct = lds;
#pragma unroll
        for (int i = 0; i < G::MI; ++i)
#pragma unroll
            for (int j = 0; j < G::NJ; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    ct[(wm + i * 16 + lk * 4 + r) * G::LDC_T + wn + j * 16 + lj] = acc[i][j][r];
        __syncthreads();
        // 8 lanes per token row (mis_ln96_head_row, common.h): 32 rows per pass
        const int l8 = tid & 7, rg = tid >> 3;
        constexpr int NCM = 4;
        float4 g[3], bt[3], wv[NCM][3];
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const int c4 = (l8 + 8 * q) * 4;
            g[q] = *reinterpret_cast<const float4*>(a.ln_g + c4);
            bt[q] = *reinterpret_cast<const float4*>(a.ln_b + c4);
#pragma unroll
            for (int n = 0; n < NCM; ++n)
                wv[n][q] = n < a.head_nc ? *reinterpret_cast<const float4*>(a.head_w + n * BN + c4) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        const int P = a.ex_P, pp = n0 / BN, p1 = pp / P, p2 = pp - p1 * P;
        const long long S = (long long)a.ex_H * P * a.ex_W * P;
#pragma unroll
        for (int it = 0; it < BMT / 32; ++it) {
            const int row = it * 32 + rg, m = m0 + row;
            if (m >= a.M) continue;        // uniform within the row group
            float4 v[3];
#pragma unroll
            for (int q = 0; q < 3; ++q) v[q] = *reinterpret_cast<const float4*>(&ct[row * G::LDC_T + (l8 + 8 * q) * 4]);
            float mu, rs, pl[NCM];
            mis_ln96_head_row<NCM>(v, g, bt, wv, a.ln_eps, mu, rs, pl);
            const int w_ = m % a.ex_W, tt = m / a.ex_W, h_ = tt % a.ex_H, b_ = tt / a.ex_H;
            const long long pix = ((long long)h_ * P + p1) * ((long long)a.ex_W * P) + (long long)w_ * P + p2;
            const long long tok = (long long)b_ * S + pix;
            if (a.C) {
#pragma unroll
                for (int q = 0; q < 3; ++q) *reinterpret_cast<float4*>(a.C + tok * BN + (l8 + 8 * q) * 4) = v[q];
            }
            if (l8 == 0) {
                a.ln_mean[tok] = mu; a.ln_rstd[tok] = rs;
#pragma unroll
                for (int n = 0; n < NCM; ++n)
                    if (n < a.head_nc) a.logits[(long long)b_ * a.logits_bs + (long long)n * S + pix] = pl[n];
            }
        }
        return;
    }
    if (a.ex_P) {
        // expand epilogue: a 16-column group stays inside one (p1, p2) (c % 16 == 0), so per (row, group) one
        // destination row is computed and 16 lanes store 64 contiguous bytes of it
        // element offset = rowpart(m) + colpart(j) + lj with
        //   rowpart = ((b*H + h)*P*W*P + w*P) * c          (per lane and accumulator row)
        //   colpart = (p1*W*P + p2) * c + cc               (wave-uniform: the scalar offset of the buffer store)
        // the whole output is < 2^31 bytes (checked by the host), rows beyond M get an out-of-range offset
        const int P = a.ex_P, c = a.ex_c;
        const unsigned total_bytes = (unsigned)((long long)a.M * a.N * 4);
        const __amdgpu_buffer_rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc((void*)a.C, 0, (int)total_bytes, 0x00020000);
        unsigned rowpart[G::MI][4];
#pragma unroll
        for (int i = 0; i < G::MI; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + wm + i * 16 + lk * 4 + r;
                const int w_ = m % a.ex_W, t = m / a.ex_W;
                const int h_ = t % a.ex_H, b_ = t / a.ex_H;
                rowpart[i][r] = m < a.M ? (unsigned)(((b_ * a.ex_H + h_) * P * a.ex_W * P + w_ * P) * c + lj) * 4u
                                        : 0x80000000u;
            }
#pragma unroll
        for (int j = 0; j < G::NJ; ++j) {
            const int n = n0 + wn + j * 16;
            if (n >= a.N) break;   // uniform
            const int pp = n / c, cc = n - pp * c;
            const int p1 = pp / P, p2 = pp - p1 * P;
            const unsigned colpart = (unsigned)(((p1 * a.ex_W * P + p2) * c + cc) * 4);
#pragma unroll
            for (int i = 0; i < G::MI; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const unsigned off = rowpart[i][r] < 0x80000000u ? rowpart[i][r] + colpart : 0x80000000u;
                    const float v = acc[i][j][r];   // (bit_cast straight from the vector element picks element 0)
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rc, (int)off, 0, 0);
                }
        }
        return;
    }
    if (a.vec4) {
        // Fast path (C, bias, E1, C2 float4-addressable; checked by the host): accumulators -> LDS tile -> float4 rows.
        // The main loop ended with a barrier, so the stage buffers are free.  D row = lk*4 + r -> m, col = lj -> n.
        float* const ct = lds;
#pragma unroll
        for (int i = 0; i < G::MI; ++i)
#pragma unroll
            for (int j = 0; j < G::NJ; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    ct[(wm + i * 16 + lk * 4 + r) * G::LDC_T + wn + j * 16 + lj] = acc[i][j][r];
        __syncthreads();
        constexpr int Q = BN / 4;
#pragma unroll 4
        for (int it = 0; it < BMT * Q / (NW * 64); ++it) {
            const int e = tid + it * (NW * 64);
            const int row = e / Q, q = e - row * Q;
            const int m = m0 + row, n = n0 + q * 4;
            if (m >= a.M || n >= a.N) continue;
            float4 v = *reinterpret_cast<const float4*>(&ct[row * G::LDC_T + q * 4]);
            if constexpr (GDBG & 1) {
                if (v.x != 1.2345f) continue;
            }
            if (a.bias) {
                const float4 b = *reinterpret_cast<const float4*>(a.bias + n);
                v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;
            }
            float* const cp = a.C + (long long)m * a.ldc + n;
            if constexpr (EP == EP_GELU_FWD) {
                *reinterpret_cast<float4*>(a.C2 + (long long)m * a.ldc2 + n) =
                    make_float4(gelu_f(v.x), gelu_f(v.y), gelu_f(v.z), gelu_f(v.w));
                if (!a.C) continue;      // the pre-activation is only the backward's: a forward nobody differentiates drops it
            } else if constexpr (EP == EP_GELU_BWD) {
                const float4 h = *reinterpret_cast<const float4*>(a.E1 + (long long)m * a.lde1 + n);
                v.x *= gelu_grad_f(h.x); v.y *= gelu_grad_f(h.y); v.z *= gelu_grad_f(h.z); v.w *= gelu_grad_f(h.w);
            } else if constexpr (EP == EP_RESIDUAL) {
                const float4 sc = *reinterpret_cast<const float4*>(a.E1 + (long long)m * a.lde1 + n);
                const float rs = a.rowscale ? a.rowscale[m / a.rps] : 1.f;
                v.x = sc.x + rs * v.x; v.y = sc.y + rs * v.y; v.z = sc.z + rs * v.z; v.w = sc.w + rs * v.w;
            } else {
                if (a.accumulate) {
                    const float4 o = *reinterpret_cast<const float4*>(cp);
                    v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
                }
            }
            *reinterpret_cast<float4*>(cp) = v;
        }
        return;
    }
    if constexpr (EP != EP_NONE) return;   // the host only launches a fused instantiation on the float4 path
#pragma unroll
    for (int i = 0; i < G::MI; ++i)
#pragma unroll
        for (int j = 0; j < G::NJ; ++j) {
            const int n = n0 + wn + j * 16 + lj;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + wm + i * 16 + lk * 4 + r;
                if (m < a.M && n < a.N) {
                    float v = acc[i][j][r];
                    float* p = a.C + (long long)m * a.ldc + n;
                    if (a.bias) v += a.bias[n];
                    if (a.accumulate) v += *p;
                    *p = v;
                }
            }
        }
}

// ------------------------------------------------------------------------------------------------ NT, short contraction
// The Linears of the first two SwinUnet stages contract over K = 96 / 192 with 10^4 .. 10^5 token rows: a 128-row tile is
// a DMA prologue, 3 .. 6 k-steps and an epilogue that moves 2 .. 4x the bytes the operands did, and with two or three
// resident workgroups per CU these phases add up instead of overlapping (ablation, M = 150528, N = 384, K = 96, plain:
// 152 us = 53 MFMA + 30 stores + 26 k-loop DMA waits + 42 prologue / staging; MFMA alone would be 71, the bytes 40).
// Here: tiles of 64 rows, k-steps of 16 (two stages of 12 KB) and the accumulators staged through LDS in two halves of
// 32 rows, so a workgroup needs 24.6 KB of LDS and 72 registers and SIX of them share a CU; one's epilogue runs under
// the others' MFMAs (152 -> 131 us; 128-row tiles with two workgroups per CU: 152, 64-row tiles of the general kernel
// with three: 151).  A deeper ring of stages (3 .. 6, fewer resident workgroups) is slower (142 .. 179 us): the waits of
// the k-loop are not what a tile spends its time on (kernel without epilogue: 32 us), residency is what overlaps the
// phases.  Tile [row][16] with the 16-byte slot XOR-swizzled by (row >> 2) & 3 (rows 4 apart would share
// banks).  Only the float4 epilogue, no split-K: the host keeps the 128-row kernel for everything else.
template <int BN>
struct NsCfg {
    static constexpr int BMS = 64, BKS = 16;
    static constexpr int NJ = BN / 32;                    // 16-column MFMA tiles per wave (wave = 32 x BN/2)
    static constexpr int PB = BN / 16;                    // 16-row DMA pieces of the B tile (A: 4, one per wave)
    static constexpr int PBW = (PB + 3) / 4;
    static constexpr int A_FLOATS = BMS * BKS, B_FLOATS = BN * BKS;
    static constexpr int STAGE = A_FLOATS + B_FLOATS;
    static constexpr int LDC_T = BN + 4;
    static constexpr int CT_FLOATS = 32 * LDC_T;          // half a tile
    static constexpr int LDS_FLOATS = 2 * STAGE > CT_FLOATS ? 2 * STAGE : CT_FLOATS;
    static constexpr int LDS_BYTES = LDS_FLOATS * 4;
};

typedef short bf16x4s __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

// PREC = 1: the k-step of 16 as bf16x3 products on v_mfma_f32_16x16x16_bf16 (4 contraction elements per lane = the lane's two
// float2 reads of the step)
template <int BN, int EP, int PREC = 0>
__global__ __launch_bounds__(256) void gemm_nt_short_kernel(const GemmArgs a) {
    using G = NsCfg<BN>;
    constexpr int BKS = G::BKS;
    float* const lds = mis_gemm_lds;
    const unsigned L = mis_xcd_remap(blockIdx.x, a.n_blocks_padded);
    if (L >= a.n_blocks) return;
    const int tn = L % a.tiles_n, tm = L / a.tiles_n;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lk = lane >> 4, lj = lane & 15;
    const int wm = (wave >> 1) * 32, wn = (wave & 1) * (BN / 2);
    const int m0 = tm * G::BMS, n0 = tn * BN;
    const unsigned lds0 = lds_addr(lds);
    const i32x4 rA = make_rsrc(a.A, (unsigned)((long long)(a.M - 1) * a.lda + a.K) * 4u);
    const i32x4 rB = make_rsrc(a.B, (unsigned)((long long)(a.N - 1) * a.ldb + a.K) * 4u);

    // DMA piece p = 16 rows x 16 k (64 lanes x 16 B): lane -> row p*16 + (lane>>2), LDS slot lane&3 holds the source slot
    // (lane&3) ^ ((row>>2)&3), and (row>>2)&3 = (lane>>4)&3 for every piece
    const int ksrc = ((lane & 3) ^ ((lane >> 4) & 3)) * 4;
    unsigned voA, voB[G::PBW];
    {
        const int row = wave * 16 + (lane >> 2);
        voA = m0 + row < a.M ? (unsigned)((long long)(m0 + row) * a.lda + ksrc) * 4u : OOB;
    }
#pragma unroll
    for (int i = 0; i < G::PBW; ++i) {
        const int row = (wave + 4 * i) * 16 + (lane >> 2);
        voB[i] = (wave + 4 * i < G::PB && n0 + row < a.N) ? (unsigned)((long long)(n0 + row) * a.ldb + ksrc) * 4u : OOB;
    }
    auto stage = [&](int buf, int k0) {
        const unsigned st = lds0 + (unsigned)buf * (G::STAGE * 4);
        const bool kout = k0 + ksrc >= a.K;                   // K % 4 == 0: a 16-byte group is all in or all out
        const unsigned kb = (unsigned)k0 * 4u;
        dma_dwordx4(st + (unsigned)(wave * 256) * 4u, kout ? OOB : voA + kb, rA);
#pragma unroll
        for (int i = 0; i < G::PBW; ++i)
            if (wave + 4 * i < G::PB)      // uniform
                dma_dwordx4(st + (unsigned)(G::A_FLOATS + (wave + 4 * i) * 256) * 4u, kout ? OOB : voB[i] + kb, rB);
    };

    f32x4 acc[2][G::NJ];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < G::NJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int swz = (lj >> 2) & 3;        // rows wm + i*16 + lj, wn + j*16 + lj: only lj matters
    auto compute = [&](const float* st) {
        const float2* __restrict__ sA2 = reinterpret_cast<const float2*>(st);
        const float2* __restrict__ sB2 = reinterpret_cast<const float2*>(st + G::A_FLOATS);
#pragma unroll
        for (int s = 0; s < BKS / 8; ++s) {
            float2 af[2], bf[G::NJ];
            // k = 8 s + 2 lk, 2 lk + 1: slot 2 s + (lk >> 1), floats (lk & 1) * 2 inside it
            const int kk = (((2 * s + (lk >> 1)) ^ swz) * 4 + (lk & 1) * 2);
#pragma unroll
            for (int i = 0; i < 2; ++i) af[i] = sA2[((wm + i * 16 + lj) * BKS + kk) >> 1];
#pragma unroll
            for (int j = 0; j < G::NJ; ++j) bf[j] = sB2[((wn + j * 16 + lj) * BKS + kk) >> 1];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < G::NJ; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i].x, bf[j].x, acc[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < G::NJ; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i].y, bf[j].y, acc[i][j], 0, 0, 0);
        }
    };

    auto mfma16 = [](const u32x2& x, const u32x2& y, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(bf16x4s, x), __builtin_bit_cast(bf16x4s, y), c, 0, 0, 0);
    };
    auto compute_bf3 = [&](const float* st) {
        const float2* __restrict__ sA2 = reinterpret_cast<const float2*>(st);
        const float2* __restrict__ sB2 = reinterpret_cast<const float2*>(st + G::A_FLOATS);
        u32x2 ah[2], am[2], al[2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int s = 0; s < BKS / 8; ++s) {
                const float2 v = sA2[((wm + i * 16 + lj) * BKS + (((2 * s + (lk >> 1)) ^ swz) * 4 + (lk & 1) * 2)) >> 1];
                unsigned h, m, l;
                if constexpr (GDBG & 32) { h = __float_as_uint(v.x); m = __float_as_uint(v.y); l = h; }
                else bf3_split_pair(v.x, v.y, h, m, l);
                ah[i][s] = h; am[i][s] = m; al[i][s] = l;
            }
#pragma unroll
        for (int j = 0; j < G::NJ; ++j) {
            u32x2 bh, bm, bl;
#pragma unroll
            for (int s = 0; s < BKS / 8; ++s) {
                const float2 v = sB2[((wn + j * 16 + lj) * BKS + (((2 * s + (lk >> 1)) ^ swz) * 4 + (lk & 1) * 2)) >> 1];
                unsigned h, m, l;
                if constexpr (GDBG & 16) { h = __float_as_uint(v.x); m = __float_as_uint(v.y); l = h; }
                else bf3_split_pair(v.x, v.y, h, m, l);
                bh[s] = h; bm[s] = m; bl[s] = l;
            }
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                f32x4 c = acc[i][j];
                c = mfma16(al[i], bh, c);
                c = mfma16(ah[i], bl, c);
                c = mfma16(am[i], bm, c);
                c = mfma16(am[i], bh, c);
                c = mfma16(ah[i], bm, c);
                c = mfma16(ah[i], bh, c);
                acc[i][j] = c;
            }
        }
    };

    stage(0, 0);
    dma_wait();
    __syncthreads();
    int s = 0;
    for (int k0 = 0; k0 < a.K; k0 += BKS, ++s) {
        if (k0 + BKS < a.K) stage((s + 1) & 1, k0 + BKS);
        if constexpr (PREC == 1) compute_bf3(lds + (s & 1) * G::STAGE);
        else compute(lds + (s & 1) * G::STAGE);
        dma_wait();
        __syncthreads();
    }

    // ---- epilogue, 32 rows at a time: the two waves that hold them -> LDS -> float4 rows by all 256 threads ----
    float* const ct = lds;
    constexpr int Q = BN / 4;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        if ((wave >> 1) == h) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < G::NJ; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        ct[(i * 16 + lk * 4 + r) * G::LDC_T + wn + j * 16 + lj] = acc[i][j][r];
        }
        __syncthreads();
#pragma unroll
        for (int it = 0; it < (32 * Q + 255) / 256; ++it) {
            const int e = tid + it * 256;
            const int row = e / Q, q = e - row * Q;
            const int m = m0 + h * 32 + row, n = n0 + q * 4;
            if (row >= 32 || m >= a.M || n >= a.N) continue;
            float4 v = *reinterpret_cast<const float4*>(&ct[row * G::LDC_T + q * 4]);
            if (a.bias) {
                const float4 b = *reinterpret_cast<const float4*>(a.bias + n);
                v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;
            }
            float* const cp = a.C + (long long)m * a.ldc + n;
            if constexpr (EP == EP_GELU_FWD) {
                *reinterpret_cast<float4*>(a.C2 + (long long)m * a.ldc2 + n) =
                    make_float4(gelu_f(v.x), gelu_f(v.y), gelu_f(v.z), gelu_f(v.w));
                if (!a.C) continue;      // the pre-activation is only the backward's: a forward nobody differentiates drops it
            } else if constexpr (EP == EP_GELU_BWD) {
                const float4 g = *reinterpret_cast<const float4*>(a.E1 + (long long)m * a.lde1 + n);
                v.x *= gelu_grad_f(g.x); v.y *= gelu_grad_f(g.y); v.z *= gelu_grad_f(g.z); v.w *= gelu_grad_f(g.w);
            } else if constexpr (EP == EP_RESIDUAL) {
                const float4 sc = *reinterpret_cast<const float4*>(a.E1 + (long long)m * a.lde1 + n);
                const float rs = a.rowscale ? a.rowscale[m / a.rps] : 1.f;
                v.x = sc.x + rs * v.x; v.y = sc.y + rs * v.y; v.z = sc.z + rs * v.z; v.w = sc.w + rs * v.w;
            } else {
                if (a.accumulate) {
                    const float4 o = *reinterpret_cast<const float4*>(cp);
                    v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
                }
            }
            *reinterpret_cast<float4*>(cp) = v;
        }
        if (h == 0) __syncthreads();
    }
}

// (Round 5 also built a PERSISTENT form for K <= 96: a workgroup keeps one 96-column panel of the pre-split B in LDS (55 KB) for
// its whole life and walks down the row tiles, A tile t + 1 landing by DMA while tile t is multiplied, whole 384-byte output
// rows through an LDS stage.  Correct on the first run and SLOWER: 150528 x 384 x 96 159 us against the short-contraction
// kernel's 118, x 288: 134 against 89, x 96: 54 against 35.  The panel, the two A buffers and the stage are 127 KB: one
// workgroup per CU, one wave per SIMD, and a lone wave pays the LDS latency, the split of its A fragments and the six-deep
// MFMA chains one after the other -- 4.3 us per tile where the arithmetic is 0.8.  The short-contraction kernel's six resident
// workgroups hide exactly that.  Removed.)

// ------------------------------------------------------------------------------------------------ NT, A operand in registers
// Round 6.  Ablations of gemm_nt_kernel<64, 96, ., 2> on the 20 Linear shapes of a SwinUnet step (scripts/gemm_variants.sh,
// profiles/r06_gemm_nt_ablation.txt): without its MFMAs the kernel takes the SAME time (1366 vs 1368 us), without the DMAs of its
// k-loop -13 %, without the A split -7 %, without the epilogue -30 %: the matrix pipe is idle two thirds of the time and what a
// tile pays for is moving its operands through LDS.  With v_mfma_f32_16x16x32_bf16 the A operand of lane (lj, g) is row lj, 8
// CONSECUTIVE contraction elements 8 g .. 8 g + 7 -- 32 contiguous bytes of a row-major activation row, and the four lane groups
// of a row read one 128-byte line: unlike the fp32 MFMA's 4-element operands (round 4's register-only attempt: 16-byte pieces
// on 16 rows per load) the row-major A can be loaded straight into the operand registers.  So here:
//   * A never touches LDS: two buffer_load_dwordx4 per 16-row tile and k-step, one k-step ahead in registers (no DMA issue, no
//     ds_read, and the barrier of a k-step does not wait for it);
//   * a wave owns 32 rows x the tile's full 96 (128) columns: B -- the pre-split planes, NATURAL element order (SplitJob::
//     natural) -- is staged through LDS once per 128 rows instead of once per 64 and read once per wave (18 ds_read_b128 per
//     72 MFMAs);
//   * the weight tile is the MFMA's ROW operand: a lane's accumulator is four consecutive columns n of one token row -- a float4
//     of C -- and the epilogue (bias, GELU, residual, accumulate) stores straight from the registers: no LDS tile, no barrier
//     behind the k-loop (the staged kernel's epilogue is 30 % of its time);
//   * 48 accumulator registers, ~150 in all, 36 KB of LDS: three waves per SIMD, three workgroups per CU -- the residency that
//     hides the phases of a short contraction (see the persistent form's obituary above).
// Same split products and the same k-block order as the other bf16x3 kernels; the element order inside a K = 32 block differs,
// so results agree with them to fp32 rounding, not bit for bit.  Float4 epilogue only (plain / GELU / residual), no split-K.
// Epilogue of the register-A kernels, straight from the accumulators: tile (i, j) of lane (lj, g) = C[mrow + 16 i + lj][n0 + 16 j +
// 4 g .. + 3], a float4 of the row (the four lane groups of a row write 64 contiguous bytes, the tiles j the row's 4 * BN): no LDS
// PRE: the lane's bias float4s (bv, zeros without a bias) and its E1 / accumulate operands (ev) were loaded by the caller -- the
// persistent kernel requests them BEFORE the next slab's A loads: vmcnt retires in order, so a load issued in the epilogue would
// wait for the whole prefetch in front of it.
// EP_RESIDUAL_LN (N = 96: a row is one tile row): x = E1 + rowscale * v goes to C and LayerNorm(x) * gamma + beta to C2, mean /
// rstd to ln_mean / ln_rstd (`x = shortcut + drop_path(branch)` followed by `self.norm2(x)` / the next block's `norm1`, reference
// ...sys.py:276-281, :244-250).  The row's 96 values sit in the four lanes (lj, g = 0 .. 3): two in-lane sums of 24 and two
// cross-group steps each for mean and variance -- no LDS pass, and the LayerNorm launch with its read of x is gone.  gamma / beta
// come from LDS (`gb`: [gamma 96][beta 96], staged by the kernel): a global load here would wait behind the A prefetch.
#ifndef MIS_GEMM_REGA_PAIR
#define MIS_GEMM_REGA_PAIR 1
#endif
// the value of the lane 8 positions away in this lane's 16-lane row (v_mov_b32_dpp row_ror:8)
__device__ __forceinline__ float rega_ror8(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xF, 0xF, false));
}
template <int NJ, int EP, bool PRE = false>
__device__ __forceinline__ void rega_store(const GemmArgs& a, const f32x4 (&acc)[2][NJ], int mrow, int n0, int lj, int g,
                                           const float4* bv = nullptr, const float4 (*ev)[NJ] = nullptr, const float* gb = nullptr) {
    if constexpr (EP == EP_LNHEAD) {
        // FinalPatchExpand_X4 + its LayerNorm + the output head (reference ...sys.py:401-409, :671, :749-752): the tile's 96
        // columns are one (p1, p2) of the pixel shuffle, a tile row one token of the shuffled tensor.  gb = [gamma 96][beta 96]
        // [head_w NC x 96] in LDS.  The row's values sit in the four lanes (lj, g): LayerNorm and the NC head dot products are
        // in-lane sums over 24 values + two cross-group steps each.
        static_assert(NJ == 6, "a 96-column row");
        const int P = a.ex_P, pp = n0 / 96, p1 = pp / P, p2 = pp - p1 * P;
        const long long S = (long long)a.ex_H * P * a.ex_W * P;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int m = mrow + 16 * i + lj;
            const bool live = m < a.M;
            float4 x[NJ];
            float s = 0.f;
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                x[j] = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
                s += (x[j].x + x[j].y) + (x[j].z + x[j].w);
            }
            const int mm = live ? m : 0;
            const int w_ = mm % a.ex_W, tt = mm / a.ex_W, h_ = tt % a.ex_H, b_ = tt / a.ex_H;
            const long long pix = ((long long)h_ * P + p1) * ((long long)a.ex_W * P) + (long long)w_ * P + p2;
            const long long tok = (long long)b_ * S + pix;
            if (live && a.C) {
#pragma unroll
                for (int j = 0; j < NJ; ++j) *reinterpret_cast<float4*>(a.C + tok * 96 + 16 * j + 4 * g) = x[j];
            }
            s += __shfl_xor(s, 16, 64);
            s += __shfl_xor(s, 32, 64);
            const float mu = s / 96.f;
            float ss = 0.f;
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                x[j] = make_float4(x[j].x - mu, x[j].y - mu, x[j].z - mu, x[j].w - mu);
                ss += (x[j].x * x[j].x + x[j].y * x[j].y) + (x[j].z * x[j].z + x[j].w * x[j].w);
            }
            ss += __shfl_xor(ss, 16, 64);
            ss += __shfl_xor(ss, 32, 64);
            const float rstd = 1.f / sqrtf(ss / 96.f + a.ln_eps);
            float pl[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int n = 16 * j + 4 * g;
                const float4 ga = *reinterpret_cast<const float4*>(gb + n), be = *reinterpret_cast<const float4*>(gb + 96 + n);
                const float4 y = make_float4(x[j].x * rstd * ga.x + be.x, x[j].y * rstd * ga.y + be.y, x[j].z * rstd * ga.z + be.z,
                                             x[j].w * rstd * ga.w + be.w);
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const float4 w = *reinterpret_cast<const float4*>(gb + 192 + c * 96 + n);      // zero rows beyond head_nc
                    pl[c] += (y.x * w.x + y.y * w.y) + (y.z * w.z + y.w * w.w);
                }
            }
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                pl[c] += __shfl_xor(pl[c], 16, 64);
                pl[c] += __shfl_xor(pl[c], 32, 64);
            }
            if (live && g == 0) {
                a.ln_mean[tok] = mu; a.ln_rstd[tok] = rstd;
#pragma unroll
                for (int c = 0; c < 4; ++c)
                    if (c < a.head_nc) a.logits[(long long)b_ * a.logits_bs + (long long)c * S + pix] = pl[c];
            }
        }
        return;
    }
    if constexpr (EP == EP_RESIDUAL_LN) {
        static_assert(NJ == 6, "a 96-column row");
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int m = mrow + 16 * i + lj;
            const bool live = m < a.M;                       // (every lane takes part in the cross-group sums)
            const float rs = (a.rowscale && live) ? a.rowscale[m / a.rps] : 1.f;
            float4 x[NJ];
            float s = 0.f;
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int n = 16 * j + 4 * g;
                float4 v = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
                float4 b = make_float4(0.f, 0.f, 0.f, 0.f);
                if constexpr (PRE) b = bv[j];
                else if (a.bias) b = *reinterpret_cast<const float4*>(a.bias + n);
                float4 sc = make_float4(0.f, 0.f, 0.f, 0.f);
                if constexpr (PRE) sc = ev[i][j];
                else if (live) sc = *reinterpret_cast<const float4*>(a.E1 + (long long)m * a.lde1 + n);
                x[j] = make_float4(sc.x + rs * (v.x + b.x), sc.y + rs * (v.y + b.y), sc.z + rs * (v.z + b.z), sc.w + rs * (v.w + b.w));
                s += (x[j].x + x[j].y) + (x[j].z + x[j].w);
                if (live) *reinterpret_cast<float4*>(a.C + (long long)m * a.ldc + n) = x[j];
            }
            s += __shfl_xor(s, 16, 64);
            s += __shfl_xor(s, 32, 64);
            const float mu = s / 96.f;
            float ss = 0.f;
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                x[j] = make_float4(x[j].x - mu, x[j].y - mu, x[j].z - mu, x[j].w - mu);
                ss += (x[j].x * x[j].x + x[j].y * x[j].y) + (x[j].z * x[j].z + x[j].w * x[j].w);
            }
            ss += __shfl_xor(ss, 16, 64);
            ss += __shfl_xor(ss, 32, 64);
            const float rstd = 1.f / sqrtf(ss / 96.f + a.ln_eps);
            if (!live) continue;
            if (g == 0) { a.ln_mean[m] = mu; a.ln_rstd[m] = rstd; }
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int n = 16 * j + 4 * g;
                const float4 ga = *reinterpret_cast<const float4*>(gb + n), be = *reinterpret_cast<const float4*>(gb + 96 + n);
                *reinterpret_cast<float4*>(a.C2 + (long long)m * a.ldc2 + n) =
                    make_float4(x[j].x * rstd * ga.x + be.x, x[j].y * rstd * ga.y + be.y, x[j].z * rstd * ga.z + be.z,
                                x[j].w * rstd * ga.w + be.w);
            }
        }
        return;
    }
    // Stores in 128-byte runs: a lane group (lj, g = 0 .. 3) holds 64 contiguous bytes of its row per tile j, and HBM gives 64-byte
    // pieces 3.1 TB/s where 128-byte pieces get 4.4 (scripts/ubench/hbm_pieces.hip).  So the lanes lj and lj ^ 8 of a 16-lane row
    // trade halves of a PAIR of tiles (j, j + 1) -- one v_mov_dpp row_ror:8 per value -- and each store instruction writes rows
    // (lj & 7) [+ 8] with eight lanes = 128 contiguous bytes per row: lanes lj < 8 the columns of tile j, lanes lj >= 8 those of
    // tile j + 1 (MIS_GEMM_REGA_PAIR=0 at build time keeps the 64-byte form).
#if MIS_GEMM_REGA_PAIR
    static_assert(NJ % 2 == 0, "tile pairs");
    const bool hi = lj >= 8;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int m = mrow + 16 * i + lj;
        const bool live = m < a.M;
        const float rs = (EP == EP_RESIDUAL && a.rowscale && live) ? a.rowscale[m / a.rps] : 1.f;
        float4 out[NJ], out2[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int n = n0 + 16 * j + 4 * g;
            float4 v = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
            if constexpr (PRE) {
                v.x += bv[j].x; v.y += bv[j].y; v.z += bv[j].z; v.w += bv[j].w;
            } else if (a.bias) {
                const float4 b = *reinterpret_cast<const float4*>(a.bias + n);
                v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;
            }
            if constexpr (EP == EP_GELU_FWD) {
                out2[j] = make_float4(gelu_f(v.x), gelu_f(v.y), gelu_f(v.z), gelu_f(v.w));
            } else if constexpr (EP == EP_GELU_BWD) {
                float4 h = make_float4(0.f, 0.f, 0.f, 0.f);
                if constexpr (PRE) h = ev[i][j]; else if (live) h = *reinterpret_cast<const float4*>(a.E1 + (long long)m * a.lde1 + n);
                v.x *= gelu_grad_f(h.x); v.y *= gelu_grad_f(h.y); v.z *= gelu_grad_f(h.z); v.w *= gelu_grad_f(h.w);
            } else if constexpr (EP == EP_RESIDUAL) {
                float4 sc = make_float4(0.f, 0.f, 0.f, 0.f);
                if constexpr (PRE) sc = ev[i][j]; else if (live) sc = *reinterpret_cast<const float4*>(a.E1 + (long long)m * a.lde1 + n);
                v.x = sc.x + rs * v.x; v.y = sc.y + rs * v.y; v.z = sc.z + rs * v.z; v.w = sc.w + rs * v.w;
            } else {
                if (a.accumulate) {
                    float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
                    if constexpr (PRE) o = ev[i][j]; else if (live) o = *reinterpret_cast<const float4*>(a.C + (long long)m * a.ldc + n);
                    v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
                }
            }
            out[j] = v;
        }
        // rows of the two store instructions of a pair, and this lane's column in both
        const int r1 = mrow + 16 * i + (lj & 7), r2 = r1 + 8;
        const bool live1 = r1 < a.M, live2 = r2 < a.M;
        auto paired = [&](const float4 (&o)[NJ], float* base, long long ld) {
#pragma unroll
            for (int j = 0; j < NJ; j += 2) {
                const float4 A = o[j], B = o[j + 1];
                const float4 send = hi ? A : B;
                const float4 recv = make_float4(rega_ror8(send.x), rega_ror8(send.y), rega_ror8(send.z), rega_ror8(send.w));
                const float4 d1 = hi ? recv : A, d2 = hi ? B : recv;
                const int n = n0 + 16 * (j + (hi ? 1 : 0)) + 4 * g;
                if (live1) *reinterpret_cast<float4*>(base + (long long)r1 * ld + n) = d1;
                if (live2) *reinterpret_cast<float4*>(base + (long long)r2 * ld + n) = d2;
            }
        };
        if constexpr (EP == EP_GELU_FWD) {
            paired(out2, a.C2, a.ldc2);
            if (a.C) paired(out, a.C, a.ldc);
        } else {
            paired(out, a.C, a.ldc);
        }
    }
#else
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int m = mrow + 16 * i + lj;
        if (m >= a.M) continue;
        const float rs = (EP == EP_RESIDUAL && a.rowscale) ? a.rowscale[m / a.rps] : 1.f;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int n = n0 + 16 * j + 4 * g;
            if (n >= a.N) continue;
            float4 v = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
            if constexpr (PRE) {
                v.x += bv[j].x; v.y += bv[j].y; v.z += bv[j].z; v.w += bv[j].w;
            } else if (a.bias) {
                const float4 b = *reinterpret_cast<const float4*>(a.bias + n);
                v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;
            }
            float* const cp = a.C + (long long)m * a.ldc + n;
            if constexpr (EP == EP_GELU_FWD) {
                *reinterpret_cast<float4*>(a.C2 + (long long)m * a.ldc2 + n) =
                    make_float4(gelu_f(v.x), gelu_f(v.y), gelu_f(v.z), gelu_f(v.w));
                if (!a.C) continue;
            } else if constexpr (EP == EP_GELU_BWD) {
                float4 h;
                if constexpr (PRE) h = ev[i][j]; else h = *reinterpret_cast<const float4*>(a.E1 + (long long)m * a.lde1 + n);
                v.x *= gelu_grad_f(h.x); v.y *= gelu_grad_f(h.y); v.z *= gelu_grad_f(h.z); v.w *= gelu_grad_f(h.w);
            } else if constexpr (EP == EP_RESIDUAL) {
                float4 sc;
                if constexpr (PRE) sc = ev[i][j]; else sc = *reinterpret_cast<const float4*>(a.E1 + (long long)m * a.lde1 + n);
                v.x = sc.x + rs * v.x; v.y = sc.y + rs * v.y; v.z = sc.z + rs * v.z; v.w = sc.w + rs * v.w;
            } else {
                if (a.accumulate) {
                    float4 o;
                    if constexpr (PRE) o = ev[i][j]; else o = *reinterpret_cast<const float4*>(cp);
                    v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
                }
            }
            *reinterpret_cast<float4*>(cp) = v;
        }
    }
#endif
}

template <int BN, int EP>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) void gemm_nt_rega_kernel(const GemmArgs a) {
    constexpr int MI = 2, NJ = BN / 16, BMT = 128;
    constexpr int RB3 = BN / 16, PB3 = 3 * RB3, NB3 = (PB3 + 3) / 4;
    constexpr int B_FLOATS = BN * BK * 3 / 2;                   // one stage: three planes [BN][32] bf16
    float* const lds = mis_gemm_lds;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, lj = lane & 15;
    const int wm = wave * 32;
    const unsigned lds0 = lds_addr(lds);
    const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.A), 0,
                                                                        (int)(((long long)(a.M - 1) * a.lda + a.K) * 4), 0x00020000);
    const i32x4 rB3 = make_rsrc(a.B3, 3u * a.b3_plane);
    // Tiles: the grid is at most three workgroups per CU; a workgroup walks the (XCD-remapped) tile list with stride gridDim.x
    // (a multiple of the XCD count, so a workgroup's tiles stay in its XCD's contiguous range) and requests the first stage of
    // its NEXT tile during the last k-step of the current one: the epilogue runs with that stage already in LDS / registers.
    auto tile_after = [&](unsigned v) -> unsigned {            // next v (>= the argument) with a real tile, or ~0u
        for (; v < a.n_blocks_padded; v += gridDim.x)
            if (mis_xcd_remap(v, a.n_blocks_padded) < a.n_blocks) return v;
        return ~0u;
    };
    unsigned v = tile_after(blockIdx.x);
    if (v == ~0u) return;
    int m0, n0;
    int voA[MI];
    unsigned voB3[NB3];
    auto place = [&](unsigned vv) {
        const unsigned L = mis_xcd_remap(vv, a.n_blocks_padded);
        const int tn = L % a.tiles_n, tm = L / a.tiles_n;
        m0 = tm * BMT;
        n0 = tn * BN;
        // A: lane (lj, g) of tile i reads row m0 + wm + 16 i + lj, floats k0 + 8 g .. + 7 (two 16-byte loads)
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            const int row = m0 + wm + 16 * i + lj;
            voA[i] = row < a.M ? (int)(((long long)row * a.lda + 8 * g) * 4) : (int)OOB;
        }
        // B planes: piece q = (plane q / RB3, rows (q % RB3) * 16 ..+16) x the 64 bytes of the k-step (as gemm_nt_kernel, PREC = 2)
#pragma unroll
        for (int i = 0; i < NB3; ++i) {
            const int q = wave + 4 * i, plane = q / RB3, row = (q % RB3) * 16 + (lane >> 2);
            voB3[i] = (q < PB3 && n0 + row < a.N)
                          ? (unsigned)plane * a.b3_plane + (unsigned)(((long long)(n0 + row) * a.K3 + (lane & 3) * 8) * 2)
                          : OOB;
        }
    };
    auto stage_b = [&](int buf, int k0) {
        const unsigned st = lds0 + (unsigned)buf * (B_FLOATS * 4);
#pragma unroll
        for (int i = 0; i < NB3; ++i)
            if (wave + 4 * i < PB3) dma_dwordx4(st + (unsigned)((wave + 4 * i) * 1024), voB3[i] + (unsigned)k0 * 2u, rB3);
    };
    auto load_a = [&](f32x4 (&r)[MI][2], int k0) {
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                // K % 4 == 0: a 16-byte group is all in or all out; beyond K the planes are zero but A must not read the next row
                const bool in = k0 + 8 * g + 4 * h < a.K;
                r[i][h] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rA, in ? voA[i] + 16 * h : (int)OOB, k0 * 4, 0));
            }
    };

    f32x4 acc[MI][NJ];
    f32x4 cur[MI][2], nxt[MI][2];

    float* const gb = lds + 2 * B_FLOATS;          // EP_RESIDUAL_LN: [gamma 96][beta 96] behind the two stages
    if constexpr (EP == EP_RESIDUAL_LN) {
        if (tid < 96) { gb[tid] = a.ln_g[tid]; gb[96 + tid] = a.ln_b[tid]; }
    }
    place(v);
    stage_b(0, 0);
    load_a(cur, 0);
    dma_wait();
    __syncthreads();
    int s = 0;
    while (true) {
        const int tm0 = m0, tn0 = n0;               // this tile's origin (place() moves m0 / n0 on during the last k-step)
        const unsigned vn = tile_after(v + gridDim.x);
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int k0 = 0; k0 < a.K; k0 += BK, ++s) {
            const bool more = k0 + BK < a.K;
            bool loaded = more;
            if (more) {
                stage_b((s + 1) & 1, k0 + BK);
                load_a(nxt, k0 + BK);
            } else if (vn != ~0u) {
                place(vn);
                stage_b((s + 1) & 1, 0);
                load_a(nxt, 0);
                loaded = true;
            }
            MisBf3 a3[MI];
#pragma unroll
            for (int i = 0; i < MI; ++i)
                a3[i] = mis_bf3_from8(cur[i][0][0], cur[i][0][1], cur[i][0][2], cur[i][0][3], cur[i][1][0], cur[i][1][1], cur[i][1][2], cur[i][1][3]);
            const u32x4* __restrict__ sB3 = reinterpret_cast<const u32x4*>(lds + (s & 1) * B_FLOATS);
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int r = (j * 16 + lj) * 4 + g;
                const u32x4 bh = sB3[r], bm = sB3[BN * 4 + r], bl = sB3[2 * BN * 4 + r];
#pragma unroll
                for (int i = 0; i < MI; ++i) {
                    // the WEIGHT tile is the MFMA's row operand: D[row = column n of C, 4 g + r][col = token lj] -- a lane ends up
                    // with four consecutive n of one token, i.e. a float4 of a C row (see the epilogue)
                    f32x4 c = acc[i][j];
                    c = bf3_mfma(bh, a3[i].l, c);
                    c = bf3_mfma(bl, a3[i].h, c);
                    c = bf3_mfma(bm, a3[i].m, c);
                    c = bf3_mfma(bh, a3[i].m, c);
                    c = bf3_mfma(bm, a3[i].h, c);
                    c = bf3_mfma(bh, a3[i].h, c);
                    acc[i][j] = c;
                }
            }
            if (loaded) {
#pragma unroll
                for (int i = 0; i < MI; ++i) { cur[i][0] = nxt[i][0]; cur[i][1] = nxt[i][1]; }
            }
            dma_wait();
            __syncthreads();
        }
        rega_store<NJ, EP>(a, acc, tm0 + wm, tn0, lj, g, nullptr, nullptr, gb);
        if (vn == ~0u) break;
        v = vn;
    }
}

template <int BN>
constexpr int rega_lds_bytes() { return 2 * (BN * BK * 3 / 2) * 4 + 768; }      // the two B stages (36 KB for BN = 96) + gamma / beta

template <int NJ>
__device__ __forceinline__ void rega_store_plain(const GemmArgs& a, const f32x4 (&acc)[2][NJ], int mrow, int n0, int lj, int g,
                                                 const float4* bv) {
#if MIS_GEMM_REGA_PAIR
    // 128-byte runs per row and store instruction (see rega_store)
    const bool hi = lj >= 8;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int r1 = mrow + 16 * i + (lj & 7), r2 = r1 + 8;
        const bool live1 = r1 < a.M, live2 = r2 < a.M;
#pragma unroll
        for (int j = 0; j < NJ; j += 2) {
            const float4 A = make_float4(acc[i][j][0] + bv[j].x, acc[i][j][1] + bv[j].y, acc[i][j][2] + bv[j].z, acc[i][j][3] + bv[j].w);
            const float4 B = make_float4(acc[i][j + 1][0] + bv[j + 1].x, acc[i][j + 1][1] + bv[j + 1].y, acc[i][j + 1][2] + bv[j + 1].z,
                                         acc[i][j + 1][3] + bv[j + 1].w);
            const float4 send = hi ? A : B;
            const float4 recv = make_float4(rega_ror8(send.x), rega_ror8(send.y), rega_ror8(send.z), rega_ror8(send.w));
            const float4 d1 = hi ? recv : A, d2 = hi ? B : recv;
            const int n = n0 + 16 * (j + (hi ? 1 : 0)) + 4 * g;
            if (live1) *reinterpret_cast<float4*>(a.C + (long long)r1 * a.ldc + n) = d1;
            if (live2) *reinterpret_cast<float4*>(a.C + (long long)r2 * a.ldc + n) = d2;
        }
    }
#else
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int m = mrow + 16 * i + lj;
        if (m >= a.M) continue;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int n = n0 + 16 * j + 4 * g;
            if (n >= a.N) continue;
            *reinterpret_cast<float4*>(a.C + (long long)m * a.ldc + n) =
                make_float4(acc[i][j][0] + bv[j].x, acc[i][j][1] + bv[j].y, acc[i][j][2] + bv[j].z, acc[i][j][3] + bv[j].w);
        }
    }
#endif
}

// The same kernel for K <= 96 (the 96-channel stage: qkv / proj / fc1 forward, proj / fc2 data gradient -- ~2 ms of a SwinUnet step
// at 10^5 token rows), PERSISTENT with the weight panel RESIDENT.  A k-loop of three steps is all prologue: the streamed kernel
// spends ~19 us on a tile whose MFMAs take 1.6 (one batch of loads per barrier).  Here a workgroup keeps its 96-column panel of
// the planes in LDS for its whole life (K3 / 32 x 18 KB = 55 KB: two workgroups per CU) and its four waves walk down the
// 32-row slabs of the matrix independently: a wave requests the WHOLE contraction of its next slab (12 x 16 bytes per lane)
// before it multiplies the current one (216 MFMAs) and stores it from the accumulators -- no barrier and no LDS write after the
// prologue, two slabs of loads in flight per wave, two waves per SIMD.  (Round 5's persistent form staged A through LDS and the
// accumulators back through it: one workgroup per CU.)
template <int EP>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void gemm_nt_rega_res_kernel(const GemmArgs a) {
    constexpr int BN = 96, MI = 2, NJ = BN / 16, KBM = 3;
    constexpr int RB3 = BN / 16, PB3 = 3 * RB3;
    constexpr int B_FLOATS = BN * BK * 3 / 2;                   // one k-block: three planes [96][32] bf16
    float* const lds = mis_gemm_lds;
    const unsigned L = mis_xcd_remap(blockIdx.x, a.n_blocks_padded);
    if (L >= a.n_blocks) return;
    const int tn = L % a.tiles_n, p = L / a.tiles_n;             // column panel, walker index
    const int walkers = a.n_blocks / a.tiles_n;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, lj = lane & 15;
    const int n0 = tn * BN;
    const int kb_n = a.K3 / BK;                                  // 1 .. 3
    const unsigned lds0 = lds_addr(lds);
    const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.A), 0,
                                                                        (int)(((long long)(a.M - 1) * a.lda + a.K) * 4), 0x00020000);
    const i32x4 rB3 = make_rsrc(a.B3, 3u * a.b3_plane);
    // ---- prologue: the panel, all k-blocks ----
    for (int q = wave; q < kb_n * PB3; q += 4) {
        const int kb = q / PB3, qq = q - kb * PB3;
        const int plane = qq / RB3, row = (qq % RB3) * 16 + (lane >> 2);
        const unsigned vo = n0 + row < a.N
                                ? (unsigned)plane * a.b3_plane + (unsigned)(((long long)(n0 + row) * a.K3 + kb * 32 + (lane & 3) * 8) * 2)
                                : OOB;
        dma_dwordx4(lds0 + (unsigned)(kb * B_FLOATS * 4 + qq * 1024), vo, rB3);
    }
    float* const gb = lds + KBM * B_FLOATS;        // EP_RESIDUAL_LN / EP_LNHEAD: [gamma 96][beta 96]([head_w 4 x 96]) behind the panel
    if constexpr (EP == EP_RESIDUAL_LN || EP == EP_LNHEAD) {
        if (tid < 96) { gb[tid] = a.ln_g[tid]; gb[96 + tid] = a.ln_b[tid]; }
    }
    if constexpr (EP == EP_LNHEAD) {
        for (int t = tid; t < 4 * 96; t += 256) gb[192 + t] = t < a.head_nc * 96 ? a.head_w[t] : 0.f;
    }
    dma_wait();
    __syncthreads();

    const int slabs = (a.M + 31) / 32;
    const int stride = walkers * 4;
    auto load_a = [&](f32x4 (&r)[KBM][MI][2], int slab) {
        const int mrow = slab * 32;
#pragma unroll
        for (int kb = 0; kb < KBM; ++kb)
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                const int row = mrow + 16 * i + lj;
                const int vo = row < a.M ? (int)(((long long)row * a.lda + 8 * g) * 4) : (int)OOB;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const bool in = kb * 32 + 8 * g + 4 * h < a.K;      // also false for k-blocks past K3
                    r[kb][i][h] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rA, in ? vo + 16 * h : (int)OOB, kb * 128, 0));
                }
            }
    };
    f32x4 cur[KBM][MI][2], nxt[KBM][MI][2];
    int slab = p * 4 + wave;
    if (slab >= slabs) return;
    float4 bv[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int n = n0 + 16 * j + 4 * g;
        bv[j] = (a.bias && n < a.N) ? *reinterpret_cast<const float4*>(a.bias + n) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    // the epilogue's second operand (pre-activation / shortcut / old C) of THIS slab, requested ahead of the next slab's A
    constexpr bool HAS_E = EP == EP_GELU_BWD || EP == EP_RESIDUAL || EP == EP_RESIDUAL_LN || EP == EP_NONE;      // (not EP_LNHEAD)
    const bool use_e = EP == EP_NONE ? a.accumulate != 0 : true;
    const float* const ebase = EP == EP_NONE ? a.C : a.E1;
    const long long lde = EP == EP_NONE ? a.ldc : a.lde1;
    float4 ev[MI][NJ];
    // EP_LNHEAD: no second A buffer (its epilogue -- LayerNorm + head on 24 values per lane -- needs the registers; the launch is
    // bound by the 925 MB it writes, or short: the teacher's)
    constexpr bool PREFETCH = EP != EP_LNHEAD;
    load_a(cur, slab);
    for (; slab < slabs; slab += stride) {
        const bool more = slab + stride < slabs;
        if constexpr (HAS_E) {
            if (use_e) {
#pragma unroll
                for (int i = 0; i < MI; ++i) {
                    const int m = slab * 32 + 16 * i + lj;
#pragma unroll
                    for (int j = 0; j < NJ; ++j) {
                        const int n = n0 + 16 * j + 4 * g;
                        ev[i][j] = (m < a.M && n < a.N) ? *reinterpret_cast<const float4*>(ebase + (long long)m * lde + n)
                                                        : make_float4(0.f, 0.f, 0.f, 0.f);
                    }
                }
            }
        }
        if (PREFETCH && more) load_a(nxt, slab + stride);
        f32x4 acc[MI][NJ];
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kb = 0; kb < KBM; ++kb) {
            if (kb >= kb_n) break;
            MisBf3 a3[MI];
#pragma unroll
            for (int i = 0; i < MI; ++i)
                a3[i] = mis_bf3_from8(cur[kb][i][0][0], cur[kb][i][0][1], cur[kb][i][0][2], cur[kb][i][0][3],
                                      cur[kb][i][1][0], cur[kb][i][1][1], cur[kb][i][1][2], cur[kb][i][1][3]);
            const u32x4* __restrict__ sB3 = reinterpret_cast<const u32x4*>(lds + kb * B_FLOATS);
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int r = (j * 16 + lj) * 4 + g;
                const u32x4 bh = sB3[r], bm = sB3[BN * 4 + r], bl = sB3[2 * BN * 4 + r];
#pragma unroll
                for (int i = 0; i < MI; ++i) {
                    f32x4 c = acc[i][j];
                    c = bf3_mfma(bh, a3[i].l, c);
                    c = bf3_mfma(bl, a3[i].h, c);
                    c = bf3_mfma(bm, a3[i].m, c);
                    c = bf3_mfma(bh, a3[i].m, c);
                    c = bf3_mfma(bm, a3[i].h, c);
                    c = bf3_mfma(bh, a3[i].h, c);
                    acc[i][j] = c;
                }
            }
        }
        if constexpr (EP == EP_NONE) {
            if (use_e) rega_store<NJ, EP, true>(a, acc, slab * 32, n0, lj, g, bv, ev);
            else rega_store_plain<NJ>(a, acc, slab * 32, n0, lj, g, bv);
        } else {
            rega_store<NJ, EP, true>(a, acc, slab * 32, n0, lj, g, bv, ev, gb);
        }
        if (more) {
            if constexpr (PREFETCH) {
#pragma unroll
                for (int kb = 0; kb < KBM; ++kb)
#pragma unroll
                    for (int i = 0; i < MI; ++i) { cur[kb][i][0] = nxt[kb][i][0]; cur[kb][i][1] = nxt[kb][i][1]; }
            } else {
                __builtin_amdgcn_sched_barrier(0);          // (the loads must not move up into the epilogue: registers)
                load_a(cur, slab + stride);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------ TN
template <int BT>
__global__ __launch_bounds__(256) void gemm_tn_kernel(const GemmArgs a) {
    constexpr int LD = BT + 16;        // 144 / 112: = 16 mod 32
    constexpr int NI = BT / 32;        // 16-wide MFMA tiles per wave and dimension (wave = BT/2 x BT/2)
    constexpr int Q = BT / 4;          // float4 per staged row
    __shared__ __attribute__((aligned(16))) float sA[BK * LD];
    __shared__ __attribute__((aligned(16))) float sB[BK * LD];

    const unsigned L = mis_xcd_remap(blockIdx.x, a.n_blocks_padded);
    if (L >= a.n_blocks) return;
    const unsigned tiles = (unsigned)(a.tiles_n * a.tiles_m);
    const int kz = L / tiles;
    const unsigned tl = L - kz * tiles;
    const int tn = tl % a.tiles_n, tm = tl / a.tiles_n;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lk = lane >> 4, lj = lane & 15;
    const int wm = (wave >> 1) * (BT / 2), wn = (wave & 1) * (BT / 2);
    const int m0 = tm * BT, n0 = tn * BT;
    const int kbeg = kz * a.kchunk;
    const int kend = kbeg + a.kchunk < a.K ? kbeg + a.kchunk : a.K;

    f32x4 acc[NI][NI];
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    // column sums of A (the bias gradient): the workgroups of the first tile column, thread = (column, half of the k-step)
    const bool colsum = a.dbias != nullptr && tn == 0;
    const int bc = tid % BT, bh = tid / BT;
    float bsum = 0.f;

    // rows = k (contraction), BT floats = Q float4 per row; branch-free, zero fill outside the matrices.  The tiles of
    // k-step s + 1 are loaded into registers before the MFMAs of k-step s and stored to LDS after them: the global-load
    // latency hides behind the workgroup's own MFMAs (no second LDS buffer: four workgroups stay resident per CU)
    constexpr int NLD = (BK * Q + 255) / 256;
    float4 ra[NLD], rb[NLD];
    auto fetch = [&](int k0) {
#pragma unroll
        for (int it = 0; it < NLD; ++it) {
            const int e = tid + it * 256;
            const int r = e / Q, q = e - r * Q;
            const int k = k0 + r;
            const bool in = e < BK * Q;
            {
                const bool ok = in && k < kend && m0 + q * 4 < a.M;
                const long long off = ok ? (long long)k * a.lda + m0 + q * 4 : 0;
                ra[it] = *reinterpret_cast<const float4*>(a.A + off);
                if (!ok) ra[it] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
            {
                const bool ok = in && k < kend && n0 + q * 4 < a.N;
                const long long off = ok ? (long long)k * a.ldb + n0 + q * 4 : 0;
                rb[it] = *reinterpret_cast<const float4*>(a.B + off);
                if (!ok) rb[it] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
    };
    fetch(kbeg);
    for (int k0 = kbeg; k0 < kend; k0 += BK) {
        __syncthreads();
#pragma unroll
        for (int it = 0; it < NLD; ++it) {
            const int e = tid + it * 256;
            const int r = e / Q, q = e - r * Q;
            if (e < BK * Q) {
                *reinterpret_cast<float4*>(sA + r * LD + q * 4) = ra[it];
                *reinterpret_cast<float4*>(sB + r * LD + q * 4) = rb[it];
            }
        }
        if (k0 + BK < kend) fetch(k0 + BK);
        __syncthreads();
        if (colsum && bh < 2) {
#pragma unroll
            for (int r = 0; r < BK / 2; ++r) bsum += sA[(bh * (BK / 2) + r) * LD + bc];
        }
#pragma unroll
        for (int s = 0; s < BK / 4; ++s) {
            float af[NI], bf[NI];
#pragma unroll
            for (int i = 0; i < NI; ++i) af[i] = sA[(s * 4 + lk) * LD + wm + i * 16 + lj];
#pragma unroll
            for (int j = 0; j < NI; ++j) bf[j] = sB[(s * 4 + lk) * LD + wn + j * 16 + lj];
#pragma unroll
            for (int i = 0; i < NI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
    }

    const bool direct = a.KS == 1;
    if (colsum) {      // workgroup-uniform
        __syncthreads();
        if (bh < 2) sB[bh * BT + bc] = bsum;
        __syncthreads();
        if (tid < BT && m0 + tid < a.M) {
            const float v = sB[tid] + sB[BT + tid];
            if (direct) a.dbias[m0 + tid] = a.dbias_acc ? a.dbias[m0 + tid] + v : v;
            else a.wsb[(long long)kz * a.M + m0 + tid] = v;
        }
    }
    float* __restrict__ out = direct ? a.C : a.ws + (long long)kz * a.M * a.N;
    const long long ldo = direct ? a.ldc : a.N;
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const int n = n0 + wn + j * 16 + lj;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + wm + i * 16 + lk * 4 + r;
                if (m < a.M && n < a.N) {
                    float v = acc[i][j][r];
                    float* p = out + (long long)m * ldo + n;
                    if (direct) {
                        if (a.bias) v += a.bias[n];
                        if (a.accumulate) v += *p;
                    }
                    *p = v;
                }
            }
        }
}

// ------------------------------------------------------------------------------------------------ TN, register-only
// dW = dY^T . X for the Linear layers whose widths are multiples of 96 (every Linear of SwinUnet and of UNETR's ViT): round 4.
// The contraction runs over ~10^5 token rows and both operands are contraction-major (a row of A / B = one token, contiguous
// along m / n): exactly the v_mfma_f32_16x16x4_f32 operand layout if lane (i = lane & 15, kk = lane >> 4) loads the float2
// A[k + kk][m0 + 32 t + 2 i .. + 1] -- component c is the A operand of the 16 x 16 tile whose row i stands for m = m0 + 32 t +
// 2 i + c.  So there is NO LDS stage, no barrier and no vector ALU work in the k-loop at all: a wave owns a 96 x 96 output tile
// (36 accumulator tiles, 144 registers) over its own range of token rows and per group of 4 rows issues 6 buffer_load_dwordx2
// (3 for A, 3 for B: 4 rows x 128 contiguous bytes each) and 36 MFMAs; loads run 4 groups ahead in a register ring.  Rows past
// the wave's range are the descriptor's range check (zeros).  The staged kernel above spends 268 vector instructions per 72
// MFMAs on the same job (operand staging through registers into LDS, scalar LDS reads) and runs the matrix pipe at 0.46.
// The 4 waves of a workgroup take 4 consecutive row ranges of the same tile and sum their tiles through LDS (fixed tree:
// deterministic), so a launch writes tiles x workgroup-slices partials, summed by gemm_reduce_kernel as before.
// The bias gradient (column sums of A) rides on the waves of the first tile column: 3 packed adds per group.
// The zeros past a wave's row range rest on the GFX9-family raw-buffer range check, which INCLUDES soffset (the row advance
// is carried there): an architecture that excludes it would return the next wave's rows -- silently wrong dW.  This file is
// built for gfx950 only; refuse anything else at compile time (tests/test_token_kernels_gpu.py puts NaNs behind the range).
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "gemm_tn_reg_kernel: buffer range check with soffset is verified for gfx950 only"
#endif
constexpr int RT = 96;                 // wave tile edge
constexpr int RD = 4;                  // groups (of 4 rows) in flight

template <bool COLSUM>
__device__ __forceinline__ void tn_reg_loop(f32x4 (&acc)[6][6], float2 (&bs)[3], __amdgpu_buffer_rsrc_t rA, __amdgpu_buffer_rsrc_t rB,
                                            int va, int vb, unsigned sa, unsigned sb, unsigned step_a, unsigned step_b, int ngroups) {
    float2 ra[RD][3], rb[RD][3];
    auto issue = [&](int d) {
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            ra[d][t] = __builtin_bit_cast(float2, __builtin_amdgcn_raw_buffer_load_b64(rA, va + t * 128, (int)sa, 0));
            rb[d][t] = __builtin_bit_cast(float2, __builtin_amdgcn_raw_buffer_load_b64(rB, vb + t * 128, (int)sb, 0));
        }
        sa += step_a; sb += step_b;
    };
#pragma unroll
    for (int d = 0; d < RD; ++d) issue(d);
    for (int g = 0; g < ngroups; g += RD) {
#pragma unroll
        for (int d = 0; d < RD; ++d) {
            float af[6], bf[6];
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                af[2 * t] = ra[d][t].x; af[2 * t + 1] = ra[d][t].y;
                bf[2 * t] = rb[d][t].x; bf[2 * t + 1] = rb[d][t].y;
                if constexpr (COLSUM) { bs[t].x += ra[d][t].x; bs[t].y += ra[d][t].y; }
            }
#pragma unroll
            for (int i = 0; i < 6; ++i)
#pragma unroll
                for (int j = 0; j < 6; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i], bf[j], acc[i][j], 0, 0, 0);
            issue(d);           // the group RD ahead, into the registers just consumed (past the range: zeros)
        }
    }
}

// bf16x3 form of the same loop (PREC = 1, see bf3_split_pair): a v_mfma_f32_16x16x32_bf16 operand is 8 contraction elements per
// lane -- here the lane's own row (kk) of 8 consecutive groups of 4 rows, i.e. a SUPER-GROUP of 32 token rows: slot j of the
// operand = group j, for A and B alike, so the values stay in the registers they were loaded into and only get cut into their
// bf16 pieces (2 x 24 pairs x 11 integer / fp32 ops per super-group beside 216 MFMAs of 18 cycles; the fp32 form: 288 of 32).
// A is split for all six tiles first and its next super-group is requested at once; B is split tile by tile between the MFMAs,
// and the rows of a column block t are requested again as soon as its two tiles are cut.
template <bool COLSUM>
__device__ __forceinline__ void tn_reg_loop_bf3(f32x4 (&acc)[6][6], float2 (&bs)[3], __amdgpu_buffer_rsrc_t rA, __amdgpu_buffer_rsrc_t rB,
                                                int va, int vb, unsigned sa, unsigned sb, unsigned step_a, unsigned step_b, int ngroups) {
    float2 ra[8][3], rb[8][3];
#pragma unroll
    for (int d = 0; d < 8; ++d)
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            ra[d][t] = __builtin_bit_cast(float2, __builtin_amdgcn_raw_buffer_load_b64(rA, va + t * 128, (int)(sa + d * step_a), 0));
            rb[d][t] = __builtin_bit_cast(float2, __builtin_amdgcn_raw_buffer_load_b64(rB, vb + t * 128, (int)(sb + d * step_b), 0));
        }
    sa += 8 * step_a; sb += 8 * step_b;
    for (int g = 0; g < ngroups; g += 8) {
        u32x4 ah[6], am[6], al[6];
#pragma unroll
        for (int t = 0; t < 3; ++t) {
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                unsigned h, m, l;
                bf3_split_pair(ra[2 * s][t].x, ra[2 * s + 1][t].x, h, m, l);
                ah[2 * t][s] = h; am[2 * t][s] = m; al[2 * t][s] = l;
                bf3_split_pair(ra[2 * s][t].y, ra[2 * s + 1][t].y, h, m, l);
                ah[2 * t + 1][s] = h; am[2 * t + 1][s] = m; al[2 * t + 1][s] = l;
            }
            if constexpr (COLSUM) {
#pragma unroll
                for (int d = 0; d < 8; ++d) { bs[t].x += ra[d][t].x; bs[t].y += ra[d][t].y; }
            }
        }
#pragma unroll
        for (int d = 0; d < 8; ++d)
#pragma unroll
            for (int t = 0; t < 3; ++t)
                ra[d][t] = __builtin_bit_cast(float2, __builtin_amdgcn_raw_buffer_load_b64(rA, va + t * 128, (int)(sa + d * step_a), 0));
        sa += 8 * step_a;
#pragma unroll
        for (int t = 0; t < 3; ++t) {
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                u32x4 bh, bm, bl;
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    unsigned h, m, l;
                    if (c == 0) bf3_split_pair(rb[2 * s][t].x, rb[2 * s + 1][t].x, h, m, l);
                    else bf3_split_pair(rb[2 * s][t].y, rb[2 * s + 1][t].y, h, m, l);
                    bh[s] = h; bm[s] = m; bl[s] = l;
                }
                const int j = 2 * t + c;
#pragma unroll
                for (int i = 0; i < 6; ++i) {
                    f32x4 v = acc[i][j];
                    v = bf3_mfma(al[i], bh, v);
                    v = bf3_mfma(ah[i], bl, v);
                    v = bf3_mfma(am[i], bm, v);
                    v = bf3_mfma(am[i], bh, v);
                    v = bf3_mfma(ah[i], bm, v);
                    v = bf3_mfma(ah[i], bh, v);
                    acc[i][j] = v;
                }
            }
#pragma unroll
            for (int d = 0; d < 8; ++d)
                rb[d][t] = __builtin_bit_cast(float2, __builtin_amdgcn_raw_buffer_load_b64(rB, vb + t * 128, (int)(sb + d * step_b), 0));
        }
        sb += 8 * step_b;
    }
}

#ifndef MIS_TN_REG_WAVES
#define MIS_TN_REG_WAVES 1
#endif
template <int PREC>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(MIS_TN_REG_WAVES, MIS_TN_REG_WAVES))) void gemm_tn_reg_kernel(const GemmArgs a) {
    float* const lds = mis_gemm_lds;                   // 2 x 36 KiB: the in-workgroup sum of the 4 waves' tiles
    const unsigned L = mis_xcd_remap(blockIdx.x, a.n_blocks_padded);
    if (L >= a.n_blocks) return;
    const unsigned tiles = (unsigned)(a.tiles_n * a.tiles_m);
    const int kz = L / tiles;                          // workgroup slice
    const unsigned tl = L - kz * tiles;
    const int tn = tl % a.tiles_n, tm = tl / a.tiles_n;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, kk = lane >> 4;
    const int m0 = tm * RT, n0 = tn * RT;
    // rows of this wave: a quarter (multiple of 4 rows) of the workgroup's slice
    const int kq = a.kchunk / 4;
    const int kbeg = kz * a.kchunk + wave * kq;
    int kend = kbeg + kq < a.K ? kbeg + kq : a.K;
    if (kend < kbeg) kend = kbeg;
    const int ngroups = (kend - kbeg + 3) / 4;
    // descriptors end at the wave's last row: the ring's loads past it (and the K tail inside a group) read zeros
    const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.A), 0, (int)((long long)kend * a.lda * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.B), 0, (int)((long long)kend * a.ldb * 4), 0x00020000);
    const int va = (int)(((long long)kk * a.lda + m0 + 2 * li) * 4), vb = (int)(((long long)kk * a.ldb + n0 + 2 * li) * 4);
    const unsigned sa = (unsigned)((long long)kbeg * a.lda * 4), sb = (unsigned)((long long)kbeg * a.ldb * 4);

    f32x4 acc[6][6];
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int j = 0; j < 6; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    float2 bs[3] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};
    const bool colsum = a.dbias != nullptr && tn == 0;          // workgroup-uniform
    if constexpr (PREC == 1) {
        if (colsum) tn_reg_loop_bf3<true>(acc, bs, rA, rB, va, vb, sa, sb, (unsigned)(a.lda * 16), (unsigned)(a.ldb * 16), ngroups);
        else tn_reg_loop_bf3<false>(acc, bs, rA, rB, va, vb, sa, sb, (unsigned)(a.lda * 16), (unsigned)(a.ldb * 16), ngroups);
    } else {
        if (colsum) tn_reg_loop<true>(acc, bs, rA, rB, va, vb, sa, sb, (unsigned)(a.lda * 16), (unsigned)(a.ldb * 16), ngroups);
        else tn_reg_loop<false>(acc, bs, rA, rB, va, vb, sa, sb, (unsigned)(a.lda * 16), (unsigned)(a.ldb * 16), ngroups);
    }

    // ---- sum of the 4 waves: (w2 -> w0, w3 -> w1), then w1 -> w0; fixed order ----
    f32x4* const l4 = reinterpret_cast<f32x4*>(lds);
    float* const lb = lds + 2 * 36 * 64 * 4;                    // bias sums: [2][64 lanes][6]
    if (wave >= 2) {
        f32x4* dst = l4 + (wave - 2) * (36 * 64) + lane;
#pragma unroll
        for (int i = 0; i < 6; ++i)
#pragma unroll
            for (int j = 0; j < 6; ++j) dst[(i * 6 + j) * 64] = acc[i][j];
        if (colsum) {
#pragma unroll
            for (int t = 0; t < 3; ++t) { lb[((wave - 2) * 64 + lane) * 6 + 2 * t] = bs[t].x; lb[((wave - 2) * 64 + lane) * 6 + 2 * t + 1] = bs[t].y; }
        }
    }
    __syncthreads();
    if (wave < 2) {
        const f32x4* src = l4 + wave * (36 * 64) + lane;
#pragma unroll
        for (int i = 0; i < 6; ++i)
#pragma unroll
            for (int j = 0; j < 6; ++j) acc[i][j] += src[(i * 6 + j) * 64];
        if (colsum) {
#pragma unroll
            for (int t = 0; t < 3; ++t) { bs[t].x += lb[(wave * 64 + lane) * 6 + 2 * t]; bs[t].y += lb[(wave * 64 + lane) * 6 + 2 * t + 1]; }
        }
    }
    __syncthreads();
    if (wave == 1) {
        f32x4* dst = l4 + lane;
#pragma unroll
        for (int i = 0; i < 6; ++i)
#pragma unroll
            for (int j = 0; j < 6; ++j) dst[(i * 6 + j) * 64] = acc[i][j];
        if (colsum) {
#pragma unroll
            for (int t = 0; t < 3; ++t) { lb[lane * 6 + 2 * t] = bs[t].x; lb[lane * 6 + 2 * t + 1] = bs[t].y; }
        }
    }
    __syncthreads();
    if (wave != 0) return;
    {
        const f32x4* src = l4 + lane;
#pragma unroll
        for (int i = 0; i < 6; ++i)
#pragma unroll
            for (int j = 0; j < 6; ++j) acc[i][j] += src[(i * 6 + j) * 64];
    }
    const bool direct = a.KS == 1;
    if (colsum) {
        // lanes (i, kk = 0..3) hold the sums of their own rows of every group: add the 4 row lanes (fixed order)
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            float x = bs[t].x + lb[lane * 6 + 2 * t], y = bs[t].y + lb[lane * 6 + 2 * t + 1];
            x = (x + __shfl_xor(x, 16, 64)) + (__shfl_xor(x, 32, 64) + __shfl_xor(x, 48, 64));
            y = (y + __shfl_xor(y, 16, 64)) + (__shfl_xor(y, 32, 64) + __shfl_xor(y, 48, 64));
            if (kk == 0) {
                const int m = m0 + 32 * t + 2 * li;
                if (direct) {
                    a.dbias[m] = a.dbias_acc ? a.dbias[m] + x : x;
                    a.dbias[m + 1] = a.dbias_acc ? a.dbias[m + 1] + y : y;
                } else {
                    a.wsb[(long long)kz * a.M + m] = x;
                    a.wsb[(long long)kz * a.M + m + 1] = y;
                }
            }
        }
    }
    // accumulator tile (i, j), register r, lane (kk, li): row 4 kk + r of the tile = m0 + 32 (i / 2) + 2 (4 kk + r) + i % 2,
    // column li = n0 + 32 (j / 2) + 2 li + j % 2: tiles (i, 2 q) and (i, 2 q + 1) make a float2 per lane, 128 bytes per 16 lanes
    float* __restrict__ out = direct ? a.C : a.ws + (long long)kz * a.M * a.N;
    const long long ldo = direct ? a.ldc : a.N;
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = m0 + 32 * (i / 2) + 2 * (4 * kk + r) + (i % 2);
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const int n = n0 + 32 * q + 2 * li;
                float2 v = make_float2(acc[i][2 * q][r], acc[i][2 * q + 1][r]);
                float2* p = reinterpret_cast<float2*>(out + (long long)m * ldo + n);
                if (direct) {
                    if (a.bias) { v.x += a.bias[n]; v.y += a.bias[n + 1]; }
                    if (a.accumulate) { const float2 o = *p; v.x += o.x; v.y += o.y; }
                }
                *p = v;
            }
        }
}

// the register-only TN form serves this shape (widths multiples of 96, 8-byte aligned float2 rows)
bool tn_reg_ok(const float* A, long long lda, const float* B, long long ldb, const float* C, long long ldc, int M, int N, int K) {
    static const bool on = [] { const char* e = getenv("MIS_GEMM_TN_REG"); return !(e && e[0] == '0'); }();
    return on && M % RT == 0 && N % RT == 0 && lda % 2 == 0 && ldb % 2 == 0 && ldc % 2 == 0 &&
           !((uintptr_t)A & 7) && !((uintptr_t)B & 7) && !((uintptr_t)C & 7) && K >= 64;
}
// workgroup slices: one resident workgroup per CU (a wave = 144 accumulator + 48 ring registers); a wave wants >= 16 groups
int tn_reg_slices(int M, int N, int K) {
    const long long tiles = (long long)(M / RT) * (N / RT);
    long long ks = (256 * MIS_TN_REG_WAVES) / tiles;
    const long long kmax = K / (4 * 64);
    if (ks > kmax) ks = kmax;
    if (ks < 1) ks = 1;
    return (int)ks;
}
// rows per workgroup slice: a multiple of 16 (4 waves x groups of 4 rows)
void tn_reg_plan(GemmArgs& a) {
    a.KS = tn_reg_slices(a.M, a.N, a.K);
    a.kchunk = (int)(mis_cdiv(mis_cdiv(a.K, a.KS), 16) * 16);
    a.KS = (int)mis_cdiv(a.K, a.kchunk);
    a.tiles_n = a.N / RT; a.tiles_m = a.M / RT;
    const long long nb = (long long)a.tiles_n * a.tiles_m * a.KS;
    a.n_blocks = (unsigned)nb;
    a.n_blocks_padded = (unsigned)(mis_cdiv(nb, MIS_NUM_XCD) * MIS_NUM_XCD);
}
int launch_tn_reg(GemmArgs& a, hipStream_t stream) {
    constexpr int LDSB = 2 * 36 * 64 * 16 + 2 * 64 * 6 * 4;
    static std::atomic<unsigned long long> attr_done{0};
    if (gemm_bf3_tn()) {
        static std::atomic<unsigned long long> attr_done1{0};
        if (mis_set_lds_attr(reinterpret_cast<const void*>(&gemm_tn_reg_kernel<1>), LDSB, attr_done1) != MIS_OK) return MIS_ERR_LAUNCH;
        hipLaunchKernelGGL(gemm_tn_reg_kernel<1>, dim3(a.n_blocks_padded), dim3(256), LDSB, stream, a);
        return mis_launch_status();
    }
    if (mis_set_lds_attr(reinterpret_cast<const void*>(&gemm_tn_reg_kernel<0>), LDSB, attr_done) != MIS_OK) return MIS_ERR_LAUNCH;
    hipLaunchKernelGGL(gemm_tn_reg_kernel<0>, dim3(a.n_blocks_padded), dim3(256), LDSB, stream, a);
    return mis_launch_status();
}

// C[m][n] (+)= bias[n] + sum_k ws[k][m][n]   (fixed order)
// block = 32 consecutive elements x 8 slice lanes: lane kl sums slices kl, kl+8, ... (four loads in flight), then a
// fixed-order LDS tree over the 8 lanes -- the reduction is latency-bound (dW of a 96 x 288 Linear: 64+ slices of
// 110 KB), so parallelism over the slices matters more than bytes
__global__ __launch_bounds__(256) void gemm_reduce_kernel(const GemmArgs a) {
    __shared__ float red[256];
    const long long total = (long long)a.M * a.N;
    const int el = threadIdx.x & 31, kl = threadIdx.x >> 5;
    const long long main_blocks = (total + 31) / 32;
    if (blockIdx.x >= main_blocks) {      // mis_gemm_dw: the per-slice column sums of A, same lane pattern
        const long long m = (blockIdx.x - main_blocks) * 32LL + el;
        float s = 0.f;
        if (m < a.M)
            for (int k = kl; k < a.KS; k += 8) s += a.wsb[(long long)k * a.M + m];
        red[threadIdx.x] = s;
        __syncthreads();
        if (kl == 0 && m < a.M) {
#pragma unroll
            for (int j = 1; j < 8; ++j) s += red[j * 32 + el];
            a.dbias[m] = a.dbias_acc ? a.dbias[m] + s : s;
        }
        return;
    }
    const long long e = blockIdx.x * 32LL + el;
    float s = 0.f;
    if (e < total) {
        const float* __restrict__ w = a.ws + e;
        int k = kl;
        for (; k + 24 < a.KS; k += 32) {
            const float v0 = w[(long long)k * total], v1 = w[(long long)(k + 8) * total];
            const float v2 = w[(long long)(k + 16) * total], v3 = w[(long long)(k + 24) * total];
            s += v0; s += v1; s += v2; s += v3;
        }
        for (; k < a.KS; k += 8) s += w[(long long)k * total];
    }
    red[threadIdx.x] = s;
    __syncthreads();
    if (kl == 0 && e < total) {
#pragma unroll
        for (int j = 1; j < 8; ++j) s += red[j * 32 + el];
        const int m = (int)(e / a.N), n = (int)(e - (long long)m * a.N);
        if (a.bias) s += a.bias[n];
        float* p = a.C + (long long)m * a.ldc + n;
        if (a.ep == EP_GELU_FWD) {
            a.C2[(long long)m * a.ldc2 + n] = gelu_f(s);
            if (!a.C) return;
        } else if (a.ep == EP_GELU_BWD) {
            s *= gelu_grad_f(a.E1[(long long)m * a.lde1 + n]);
        } else if (a.ep == EP_RESIDUAL) {
            s = a.E1[(long long)m * a.lde1 + n] + (a.rowscale ? a.rowscale[m / a.rps] : 1.f) * s;
        }
        *p = a.accumulate ? *p + s : s;
    }
}

// tile edge of the TN form / tile width of the NT form: 96 when it removes padding
int tn_tile(int M, int N) { return (M % 96 == 0 && N % 96 == 0 && (M % 128 != 0 || N % 128 != 0)) ? 96 : 128; }
int nt_tile_n(int N) { return (N % 96 == 0 && N % 128 != 0) ? 96 : 128; }
// pre-split B: a 64 x 96 tile keeps three workgroups per CU (52 KB of LDS; 64 x 128 needs 64 KB: two)
int nt_tile_n_b3(int N) {       // (MIS_GEMM_B3_RULE=0 sweeps only)
    static const int pref = getenv("MIS_GEMM_B3_BN") ? atoi(getenv("MIS_GEMM_B3_BN")) : 96;
    if (pref == 96 && N % 96 == 0) return 96;
    return nt_tile_n(N);
}

// NT form: rows per tile (128 / 64) and K slices.  Split-K only when the tiles cannot fill the chip (deep SwinUnet stages,
// the ViT of UNETR: 60 .. 114 tiles of 128 rows) and K is long: every slice costs a write + read of the M x N partial
// and the reduction launch (~20 us).  One workgroup per CU is the unit: a second resident workgroup halves each one's
// speed, so slices beyond 256 workgroups buy nothing and a partial second round doubles the time (measured: 5 slices of
// 114 tiles 158 us).  64-row tiles double the tile count instead: UNETR's M = 1728, N = K = 768 Linears are 84 tiles x 3
// slices + reduction (49 + 20 us) or 162 tiles unsplit -- the cost model below (workgroup time ~ rows x K / slices, +
// a fixed share per round, + the reduction) picks between them.
void nt_choice(int M, int N, int K, int& bm, int& ks, int rows_arg = 0, int bn_arg = 0) {
    static const int force_env = getenv("MIS_GEMM_BM") ? atoi(getenv("MIS_GEMM_BM")) : 0;
    const int force = rows_arg ? rows_arg : force_env;
    const long long tn = mis_cdiv(N, bn_arg ? bn_arg : nt_tile_n(N));
    if (force != 128 && mis_cdiv(M, 128) * tn >= 512) {      // plenty of tiles: 64 rows = three resident workgroups per CU
        bm = 64; ks = 1;                                      // instead of two (SwinUnet 36.2 -> 35.6 ms per step)
        return;
    }
    long long best = -1;
    for (int rows = 128; rows >= 64; rows -= 64) {
        if ((force == 64 || force == 128) && rows != force) continue;
        const long long tiles = mis_cdiv(M, rows) * tn;
        long long s = 256 / tiles;
        const long long kmax = K / (8 * BK);      // at least 8 k-steps per slice
        if (s > kmax) s = kmax;
        if (s < 2) s = 1;
        const long long rounds = mis_cdiv(tiles * s, 256);
        // units: one row of a tile over one k; 128 rows x 768 k = 40 us measured -> 2458 units / us.  bf16x3 products run the
        // k-loop ~2x faster (MIS_GEMM_NT_BF3_SCALE percent of the fp32 k-loop cost), the fixed costs stay: fewer slices pay
        static const int bf3_pct = getenv("MIS_GEMM_NT_BF3_SCALE") ? atoi(getenv("MIS_GEMM_NT_BF3_SCALE")) : 50;
        const long long kloop = (long long)rows * mis_cdiv(K, s) * (gemm_bf3() ? bf3_pct : 100) / 100;
        long long cost = rounds * (kloop + 128 * 96);
        if (s > 1) cost += 49152 + (long long)(0.0039 * (double)s * M * N);
        if (rows == 128) cost += cost / 12;      // near-ties go to 64 rows (per-shape A/B, scripts/gemm_shapes_ab.py)
        if (best < 0 || cost < best) { best = cost; bm = rows; ks = (int)s; }
    }
}

int nt_tile_m(int M, int N, int K);

int pick_ks(int M, int N, int K, int trans) {
    if (!trans) {
        if (nt_tile_m(M, N, K) == 64 && K <= 192) return 1;      // the short contractions: never split
        int bm, ks;
        nt_choice(M, N, K, bm, ks);
        return ks;
    }
    const int bt = tn_tile(M, N);
    const long long tiles = mis_cdiv(M, bt) * mis_cdiv(N, bt);
    if (tiles >= 256) return 1;
    // slices x tiles ~ three workgroups per CU for the short contractions (deep Swin stages, UNETR's 1 728 token rows); with
    // the register prefetch of the k-loop a long contraction (>= 32 768 rows) runs better on half as many, longer slices
    // (less partial traffic): SwinUnet 28.9 -> 28.5 ms per step, UNETR keeps 43.5 (44.5 with 384 everywhere)
    static const int forced = getenv("MIS_GEMM_TN_SLOTS") ? atoi(getenv("MIS_GEMM_TN_SLOTS")) : 0;
    const int slots = forced ? forced : (K >= 32768 ? 384 : 768);
    long long ks = slots / tiles;
    const long long kmax = mis_cdiv(K, 4 * BK);   // at least 4 k-steps per slice
    if (ks > kmax) ks = kmax;
    if (ks < 1) ks = 1;
    if (ks > 256) ks = 256;
    return (int)ks;
}

bool a16(const void* p) { return ((uintptr_t)p & 15) == 0; }

// (Round 4 also tried the register-only form for the NT GEMMs -- forward and dX: lanes load float4 A[m0 + 16 t + i][k0 + 4 kk ..]
// straight into the MFMA layout, 64 x 96 wave tiles, two waves per SIMD, epilogues on registers.  Correct, and SLOWER than the
// staged kernels on every SwinUnet shape: 2277 us against 1750 us for the 20 Linear shapes of a step, 0.26-0.47 of the pipe on the
// K = 96 / 192 layers.  Row-major operands put the 16 lanes of a load group on 16 different rows -- 64 separate 16-byte accesses
// per instruction for the texture-address unit, ~5000 of its cycles per 6144-cycle k-step of a CU's 8 waves; the TN form above
// works because ITS operands are contraction-major and a lane group reads 128 contiguous bytes.  The NT GEMMs need the transpose
// that the LDS stage provides.  Removed; scripts/gemm_nt_bench.py is the measurement.)

template <int BMT, int BN, int EP, int PREC, int NW = 4>
int launch_nt_waves(const GemmArgs& a, hipStream_t stream) {
    static std::atomic<unsigned long long> attr_done{0};   // per instantiation, one bit per device
    using G = NtCfg<BMT, BN, PREC, NW>;
    if (mis_set_lds_attr(reinterpret_cast<const void*>(&gemm_nt_kernel<BMT, BN, EP, PREC, NW>), G::LDS_BYTES, attr_done) != MIS_OK)
        return MIS_ERR_LAUNCH;
    hipLaunchKernelGGL((gemm_nt_kernel<BMT, BN, EP, PREC, NW>), dim3(a.n_blocks_padded), dim3(NW * 64), G::LDS_BYTES, stream, a);
    return mis_launch_status();
}

// MIS_GEMM_NT_WAVES=8: eight waves per workgroup for the 64 x 96 tiles of the pre-split form.  Measured (profiles/r06_gemm_nt_waves.txt):
// alone, most shapes gain 3 .. 10 % (six waves per SIMD hide the k-step's DMA wait and barrier), the split-K shapes lose 4 .. 11 %,
// 1305 -> 1293 us over the 20 shapes of a step; INSIDE the step, where the weight-gradient and teacher streams already fill the idle
// issue slots, it is 0.1 .. 0.15 ms SLOWER (SwinUnet 18.69 -> 18.80 ms, cross teaching 18.95 -> 19.12).  Default: four.
inline bool nt_eight_waves() {
    static const int w = getenv("MIS_GEMM_NT_WAVES") ? atoi(getenv("MIS_GEMM_NT_WAVES")) : 4;
    return w == 8;
}

template <int BMT, int BN, int EP, int PREC>
int launch_nt_prec(const GemmArgs& a, hipStream_t stream) {
    if constexpr (BMT == 64 && BN == 96 && PREC == 2 && EP != EP_LNHEAD) {
        if (nt_eight_waves()) return launch_nt_waves<BMT, BN, EP, PREC, 8>(a, stream);
    }
    return launch_nt_waves<BMT, BN, EP, PREC, 4>(a, stream);
}

template <int BMT, int BN, int EP>
int launch_nt_ep(const GemmArgs& a, hipStream_t stream) {
    if (a.B3) return launch_nt_prec<BMT, BN, EP, 2>(a, stream);
    return gemm_bf3() ? launch_nt_prec<BMT, BN, EP, 1>(a, stream) : launch_nt_prec<BMT, BN, EP, 0>(a, stream);
}

// split-K slices write raw partials (the epilogue runs in gemm_reduce_kernel): always the plain instantiation
template <int BMT, int BN>
int launch_nt_bm(const GemmArgs& a, hipStream_t stream) {
    if (a.KS > 1 || a.ep == EP_NONE) return launch_nt_ep<BMT, BN, EP_NONE>(a, stream);
    if (a.ep == EP_GELU_FWD) return launch_nt_ep<BMT, BN, EP_GELU_FWD>(a, stream);
    if (a.ep == EP_GELU_BWD) return launch_nt_ep<BMT, BN, EP_GELU_BWD>(a, stream);
    return launch_nt_ep<BMT, BN, EP_RESIDUAL>(a, stream);
}

// rows per tile of the NT form: 64 for the short contractions of the first stages (K = 96 / 192 over 10^4 .. 10^5 token
// rows), where a tile is a prologue, 3 .. 6 k-steps and an epilogue: three resident workgroups per CU overlap them
int nt_tile_m(int M, int N, int K) {
    if (K <= 192 && mis_cdiv(M, 64) * mis_cdiv(N, nt_tile_n(N)) >= 1536) return 64;
    int bm, ks;
    nt_choice(M, N, K, bm, ks);
    return bm;
}

// Tile shape and k slices of the pre-split form, from per-shape sweeps of the SwinUnet Linears at 16, 24, 32 (256 x 256) and 48
// images (scripts/gemm_nt_bench.py --split --batch B with MIS_GEMM_B3_RULE=0 MIS_GEMM_BM / MIS_GEMM_B3_BN; 20 shapes each).
// K <= 96 stays with the short-contraction kernel: a 64-row tile re-fetches its B panel for three k-steps of work and the
// planes are 1.5x the bytes (118 -> 133 us at 150528 x 384 x 96).  Otherwise 64 x 96 tiles (52 KB of LDS: three workgroups
// per CU; best or within noise of the best for 60 of the 68 shapes), except wide N with many tiles (N >= 1536 a multiple of
// 128, >= 1000 tiles of 64 x 96): 128 x 128, which halves the B-panel traffic per output and the re-reads of A (2352 x 3072
// x 768: 79 -> 63 us; 9408 x 1536 x 384: 74 -> 69).  Sums of the 20 shapes, fp32-B kernels -> this rule: 48 images 1432 ->
// 1307 us, 32: 1352 -> 1184, 24: 897 -> 832, 16: 708 -> 653.  MIS_GEMM_B3_RULE=0: the tile choice of the fp32-B path.
bool b3_choice(int M, int N, int K, int& bm, int& bn, int& ks) {
    static const bool rule = !(getenv("MIS_GEMM_B3_RULE") && getenv("MIS_GEMM_B3_RULE")[0] == '0');
    static const int kmin = getenv("MIS_GEMM_B3_KMIN") ? atoi(getenv("MIS_GEMM_B3_KMIN")) : 97;
    if (K < kmin) return false;
    if (!rule) {
        bn = nt_tile_n_b3(N);
        bm = nt_tile_m(M, N, K);
        ks = pick_ks(M, N, K, 0);
        return true;
    }
    const bool wide = N % 128 == 0 && N >= 1536 && mis_cdiv(M, 64) * mis_cdiv(N, 96) >= 1000;
    bn = wide ? 128 : (N % 96 == 0 ? 96 : nt_tile_n(N));
    nt_choice(M, N, K, bm, ks, wide ? 128 : 64, bn);
    return true;
}

// the short-contraction kernel serves: float4 epilogue, no split-K, no pixel-shuffle store, K <= 192, many tiles
bool nt_short(const GemmArgs& a) {
    static const int force = getenv("MIS_GEMM_SHORT") ? atoi(getenv("MIS_GEMM_SHORT")) : -1;
    if (force == 0 || a.KS > 1 || !a.vec4 || a.ex_P || a.K % 4 || a.B3) return false;
    // bf16x3: the K = 32 products of the general kernel win from K = 192 on (72 vs 75 us at 37632 x 576 x 192); at K = 96 the
    // short kernel's residency still pays (150528 x 384 x 96: 120 vs 132 us, x 288: 93 vs 92, scripts/gemm_nt_bench.py)
    if (gemm_bf3() && force != 1 && a.K > 96) return false;
    if (force == 1) return true;
    return a.K <= 192 && mis_cdiv(a.M, 64) * mis_cdiv(a.N, nt_tile_n(a.N)) >= 1536;
}

template <int BN, int EP>
int launch_nt_short_ep(const GemmArgs& a, hipStream_t stream) {
    using G = NsCfg<BN>;
    static std::atomic<unsigned long long> attr_done{0};
    if (gemm_bf3()) {
        static std::atomic<unsigned long long> attr_done1{0};
        if (mis_set_lds_attr(reinterpret_cast<const void*>(&gemm_nt_short_kernel<BN, EP, 1>), G::LDS_BYTES, attr_done1) != MIS_OK)
            return MIS_ERR_LAUNCH;
        hipLaunchKernelGGL((gemm_nt_short_kernel<BN, EP, 1>), dim3(a.n_blocks_padded), dim3(256), G::LDS_BYTES, stream, a);
        return mis_launch_status();
    }
    if (mis_set_lds_attr(reinterpret_cast<const void*>(&gemm_nt_short_kernel<BN, EP, 0>), G::LDS_BYTES, attr_done) != MIS_OK)
        return MIS_ERR_LAUNCH;
    hipLaunchKernelGGL((gemm_nt_short_kernel<BN, EP, 0>), dim3(a.n_blocks_padded), dim3(256), G::LDS_BYTES, stream, a);
    return mis_launch_status();
}

template <int BN>
int launch_nt_short(const GemmArgs& a, hipStream_t stream) {
    if (a.ep == EP_NONE) return launch_nt_short_ep<BN, EP_NONE>(a, stream);
    if (a.ep == EP_GELU_FWD) return launch_nt_short_ep<BN, EP_GELU_FWD>(a, stream);
    if (a.ep == EP_GELU_BWD) return launch_nt_short_ep<BN, EP_GELU_BWD>(a, stream);
    return launch_nt_short_ep<BN, EP_RESIDUAL>(a, stream);
}

// fills tiles_m / n_blocks for the NT form and launches
template <int BN>
int launch_nt(GemmArgs& a, hipStream_t stream) {
    if (nt_short(a)) {
        a.tiles_m = (int)mis_cdiv(a.M, 64);
        const long long nbs = (long long)a.tiles_n * a.tiles_m;
        if (nbs > 0x7fffffffLL) return MIS_ERR_UNSUPPORTED;
        a.n_blocks = (unsigned)nbs;
        a.n_blocks_padded = (unsigned)(mis_cdiv(nbs, MIS_NUM_XCD) * MIS_NUM_XCD);
        return launch_nt_short<BN>(a, stream);
    }
    const int bm = a.bm_force ? a.bm_force : nt_tile_m(a.M, a.N, a.K);
    a.tiles_m = (int)mis_cdiv(a.M, bm);
    const long long nb = (long long)a.tiles_n * a.tiles_m * a.KS;
    if (nb > 0x7fffffffLL) return MIS_ERR_UNSUPPORTED;
    a.n_blocks = (unsigned)nb;
    a.n_blocks_padded = (unsigned)(mis_cdiv(nb, MIS_NUM_XCD) * MIS_NUM_XCD);
    return bm == 64 ? launch_nt_bm<64, BN>(a, stream) : launch_nt_bm<BM, BN>(a, stream);
}

// the register-A kernel serves: bf16x3 products, pre-split B in the natural element order, N a multiple of 96, the float4
// epilogue, no split-K, no pixel-shuffle store, enough 128-row tiles for three workgroups per CU (MIS_GEMM_REGA=0: never)
bool nt_rega_shape(int M, int N, int K) {
    static const int on = getenv("MIS_GEMM_REGA") ? atoi(getenv("MIS_GEMM_REGA")) : 1;
    static const int kmin = getenv("MIS_GEMM_REGA_KMIN") ? atoi(getenv("MIS_GEMM_REGA_KMIN")) : 96;
    static const int tmin = getenv("MIS_GEMM_REGA_TILES") ? atoi(getenv("MIS_GEMM_REGA_TILES")) : 512;
    static const int mmin = getenv("MIS_GEMM_REGA_MMIN") ? atoi(getenv("MIS_GEMM_REGA_MMIN")) : 30000;
    if (!on || !gemm_bf3() || N % 96 || N % 4 || K % 4 || K < kmin) return false;
    if ((long long)M * 4 >= (1LL << 31)) return false;
    // per-shape sweep at 48 images (scripts/gemm_nt_bench.py --rega, profiles/r06_gemm_nt_rega.txt): ahead of the staged kernels
    // for the 10^5-row stage and, at 4 x 10^4 rows, where the output is wider than the contraction; behind them below that
    static const int msq = getenv("MIS_GEMM_REGA_MSQ") ? atoi(getenv("MIS_GEMM_REGA_MSQ")) : 100000;
    // N == 96 (proj / fc2 of a 96-channel block): the register-A kernels carry the residual add AND the next LayerNorm in their
    // epilogue (mis_gemm_nt_residual_ln) -- a LayerNorm pass less outweighs the few us the staged kernel is ahead by at 7 x 10^4 rows
    static const int n96 = getenv("MIS_GEMM_REGA_N96") ? atoi(getenv("MIS_GEMM_REGA_N96")) : 1;
    if (M < mmin || (M < msq && N <= K && !(n96 && N == 96))) return false;
    return mis_cdiv(M, 128) * (N / 96) >= tmin;
}

template <int BN, int EP>
int launch_nt_rega_ep(const GemmArgs& a, hipStream_t stream) {
    static std::atomic<unsigned long long> attr_done{0};
    constexpr int LDSB = rega_lds_bytes<BN>();
    if (mis_set_lds_attr(reinterpret_cast<const void*>(&gemm_nt_rega_kernel<BN, EP>), LDSB, attr_done) != MIS_OK) return MIS_ERR_LAUNCH;
    hipLaunchKernelGGL((gemm_nt_rega_kernel<BN, EP>), dim3((unsigned)a.KS), dim3(256), LDSB, stream, a);
    return mis_launch_status();
}

template <int EP>
int launch_nt_rega_res_ep(const GemmArgs& a, hipStream_t stream) {
    static std::atomic<unsigned long long> attr_done{0};
    const int ldsb = 3 * (96 * BK * 3 / 2) * 4 + (192 + 384) * 4;      // the panel (up to three k-blocks) + gamma / beta / head weights
    if (mis_set_lds_attr(reinterpret_cast<const void*>(&gemm_nt_rega_res_kernel<EP>), ldsb, attr_done) != MIS_OK)
        return MIS_ERR_LAUNCH;
    hipLaunchKernelGGL((gemm_nt_rega_res_kernel<EP>), dim3(a.n_blocks_padded), dim3(256), ldsb, stream, a);
    return mis_launch_status();
}

int launch_nt_rega(GemmArgs& a, hipStream_t stream) {
    static const int res_on = getenv("MIS_GEMM_REGA_RES") ? atoi(getenv("MIS_GEMM_REGA_RES")) : 1;
    if (res_on && a.K3 <= 96) {
        // persistent: two workgroups per CU, split evenly over the column panels; a walker's four waves want >= 2 slabs each
        a.tiles_n = a.N / 96;
        long long walkers = 512 / a.tiles_n;
        const long long slabs = mis_cdiv(a.M, 32);
        if (walkers > mis_cdiv(slabs, 8)) walkers = mis_cdiv(slabs, 8);
        if (walkers < 1) walkers = 1;
        a.tiles_m = (int)walkers;
        a.n_blocks = (unsigned)(walkers * a.tiles_n);
        a.n_blocks_padded = (unsigned)(mis_cdiv((long long)a.n_blocks, MIS_NUM_XCD) * MIS_NUM_XCD);
        a.KS = 1;
        if (a.ep == EP_NONE) return launch_nt_rega_res_ep<EP_NONE>(a, stream);
        if (a.ep == EP_GELU_FWD) return launch_nt_rega_res_ep<EP_GELU_FWD>(a, stream);
        if (a.ep == EP_GELU_BWD) return launch_nt_rega_res_ep<EP_GELU_BWD>(a, stream);
        if (a.ep == EP_RESIDUAL_LN) return launch_nt_rega_res_ep<EP_RESIDUAL_LN>(a, stream);
        if (a.ep == EP_LNHEAD) return launch_nt_rega_res_ep<EP_LNHEAD>(a, stream);
        return launch_nt_rega_res_ep<EP_RESIDUAL>(a, stream);
    }
    if (a.ep == EP_LNHEAD) return MIS_ERR_UNSUPPORTED;
    a.tiles_n = a.N / 96;
    a.tiles_m = (int)mis_cdiv(a.M, 128);
    const long long nb = (long long)a.tiles_n * a.tiles_m;
    if (nb > 0x7fffffffLL) return MIS_ERR_UNSUPPORTED;
    a.n_blocks = (unsigned)nb;
    a.n_blocks_padded = (unsigned)(mis_cdiv(nb, MIS_NUM_XCD) * MIS_NUM_XCD);
    a.KS = 1;
    // persistent above three workgroups per CU (KS carries the grid size; MIS_GEMM_REGA_WALK=0: one workgroup per tile)
    static const int walk = getenv("MIS_GEMM_REGA_WALK") ? atoi(getenv("MIS_GEMM_REGA_WALK")) : 768;
    a.KS = (walk > 0 && a.n_blocks_padded > (unsigned)walk) ? walk / MIS_NUM_XCD * MIS_NUM_XCD : (int)a.n_blocks_padded;
    if (a.ep == EP_NONE) return launch_nt_rega_ep<96, EP_NONE>(a, stream);
    if (a.ep == EP_GELU_FWD) return launch_nt_rega_ep<96, EP_GELU_FWD>(a, stream);
    if (a.ep == EP_GELU_BWD) return launch_nt_rega_ep<96, EP_GELU_BWD>(a, stream);
    if (a.ep == EP_RESIDUAL_LN) return launch_nt_rega_ep<96, EP_RESIDUAL_LN>(a, stream);
    return launch_nt_rega_ep<96, EP_RESIDUAL>(a, stream);
}

}  // namespace

int mis_split_precision_mask() { return gemm_bf3_state().load(std::memory_order_relaxed); }

// Arithmetic of the nn.Linear GEMMs: mask bit 0 = forward / dX (NT), bit 1 = dW (TN, widths % 96 == 0) as bf16x3 split
// products on v_mfma_f32_16x16x32_bf16 (exact 3-way split of the fp32 operands, six piece products, fp32 accumulation: error
// against float64 no larger than the fp32 MFMA kernels'); 0 = v_mfma_f32_16x16x4_f32 everywhere.  Returns the previous mask;
// mask < 0 only queries.
extern "C" int mis_gemm_set_split_precision(int mask) {
    const int prev = gemm_bf3_state().load();
    if (mask >= 0) gemm_bf3_state().store(mask & 7);
    return prev;
}

// NT kernel instantiation mis_gemm / mis_gemm_ex run this shape with (16-byte aligned operands, N % 4 == 0), as
// rocprofv3 prints it minus the anonymous-namespace prefix: for bench.py's attribution
extern "C" int mis_gemm_nt_kernel_name(int M, int N, int K, int epilogue, char* name, int name_len) {
    if (M <= 0 || N <= 0 || K <= 0 || !name || name_len <= 0) return MIS_ERR_ARG;
    GemmArgs a{};
    a.M = M; a.N = N; a.K = K; a.vec4 = N % 4 == 0; a.KS = pick_ks(M, N, K, 0);
    const int bn = nt_tile_n(N);
    if (nt_short(a)) snprintf(name, name_len, "gemm_nt_short_kernel<%d, %d, %d>", bn, epilogue, gemm_bf3() ? 1 : 0);
    else snprintf(name, name_len, "gemm_nt_kernel<%d, %d, %d, %d, 4>", nt_tile_m(M, N, K), bn, a.KS > 1 ? 0 : epilogue, gemm_bf3() ? 1 : 0);
    return MIS_OK;
}

// ... and the instantiation mis_gemm_nt_split runs it with
extern "C" int mis_gemm_nt_split_kernel_name(int M, int N, int K, int epilogue, char* name, int name_len) {
    if (M <= 0 || N <= 0 || K <= 0 || !name || name_len <= 0) return MIS_ERR_ARG;
    int bm, bn, ks;
    if (!b3_choice(M, N, K, bm, bn, ks)) return MIS_ERR_UNSUPPORTED;
    const int ep = ks > 1 ? 0 : epilogue;
    snprintf(name, name_len, "gemm_nt_kernel<%d, %d, %d, 2, %d>", bm, bn, ep, (bm == 64 && bn == 96 && ep != EP_LNHEAD && nt_eight_waves()) ? 8 : 4);
    return MIS_OK;
}
static int split_k3(int K);
// ... on natural-order planes (layout 1): the register-A kernel
extern "C" int mis_gemm_nt_split_layout_kernel_name(int M, int N, int K, int epilogue, int layout, char* name, int name_len) {
    if (!layout) return mis_gemm_nt_split_kernel_name(M, N, K, epilogue, name, name_len);
    if (M <= 0 || N <= 0 || K <= 0 || !name || name_len <= 0) return MIS_ERR_ARG;
    if (!nt_rega_shape(M, N, K)) return MIS_ERR_UNSUPPORTED;
    static const int res_on = getenv("MIS_GEMM_REGA_RES") ? atoi(getenv("MIS_GEMM_REGA_RES")) : 1;
    if (res_on && split_k3(K) <= 96) snprintf(name, name_len, "gemm_nt_rega_res_kernel<%d>", epilogue);
    else snprintf(name, name_len, "gemm_nt_rega_kernel<96, %d>", epilogue);
    return MIS_OK;
}

// TN kernel mis_gemm(trans = 1) / mis_gemm_dw run this shape with (operands as the callers pass them: row strides in
// floats, base pointers), as rocprofv3 prints it minus the anonymous-namespace prefix: for bench.py's attribution
extern "C" int mis_gemm_tn_kernel_name(const float* A, long long lda, const float* B, long long ldb, const float* C,
                                       long long ldc, int M, int N, int K, char* name, int name_len) {
    if (M <= 0 || N <= 0 || K <= 0 || !name || name_len <= 0) return MIS_ERR_ARG;
    if (tn_reg_ok(A, lda, B, ldb, C, ldc, M, N, K) && (long long)K * lda * 4 < (1LL << 31) && (long long)K * ldb * 4 < (1LL << 31))
        snprintf(name, name_len, "gemm_tn_reg_kernel<%d>", gemm_bf3_tn() ? 1 : 0);
    else snprintf(name, name_len, "gemm_tn_kernel<%d>", tn_tile(M, N));
    return MIS_OK;
}

extern "C" long long mis_gemm_workspace_bytes(int M, int N, int K, int trans) {
    if (M <= 0 || N <= 0 || K <= 0) return MIS_ERR_ARG;
    int ks = pick_ks(M, N, K, trans);
    if (trans && M % RT == 0 && N % RT == 0) { const int kr = tn_reg_slices(M, N, K); if (kr > ks) ks = kr; }   // either TN form
    return ks > 1 ? (long long)ks * M * N * 4 : 0;
}

// mis_gemm (NT form only) with a fused epilogue, v = A.B^T + bias:
//   epilogue 1  C = v, C2 = gelu(v)                      Mlp.fc1 + GELU (reference ...sys.py:14-15,20-21): the
//                                                        pre-activation stays for the backward, the GELU pass is gone;
//                                                        C NULL: only C2 (a forward nobody differentiates)
//   epilogue 2  C = v * gelu'(E1)                        dX of Mlp.fc2 straight into the gradient of fc1's output
//   epilogue 3  C = E1 + rowscale[m / rows_per_scale] * v  proj / fc2 + DropPath + residual add (:276, :281);
//                                                        rowscale NULL = 1
// E1 / C2 are [M][N] views with their own row strides.  accumulate is not supported here.
extern "C" int mis_gemm_ex(const float* A, long long lda, const float* B, long long ldb, float* C, long long ldc,
                           const float* bias, int M, int N, int K, int epilogue, const float* E1, long long lde1,
                           float* C2, long long ldc2, const float* rowscale, long long rows_per_scale,
                           float* workspace, long long workspace_bytes, hipStream_t stream) {
    if (!A || !B || (!C && epilogue != EP_GELU_FWD) || M <= 0 || N <= 0 || K <= 0) return MIS_ERR_ARG;      // epilogue 1: C may be NULL
    if (epilogue < EP_GELU_FWD || epilogue > EP_RESIDUAL) return MIS_ERR_ARG;
    if (epilogue == EP_GELU_FWD ? (!C2 || ldc2 < N) : (!E1 || lde1 < N)) return MIS_ERR_ARG;
    if (epilogue == EP_RESIDUAL && rowscale && rows_per_scale <= 0) return MIS_ERR_ARG;
    if (!a16(A) || !a16(B) || lda % 4 || ldb % 4 || K % 4) return MIS_ERR_UNSUPPORTED;
    // the fused epilogues live in the float4 (LDS-staged) epilogue
    if (N % 4 || ldc % 4 || !a16(C) || (bias && !a16(bias))) return MIS_ERR_UNSUPPORTED;
    if (epilogue == EP_GELU_FWD ? (ldc2 % 4 || !a16(C2)) : (lde1 % 4 || !a16(E1))) return MIS_ERR_UNSUPPORTED;
    if ((long long)M * lda * 4 >= (1LL << 31) || (long long)N * ldb * 4 >= (1LL << 31)) return MIS_ERR_UNSUPPORTED;
    GemmArgs a{A, lda, B, ldb, C, ldc, bias, workspace, M, N, K, 1, K, 0};
    a.ex_P = 0;
    a.ep = epilogue; a.E1 = E1; a.lde1 = lde1; a.C2 = C2; a.ldc2 = ldc2; a.rowscale = rowscale; a.rps = rows_per_scale;
    a.vec4 = 1;
    a.KS = pick_ks(M, N, K, 0);
    if (a.KS > 1) {
        if (!workspace || workspace_bytes < (long long)a.KS * M * N * 4) return MIS_ERR_WORKSPACE;
        a.kchunk = (int)(mis_cdiv(mis_cdiv(K, a.KS), BK) * BK);
        a.KS = (int)mis_cdiv(K, a.kchunk);
    }
    const int bn = nt_tile_n(N);
    a.tiles_n = (int)mis_cdiv(N, bn);
    a.tiles_m = (int)mis_cdiv(M, BM);
    const long long nb = (long long)a.tiles_n * a.tiles_m * a.KS;
    if (nb > 0x7fffffffLL) return MIS_ERR_UNSUPPORTED;
    a.n_blocks = (unsigned)nb;
    a.n_blocks_padded = (unsigned)(mis_cdiv(nb, MIS_NUM_XCD) * MIS_NUM_XCD);
    int st = bn == 96 ? launch_nt<96>(a, stream) : launch_nt<128>(a, stream);
    if (st) return st;
    if (a.KS > 1)
        hipLaunchKernelGGL(gemm_reduce_kernel, dim3((unsigned)mis_cdiv((long long)M * N, 32)), dim3(256), 0, stream, a);
    return mis_launch_status();
}

extern "C" int mis_gemm(const float* A, long long lda, const float* B, long long ldb, float* C, long long ldc,
                        const float* bias, int M, int N, int K, int trans, int accumulate, float* workspace,
                        long long workspace_bytes, hipStream_t stream) {
    if (!A || !B || !C || M <= 0 || N <= 0 || K <= 0) return MIS_ERR_ARG;
    if (!a16(A) || !a16(B) || lda % 4 || ldb % 4) return MIS_ERR_UNSUPPORTED;
    if (!trans && K % 4) return MIS_ERR_UNSUPPORTED;
    if (trans && (M % 4 || N % 4)) return MIS_ERR_UNSUPPORTED;
    // the DMA descriptors address each operand with 32-bit byte offsets
    const long long rowsA = trans ? K : M, rowsB = trans ? K : N;
    if (rowsA * lda * 4 >= (1LL << 31) || rowsB * ldb * 4 >= (1LL << 31)) return MIS_ERR_UNSUPPORTED;
    GemmArgs a{A, lda, B, ldb, C, ldc, bias, workspace, M, N, K, 1, K, accumulate};
    a.ex_P = 0;
    a.ep = EP_NONE;
    a.vec4 = (!trans && N % 4 == 0 && ldc % 4 == 0 && a16(C) && (!bias || a16(bias))) ? 1 : 0;
    if (trans && tn_reg_ok(A, lda, B, ldb, C, ldc, M, N, K)) {
        a.dbias = nullptr;
        tn_reg_plan(a);
        if (a.KS > 1 && (!workspace || workspace_bytes < (long long)a.KS * M * N * 4)) return MIS_ERR_WORKSPACE;
        const int st = launch_tn_reg(a, stream);
        if (st) return st;
        if (a.KS > 1)
            hipLaunchKernelGGL(gemm_reduce_kernel, dim3((unsigned)mis_cdiv((long long)M * N, 32)), dim3(256), 0, stream, a);
        return mis_launch_status();
    }
    a.KS = pick_ks(M, N, K, trans);
    if (a.KS > 1) {
        if (!workspace || workspace_bytes < (long long)a.KS * M * N * 4) return MIS_ERR_WORKSPACE;
        a.kchunk = (int)(mis_cdiv(mis_cdiv(K, a.KS), BK) * BK);
        a.KS = (int)mis_cdiv(K, a.kchunk);
    }
    const int bt = trans ? tn_tile(M, N) : 0;
    const int bn = trans ? bt : nt_tile_n(N), bm = trans ? bt : BM;
    a.tiles_n = (int)mis_cdiv(N, bn);
    a.tiles_m = (int)mis_cdiv(M, bm);
    // one linear, XCD-remapped index over (k-slice, tile): slices of a tile and tiles of a row-block are
    // spread over all XCDs in contiguous runs (a 1-tile split-K dW must not land on a single XCD)
    const long long nb = (long long)a.tiles_n * a.tiles_m * a.KS;
    if (nb > 0x7fffffffLL) return MIS_ERR_UNSUPPORTED;
    a.n_blocks = (unsigned)nb;
    a.n_blocks_padded = (unsigned)(mis_cdiv(nb, MIS_NUM_XCD) * MIS_NUM_XCD);
    int st;
    if (trans) {
        if (bt == 96)
            hipLaunchKernelGGL(gemm_tn_kernel<96>, dim3(a.n_blocks_padded), dim3(256), 0, stream, a);
        else
            hipLaunchKernelGGL(gemm_tn_kernel<128>, dim3(a.n_blocks_padded), dim3(256), 0, stream, a);
        st = mis_launch_status();
    } else {
        st = bn == 96 ? launch_nt<96>(a, stream) : launch_nt<128>(a, stream);
    }
    if (st) return st;
    if (a.KS > 1) {
        hipLaunchKernelGGL(gemm_reduce_kernel, dim3((unsigned)mis_cdiv((long long)M * N, 32)), dim3(256), 0, stream, a);
    }
    return mis_launch_status();
}

// Weight and bias gradient of nn.Linear in one pass over dy (reference swin_transformer_unet_skip_expand_decoder_sys.py
// :14,16,107,109 through autograd): dW[M,N] (+)= dy[K,M]^T . x[K,N], db[M] (+)= sum_k dy[k][m] -- mis_gemm's TN form
// whose first tile column also sums the dy tile it staged (mis_colsum's partial + final launches are gone).
// Deterministic: fixed slices, fixed-order sums.  workspace >= mis_gemm_dw_workspace_bytes(M, N, K).
extern "C" long long mis_gemm_dw_workspace_bytes(int M, int N, int K) {
    if (M <= 0 || N <= 0 || K <= 0) return MIS_ERR_ARG;
    int ks = pick_ks(M, N, K, 1);
    if (M % RT == 0 && N % RT == 0) { const int kr = tn_reg_slices(M, N, K); if (kr > ks) ks = kr; }             // either TN form
    return ks > 1 ? (long long)ks * M * (N + 1) * 4 : 0;
}

// reduce != 0: dW / db are finished here (gemm_reduce_kernel).  reduce == 0 (mis_gemm_dw_parts): a split contraction leaves its
// partials in `workspace` -- [KS][M][N] floats, then [KS][M] column sums of dy -- and *slices = KS (0: the contraction was not
// split and dW / db are complete); the caller sums the slices later (mis_colsum_batch), off the launch sequence of the backward.
static int gemm_dw_impl(const float* dy, long long lddy, const float* x, long long ldx, float* dW, long long lddw,
                        float* db, int M, int N, int K, int accumulate, float* workspace, long long workspace_bytes,
                        hipStream_t stream, int reduce, int* slices) {
    if (!dy || !x || !dW || (!db && reduce) || M <= 0 || N <= 0 || K <= 0) return MIS_ERR_ARG;
    if (!a16(dy) || !a16(x) || lddy % 4 || ldx % 4 || M % 4 || N % 4) return MIS_ERR_UNSUPPORTED;
    GemmArgs a{dy, lddy, x, ldx, dW, lddw, nullptr, workspace, M, N, K, 1, K, accumulate};
    a.ex_P = 0;
    a.ep = EP_NONE;
    a.vec4 = 0;
    a.dbias = db;
    a.dbias_acc = accumulate;
    if (slices) *slices = 0;
    const unsigned red_blocks = (unsigned)(mis_cdiv((long long)M * N, 32) + (db ? mis_cdiv(M, 32) : 0));
    if (tn_reg_ok(dy, lddy, x, ldx, dW, lddw, M, N, K) && (long long)K * lddy * 4 < (1LL << 31) && (long long)K * ldx * 4 < (1LL << 31)) {
        tn_reg_plan(a);
        if (a.KS > 1) {
            if (!workspace || workspace_bytes < (long long)a.KS * M * (N + 1) * 4) return MIS_ERR_WORKSPACE;
            a.wsb = workspace + (long long)a.KS * M * N;
        }
        const int st = launch_tn_reg(a, stream);
        if (st) return st;
        if (a.KS > 1) {
            if (!reduce) { *slices = a.KS; return MIS_OK; }
            hipLaunchKernelGGL(gemm_reduce_kernel, dim3(red_blocks), dim3(256), 0, stream, a);
        }
        return mis_launch_status();
    }
    a.KS = pick_ks(M, N, K, 1);
    if (a.KS > 1) {
        if (!workspace || workspace_bytes < (long long)a.KS * M * (N + 1) * 4) return MIS_ERR_WORKSPACE;
        a.kchunk = (int)(mis_cdiv(mis_cdiv(K, a.KS), BK) * BK);
        a.KS = (int)mis_cdiv(K, a.kchunk);
        a.wsb = workspace + (long long)a.KS * M * N;
    }
    const int bt = tn_tile(M, N);
    a.tiles_n = (int)mis_cdiv(N, bt);
    a.tiles_m = (int)mis_cdiv(M, bt);
    const long long nb = (long long)a.tiles_n * a.tiles_m * a.KS;
    if (nb > 0x7fffffffLL) return MIS_ERR_UNSUPPORTED;
    a.n_blocks = (unsigned)nb;
    a.n_blocks_padded = (unsigned)(mis_cdiv(nb, MIS_NUM_XCD) * MIS_NUM_XCD);
    if (bt == 96)
        hipLaunchKernelGGL(gemm_tn_kernel<96>, dim3(a.n_blocks_padded), dim3(256), 0, stream, a);
    else
        hipLaunchKernelGGL(gemm_tn_kernel<128>, dim3(a.n_blocks_padded), dim3(256), 0, stream, a);
    const int st = mis_launch_status();
    if (st) return st;
    if (a.KS > 1) {
        if (!reduce) { *slices = a.KS; return MIS_OK; }
        hipLaunchKernelGGL(gemm_reduce_kernel, dim3(red_blocks), dim3(256), 0, stream, a);
    }
    return mis_launch_status();
}

extern "C" int mis_gemm_dw(const float* dy, long long lddy, const float* x, long long ldx, float* dW, long long lddw,
                           float* db, int M, int N, int K, int accumulate, float* workspace, long long workspace_bytes,
                           hipStream_t stream) {
    return gemm_dw_impl(dy, lddy, x, ldx, dW, lddw, db, M, N, K, accumulate, workspace, workspace_bytes, stream, 1, nullptr);
}

// mis_gemm_dw without its finishing launch (db may be NULL: weight gradient only).  *slices > 0: `workspace` (the caller's own
// until the sums have run) holds `*slices` partial matrices [M][N] followed by `*slices` partial rows [M] of the bias gradient;
// dW / db are untouched (`accumulate` is the final sum's business).  *slices == 0: dW / db are complete.
extern "C" int mis_gemm_dw_parts(const float* dy, long long lddy, const float* x, long long ldx, float* dW, long long lddw,
                                 float* db, int M, int N, int K, int accumulate, float* workspace, long long workspace_bytes,
                                 int* slices, hipStream_t stream) {
    if (!slices) return MIS_ERR_ARG;
    return gemm_dw_impl(dy, lddy, x, ldx, dW, lddw, db, M, N, K, accumulate, workspace, workspace_bytes, stream, 0, slices);
}

// ---- the NT form with a pre-split B operand ----------------------------------------------------------------------------
// mis_gemm_split_b cuts B [N][K] (an nn.Linear weight, or its transpose for the data gradient) into the bf16 piece planes once;
// mis_gemm_nt_split is mis_gemm (trans = 0) / mis_gemm_ex / mis_gemm_expand on them: same products, same sums, same results
// bit for bit as the bf16x3 kernels that split B per tile -- minus two thirds of their vector instructions.
static int split_k3(int K) { return (int)(mis_cdiv(K, BK) * BK); }

extern "C" long long mis_gemm_split_bytes(int N, int K) {
    if (N <= 0 || K <= 0) return MIS_ERR_ARG;
    return 3LL * N * split_k3(K) * 2;
}

static long long split_fill(SplitJob& j, const float* B, long long ldb, int N, int K, void* B3, long long first, int natural = 0) {
    if (!B || !B3 || N <= 0 || K <= 0 || first < 0) return MIS_ERR_ARG;
    if (K % 4 || ldb % 4 || ldb < K || !a16(B) || !a16(B3)) return MIS_ERR_UNSUPPORTED;
    if (mis_gemm_split_bytes(N, K) >= (1LL << 31)) return MIS_ERR_UNSUPPORTED;      // the GEMM addresses the planes with 32 bits
    j.B = B; j.ldb = ldb; j.B3 = B3; j.N = N; j.K = K; j.K3 = split_k3(K); j.first = (int)first; j.natural = natural ? 1 : 0;
    const long long units = mis_cdiv((long long)N * (j.K3 / 32) * 4, 256);
    if (first + units > 0x7fffffffLL) return MIS_ERR_ARG;
    return units;
}

extern "C" int mis_gemm_split_b_layout(const float* B, long long ldb, int N, int K, void* B3, int natural, hipStream_t stream) {
    SplitJob j;
    memset(&j, 0, sizeof(j));
    const long long units = split_fill(j, B, ldb, N, K, B3, 0, natural);
    if (units < 0) return (int)units;
    hipLaunchKernelGGL(gemm_split_kernel, dim3((unsigned)units), dim3(256), 0, stream, j);
    return mis_launch_status();
}

extern "C" int mis_gemm_split_b(const float* B, long long ldb, int N, int K, void* B3, hipStream_t stream) {
    return mis_gemm_split_b_layout(B, ldb, N, K, B3, 0, stream);
}

// batched form, as mis_transpose_job / mis_transpose_batch: records filled on the host, one launch for all of them
extern "C" long long mis_gemm_split_job_bytes(void) { return (long long)sizeof(SplitJob); }

extern "C" long long mis_gemm_split_job_layout(void* job, const float* B, long long ldb, int N, int K, void* B3, long long first,
                                               int natural) {
    if (!job) return MIS_ERR_ARG;
    SplitJob j;
    memset(&j, 0, sizeof(j));
    const long long units = split_fill(j, B, ldb, N, K, B3, first, natural);
    if (units < 0) return units;
    memcpy(job, &j, sizeof(j));
    return units;
}

extern "C" long long mis_gemm_split_job(void* job, const float* B, long long ldb, int N, int K, void* B3, long long first) {
    return mis_gemm_split_job_layout(job, B, ldb, N, K, B3, first, 0);
}

// 1: mis_gemm_nt_split runs this shape on the register-A kernel, which reads the planes in the NATURAL element order
// (mis_gemm_split_*_layout with natural = 1, mis_gemm_nt_split_layout with layout = 1); 0: the staged kernels' order
extern "C" int mis_gemm_nt_split_natural(int M, int N, int K) {
    return (M > 0 && N > 0 && K > 0 && nt_rega_shape(M, N, K)) ? 1 : 0;
}

extern "C" int mis_gemm_split_batch(const void* jobs, int n, long long units, hipStream_t stream) {
    if (!jobs || n <= 0 || units <= 0 || units > 0x7fffffffLL) return MIS_ERR_ARG;
    hipLaunchKernelGGL(gemm_split_batch_kernel, dim3((unsigned)units), dim3(256), 0, stream,
                       reinterpret_cast<const SplitJob*>(jobs), n);
    return mis_launch_status();
}

extern "C" long long mis_gemm_nt_split_workspace_bytes(int M, int N, int K) {
    if (M <= 0 || N <= 0 || K <= 0) return MIS_ERR_ARG;
    int bm, bn, ks;
    if (!b3_choice(M, N, K, bm, bn, ks)) return 0;
    return ks > 1 ? (long long)ks * M * N * 4 : 0;
}

// C [M][N] (+)= A [M][K] . B^T + bias with B given as mis_gemm_split_b's planes.  epilogue 0: plain (accumulate allowed);
// 1 .. 3: the fused epilogues of mis_gemm_ex (E1 / C2 / rowscale as there); ex_P > 0: the pixel-shuffle store of
// mis_gemm_expand (C dense [B H P W P][ex_c], M = B ex_H ex_W, N = P P ex_c, no bias, no epilogue, no split-K).
// workspace >= mis_gemm_nt_split_workspace_bytes(M, N, K).  MIS_ERR_UNSUPPORTED (also: contractions the rule leaves to the
// short-contraction kernel): the caller uses the fp32-B entry points.
static int gemm_nt_split_impl(const float* A, long long lda, const void* B3, float* C, long long ldc, const float* bias,
                              int M, int N, int K, int accumulate, int epilogue, const float* E1, long long lde1,
                              float* C2, long long ldc2, const float* rowscale, long long rows_per_scale, int ex_H,
                              int ex_W, int ex_P, int ex_c, float* workspace, long long workspace_bytes,
                              hipStream_t stream, int layout) {
    if (!A || !B3 || (!C && epilogue != EP_GELU_FWD) || M <= 0 || N <= 0 || K <= 0) return MIS_ERR_ARG;
    if (epilogue < EP_NONE || epilogue > EP_RESIDUAL) return MIS_ERR_ARG;
    if (!a16(A) || !a16(B3) || lda % 4 || K % 4) return MIS_ERR_UNSUPPORTED;
    if ((long long)M * lda * 4 >= (1LL << 31) || mis_gemm_split_bytes(N, K) >= (1LL << 31)) return MIS_ERR_UNSUPPORTED;
    GemmArgs a{A, lda, nullptr, 0, C, ldc, bias, workspace, M, N, K, 1, K, accumulate};
    a.B3 = B3; a.K3 = split_k3(K); a.b3_plane = (unsigned)((long long)N * a.K3 * 2);
    a.ex_P = 0;
    a.ep = epilogue;
    a.vec4 = (N % 4 == 0 && ldc % 4 == 0 && a16(C) && (!bias || a16(bias))) ? 1 : 0;
    if (ex_P > 0) {
        if (epilogue != EP_NONE || bias || accumulate) return MIS_ERR_ARG;
        if (ex_H <= 0 || ex_W <= 0 || ex_c <= 0 || ex_c % 16 || M % (ex_H * ex_W) || N != ex_P * ex_P * ex_c) return MIS_ERR_UNSUPPORTED;
        if ((long long)M * N * 4 >= (1LL << 31)) return MIS_ERR_UNSUPPORTED;
        a.ex_P = ex_P; a.ex_H = ex_H; a.ex_W = ex_W; a.ex_c = ex_c; a.ldc = N;
    } else if (epilogue != EP_NONE) {
        if (accumulate) return MIS_ERR_ARG;
        if (epilogue == EP_GELU_FWD ? (!C2 || ldc2 < N) : (!E1 || lde1 < N)) return MIS_ERR_ARG;
        if (epilogue == EP_RESIDUAL && rowscale && rows_per_scale <= 0) return MIS_ERR_ARG;
        if (!a.vec4) return MIS_ERR_UNSUPPORTED;
        if (epilogue == EP_GELU_FWD ? (ldc2 % 4 || !a16(C2)) : (lde1 % 4 || !a16(E1))) return MIS_ERR_UNSUPPORTED;
        a.E1 = E1; a.lde1 = lde1; a.C2 = C2; a.ldc2 = ldc2; a.rowscale = rowscale; a.rps = rows_per_scale;
    }
    if (layout) {
        // natural-order planes: only the register-A kernel reads them (the caller asked mis_gemm_nt_split_natural)
        if (!nt_rega_shape(M, N, K) || a.ex_P || !a.vec4) return MIS_ERR_UNSUPPORTED;
        return launch_nt_rega(a, stream);
    }
    int bm, bn, ks;
    if (!b3_choice(M, N, K, bm, bn, ks)) return MIS_ERR_UNSUPPORTED;
    if (a.ex_P && ks != 1) return MIS_ERR_UNSUPPORTED;
    a.KS = ks;
    a.bm_force = bm;
    if (a.KS > 1) {
        if (!workspace || workspace_bytes < (long long)a.KS * M * N * 4) return MIS_ERR_WORKSPACE;
        a.kchunk = (int)(mis_cdiv(mis_cdiv(K, a.KS), BK) * BK);
        a.KS = (int)mis_cdiv(K, a.kchunk);
    }
    a.tiles_n = (int)mis_cdiv(N, bn);
    a.tiles_m = (int)mis_cdiv(M, BM);
    const long long nb = (long long)a.tiles_n * a.tiles_m * a.KS;
    if (nb > 0x7fffffffLL) return MIS_ERR_UNSUPPORTED;
    a.n_blocks = (unsigned)nb;
    a.n_blocks_padded = (unsigned)(mis_cdiv(nb, MIS_NUM_XCD) * MIS_NUM_XCD);
    const int st = bn == 96 ? launch_nt<96>(a, stream) : launch_nt<128>(a, stream);
    if (st) return st;
    if (a.KS > 1)
        hipLaunchKernelGGL(gemm_reduce_kernel, dim3((unsigned)mis_cdiv((long long)M * N, 32)), dim3(256), 0, stream, a);
    return mis_launch_status();
}

extern "C" int mis_gemm_nt_split(const float* A, long long lda, const void* B3, float* C, long long ldc, const float* bias,
                                 int M, int N, int K, int accumulate, int epilogue, const float* E1, long long lde1,
                                 float* C2, long long ldc2, const float* rowscale, long long rows_per_scale, int ex_H,
                                 int ex_W, int ex_P, int ex_c, float* workspace, long long workspace_bytes,
                                 hipStream_t stream) {
    return gemm_nt_split_impl(A, lda, B3, C, ldc, bias, M, N, K, accumulate, epilogue, E1, lde1, C2, ldc2, rowscale, rows_per_scale,
                              ex_H, ex_W, ex_P, ex_c, workspace, workspace_bytes, stream, 0);
}

// mis_gemm_nt_split on planes in the layout `layout` (0: as mis_gemm_nt_split; 1: natural order, shapes for which
// mis_gemm_nt_split_natural says 1 -- the register-A kernel: A straight from HBM into the MFMA operand registers, round 6)
extern "C" int mis_gemm_nt_split_layout(const float* A, long long lda, const void* B3, float* C, long long ldc, const float* bias,
                                        int M, int N, int K, int accumulate, int epilogue, const float* E1, long long lde1,
                                        float* C2, long long ldc2, const float* rowscale, long long rows_per_scale, int ex_H,
                                        int ex_W, int ex_P, int ex_c, float* workspace, long long workspace_bytes, int layout,
                                        hipStream_t stream) {
    return gemm_nt_split_impl(A, lda, B3, C, ldc, bias, M, N, K, accumulate, epilogue, E1, lde1, C2, ldc2, rowscale, rows_per_scale,
                              ex_H, ex_W, ex_P, ex_c, workspace, workspace_bytes, stream, layout);
}

// mis_gemm_expand_ln_head on natural-order planes of the expand weight (round 6): the persistent resident-panel register-A kernel
// -- K <= 96, c = 96, the 16 (p1, p2) column panels of FinalPatchExpand_X4 stay in LDS, a wave's accumulators ARE token rows of
// the shuffled tensor, so LayerNorm and the output head run in registers.  Same arguments otherwise.  MIS_ERR_UNSUPPORTED:
// mis_gemm_nt_split_natural(M, N, K) is 0, K > 96, c != 96, NC outside 2 .. 4.
extern "C" int mis_gemm_expand_ln_head_split(const float* x, long long lda, const void* B3, float* out, int B, int H, int Wd, int K,
                                             int P, int c, const float* gamma, const float* beta, const float* head_w, int NC,
                                             float eps, float* mean, float* rstd, float* logits, long long logits_bs,
                                             hipStream_t stream) {
    if (!x || !B3 || !gamma || !beta || !head_w || !mean || !rstd || !logits || B <= 0 || H <= 0 || Wd <= 0 || K <= 0 || P <= 0)
        return MIS_ERR_ARG;
    if (c != 96 || NC < 2 || NC > 4) return MIS_ERR_UNSUPPORTED;
    const long long M = (long long)B * H * Wd, N = (long long)P * P * c;
    if (M * 4 >= (1LL << 31) || !nt_rega_shape((int)M, (int)N, K) || split_k3(K) > 96) return MIS_ERR_UNSUPPORTED;
    if (!a16(x) || !a16(B3) || lda % 4 || K % 4 || (out && !a16(out))) return MIS_ERR_UNSUPPORTED;
    if (M * lda * 4 >= (1LL << 31) || mis_gemm_split_bytes((int)N, K) >= (1LL << 31)) return MIS_ERR_UNSUPPORTED;
    if (logits_bs < (long long)NC * H * P * Wd * P) return MIS_ERR_ARG;
    GemmArgs a{x, lda, nullptr, 0, out, N, nullptr, nullptr, (int)M, (int)N, K, 1, K, 0};
    a.B3 = B3; a.K3 = split_k3(K); a.b3_plane = (unsigned)(N * a.K3 * 2);
    a.ex_P = P; a.ex_H = H; a.ex_W = Wd; a.ex_c = c; a.vec4 = 1;
    a.ep = EP_LNHEAD;
    a.ln_g = gamma; a.ln_b = beta; a.head_w = head_w; a.ln_mean = mean; a.ln_rstd = rstd; a.logits = logits;
    a.logits_bs = logits_bs; a.head_nc = NC; a.ln_eps = eps;
    return launch_nt_rega(a, stream);
}

// proj / fc2 of a 96-channel Swin block with everything up to the next LayerNorm in the epilogue (register-A kernels, natural-order
// planes): X = E1 + rowscale[m / rows_per_scale] * (A . B^T + bias) -> X (the block's residual stream, row stride ldx), and
// Y = LayerNorm(X) * gamma + beta (row stride ldy), mean / rstd [M] for the backward -- `x = shortcut + self.drop_path(x)` then
// `self.norm2(x)` (reference ...sys.py:276-281) or the next block's `norm1` (:244-250).  N must be 96 (a row = one tile row).
// MIS_ERR_UNSUPPORTED: mis_gemm_nt_split_natural(M, 96, K) is 0 or an operand is not float4-addressable.
extern "C" int mis_gemm_nt_residual_ln(const float* A, long long lda, const void* B3, const float* bias, int M, int N, int K,
                                       const float* E1, long long lde1, const float* rowscale, long long rows_per_scale,
                                       float* X, long long ldx, const float* gamma, const float* beta, float eps, float* Y,
                                       long long ldy, float* mean, float* rstd, hipStream_t stream) {
    if (!A || !B3 || !E1 || !X || !gamma || !beta || !Y || !mean || !rstd || M <= 0 || K <= 0) return MIS_ERR_ARG;
    if (rowscale && rows_per_scale <= 0) return MIS_ERR_ARG;
    if (N != 96 || !nt_rega_shape(M, N, K)) return MIS_ERR_UNSUPPORTED;
    if (!a16(A) || !a16(B3) || !a16(E1) || !a16(X) || !a16(Y) || (bias && !a16(bias)) || lda % 4 || lde1 % 4 || ldx % 4 || ldy % 4 || K % 4)
        return MIS_ERR_UNSUPPORTED;
    if ((long long)M * lda * 4 >= (1LL << 31) || mis_gemm_split_bytes(N, K) >= (1LL << 31)) return MIS_ERR_UNSUPPORTED;
    GemmArgs a{A, lda, nullptr, 0, X, ldx, bias, nullptr, M, N, K, 1, K, 0};
    a.B3 = B3; a.K3 = split_k3(K); a.b3_plane = (unsigned)((long long)N * a.K3 * 2);
    a.ex_P = 0; a.vec4 = 1;
    a.ep = EP_RESIDUAL_LN;
    a.E1 = E1; a.lde1 = lde1; a.C2 = Y; a.ldc2 = ldy; a.rowscale = rowscale; a.rps = rowscale ? rows_per_scale : 1;
    a.ln_g = gamma; a.ln_b = beta; a.ln_mean = mean; a.ln_rstd = rstd; a.ln_eps = eps;
    return launch_nt_rega(a, stream);
}

// nn.Linear of PatchExpand / FinalPatchExpand_X4 fused with their pixel shuffle (reference
// swin_transformer_unet_skip_expand_decoder_sys.py:373-380, :401-408): out[(b, h*P + p1, w*P + p2)][c] =
// sum_k x[(b, h, w)][k] * W[(p1*P + p2)*c + c'][k] -- the GEMM's epilogue writes the shuffled layout, so the expanded
// token-major tensor never exists in forward.  x [B*H*W][K] (row stride lda), W [P*P*c][K], out [B*H*P*W*P][c] dense.
// No bias (the reference's expand layers have none), no split-K: MIS_ERR_UNSUPPORTED when mis_gemm would split
// this shape (callers then use mis_gemm + mis_token_rearrange) or c % 16 != 0.
extern "C" int mis_gemm_expand(const float* x, long long lda, const float* W, long long ldb, float* out, int B, int H,
                               int Wd, int K, int P, int c, hipStream_t stream) {
    if (!x || !W || !out || B <= 0 || H <= 0 || Wd <= 0 || K <= 0 || P <= 0 || c <= 0) return MIS_ERR_ARG;
    const long long M = (long long)B * H * Wd, N = (long long)P * P * c;
    if (!a16(x) || !a16(W) || lda % 4 || ldb % 4 || K % 4 || c % 16) return MIS_ERR_UNSUPPORTED;
    if (M * lda * 4 >= (1LL << 31) || N * ldb * 4 >= (1LL << 31) || M * N * 4 >= (1LL << 31))
        return MIS_ERR_UNSUPPORTED;
    if (pick_ks((int)M, (int)N, K, 0) != 1) return MIS_ERR_UNSUPPORTED;
    GemmArgs a{x, lda, W, ldb, out, N, nullptr, nullptr, (int)M, (int)N, K, 1, K, 0};
    a.ex_P = P; a.ex_H = H; a.ex_W = Wd; a.ex_c = c;
    const int bn = nt_tile_n((int)N);
    a.tiles_n = (int)mis_cdiv(N, bn);
    a.tiles_m = (int)mis_cdiv(M, BM);
    const long long nb = (long long)a.tiles_n * a.tiles_m;
    if (nb > 0x7fffffffLL) return MIS_ERR_UNSUPPORTED;
    a.n_blocks = (unsigned)nb;
    a.n_blocks_padded = (unsigned)(mis_cdiv(nb, MIS_NUM_XCD) * MIS_NUM_XCD);
    return bn == 96 ? launch_nt<96>(a, stream) : launch_nt<128>(a, stream);
}

// mis_gemm_expand (P x P pixel shuffle, c = 96) with FinalPatchExpand_X4's LayerNorm and the bias-free 1 x 1 output convolution in
// the GEMM's epilogue (reference swin_transformer_unet_skip_expand_decoder_sys.py:401-409 `x = self.expand(x); rearrange;
// x = self.norm(x)`, :749-752 `self.output(x)`): per token of the shuffled tensor mean / rstd (saved for the backward, index =
// token of the shuffled tensor) and logits [B][NC][H P][W P] (batch stride logits_bs floats).  `out` (the shuffled tokens, dense
// [B H P W P][96]) is what the backward reads; NULL = not kept (a forward nobody differentiates: the EMA teacher).  Same bits as
// mis_gemm_expand + mis_ln_head_fwd.  MIS_ERR_UNSUPPORTED: c != 96, NC outside 2 .. 4, or a shape mis_gemm_expand refuses.
extern "C" int mis_gemm_expand_ln_head(const float* x, long long lda, const float* W, long long ldb, float* out, int B, int H,
                                       int Wd, int K, int P, int c, const float* gamma, const float* beta, const float* head_w,
                                       int NC, float eps, float* mean, float* rstd, float* logits, long long logits_bs,
                                       hipStream_t stream) {
    if (!x || !W || !gamma || !beta || !head_w || !mean || !rstd || !logits || B <= 0 || H <= 0 || Wd <= 0 || K <= 0 || P <= 0)
        return MIS_ERR_ARG;
    if (c != 96 || NC < 2 || NC > 4) return MIS_ERR_UNSUPPORTED;
    const long long M = (long long)B * H * Wd, N = (long long)P * P * c;
    if (!a16(x) || !a16(W) || lda % 4 || ldb % 4 || K % 4 || (out && !a16(out)) || !a16(gamma) || !a16(beta) || !a16(head_w))
        return MIS_ERR_UNSUPPORTED;
    if (M * lda * 4 >= (1LL << 31) || N * ldb * 4 >= (1LL << 31) || M * N * 4 >= (1LL << 31)) return MIS_ERR_UNSUPPORTED;
    if (logits_bs < (long long)NC * H * P * Wd * P) return MIS_ERR_ARG;
    if (pick_ks((int)M, (int)N, K, 0) != 1) return MIS_ERR_UNSUPPORTED;
    GemmArgs a{x, lda, W, ldb, out, N, nullptr, nullptr, (int)M, (int)N, K, 1, K, 0};
    a.ex_P = P; a.ex_H = H; a.ex_W = Wd; a.ex_c = c;
    a.ep = EP_LNHEAD;
    a.ln_g = gamma; a.ln_b = beta; a.head_w = head_w; a.ln_mean = mean; a.ln_rstd = rstd; a.logits = logits;
    a.logits_bs = logits_bs; a.head_nc = NC; a.ln_eps = eps;
    a.tiles_n = (int)(N / 96);
    a.tiles_m = (int)mis_cdiv(M, 64);
    const long long nb = (long long)a.tiles_n * a.tiles_m;
    if (nb > 0x7fffffffLL) return MIS_ERR_UNSUPPORTED;
    a.n_blocks = (unsigned)nb;
    a.n_blocks_padded = (unsigned)(mis_cdiv(nb, MIS_NUM_XCD) * MIS_NUM_XCD);
    return gemm_bf3() ? launch_nt_prec<64, 96, EP_LNHEAD, 1>(a, stream) : launch_nt_prec<64, 96, EP_LNHEAD, 0>(a, stream);
}
