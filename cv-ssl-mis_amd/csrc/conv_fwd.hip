// Direct convolution forward (and, with flipped/transposed packed weights, the
// data-gradient) for the UNet / unet_3D / V-Net blocks of the Mean-Teacher step.
//
// Replaces: nn.Conv2d(k=3,pad=1) / nn.Conv2d(k=1)  (reference code/networks/unet.py:37,41,73,138)
//           nn.Conv3d(k=3,pad=1) / nn.Conv3d(k=1)  (reference code/networks/utils.py:104,107; unet_3D.py:59;
//                                                   vnet.py:16,73,100,175)
//
// Design (gfx950): NCDHW fp32 in HBM (2D = D==1).  One 256-thread workgroup
// owns a TZ x TY x TX output tile of one image for CO_B output channels.  For
// each chunk of CI_B input channels the haloed input tile and the matching
// packed weights live in LDS; the contraction over (ci, tap) runs on the
// fp32-input matrix pipe (v_mfma_f32_16x16x4_f32: M = 16 output channels,
// N = 16 pixels, K = 4 input channels of one tap).  That MFMA is bit-for-bit an
// fp32 fmaf chain, so numerics are those of an fp32 direct convolution, at the
// full fp32 rate.  There is no im2col buffer in HBM: the "im2col" is only the
// LDS addressing (per-lane pixel offset + compile-time tap offset).
//
// Staging is LDS-DMA (buffer_load_dword ... lds / buffer_load_dwordx4 ... lds) into a DOUBLE-buffered
// stage: the loads of chunk k+1 are issued before the MFMA loop of chunk k and land in the other
// buffer while the matrix pipe works, with one barrier per chunk.  The register file holds 64
// accumulator + ~100 other registers per lane, i.e. only 2-3 waves per SIMD are resident, too few for
// workgroup interleaving alone to hide a register-staged (global -> VGPR -> ds_write) copy: that version
// of this kernel left the matrix pipe idle ~30 % of the time (profiles/r01_unet3d_pmc_*.csv).
// The DMA destination is wave-uniform base + lane * size, so the LDS image is built from 64-dword
// pieces of one channel's haloed tile; every lane gathers from its own global address, and halo /
// channel padding comes for free from the buffer descriptor's range check (out-of-range -> 0).
//
// MFMA 16x16x4 f32 operand maps (cdna guide s.3):
//   A[i = lane&15][k = lane>>4]   -> weight  w[co0 + i][ci0 + k][tap]
//   B[k = lane>>4][j = lane&15]   -> input   x[ci0 + k][pixel(j) shifted by tap]
//   D[row = (lane>>4)*4 + r][col = lane&15]  -> y[co0 + row][pixel(col)]
// Pixel groups are paired: column j of the "even" MFMA is tile pixel 2j, of the "odd" MFMA pixel
// 2j+1.  One ds_read_b64 at pixel 2j therefore yields the B operand of two (group, kx) pairs, which
// cuts the LDS read cycles per MFMA ~3x.  The epilogue stores (even, odd) as one float2: 128
// contiguous bytes per channel per 16 lanes.
#include "common.h"
#include <stdio.h>
#include <stdlib.h>

namespace {

struct ConvFwdArgs {
    const float* x; long long x_bs;
    const float* wp;      // packed [Cin_pad4][TAPS][Cout_pad16]
    const float* bias;    // [Cout] or nullptr
    float* y; long long y_bs;
    int N, Cin, Cout, Cin_pad, Cout_pad, D, H, W;
    int tiles_z, tiles_y, tiles_x, co_blocks;
    unsigned n_blocks, n_blocks_padded;
    int st2;  // 1: output rows may be stored as aligned float2 (W even, 8-byte aligned rows)
    float2* stat; long long stat_sc, stat_sn;   // optional per-tile (sum, sumsq) of the output, see mis_conv_fwd_stats
    int stagger, stagger_mod;                   // experiment: de-phase the first round of workgroups
};

using namespace mis_dma;   // LDS-DMA helpers (common.h)

template <int KD_, int KH_, int KW_, int TZ_, int TY_, int TX_, int CO_B_, int CI_B_, int NT_>
struct Cfg {
    static constexpr int KD = KD_, KH = KH_, KW = KW_, TZ = TZ_, TY = TY_, TX = TX_;
    static constexpr int CO_B = CO_B_, CI_B = CI_B_, NT = NT_;
    static constexpr int TAPS = KD * KH * KW;
    static constexpr int HZ = TZ + KD - 1, HY = TY + KH - 1, HX = TX + KW - 1;
    static constexpr int CS_RAW = HZ * HY * HX;
    static constexpr int NCH = (CS_RAW + 63) / 64;   // 64-dword DMA pieces per channel
    // channel stride in LDS: holds NCH whole pieces and is == 32 (mod 64): ds_read_b64 banks are
    // dword-address mod 64 over a 32-lane group, whose two k-lane halves (ci, ci+1) then use disjoint
    // halves of the banks.
    static constexpr int CS = NCH * 64 + 32;
    static constexpr int NP = NT_ / 2;   // even/odd pixel-group pairs per wave
    static constexpr int M = CO_B / 16;
    static constexpr int PIX = TZ * TY * TX;
    static constexpr int IN_FLOATS = CI_B * CS;
    static constexpr int W_FLOATS = CI_B * TAPS * CO_B;
    static constexpr int W_PIECES = (W_FLOATS + 255) / 256;   // 1 KiB DMA pieces (64 lanes x 16 B)
    static constexpr int WPW = (W_PIECES + 3) / 4;            // pieces per wave
    static constexpr int STAGE = IN_FLOATS + W_PIECES * 256;  // floats per stage buffer
    static constexpr int LDS_BYTES = 2 * STAGE * 4;
    static constexpr int CPW = CI_B / 4;                      // input channels per wave
    static_assert(PIX == 64 * NT, "tile must hold 4 waves x NT x 16 pixels");
    static_assert(NT % 2 == 0 && TX % 2 == 0 && HX % 2 == 0, "even/odd pixel pairing needs even rows");
    static_assert(CO_B % 16 == 0 && CI_B % 4 == 0, "MFMA 16x16x4 granularity");
    static_assert(LDS_BYTES <= 80 * 1024, "two workgroups per CU");
    static_assert(STAGE % 2 == 0, "float2 addressing of the stage buffers");
};

extern __shared__ __attribute__((aligned(16))) float mis_conv_lds[];

template <class C>
__global__ __launch_bounds__(256) void conv_fwd_kernel(const ConvFwdArgs a) {
    float* const lds = mis_conv_lds;

    const unsigned L = mis_xcd_remap(blockIdx.x, a.n_blocks_padded);
    if (L >= a.n_blocks) return;
    if (a.stagger > 0 && blockIdx.x < 1024) {
        const int k = (blockIdx.x >> 8) % a.stagger_mod;
        for (int i = 0; i < k * a.stagger; ++i) __builtin_amdgcn_s_sleep(16);   // 1024 clocks per unit
    }
    unsigned t = L;
    const int cob = t % a.co_blocks; t /= a.co_blocks;
    const int tx = t % a.tiles_x;    t /= a.tiles_x;
    const int ty = t % a.tiles_y;    t /= a.tiles_y;
    const int tz = t % a.tiles_z;    t /= a.tiles_z;
    const int n = t;
    const int z0 = tz * C::TZ, y0 = ty * C::TY, x0 = tx * C::TX, co0 = cob * C::CO_B;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lk = lane >> 4, lj = lane & 15;
    const long long S = (long long)a.D * a.H * a.W;
    const unsigned s_bytes = (unsigned)S * 4u;

    // buffer descriptors (raw, stride 0): x of this image [Cin][S], packed weights [Cin_pad][TAPS][Cout_pad]
    const i32x4 rx = make_rsrc(a.x + (long long)n * a.x_bs, (unsigned)a.Cin * s_bytes);
    const i32x4 rw = make_rsrc(a.wp, (unsigned)a.Cin_pad * C::TAPS * a.Cout_pad * 4u);
    const unsigned lds0 = lds_addr(lds);

    // per-lane source byte offsets of this lane's element of every DMA piece (chunk-invariant):
    // piece p of a channel = halo elements [64p, 64p+64) of s_in[ci][hz][hy][hx]
    unsigned voff[C::NCH];
#pragma unroll
    for (int p = 0; p < C::NCH; ++p) {
        const int e = p * 64 + lane;
        const int hz = e / (C::HY * C::HX), r2 = e - hz * (C::HY * C::HX);
        const int hy = r2 / C::HX, hx = r2 - hy * C::HX;
        const int gz = z0 + hz - C::KD / 2, gy = y0 + hy - C::KH / 2, gx = x0 + hx - C::KW / 2;
        const bool ok = e < C::CS_RAW && (unsigned)gz < (unsigned)a.D && (unsigned)gy < (unsigned)a.H &&
                        (unsigned)gx < (unsigned)a.W;
        voff[p] = ok ? (unsigned)((gz * a.H + gy) * a.W + gx) * 4u : OOB;
    }
    // weights: piece j = 64 float4 of s_w[ci][tap][co]; rows of CO_B floats, Cout_pad apart in HBM
    unsigned wvoff[C::WPW];
#pragma unroll
    for (int i = 0; i < C::WPW; ++i) {
        constexpr int VPR = C::CO_B / 4;
        const int e = (wave + 4 * i) * 64 + lane;
        const int row = e / VPR, q = e - row * VPR;
        const bool ok = e < C::W_FLOATS / 4 && co0 + q * 4 < a.Cout_pad;
        wvoff[i] = ok ? (unsigned)(row * a.Cout_pad + co0 + q * 4) * 4u : OOB;
    }

    auto stage = [&](int buf, int c0) {
        const unsigned st = lds0 + (unsigned)buf * (C::STAGE * 4);   // LDS byte address of the stage buffer
#pragma unroll
        for (int i = 0; i < C::CPW; ++i) {
            const int ci = wave + 4 * i;
            const unsigned cbase = (unsigned)(c0 + ci) * s_bytes;   // >= num_records for c >= Cin -> zeros
#pragma unroll
            for (int p = 0; p < C::NCH; ++p)
                dma_dword(st + (unsigned)(ci * C::CS + p * 64) * 4u, voff[p] + cbase, rx);
        }
        const unsigned wbase = (unsigned)c0 * (unsigned)(C::TAPS * 4) * (unsigned)a.Cout_pad;
#pragma unroll
        for (int i = 0; i < C::WPW; ++i) {
            const int j = wave + 4 * i;
            if (j < C::W_PIECES) dma_dwordx4(st + (unsigned)(C::IN_FLOATS + j * 256) * 4u, wvoff[i] + wbase, rw);
        }
    };

    // Pixel groups come in even/odd PAIRS: pair q of this wave covers 32 consecutive tile pixels,
    // the even MFMA column j is pixel 2j, the odd one pixel 2j+1.  One 8-byte LDS read at pixel 2j
    // then feeds two MFMA operands (taps kx and kx+1 of the even group == taps kx-1.. of the odd
    // group): 2 ds_read_b64 per pair and (kz,ky) row instead of 6 ds_read_b32.
    // po2[q]: per-lane offset (in float2 units) of the pair's even pixel, incl. the k-lane channel.
    int po2[C::NP];
#pragma unroll
    for (int q = 0; q < C::NP; ++q) {
        const int p = (wave * C::NP + q) * 32 + 2 * lj;
        const int px = p % C::TX, py = (p / C::TX) % C::TY, pz = p / (C::TX * C::TY);
        po2[q] = ((pz * C::HY + py) * C::HX + px + lk * C::CS) >> 1;
    }
    const int woff = C::IN_FLOATS + lk * C::TAPS * C::CO_B + lj;   // A operand: weights [ci][tap][co]

    f32x4 acc[C::M][C::NT];
#pragma unroll
    for (int m = 0; m < C::M; ++m)
#pragma unroll
        for (int i = 0; i < C::NT; ++i) acc[m][i] = f32x4{0.f, 0.f, 0.f, 0.f};

    // Operands of one (kz,ky) tap row: B = 2 x NP float2 (4 consecutive halo pixels per pair), A = KW x M.
    // The operands of row r+1 are fetched from LDS before the MFMAs of row r are issued (register double
    // buffer), so the matrix pipe never waits for an LDS round trip inside a chunk.
    struct RowOps { float2 r0[C::NP], r2[C::NP]; float av[C::KW][C::M]; };
    constexpr int R = C::KD * C::KH;

    // All CI_B channels of a chunk are always contracted: channels >= Cin_pad hold zeros (the DMA's range
    // check zero-fills both their pixels and their weight rows), so a ragged last chunk needs no special case
    // and the whole chunk unrolls with compile-time register ping-pong.
    constexpr int T = (C::CI_B / 4) * R;   // tap rows per chunk

    auto compute = [&](const float* st) {
        const float2* __restrict__ s_in2 = reinterpret_cast<const float2*>(st);
        auto fetch = [&](RowOps& o, int t) {
            const int cq = t / R, row = t % R;
            const int kz = row / C::KH, ky = row % C::KH;
            const int rowoff2 = ((kz * C::HY + ky) * C::HX) >> 1;
#pragma unroll
            for (int q = 0; q < C::NP; ++q) {
                o.r0[q] = s_in2[po2[q] + cq * 2 * C::CS + rowoff2];
                if (C::KW == 3) o.r2[q] = s_in2[po2[q] + cq * 2 * C::CS + rowoff2 + 1];
            }
#pragma unroll
            for (int kx = 0; kx < C::KW; ++kx)
#pragma unroll
                for (int m = 0; m < C::M; ++m)
                    o.av[kx][m] = st[woff + cq * 4 * C::TAPS * C::CO_B + (row * C::KW + kx) * C::CO_B + m * 16];
        };
        RowOps ops[2];
        fetch(ops[0], 0);
#pragma unroll
        for (int t = 0; t < T; ++t) {
            if (t + 1 < T) fetch(ops[(t + 1) & 1], t + 1);
            __builtin_amdgcn_sched_barrier(0);   // keep the LDS reads of row t+1 ahead of the MFMAs of row t
            const RowOps& cur = ops[t & 1];
#pragma unroll
            for (int kx = 0; kx < C::KW; ++kx) {
#pragma unroll
                for (int q = 0; q < C::NP; ++q) {
                    const float be = kx == 0 ? cur.r0[q].x : (kx == 1 ? cur.r0[q].y : cur.r2[q].x);
                    const float bo = kx == 0 ? cur.r0[q].y : (kx == 1 ? cur.r2[q].x : cur.r2[q].y);
#pragma unroll
                    for (int m = 0; m < C::M; ++m) {
                        acc[m][2 * q] =
                            __builtin_amdgcn_mfma_f32_16x16x4f32(cur.av[kx][m], be, acc[m][2 * q], 0, 0, 0);
                        acc[m][2 * q + 1] =
                            __builtin_amdgcn_mfma_f32_16x16x4f32(cur.av[kx][m], bo, acc[m][2 * q + 1], 0, 0, 0);
                    }
                }
            }
        }
    };

    // ---- software pipeline over chunks of CI_B input channels: DMA(k+1) || MFMA(k) ----
    const int nchunks = (a.Cin_pad + C::CI_B - 1) / C::CI_B;
    stage(0, 0);
    dma_wait();
    __syncthreads();   // chunk 0 has landed for every wave
    for (int k = 0; k < nchunks; ++k) {
        if (k + 1 < nchunks) stage((k + 1) & 1, (k + 1) * C::CI_B);
        compute(lds + (k & 1) * C::STAGE);
        dma_wait();
        __syncthreads();   // chunk k+1 landed in the other buffer and every wave is done reading chunk k
    }

    // ---- epilogue: bias + store.  D: row = lk*4 + r -> channel, col = lj -> pixel pair (2j, 2j+1) ----
    // Fast path (tile and channel block fully inside, float2-aligned rows): one buffer_store_dwordx2 per
    // (pair, channel) with the per-lane offset computed once per pair and the channel in the scalar offset;
    // no per-store predicates or 64-bit address arithmetic (the generic path below costs ~15 % of the
    // 2-D kernel's time, measured by removing the stores).
    if (a.st2 && z0 + C::TZ <= a.D && y0 + C::TY <= a.H && x0 + C::TX <= a.W && co0 + C::CO_B <= a.Cout) {
        const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(
            (void*)(a.y + (long long)n * a.y_bs + (long long)co0 * S), 0, (int)((unsigned)C::CO_B * s_bytes), 0x00020000);
        float bv[C::M][4];
#pragma unroll
        for (int m = 0; m < C::M; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r) bv[m][r] = a.bias ? a.bias[co0 + m * 16 + lk * 4 + r] : 0.f;
        const unsigned lane_c = (unsigned)(lk * 4) * s_bytes;
        float st1[C::M][4], st2[C::M][4];
#pragma unroll
        for (int m = 0; m < C::M; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r) st1[m][r] = st2[m][r] = 0.f;
#pragma unroll
        for (int q = 0; q < C::NP; ++q) {
            const int p = (wave * C::NP + q) * 32 + 2 * lj;
            const int px = p % C::TX, py = (p / C::TX) % C::TY, pz = p / (C::TX * C::TY);
            const unsigned vo = lane_c + (unsigned)(((z0 + pz) * a.H + (y0 + py)) * a.W + x0 + px) * 4u;
#pragma unroll
            for (int m = 0; m < C::M; ++m) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    typedef float f32x2 __attribute__((ext_vector_type(2)));
                    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
                    const f32x2 v = {acc[m][2 * q][r] + bv[m][r], acc[m][2 * q + 1][r] + bv[m][r]};
                    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, v), ry, (int)vo,
                                                          (int)((unsigned)(m * 16 + r) * s_bytes), 0);
                    if (a.stat) {   // uniform: statistics of the normalisation that consumes this output
                        st1[m][r] += v[0] + v[1];
                        st2[m][r] += v[0] * v[0] + v[1] * v[1];
                    }
                }
            }
        }
        if (a.stat) {
            // (sum, sum of squares) of this tile per output channel: 16 lanes share a channel -> xor-shuffles,
            // 4 waves -> LDS (the stage buffers are idle now), fixed order -> deterministic.  One float2 per
            // (channel, image, tile) replaces the separate statistics pass over the conv output.
#pragma unroll
            for (int m = 0; m < C::M; ++m)
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int o = 1; o < 16; o <<= 1) {
                        st1[m][r] += __shfl_xor(st1[m][r], o, 64);
                        st2[m][r] += __shfl_xor(st2[m][r], o, 64);
                    }
            float2* red = reinterpret_cast<float2*>(lds);
            if (lj == 0) {
#pragma unroll
                for (int m = 0; m < C::M; ++m)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        red[wave * C::CO_B + m * 16 + lk * 4 + r] = make_float2(st1[m][r], st2[m][r]);
            }
            __syncthreads();
            if (tid < C::CO_B) {
                const float2 p0 = red[tid], p1 = red[C::CO_B + tid], p2 = red[2 * C::CO_B + tid], p3 = red[3 * C::CO_B + tid];
                const long long tile = ((long long)tz * a.tiles_y + ty) * a.tiles_x + tx;
                a.stat[(long long)(co0 + tid) * a.stat_sc + (long long)n * a.stat_sn + tile] =
                    make_float2((p0.x + p1.x) + (p2.x + p3.x), (p0.y + p1.y) + (p2.y + p3.y));
            }
        }
        return;
    }
    float* __restrict__ yout = a.y + (long long)n * a.y_bs;
#pragma unroll
    for (int q = 0; q < C::NP; ++q) {
        const int p = (wave * C::NP + q) * 32 + 2 * lj;
        const int px = p % C::TX, py = (p / C::TX) % C::TY, pz = p / (C::TX * C::TY);
        const int gz = z0 + pz, gy = y0 + py, gx = x0 + px;
        const bool row_ok = gz < a.D && gy < a.H;
        const long long sp = ((long long)gz * a.H + gy) * a.W + gx;
#pragma unroll
        for (int m = 0; m < C::M; ++m) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int co = co0 + m * 16 + lk * 4 + r;
                if (row_ok && co < a.Cout) {
                    const float bv = a.bias ? a.bias[co] : 0.f;
                    const float ve = acc[m][2 * q][r] + bv, vo = acc[m][2 * q + 1][r] + bv;
                    float* dst = yout + (long long)co * S + sp;
                    if (a.st2 && gx + 1 < a.W) {
                        *reinterpret_cast<float2*>(dst) = make_float2(ve, vo);
                    } else {
                        if (gx < a.W) dst[0] = ve;
                        if (gx + 1 < a.W) dst[1] = vo;
                    }
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// 2-D 3x3 convolution with Cout <= 4 and Cin <= 16 (the UNet's `out_conv` 16 -> num_classes, reference
// unet.py:138): in the 16-row MFMA form three quarters of the matrix rows are padding (measured 21 TF, 0.26 ms per
// step).  v_mfma_f32_4x4x1_16B_f32 has no such padding: 16 independent 4x4 outer products per instruction = 4 output
// channels x 64 consecutive pixels, one (cin, tap) term per instruction.  The weight operand comes from ONE of the 16
// blocks and is broadcast to the others (cbsz = 4, abid = term % 16), so a lane keeps 144/16 = 9 weight registers;
// the pixel operand is one conflict-free LDS read (lane = pixel) of the DMA-staged haloed tile per instruction.
// Workgroup = 8 rows x 64 columns of one image, each of the 4 waves takes two rows.
// ---------------------------------------------------------------------------------------------------------
namespace small {
constexpr int TY = 8, TX = 64, HY = TY + 2, HX = TX + 2, RAW = HY * HX, NCH = (RAW + 63) / 64, XS = NCH * 64;
constexpr int CI = 16, LDS_BYTES = CI * XS * 4;      // 16 channels x 9 taps = 144 terms per output
}  // namespace small

// Two vertically adjacent output rows per wave: input row ir (0..3 relative to the upper output row) is tap row
// dy = ir of the upper and dy = ir - 1 of the lower output row, so the 4 x 3 reads per channel feed 18 instructions
// (12 LDS reads instead of 18).  STEP enumerates (channel, input row, dx).
template <int STEP>
__device__ __forceinline__ void small_terms(const float* __restrict__ sx, int lane, const float* wreg, f32x4& acc0,
                                            f32x4& acc1) {
    if constexpr (STEP < small::CI * 12) {
        constexpr int c = STEP / 12, ir = (STEP % 12) / 3, dx = STEP % 3;
        const float b = sx[c * small::XS + ir * small::HX + dx + lane];
        if constexpr (ir <= 2) {
            constexpr int term = c * 9 + ir * 3 + dx;
            acc0 = __builtin_amdgcn_mfma_f32_4x4x1f32(wreg[term / 16], b, acc0, 4, term % 16, 0);
        }
        if constexpr (ir >= 1) {
            constexpr int term = c * 9 + (ir - 1) * 3 + dx;
            acc1 = __builtin_amdgcn_mfma_f32_4x4x1f32(wreg[term / 16], b, acc1, 4, term % 16, 0);
        }
        small_terms<STEP + 1>(sx, lane, wreg, acc0, acc1);
    }
}

__global__ __launch_bounds__(256) void conv_small_cout_kernel(const ConvFwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float mis_small_lds[];
    float* const sx = mis_small_lds;
    const unsigned L = mis_xcd_remap(blockIdx.x, a.n_blocks_padded);
    if (L >= a.n_blocks) return;
    const int tx = L % a.tiles_x, ty = (L / a.tiles_x) % a.tiles_y, n = L / (a.tiles_x * a.tiles_y);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int y0 = ty * small::TY, x0 = tx * small::TX;
    const unsigned s_bytes = (unsigned)(a.H * a.W) * 4u;
    const i32x4 rx = make_rsrc(a.x + (long long)n * a.x_bs, (unsigned)a.Cin * s_bytes);
    const unsigned lds0 = lds_addr(sx);
#pragma unroll
    for (int p = 0; p < small::NCH; ++p) {
        const int e = p * 64 + lane;
        const int hy = e / small::HX, hx = e - hy * small::HX;
        const int gy = y0 + hy - 1, gx = x0 + hx - 1;
        const bool ok = e < small::RAW && (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W;
        const unsigned vo = ok ? (unsigned)(gy * a.W + gx) * 4u : OOB;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = wave + 4 * i;      // channels >= Cin lie beyond the descriptor: zeros
            dma_dword(lds0 + (unsigned)(c * small::XS + p * 64) * 4u, vo + (unsigned)c * s_bytes, rx);
        }
    }
    // weights of this lane: output channel lane%4 of terms 16k + lane/4 (packed layout wp[(ci*9 + tap)*Cout_pad + co])
    float wreg[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        const int term = 16 * k + (lane >> 2);
        const int ci = term / 9;
        wreg[k] = ci < a.Cin_pad ? a.wp[(long long)term * a.Cout_pad + (lane & 3)] : 0.f;
    }
    dma_wait();
    __syncthreads();
    f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
    small_terms<0>(sx + (wave * 2) * small::HX, lane, wreg, acc[0], acc[1]);
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int gy = y0 + wave * 2 + r, gx = x0 + lane;
        if (gy < a.H && gx < a.W) {
            float* __restrict__ o = a.y + (long long)n * a.y_bs + (long long)gy * a.W + gx;
#pragma unroll
            for (int m = 0; m < 4; ++m)
                if (m < a.Cout) o[(long long)m * a.H * a.W] = acc[r][m] + (a.bias ? a.bias[m] : 0.f);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// First layer of the 3-D networks: Conv3d(1 -> 16, k = 3) on the full-resolution volume (reference unet_3D.py:28 conv1,
// vnet.py:123 block_one, unetr.py encoder1).  In the generic form the single input channel is padded to the MFMA's K = 4
// (27 MFMAs per 16 voxels x 16 channels, three quarters of them on zeros).  Here the 27 taps ARE the contraction:
//   D[co][voxel] = sum_tap w[co][tap] * x[voxel + tap]:   7 MFMAs (K = 28) per 16 voxels
//   A[i = lane&15][k = lane>>4] = w[co = i][tap 4 g + k]           (7 registers, loaded once)
//   B[k = lane>>4][j = lane&15] = x[voxel j + offset(tap 4 g + k)]  (one ds_read_b32 of the haloed tile per MFMA)
// Same tiles (4 x 8 x 16 voxels) and the same per-tile (sum, sumsq) epilogue as Cfg<3,3,3,4,8,16,16,4,8>, so the
// statistics plumbing does not change.  A wave takes one z plane: 8 rows of 16 voxels.
// ---------------------------------------------------------------------------------------------------------
// KD = 3: Conv3d(1 -> 16, 3, pad 1) on tiles of 4 x 8 x 16 voxels, a wave per z plane (27 taps, 7 MFMAs per 16 voxels);
// KD = 1: Conv2d(1 -> 16, 3, pad 1) (reference unet.py:37 in_conv of the 2-D UNet) on tiles of 32 x 16 pixels, a wave per
// 8 rows (9 taps, 3 MFMAs per 16 pixels)
template <int KD>
struct Cin1 {
    static constexpr int TZ = KD == 3 ? 4 : 1, TY = KD == 3 ? 8 : 32, TX = 16, WROWS = 8;     // rows of a wave
    static constexpr int HZ = TZ + KD - 1, HY = TY + 2, HX = TX + 2, HALO = HZ * HY * HX;
    static constexpr int TAPS = 9 * KD, NG = (TAPS + 3) / 4;
};

template <int KD>
__global__ __launch_bounds__(256) void conv_fwd_cin1_kernel(const ConvFwdArgs a) {
    using C = Cin1<KD>;
    __shared__ float sx[C::HALO];
    __shared__ float2 red[4 * 16];
    const unsigned L = mis_xcd_remap(blockIdx.x, a.n_blocks_padded);
    if (L >= a.n_blocks) return;
    unsigned t = L;
    const int tx = t % a.tiles_x; t /= a.tiles_x;
    const int ty = t % a.tiles_y; t /= a.tiles_y;
    const int tz = t % a.tiles_z; t /= a.tiles_z;
    const int n = t;
    const int z0 = tz * C::TZ, y0 = ty * C::TY, x0 = tx * C::TX;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lk = lane >> 4, lj = lane & 15;
    const long long S = (long long)a.D * a.H * a.W;
    const float* __restrict__ xn = a.x + (long long)n * a.x_bs;
    for (int e = tid; e < C::HALO; e += 256) {
        const int hz = e / (C::HY * C::HX), r2 = e - hz * (C::HY * C::HX), hy = r2 / C::HX, hx = r2 - hy * C::HX;
        const int gz = z0 + hz - (KD == 3 ? 1 : 0), gy = y0 + hy - 1, gx = x0 + hx - 1;
        const bool ok = (unsigned)gz < (unsigned)a.D && (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W;
        sx[e] = ok ? xn[((long long)gz * a.H + gy) * a.W + gx] : 0.f;
    }
    // weights (packed [1 -> 4][taps][16]: wp[tap * 16 + co]) and the LDS offset of this lane's tap of every group
    float wa[C::NG];
    int toff[C::NG];
#pragma unroll
    for (int g = 0; g < C::NG; ++g) {
        const int tap = 4 * g + lk;
        wa[g] = tap < C::TAPS ? a.wp[tap * a.Cout_pad + lj] : 0.f;
        const int tc = tap < C::TAPS ? tap : C::TAPS - 1;
        toff[g] = ((tc / 9) * C::HY + (tc / 3) % 3) * C::HX + tc % 3;
    }
    float bv[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) bv[r] = a.bias ? a.bias[lk * 4 + r] : 0.f;
    __syncthreads();
    // this wave's rows: plane z0 + wave (3-D) / rows y0 + 8 wave ... (2-D)
    const int wz = KD == 3 ? wave : 0, wy = KD == 3 ? 0 : wave * C::WROWS;
    const float* __restrict__ sb = sx + (wz * C::HY + wy) * C::HX + lj;
    float* __restrict__ yo = a.y + (long long)n * a.y_bs + ((long long)(z0 + wz) * a.H + y0 + wy) * a.W + x0 + lj;
    float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 2
    for (int row = 0; row < C::WROWS; ++row) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int g = 0; g < C::NG; ++g)
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[g], sb[row * C::HX + toff[g]], acc, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float v = acc[r] + bv[r];
            yo[(long long)(lk * 4 + r) * S + (long long)row * a.W] = v;
            s1[r] += v; s2[r] += v * v;
        }
    }
    if (a.stat) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int o = 1; o < 16; o <<= 1) { s1[r] += __shfl_xor(s1[r], o, 64); s2[r] += __shfl_xor(s2[r], o, 64); }
        if (lj == 0) {
#pragma unroll
            for (int r = 0; r < 4; ++r) red[wave * 16 + lk * 4 + r] = make_float2(s1[r], s2[r]);
        }
        __syncthreads();
        if (tid < 16) {
            const float2 p0 = red[tid], p1 = red[16 + tid], p2 = red[32 + tid], p3 = red[48 + tid];
            const long long tile = ((long long)tz * a.tiles_y + ty) * a.tiles_x + tx;
            a.stat[(long long)tid * a.stat_sc + (long long)n * a.stat_sn + tile] =
                make_float2((p0.x + p1.x) + (p2.x + p3.x), (p0.y + p1.y) + (p2.y + p3.y));
        }
    }
}

// 3 (3x3x3 on a volume), 1 (3x3 on images), 0: not the first-layer form
int cin1_kd(const ConvFwdArgs& a, int kd, int kh, int kw) {
    if (kh != 3 || kw != 3 || a.Cin != 1 || a.Cout != 16) return 0;
    if (kd == 3 && a.D % Cin1<3>::TZ == 0 && a.H % Cin1<3>::TY == 0 && a.W % Cin1<3>::TX == 0) return 3;
    if (kd == 1 && a.D == 1 && a.H % Cin1<1>::TY == 0 && a.W % Cin1<1>::TX == 0) return 1;
    return 0;
}

template <int KD>
int launch_cin1(ConvFwdArgs a, hipStream_t stream) {
    using C = Cin1<KD>;
    a.tiles_z = a.D / C::TZ; a.tiles_y = a.H / C::TY; a.tiles_x = a.W / C::TX;
    const long long nb = (long long)a.N * a.tiles_z * a.tiles_y * a.tiles_x;
    if (nb <= 0 || nb > 0x7fffffffLL) return MIS_ERR_ARG;
    a.n_blocks = (unsigned)nb;
    a.n_blocks_padded = (unsigned)(mis_cdiv(nb, MIS_NUM_XCD) * MIS_NUM_XCD);
    hipLaunchKernelGGL(conv_fwd_cin1_kernel<KD>, dim3(a.n_blocks_padded), dim3(256), 0, stream, a);
    return mis_launch_status();
}

int launch_small_cout(ConvFwdArgs a, hipStream_t stream) {
    a.tiles_y = (int)mis_cdiv(a.H, small::TY);
    a.tiles_x = (int)mis_cdiv(a.W, small::TX);
    const long long nb = (long long)a.N * a.tiles_y * a.tiles_x;
    if (nb <= 0 || nb > 0x7fffffffLL) return MIS_ERR_ARG;
    a.n_blocks = (unsigned)nb;
    a.n_blocks_padded = (unsigned)(mis_cdiv(nb, MIS_NUM_XCD) * MIS_NUM_XCD);
    hipLaunchKernelGGL(conv_small_cout_kernel, dim3(a.n_blocks_padded), dim3(256), small::LDS_BYTES, stream, a);
    return mis_launch_status();
}

template <class C>
int launch_cfg(ConvFwdArgs a, hipStream_t stream) {
    a.tiles_z = (int)mis_cdiv(a.D, C::TZ);
    a.tiles_y = (int)mis_cdiv(a.H, C::TY);
    a.tiles_x = (int)mis_cdiv(a.W, C::TX);
    a.co_blocks = (int)mis_cdiv(a.Cout_pad, C::CO_B);
    const long long nb = (long long)a.N * a.tiles_z * a.tiles_y * a.tiles_x * a.co_blocks;
    if (nb <= 0 || nb > 0x7fffffffLL) return MIS_ERR_ARG;
    a.n_blocks = (unsigned)nb;
    a.n_blocks_padded = (unsigned)(mis_cdiv(nb, MIS_NUM_XCD) * MIS_NUM_XCD);
    static const int stagger = getenv("MIS_CF_STAGGER") ? atoi(getenv("MIS_CF_STAGGER")) : 0;
    static const int stagger_mod = getenv("MIS_CF_STAGGER_MOD") ? atoi(getenv("MIS_CF_STAGGER_MOD")) : 4;
    a.stagger = stagger; a.stagger_mod = stagger_mod;
    static std::atomic<unsigned long long> attr_done{0};   // per instantiation, one bit per device
    if (mis_set_lds_attr(reinterpret_cast<const void*>(&conv_fwd_kernel<C>), C::LDS_BYTES, attr_done) != MIS_OK) return MIS_ERR_LAUNCH;
    hipLaunchKernelGGL(conv_fwd_kernel<C>, dim3(a.n_blocks_padded), dim3(256), C::LDS_BYTES, stream, a);
    return mis_launch_status();
}

}  // namespace

// Packed-weight geometry the kernels expect (see pack.hip / include/mis_hip.h).
extern "C" int mis_conv_cin_pad(int cin) { return (cin + 3) / 4 * 4; }
extern "C" int mis_conv_cout_pad(int cout) { return (cout + 15) / 16 * 16; }

namespace {

// One table for launch and for the profiling label, so both always agree.
int dispatch_fwd(ConvFwdArgs a, int kd, int kh, int kw, hipStream_t stream, char* name, int name_len,
                 long long* stat_tiles = nullptr) {
#define MIS_CF(KD, KH, KW, TZ, TY, TX, COB, CIB, NT)                                                  \
    do {                                                                                              \
        if (stat_tiles) {   /* tiles per image if EVERY workgroup takes the fused-statistics epilogue, else 0 */ \
            const bool all = a.st2 && a.D % TZ == 0 && a.H % TY == 0 && a.W % TX == 0 && a.Cout % COB == 0;  \
            *stat_tiles = all ? (long long)(a.D / TZ) * (a.H / TY) * (a.W / TX) : 0;                  \
            return MIS_OK;                                                                            \
        }                                                                                             \
        if (name) {                                                                                   \
            snprintf(name, name_len, "conv_fwd_kernel<Cfg<%d, %d, %d, %d, %d, %d, %d, %d, %d>>", KD, KH, KW, \
                     TZ, TY, TX, COB, CIB, NT);                                                       \
            return MIS_OK;                                                                            \
        }                                                                                             \
        return launch_cfg<Cfg<KD, KH, KW, TZ, TY, TX, COB, CIB, NT>>(a, stream);                      \
    } while (0)
    // 32-channel blocks unless that would pad the channel count (Cout_pad = 48: the data gradient of
    // unet_3D's 48 -> 16 decoder conv): three exact 16-channel blocks beat 32 + a half-empty 32
    // (measured 2.4 ms vs 3.1 ms for 16 -> 48 at 96^3 x 8).
    const bool wide = a.Cout_pad >= 32 && a.Cout_pad % 32 == 0;
    if (const int c1 = cin1_kd(a, kd, kh, kw)) {      // first layer: the taps as the MFMA's contraction
        if (stat_tiles) {
            *stat_tiles = c1 == 3 ? (long long)(a.D / Cin1<3>::TZ) * (a.H / Cin1<3>::TY) * (a.W / Cin1<3>::TX)
                                  : (long long)(a.H / Cin1<1>::TY) * (a.W / Cin1<1>::TX);
            return MIS_OK;
        }
        if (name) { snprintf(name, name_len, c1 == 3 ? "conv_fwd_cin1_kernel<3>" : "conv_fwd_cin1_kernel<1>"); return MIS_OK; }
        return c1 == 3 ? launch_cin1<3>(a, stream) : launch_cin1<1>(a, stream);
    }
    if (kd == 3 && kh == 3 && kw == 3) {
        if (a.W % 16 == 0 || a.W >= 64) {
            if (wide) MIS_CF(3, 3, 3, 4, 8, 16, 32, 4, 8); else MIS_CF(3, 3, 3, 4, 8, 16, 16, 4, 8);
        } else if (a.W % 8 == 0 && a.W >= 16) {
            // 24^3 volumes: 8x8x8 tiles, or 4x8x8 when those would not fill the 512 resident slots
            // (the 4-volume teacher batch: 216 workgroups)
            const long long nb888 = (long long)a.N * mis_cdiv(a.D, 8) * mis_cdiv(a.H, 8) * mis_cdiv(a.W, 8) *
                                    mis_cdiv(a.Cout_pad, wide ? 32 : 16);
            if (nb888 < 512) {
                if (wide) MIS_CF(3, 3, 3, 4, 8, 8, 32, 4, 4); else MIS_CF(3, 3, 3, 4, 8, 8, 16, 4, 4);
            }
            if (wide) MIS_CF(3, 3, 3, 8, 8, 8, 32, 4, 8); else MIS_CF(3, 3, 3, 8, 8, 8, 16, 4, 8);
        } else if (a.W > 12) {
            if (wide) MIS_CF(3, 3, 3, 4, 4, 16, 32, 4, 4); else MIS_CF(3, 3, 3, 4, 4, 16, 16, 4, 4);
        } else if (a.W > 8) {
            // deep, small-volume layers (12^3): few pixels, many channels -> small tiles and 16-channel
            // blocks so that the grid still covers the 256 CUs several times over
            MIS_CF(3, 3, 3, 2, 4, 16, 16, 4, 2);
        } else {
            MIS_CF(3, 3, 3, 2, 8, 8, 16, 4, 2);
        }
    }
    if (kd == 1 && kh == 3 && kw == 3) {
        if (a.D != 1) return MIS_ERR_UNSUPPORTED;
        if (a.Cout <= 4 && a.Cin_pad <= small::CI && a.H * (long long)a.W * (a.Cin + 1) * 4 < (1LL << 31)) {
            if (stat_tiles) { *stat_tiles = 0; return MIS_OK; }      // no fused statistics in this form
            if (name) { snprintf(name, name_len, "conv_small_cout_kernel"); return MIS_OK; }
            return launch_small_cout(a, stream);
        }
        // 16x32 tiles only for the large images: at 64^2 and below they leave too few workgroups (a 24-image
        // teacher batch at 32^2 gives 192 for 512 resident slots); 16x16 tiles double the count
        // (config 2: 3116 -> 3180 images/s).
        if (a.W >= 128) {
            if (wide) MIS_CF(1, 3, 3, 1, 16, 32, 32, 8, 8); else MIS_CF(1, 3, 3, 1, 16, 32, 16, 8, 8);
        } else {
            if (wide) MIS_CF(1, 3, 3, 1, 16, 16, 32, 8, 4); else MIS_CF(1, 3, 3, 1, 16, 16, 16, 8, 4);
        }
    }
    if (kd == 1 && kh == 1 && kw == 1) {
        if (a.D > 1) {
            if (wide) MIS_CF(1, 1, 1, 4, 8, 16, 32, 16, 8); else MIS_CF(1, 1, 1, 4, 8, 16, 16, 16, 8);
        } else if (a.W >= 32) {
            if (wide) MIS_CF(1, 1, 1, 1, 16, 32, 32, 16, 8); else MIS_CF(1, 1, 1, 1, 16, 32, 16, 16, 8);
        } else {
            if (wide) MIS_CF(1, 1, 1, 1, 16, 16, 32, 16, 4); else MIS_CF(1, 1, 1, 1, 16, 16, 16, 16, 4);
        }
    }
#undef MIS_CF
    return MIS_ERR_UNSUPPORTED;
}

ConvFwdArgs make_fwd_args(const float* x, long long x_bs, const float* wp, const float* bias, float* y,
                          long long y_bs, int N, int Cin, int Cout, int D, int H, int W) {
    ConvFwdArgs a{};
    a.x = x; a.x_bs = x_bs; a.wp = wp; a.bias = bias; a.y = y; a.y_bs = y_bs;
    a.N = N; a.Cin = Cin; a.Cout = Cout; a.D = D; a.H = H; a.W = W;
    a.Cin_pad = mis_conv_cin_pad(Cin);
    a.Cout_pad = mis_conv_cout_pad(Cout);
    a.st2 = (W % 2 == 0 && y_bs % 2 == 0 && ((uintptr_t)y & 7) == 0) ? 1 : 0;
    return a;
}

}  // namespace

extern "C" int mis_conv_fwd(const float* x, long long x_bs, const float* wp, const float* bias,
                            float* y, long long y_bs, int N, int Cin, int Cout, int D, int H, int W,
                            int kd, int kh, int kw, hipStream_t stream) {
    if (!x || !wp || !y || N <= 0 || Cin <= 0 || Cout <= 0 || D <= 0 || H <= 0 || W <= 0) return MIS_ERR_ARG;
    const long long S = (long long)D * H * W;
    if (x_bs < (long long)Cin * S || y_bs < (long long)Cout * S) return MIS_ERR_ARG;
    // the DMA descriptors address one image's channels / the packed weights with 32-bit byte offsets
    if (((long long)Cin + 32) * S * 4 >= (1LL << 30) || ((uintptr_t)wp & 15) != 0) return MIS_ERR_UNSUPPORTED;
    return dispatch_fwd(make_fwd_args(x, x_bs, wp, bias, y, y_bs, N, Cin, Cout, D, H, W), kd, kh, kw, stream,
                        nullptr, 0);
}

// Number of partial-statistics tiles per image the fused form below writes for this geometry, or 0 when the geometry
// is not eligible (some workgroup would take the generic epilogue: ragged tiles / channel blocks, odd W).
extern "C" long long mis_conv_fwd_stat_tiles(int N, int Cin, int Cout, int D, int H, int W, int kd, int kh, int kw) {
    if (N <= 0 || Cin <= 0 || Cout <= 0 || D <= 0 || H <= 0 || W <= 0) return MIS_ERR_ARG;
    long long tiles = 0;
    const int st = dispatch_fwd(make_fwd_args(nullptr, 0, nullptr, nullptr, nullptr, 0, N, Cin, Cout, D, H, W), kd, kh,
                                kw, nullptr, nullptr, 0, &tiles);
    return st ? st : tiles;
}

// mis_conv_fwd that also emits, per (channel, image, tile), the (sum, sum of squares) of its output:
//   stat[co * stat_sc + n * stat_sn + tile] = float2, tile < mis_conv_fwd_stat_tiles(...)
// so that the normalisation consuming y needs no statistics pass of its own (mis_norm_stats_finalize).
// BatchNorm: stat_sc = N*T, stat_sn = T; InstanceNorm: stat_sc = T, stat_sn = Cout*T.
extern "C" int mis_conv_fwd_stats(const float* x, long long x_bs, const float* wp, const float* bias, float* y,
                                  long long y_bs, int N, int Cin, int Cout, int D, int H, int W, int kd, int kh, int kw,
                                  float* stat, long long stat_sc, long long stat_sn, hipStream_t stream) {
    if (!x || !wp || !y || !stat || N <= 0 || Cin <= 0 || Cout <= 0 || D <= 0 || H <= 0 || W <= 0) return MIS_ERR_ARG;
    const long long S = (long long)D * H * W;
    if (x_bs < (long long)Cin * S || y_bs < (long long)Cout * S) return MIS_ERR_ARG;
    if (((long long)Cin + 32) * S * 4 >= (1LL << 30) || ((uintptr_t)wp & 15) != 0 || ((uintptr_t)stat & 7) != 0)
        return MIS_ERR_UNSUPPORTED;
    ConvFwdArgs a = make_fwd_args(x, x_bs, wp, bias, y, y_bs, N, Cin, Cout, D, H, W);
    long long tiles = 0;
    int st = dispatch_fwd(a, kd, kh, kw, nullptr, nullptr, 0, &tiles);
    if (st) return st;
    if (tiles <= 0) return MIS_ERR_UNSUPPORTED;   // the caller must use mis_conv_fwd + mis_norm_stats here
    a.stat = reinterpret_cast<float2*>(stat); a.stat_sc = stat_sc; a.stat_sn = stat_sn;
    return dispatch_fwd(a, kd, kh, kw, stream, nullptr, 0);
}

// Name of the kernel instantiation mis_conv_fwd would launch for this geometry (as rocprofv3 prints
// it, minus the anonymous-namespace prefix): lets bench.py attribute event timings to kernels.
extern "C" int mis_conv_fwd_kernel_name(int N, int Cin, int Cout, int D, int H, int W, int kd, int kh, int kw,
                                        char* name, int name_len) {
    if (!name || name_len <= 0 || N <= 0 || Cin <= 0 || Cout <= 0 || D <= 0 || H <= 0 || W <= 0) return MIS_ERR_ARG;
    return dispatch_fwd(make_fwd_args(nullptr, 0, nullptr, nullptr, nullptr, 0, N, Cin, Cout, D, H, W), kd, kh, kw,
                        nullptr, name, name_len);
}
