// Convolution weight-gradient for the UNet / unet_3D / V-Net blocks.
//
// Replaces the autograd backward of nn.Conv2d / nn.Conv3d w.r.t. weight
// (reference code/networks/unet.py:37,41,73,138; code/networks/utils.py:104,107;
//  unet_3D.py:59; vnet.py:16,73,100,175) reached from loss.backward() in train_mean_teacher_{2D,3D}.py.
//
//   dw[co][ci][tap] = sum_{n,p} dy[n][co][p] * x[n][ci][p + tap - pad]
//
// Design (gfx950): a GEMM with M = 16 output channels, N = 16 input channels
// (one per tap), K = pixels, on v_mfma_f32_16x16x4_f32 (exact fp32 fmaf chain):
//   A[i = lane&15][k = lane>>4] = dy[co0+i][pixel 4q+k]
//   B[k = lane>>4][j = lane&15] = x [ci0+j][pixel 4q+k shifted by tap]
//   D[row -> co][col -> ci], one f32x4 accumulator per tap (27 for 3x3x3).
// A workgroup owns one (co-tile, ci-tile) pair and walks a strided list of
// pixel tiles (split-K); its 4 waves split each tile's pixel quads.  Partials
// go to a caller-provided workspace and are summed by a second kernel in a
// fixed order, so the result is run-to-run deterministic (no float atomics).
//
// The haloed x tile and the dy tile of a pixel tile are brought into LDS by LDS-DMA
// (buffer_load_dword ... lds, 64-dword pieces of one channel, see common.h): a wave issues its whole
// share of a tile back to back and waits once, instead of ~10 serialised register-staged batches
// (61 floats per lane per tile), and halo / channel padding is the descriptor's range check.
// With NBUF = 2 the DMA of the next tile overlaps the MFMAs of the current one.
#include <cstdio>
#include "common.h"
#include <stdlib.h>

namespace {

using namespace mis_dma;

struct WgradArgs {
    const float* x; long long x_bs;
    const float* dy; long long dy_bs;
    float* ws;  // [KS][pairs][TAPS][256]
    int N, Cin, Cout, D, H, W;
    int tiles_z, tiles_y, tiles_x, tiles_total;
    int ci_tiles, pairs, KS;
    unsigned n_blocks_padded;
    int stagger_shift;
    int stagger;   // s_sleep units (64 clocks each) by which every second resident workgroup of a CU starts late
};

template <int KD_, int KH_, int KW_, int TZ_, int TY_, int TX_, int NBUF_>
struct WCfg {
    static constexpr int KD = KD_, KH = KH_, KW = KW_, TZ = TZ_, TY = TY_, TX = TX_, NBUF = NBUF_;
    static constexpr int TAPS = KD * KH * KW;
    static constexpr int HZ = TZ + KD - 1, HY = TY + KH - 1, HX = TX + KW - 1;
    static constexpr int XS_RAW = HZ * HY * HX;
    static constexpr int PIX = TZ * TY * TX;
    static constexpr int NCHX = (XS_RAW + 63) / 64, NCHD = (PIX + 63) / 64;   // 64-dword DMA pieces / channel
    // channel strides = 4 * odd and >= the DMA pieces: operands are fetched with 8-byte LDS reads (banks =
    // dword address mod 64 over a 32-lane group); lane (channel c, k-lane k) reads dwords
    // c*stride + 2k + {0,1}, and c*4*odd mod 64 enumerates the 16 multiples of 4, so the group touches
    // 64 distinct banks.
    static constexpr int XS = NCHX * 64 + 4;
    static constexpr int DS = NCHD * 64 + 4;
    static constexpr int X_FLOATS = 16 * XS, DY_FLOATS = 16 * DS;
    static constexpr int STAGE = X_FLOATS + DY_FLOATS;
    static constexpr int RED_FLOATS = TAPS * 256;
    static constexpr int LDS_FLOATS = NBUF * STAGE > RED_FLOATS ? NBUF * STAGE : RED_FLOATS;
    static constexpr int LDS_BYTES = LDS_FLOATS * 4;
    static_assert(TX % 8 == 0 && PIX % 32 == 0 && HX % 2 == 0, "even/odd pixel-quad pairs");
    static_assert(NBUF == 1 || NBUF == 2, "single or double buffered");
    static_assert(LDS_BYTES <= 160 * 1024, "LDS per CU");
};

extern __shared__ __attribute__((aligned(16))) float mis_wgrad_lds[];

template <class C>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(const WgradArgs a) {
    float* const smem = mis_wgrad_lds;

    // XCD-aware order: the workgroups of one XCD (private L2) take consecutive logical ids, i.e. all
    // channel-tile pairs of one pixel tile, then the neighbouring pixel tile: they share the dy / x tiles
    // and the halos in that L2 instead of each XCD fetching its own copy from HBM.
    const unsigned L = mis_xcd_remap(blockIdx.x, a.n_blocks_padded);
    if (L >= (unsigned)(a.pairs * a.KS)) return;
    const int pair = L % a.pairs, ks = L / a.pairs;
    const int mt = pair / a.ci_tiles, jt = pair % a.ci_tiles;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lk = lane >> 4, lj = lane & 15;
    const long long S = (long long)a.D * a.H * a.W;
    const unsigned s_bytes = (unsigned)S * 4u;
    const unsigned lds0 = lds_addr(smem);
    const int ci0 = jt * 16, co0 = mt * 16;

    f32x4 acc[C::TAPS];
#pragma unroll
    for (int t = 0; t < C::TAPS; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};

    // DMA of pixel tile `tile` into stage buffer `buf`: wave w brings channels w, w+4, w+8, w+12 of both tiles
    auto issue = [&](int tile, int buf) {
        int t = tile;
        const int tx = t % a.tiles_x; t /= a.tiles_x;
        const int ty = t % a.tiles_y; t /= a.tiles_y;
        const int tz = t % a.tiles_z; t /= a.tiles_z;
        const int n = t;
        const int z0 = tz * C::TZ, y0 = ty * C::TY, x0 = tx * C::TX;
        const i32x4 rx = make_rsrc(a.x + (long long)n * a.x_bs, (unsigned)a.Cin * s_bytes);
        const i32x4 rd = make_rsrc(a.dy + (long long)n * a.dy_bs, (unsigned)a.Cout * s_bytes);
        const unsigned st = lds0 + (unsigned)buf * (C::STAGE * 4);
#pragma unroll
        for (int p = 0; p < C::NCHX; ++p) {
            const int e = p * 64 + lane;
            const int hz = e / (C::HY * C::HX), r2 = e - hz * (C::HY * C::HX);
            const int hy = r2 / C::HX, hx = r2 - hy * C::HX;
            const int gz = z0 + hz - C::KD / 2, gy = y0 + hy - C::KH / 2, gx = x0 + hx - C::KW / 2;
            const bool ok = e < C::XS_RAW && (unsigned)gz < (unsigned)a.D && (unsigned)gy < (unsigned)a.H &&
                            (unsigned)gx < (unsigned)a.W;
            const unsigned vo = ok ? (unsigned)((gz * a.H + gy) * a.W + gx) * 4u : OOB;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int c = wave + 4 * i;   // channel >= Cin: beyond num_records -> zeros
                dma_dword(st + (unsigned)(c * C::XS + p * 64) * 4u, vo + (unsigned)(ci0 + c) * s_bytes, rx);
            }
        }
#pragma unroll
        for (int p = 0; p < C::NCHD; ++p) {
            const int e = p * 64 + lane;
            const int px = e % C::TX, py = (e / C::TX) % C::TY, pz = e / (C::TX * C::TY);
            const int gz = z0 + pz, gy = y0 + py, gx = x0 + px;
            const bool ok = e < C::PIX && gz < a.D && gy < a.H && gx < a.W;
            const unsigned vo = ok ? (unsigned)((gz * a.H + gy) * a.W + gx) * 4u : OOB;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int c = wave + 4 * i;
                dma_dword(st + (unsigned)(C::X_FLOATS + c * C::DS + p * 64) * 4u, vo + (unsigned)(co0 + c) * s_bytes,
                          rd);
            }
        }
    };

    // 8 consecutive pixels per step = an "even" K-quad (pixels p0+2k) and an "odd" one (p0+2k+1):
    // one 8-byte LDS read at pixel p0+2k feeds both quads (and two kx taps), as in conv_fwd.hip.
    auto compute = [&](const float* st) {
        const float2* __restrict__ s_x2 = reinterpret_cast<const float2*>(st);
        const float2* __restrict__ s_dy2 = reinterpret_cast<const float2*>(st + C::X_FLOATS);
        for (int g = wave; g < C::PIX / 8; g += 4) {
            const int p0 = g * 8;
            const int px0 = p0 % C::TX, py = (p0 / C::TX) % C::TY, pz = p0 / (C::TX * C::TY);
            const float2 av = s_dy2[(lj * C::DS + p0 + 2 * lk) >> 1];
            const int xb2 = (lj * C::XS + (pz * C::HY + py) * C::HX + px0 + 2 * lk) >> 1;
            float be[C::TAPS], bo[C::TAPS];
#pragma unroll
            for (int row = 0; row < C::KD * C::KH; ++row) {
                const int kz = row / C::KH, ky = row % C::KH;
                const int ro2 = ((kz * C::HY + ky) * C::HX) >> 1;
                const float2 r0 = s_x2[xb2 + ro2];
                float2 r2 = r0;
                if (C::KW == 3) r2 = s_x2[xb2 + ro2 + 1];
#pragma unroll
                for (int kx = 0; kx < C::KW; ++kx) {
                    be[row * C::KW + kx] = kx == 0 ? r0.x : (kx == 1 ? r0.y : r2.x);
                    bo[row * C::KW + kx] = kx == 0 ? r0.y : (kx == 1 ? r2.x : r2.y);
                }
            }
#pragma unroll
            for (int tap = 0; tap < C::TAPS; ++tap)
                acc[tap] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, be[tap], acc[tap], 0, 0, 0);
#pragma unroll
            for (int tap = 0; tap < C::TAPS; ++tap)
                acc[tap] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, bo[tap], acc[tap], 0, 0, 0);
        }
    };

    if constexpr (C::NBUF == 2) {
        // Software pipeline over this workgroup's tiles: DMA(i+1) || MFMA(i), one barrier per tile.  The DMA
        // instructions of tile i+1 (and their address arithmetic) are ISSUED between the MFMAs of the first
        // half of tile i.  NOT selected by the dispatch table this round: the second stage buffer halves the
        // tile (LDS) and, for 3x3x3, the unrolled groups push the kernel to 1 wave/SIMD; measured 93 TF (3-D)
        // and 94 TF (2-D) against 105 / 95 TF for the single-buffered NBUF = 1 path with full-size tiles.
        // (Removing the DMA altogether gives 132 / 119 TF: the copy still costs ~20 %)
        constexpr int GI = C::PIX / 32;                       // pixel groups per wave and tile
        constexpr int NPC = C::NCHX + C::NCHD;                // DMA pieces per channel (x then dy)
        constexpr int ITEMS = 4 * NPC;                        // DMA instructions per wave and tile
        constexpr int GISSUE = GI > 1 ? GI / 2 : 1;           // groups that carry DMA issue: the first half of a
                                                              // tile, so the rest of its MFMAs cover the latency
        constexpr int IPG = (ITEMS + GISSUE - 1) / GISSUE;    // ... per issuing pixel group
        constexpr int NM = 2 * C::TAPS;                       // MFMAs per pixel group
        if (ks < a.tiles_total) issue(ks, 0);
        int i = 0;
        for (int tile = ks; tile < a.tiles_total; tile += a.KS, ++i) {
            dma_wait();
            __syncthreads();   // tile i has landed for every wave; everyone is done reading the other buffer
            const int nxt = tile + a.KS;
            const bool has_next = nxt < a.tiles_total;
            int t = has_next ? nxt : tile;
            const int tx = t % a.tiles_x; t /= a.tiles_x;
            const int ty = t % a.tiles_y; t /= a.tiles_y;
            const int tz = t % a.tiles_z; t /= a.tiles_z;
            const int z0 = tz * C::TZ, y0 = ty * C::TY, x0 = tx * C::TX;
            const i32x4 rx = make_rsrc(a.x + (long long)t * a.x_bs, (unsigned)a.Cin * s_bytes);
            const i32x4 rd = make_rsrc(a.dy + (long long)t * a.dy_bs, (unsigned)a.Cout * s_bytes);
            const unsigned nst = lds0 + (unsigned)((i + 1) & 1) * (C::STAGE * 4);
            unsigned vo = 0;
            // DMA instruction `item` of the next tile: piece j = item / 4 (x pieces first), channel wave + 4*(item % 4)
            auto issue_item = [&](int item) {
                const int j = item / 4, ic = item % 4;
                const int c = wave + 4 * ic;
                if (j < C::NCHX) {
                    if (ic == 0) {
                        const int e = j * 64 + lane;
                        const int hz = e / (C::HY * C::HX), r2 = e - hz * (C::HY * C::HX);
                        const int hy = r2 / C::HX, hx = r2 - hy * C::HX;
                        const int gz = z0 + hz - C::KD / 2, gy = y0 + hy - C::KH / 2, gx = x0 + hx - C::KW / 2;
                        const bool ok = e < C::XS_RAW && (unsigned)gz < (unsigned)a.D &&
                                        (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W;
                        vo = ok ? (unsigned)((gz * a.H + gy) * a.W + gx) * 4u : OOB;
                    }
                    dma_dword(nst + (unsigned)(c * C::XS + j * 64) * 4u, vo + (unsigned)(ci0 + c) * s_bytes, rx);
                } else {
                    const int p = j - C::NCHX;
                    if (ic == 0) {
                        const int e = p * 64 + lane;
                        const int px = e % C::TX, py = (e / C::TX) % C::TY, pz = e / (C::TX * C::TY);
                        const int gz = z0 + pz, gy = y0 + py, gx = x0 + px;
                        const bool ok = e < C::PIX && gz < a.D && gy < a.H && gx < a.W;
                        vo = ok ? (unsigned)((gz * a.H + gy) * a.W + gx) * 4u : OOB;
                    }
                    dma_dword(nst + (unsigned)(C::X_FLOATS + c * C::DS + p * 64) * 4u,
                              vo + (unsigned)(co0 + c) * s_bytes, rd);
                }
            };
            const float* st = smem + (i & 1) * C::STAGE;
            const float2* __restrict__ s_x2 = reinterpret_cast<const float2*>(st);
            const float2* __restrict__ s_dy2 = reinterpret_cast<const float2*>(st + C::X_FLOATS);
#pragma unroll
            for (int gi = 0; gi < GI; ++gi) {
                const int p0 = (wave + 4 * gi) * 8;
                const int px0 = p0 % C::TX, py = (p0 / C::TX) % C::TY, pz = p0 / (C::TX * C::TY);
                const float2 av = s_dy2[(lj * C::DS + p0 + 2 * lk) >> 1];
                const int xb2 = (lj * C::XS + (pz * C::HY + py) * C::HX + px0 + 2 * lk) >> 1;
                float be[C::TAPS], bo[C::TAPS];
#pragma unroll
                for (int row = 0; row < C::KD * C::KH; ++row) {
                    const int kz = row / C::KH, ky = row % C::KH;
                    const int ro2 = ((kz * C::HY + ky) * C::HX) >> 1;
                    const float2 r0 = s_x2[xb2 + ro2];
                    float2 r2 = r0;
                    if (C::KW == 3) r2 = s_x2[xb2 + ro2 + 1];
#pragma unroll
                    for (int kx = 0; kx < C::KW; ++kx) {
                        be[row * C::KW + kx] = kx == 0 ? r0.x : (kx == 1 ? r0.y : r2.x);
                        bo[row * C::KW + kx] = kx == 0 ? r0.y : (kx == 1 ? r2.x : r2.y);
                    }
                }
                int k = 0;   // next DMA item of this group (compile-time after unrolling)
#pragma unroll
                for (int mi = 0; mi < NM; ++mi) {
                    const int tap = mi % C::TAPS;
                    acc[tap] = __builtin_amdgcn_mfma_f32_16x16x4f32(mi < C::TAPS ? av.x : av.y,
                                                                    mi < C::TAPS ? be[tap] : bo[tap], acc[tap], 0, 0, 0);
                    // spread this group's IPG DMA instructions evenly over its NM MFMAs
                    if (k < IPG && gi * IPG + k < ITEMS && mi == (k * NM) / IPG) {
                        if (has_next) issue_item(gi * IPG + k);
                        __builtin_amdgcn_sched_barrier(0);
                        ++k;
                    }
                }
                __builtin_amdgcn_sched_barrier(0);   // no hoisting of the next group's operand reads (registers)
            }
        }
    } else {
        // Two workgroups share a CU and each alternates "DMA a tile" / "MFMA over it".  Started together they stay in
        // phase: both copy (matrix pipe idle), then both compute (sharing the pipe) -- measured 70 % MFMA-busy.  Starting
        // the second resident workgroup of a CU (dispatch order: b and b + 256 land on the same CU) half a period late
        // lets one workgroup's copy run under the other's MFMAs.  Timing only, never results.
        if (a.stagger > 0 && ((blockIdx.x >> a.stagger_shift) & 1)) {
            for (int i = 0; i < a.stagger; i += 127) __builtin_amdgcn_s_sleep(127);
        }
        for (int tile = ks; tile < a.tiles_total; tile += a.KS) {
            __syncthreads();   // previous tile fully consumed
            issue(tile, 0);
            dma_wait();
            __syncthreads();
            compute(smem);
        }
    }

    // ---- combine the 4 waves through LDS (wave 0 accumulates), then write the partial ----
#pragma unroll 1
    for (int w = 1; w < 4; ++w) {
        __syncthreads();
        if (wave == w) {
#pragma unroll
            for (int t = 0; t < C::TAPS; ++t) *reinterpret_cast<f32x4*>(&smem[(t * 64 + lane) * 4]) = acc[t];
        }
        __syncthreads();
        if (wave == 0) {
#pragma unroll
            for (int t = 0; t < C::TAPS; ++t) acc[t] += *reinterpret_cast<const f32x4*>(&smem[(t * 64 + lane) * 4]);
        }
    }
    if (wave == 0) {
        float* __restrict__ out = a.ws + ((long long)ks * a.pairs + pair) * (C::TAPS * 256);
#pragma unroll
        for (int t = 0; t < C::TAPS; ++t) *reinterpret_cast<f32x4*>(&out[(t * 64 + lane) * 4]) = acc[t];
    }
}

struct WredArgs {
    const float* ws;
    float* dw;  // [Cout][Cin][TAPS]
    int Cin, Cout, TAPS, ci_tiles, pairs, KS, accumulate;
};

// One workgroup of 16 waves per (channel-tile pair, tap): wave w sums the partial tiles k = w, w+16, ... as
// coalesced float4 rows (a partial tile is the MFMA accumulator image: lane -> 4 consecutive floats), the 16 wave
// sums are combined through LDS in a fixed order, and lane l writes its four dw entries
// (co = 16*mt + 4*(l>>4) + r, ci = 16*jt + (l&15)).  Deterministic; ~2x faster than one thread per output with
// strided 4-byte reads when KS is in the hundreds.
constexpr int RED_WAVES = 16;

__global__ __launch_bounds__(64 * RED_WAVES) void conv_wgrad_reduce_kernel(const WredArgs a) {
    __shared__ f32x4 red[RED_WAVES][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int pair = blockIdx.x / a.TAPS, tap = blockIdx.x - pair * a.TAPS;
    const float* __restrict__ base = a.ws + ((long long)pair * a.TAPS + tap) * 256 + lane * 4;
    const long long stride = (long long)a.pairs * a.TAPS * 256;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    for (int k = wave; k < a.KS; k += RED_WAVES) s += *reinterpret_cast<const f32x4*>(base + k * stride);
    red[wave][lane] = s;
    __syncthreads();
    if (wave != 0) return;
#pragma unroll
    for (int w = 1; w < RED_WAVES; ++w) s += red[w][lane];
    const int mt = pair / a.ci_tiles, jt = pair - mt * a.ci_tiles;
    const int ci = jt * 16 + (lane & 15);
    if (ci >= a.Cin) return;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int co = mt * 16 + (lane >> 4) * 4 + r;
        if (co < a.Cout) {
            float* p = a.dw + ((long long)co * a.Cin + ci) * a.TAPS + tap;
            *p = a.accumulate ? *p + s[r] : s[r];
        }
    }
}

template <class C>
void fill_tiles(WgradArgs& a) {
    a.tiles_z = (int)mis_cdiv(a.D, C::TZ);
    a.tiles_y = (int)mis_cdiv(a.H, C::TY);
    a.tiles_x = (int)mis_cdiv(a.W, C::TX);
    a.tiles_total = a.N * a.tiles_z * a.tiles_y * a.tiles_x;
}

template <class C>
int launch_wgrad(WgradArgs a, float* dw, int accumulate, hipStream_t stream) {
    static std::atomic<unsigned long long> attr_done{0};   // per instantiation, one bit per device
    if (mis_set_lds_attr(reinterpret_cast<const void*>(&conv_wgrad_kernel<C>), C::LDS_BYTES, attr_done) != MIS_OK) return MIS_ERR_LAUNCH;
    a.n_blocks_padded = (unsigned)(mis_cdiv((long long)a.pairs * a.KS, MIS_NUM_XCD) * MIS_NUM_XCD);
    static const int stagger = getenv("MIS_WG_STAGGER") ? atoi(getenv("MIS_WG_STAGGER")) : 0;
    a.stagger = stagger;
    static const int stagger_shift = getenv("MIS_WG_STAGGER_SHIFT") ? atoi(getenv("MIS_WG_STAGGER_SHIFT")) : 8;
    a.stagger_shift = stagger_shift;
    hipLaunchKernelGGL(conv_wgrad_kernel<C>, dim3(a.n_blocks_padded), dim3(256), C::LDS_BYTES, stream, a);
    int st = mis_launch_status();
    if (st) return st;
    WredArgs r{a.ws, dw, a.Cin, a.Cout, C::TAPS, a.ci_tiles, a.pairs, a.KS, accumulate};
    hipLaunchKernelGGL(conv_wgrad_reduce_kernel, dim3((unsigned)(a.pairs * C::TAPS)), dim3(64 * RED_WAVES), 0, stream, r);
    return mis_launch_status();
}

// Split-K factor.  All workgroups of a launch do the same work, so the launch takes
//   rounds(ks) x per-workgroup time = ceil(pairs*ks / slots) x (ceil(tiles/ks) + E)
// where slots = resident workgroups on the chip and E = the fixed cost of a workgroup (prologue, 4-wave
// reduction, partial write) in tile units.  The minimum over ks avoids a mostly-empty last round
// (e.g. 1024 workgroups on 768 slots run as long as 1536 would).  Pure function of its arguments:
// the workspace query and the launch always agree.
int pick_ks(int pairs, int tiles_total, int slots) {
    struct Memo { int pairs, tiles, slots, ks; };
    static thread_local Memo memo[64];
    static thread_local int n_memo = 0;
    for (int i = 0; i < n_memo; ++i)
        if (memo[i].pairs == pairs && memo[i].tiles == tiles_total && memo[i].slots == slots) return memo[i].ks;
    const int E = 3;
    int kmax = tiles_total < 2048 ? tiles_total : 2048;
    if (kmax < 1) kmax = 1;
    long long best_t = -1;
    int best = 1;
    for (int ks = 1; ks <= kmax; ++ks) {
        const long long rounds = ((long long)pairs * ks + slots - 1) / slots;
        const long long t = rounds * ((tiles_total + ks - 1) / ks + E);
        if (best_t < 0 || t < best_t) { best_t = t; best = ks; }
    }
    if (n_memo < 64) memo[n_memo++] = Memo{pairs, tiles_total, slots, best};
    return best;
}

// resident workgroups per CU of this instantiation: LDS (160 KiB / CU) and registers (512 / SIMD lane)
template <class C>
int slots_on_chip() {
    const int by_lds = (160 * 1024) / (C::LDS_BYTES > 1024 ? C::LDS_BYTES : 1024);
    const int by_regs = C::TAPS >= 27 ? 2 : (C::TAPS >= 9 ? 5 : 8);   // accumulators: TAPS x 4 AGPRs (+ operands)
    const int per_cu = by_lds < by_regs ? by_lds : by_regs;
    return 256 * (per_cu < 1 ? 1 : per_cu);
}

template <class C>
long long ws_floats(WgradArgs a) {
    fill_tiles<C>(a);
    const int ks = pick_ks(a.pairs, a.tiles_total, slots_on_chip<C>());
    return (long long)ks * a.pairs * C::TAPS * 256;
}

template <class C>
int run(WgradArgs a, float* dw, long long ws_bytes, int accumulate, hipStream_t stream) {
    fill_tiles<C>(a);
    a.KS = pick_ks(a.pairs, a.tiles_total, slots_on_chip<C>());
    if ((long long)a.KS * a.pairs * C::TAPS * 256 * 4 > ws_bytes) return MIS_ERR_WORKSPACE;
    return launch_wgrad<C>(a, dw, accumulate, stream);
}

#ifndef MIS_WG_3D_MAIN
#define MIS_WG_3D_MAIN 3, 3, 3, 2, 8, 16, 1
#endif
#ifndef MIS_WG_2D_MAIN
#define MIS_WG_2D_MAIN 1, 3, 3, 1, 8, 32, 1
#endif

// mode 0: workspace query (returns floats through *out_ws), mode 1: run
#define MIS_WG_STR2(...) #__VA_ARGS__
#define MIS_WG_STR(...) MIS_WG_STR2(__VA_ARGS__)
int dispatch(WgradArgs a, int kd, int kh, int kw, float* dw, long long ws_bytes, int accumulate,
             hipStream_t stream, long long* out_ws, const char** out_name = nullptr) {
#define MIS_WG(...)                                                              \
    do {                                                                         \
        using C_ = WCfg<__VA_ARGS__>;                                            \
        if (out_name) { *out_name = "conv_wgrad_kernel<WCfg<" MIS_WG_STR(__VA_ARGS__) "> >"; return MIS_OK; } \
        if (out_ws) { *out_ws = ws_floats<C_>(a) * 4; return MIS_OK; }           \
        return run<C_>(a, dw, ws_bytes, accumulate, stream);                     \
    } while (0)
    if (kd == 3 && kh == 3 && kw == 3) {
        if (a.W % 16 == 0 || a.W >= 64) MIS_WG(MIS_WG_3D_MAIN);
        else if (a.W > 12) MIS_WG(3, 3, 3, 4, 8, 8, 1);
        else if (a.W > 8) MIS_WG(3, 3, 3, 2, 12, 8, 1);   // 12^3 volumes: 75 % tile efficiency instead of 56 %
        else MIS_WG(3, 3, 3, 2, 6, 8, 1);                 // 6^3 volumes
    }
    if (kd == 1 && kh == 3 && kw == 3) {
        if (a.D != 1) return MIS_ERR_UNSUPPORTED;
        if (a.W >= 32) MIS_WG(MIS_WG_2D_MAIN);
        else MIS_WG(1, 3, 3, 1, 16, 16, 1);
    }
    if (kd == 1 && kh == 1 && kw == 1) {
        if (a.D > 1) MIS_WG(1, 1, 1, 2, 8, 16, 1);
        else if (a.W >= 32) MIS_WG(1, 1, 1, 1, 8, 32, 1);
        else MIS_WG(1, 1, 1, 1, 16, 16, 1);
    }
#undef MIS_WG
    return MIS_ERR_UNSUPPORTED;
}

WgradArgs make_args(const float* x, long long x_bs, const float* dy, long long dy_bs, float* ws, int N,
                    int Cin, int Cout, int D, int H, int W) {
    WgradArgs a{};
    a.x = x; a.x_bs = x_bs; a.dy = dy; a.dy_bs = dy_bs; a.ws = ws;
    a.N = N; a.Cin = Cin; a.Cout = Cout; a.D = D; a.H = H; a.W = W;
    a.ci_tiles = (Cin + 15) / 16;
    a.pairs = ((Cout + 15) / 16) * a.ci_tiles;
    return a;
}

}  // namespace

// conv_wgrad_cin1.hip: the first layer of the 3-D networks (1 -> 16 channels, taps on the MFMA's N side)
bool mis_wgrad_cin1_eligible(int N, int Cin, int Cout, int D, int H, int W, int kd, int kh, int kw);
long long mis_wgrad_cin1_workspace_bytes(int N, int D, int H, int W);
int mis_wgrad_cin1(const float* x, long long x_bs, const float* dy, long long dy_bs, float* dw, float* ws,
                   long long ws_bytes, int N, int D, int H, int W, int accumulate, hipStream_t stream);

extern "C" long long mis_conv_wgrad_workspace_bytes(int N, int Cin, int Cout, int D, int H, int W, int kd,
                                                    int kh, int kw) {
    if (N <= 0 || Cin <= 0 || Cout <= 0 || D <= 0 || H <= 0 || W <= 0) return MIS_ERR_ARG;
    if (mis_wgrad_cin1_eligible(N, Cin, Cout, D, H, W, kd, kh, kw)) {
        // either kernel may run (the special one needs 16-byte aligned dy): room for both
        WgradArgs g = make_args(nullptr, 0, nullptr, 0, nullptr, N, Cin, Cout, D, H, W);
        long long generic = 0;
        const int st = dispatch(g, kd, kh, kw, nullptr, 0, 0, nullptr, &generic);
        if (st) return st;
        const long long special = mis_wgrad_cin1_workspace_bytes(N, D, H, W);
        return generic > special ? generic : special;
    }
    WgradArgs a = make_args(nullptr, 0, nullptr, 0, nullptr, N, Cin, Cout, D, H, W);
    long long out = 0;
    int st = dispatch(a, kd, kh, kw, nullptr, 0, 0, nullptr, &out);
    return st ? st : out;
}

// kernel mis_conv_wgrad launches for this geometry as rocprofv3 prints it (minus the anonymous-namespace prefix)
extern "C" int mis_conv_wgrad_kernel_name(int N, int Cin, int Cout, int D, int H, int W, int kd, int kh, int kw, char* name,
                                          int name_len) {
    if (!name || name_len <= 0 || N <= 0 || Cin <= 0 || Cout <= 0 || D <= 0 || H <= 0 || W <= 0) return MIS_ERR_ARG;
    if (mis_wgrad_cin1_eligible(N, Cin, Cout, D, H, W, kd, kh, kw)) {
        snprintf(name, name_len, "wgrad_cin1_kernel<%d, false>", kd == 1 ? 1 : 3);
        return MIS_OK;
    }
    WgradArgs a = make_args(nullptr, 0, nullptr, 0, nullptr, N, Cin, Cout, D, H, W);
    const char* n = nullptr;
    const int st = dispatch(a, kd, kh, kw, nullptr, 0, 0, nullptr, nullptr, &n);
    if (st) return st;
    snprintf(name, name_len, "%s", n);
    return MIS_OK;
}

extern "C" int mis_conv_wgrad(const float* x, long long x_bs, const float* dy, long long dy_bs, float* dw,
                              float* workspace, long long workspace_bytes, int N, int Cin, int Cout, int D,
                              int H, int W, int kd, int kh, int kw, int accumulate, hipStream_t stream) {
    if (!x || !dy || !dw || !workspace || N <= 0 || Cin <= 0 || Cout <= 0 || D <= 0 || H <= 0 || W <= 0)
        return MIS_ERR_ARG;
    const long long S = (long long)D * H * W;
    if (x_bs < (long long)Cin * S || dy_bs < (long long)Cout * S) return MIS_ERR_ARG;
    // the DMA descriptors address one image's channels with 32-bit byte offsets
    const long long cmax = (Cin > Cout ? Cin : Cout) + 32;
    if (cmax * S * 4 >= (1LL << 30)) return MIS_ERR_UNSUPPORTED;
    if (mis_wgrad_cin1_eligible(N, Cin, Cout, D, H, W, kd, kh, kw) && ((uintptr_t)dy & 15) == 0 && dy_bs % 4 == 0)
        return mis_wgrad_cin1(x, x_bs, dy, dy_bs, dw, workspace, workspace_bytes, N, D, H, W, accumulate, stream);
    WgradArgs a = make_args(x, x_bs, dy, dy_bs, workspace, N, Cin, Cout, D, H, W);
    return dispatch(a, kd, kh, kw, dw, workspace_bytes, accumulate, stream, nullptr);
}
