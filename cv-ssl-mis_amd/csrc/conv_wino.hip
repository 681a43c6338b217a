// Winograd F(2x2x2, 3x3x3) convolution forward (and, with the flipped / transposed transformed filter, the data
// gradient) for the stride-1 'same' 3x3x3 convolutions of unet_3D / V-Net.
//
// Replaces: nn.Conv3d(k=3, pad=1) of the reference's UnetConv3 / UnetUp3_CT / ConvBlock
//           (code/networks/utils.py:99-123, code/networks/unet_3D.py:28-57, code/networks/vnet.py:15-22).
//
// Why: the direct kernel (conv_fwd.hip) runs the fp32 matrix pipe at 0.77 of its 157 TF peak and the step is bound by
// it.  The minimal-filtering form needs 64 multiplies per 2x2x2 outputs instead of 216 (3.375x fewer MFMA flops); the
// arithmetic stays fp32 end to end (the transforms only add / subtract inputs and outputs; the filter transform has
// the factors 1/2), so the result differs from the direct form by fp32 rounding only (tests: tolerance vs fp64).
//
//   Y = A^T [ sum_ci (G g G^T) (.) (B^T d B) ] A      per dimension:
//   B^T d = (d0 - d2, d1 + d2, d2 - d1, d1 - d3)      input tile of 4 (2 outputs + halo)
//   G g   = (g0, (g0 + g1 + g2) / 2, (g0 - g1 + g2) / 2, g2)
//   A^T m = (m0 + m1 + m2, m1 - m2 - m3)
//
// Mapping (gfx950): a wave owns a GROUP of 16 output tiles (2x2x2 voxels each) x 16 output channels.  For each of the
// 64 transform points xi one v_mfma_f32_16x16x4_f32 per 4 input channels contracts
//   A[i = lane&15][k = lane>>4] = W_xi[co0 + i][ci0 + k]      (LDS; transformed once per step: pack.hip modes 4 / 5)
//   B[k = lane>>4][j = lane&15] = U_xi[ci0 + k][tile j]       (registers: the lane transforms its own 4x4x4 patch)
//   D[row = (lane>>4)*4 + r][col = lane&15]                   -> 64 x 4 accumulator registers (AGPRs)
// so a lane holds all 64 points of 4 (channel, tile) outputs and inverse-transforms them in registers.  The wave runs
// alone on its SIMD (512 registers).  Per chunk of 4 input channels: 64 MFMAs in program-ordered slots that also carry
// the LDS reads of the next chunk's patch, the filter points (fetched two groups ahead) and the DMA issue; then the
// transform of the next chunk (96 v_pk_add_f32) -- VALU work does not overlap the issuing wave's own MFMAs on this
// hardware (scripts/ubench/mfma_overlap.hip), wherever it is placed.  Chunks (haloed box of the workgroup + the
// 64 x 16 x 4 filter points of each 16-channel block) arrive by LDS-DMA into a ring of NBUF stage buffers, issued
// NBUF-1 chunks ahead; one barrier per chunk.  Workgroups are persistent (one per CU): XCD x owns a contiguous range of
// boxes that its 32 workgroups walk interleaved, and the first stages of the next box are in flight during the
// epilogue of the current one.
#include "common.h"
#include "wino.h"
#include <stdio.h>
#include <stdlib.h>
#include <type_traits>

namespace {

using namespace mis_dma;
using namespace mis_wino;

struct WinoArgs {
    const float* x; long long x_bs;
    const float* wt;      // [co_blocks][Cin_pad/4][16][64 lanes][4]  (mis_wino_pack)
    const float* bias;    // [Cout] or nullptr
    float* y; long long y_bs;
    int N, Cin, Cout, D, H, W;
    int nci4;             // input-channel chunks (Cin_pad / 4)
    int boxes_z, boxes_y, boxes_x, co_groups;
    unsigned n_blocks, n_blocks_padded;
    float2* stat; long long stat_sc, stat_sn;
    // NB instantiation (data gradient feeding the backward of a ReLU'd InstanceNorm): the norm's input, its per-(n, c)
    // means = the activation thresholds, the activation's negative slope; `stat` then receives (sum dz, sum dz * x)
    const float* xn; long long xn_bs;
    const float* thr;
    float slope;
    // SPLIT instantiation (few boxes: the 6^3 level, half batches at 12^3): the input channels are cut into `ks` slices of
    // nci4 / ks chunks, a (box, slice) pair is one entry of the workgroups' walk; y then points at the partial outputs
    // [ks][N][...] (slice stride y_ks floats), no bias, no statistics: mis_wino_split_reduce finishes
    int ks; long long y_ks;
};

// group = GZ x GY x GX tiles (16) per wave; workgroup = WZ x WY x WX groups x COB blocks of 16 output channels (4 waves)
// FLAT = 1: the box is GZ x GY x GX tiles (<= 64, any shape) and the 4 waves take its tiles 16 at a time in (z, y, x) order
// (lanes past the last tile idle): boxes that do not split into 4 x 16-tile groups, e.g. 3 x 3 x 6 tiles for a 12^3 volume
// (54 of 64 lanes busy instead of the 27 of 64 that partly filled 8 x 8 x 8 boxes give)
template <int GZ_, int GY_, int GX_, int WZ_, int WY_, int WX_, int COB_, int NBUF_, int DW_ = 0, int FLAT_ = 0>
struct WinoCfg {
    static constexpr int GZ = GZ_, GY = GY_, GX = GX_, WZ = WZ_, WY = WY_, WX = WX_, COB = COB_, NBUF = NBUF_, FLAT = FLAT_;
    static constexpr int BZ = GZ * WZ, BY = GY * WY, BX = GX * WX;          // tiles per box
    static constexpr int OZ = 2 * BZ, OY = 2 * BY, OX = 2 * BX;             // output voxels per box
    static constexpr int HZ = OZ + 2, HY = OY + 2;
    // LDS rows hold x in [x0 - 4, x0 + OX + 4): whole 16-byte groups of the image row (W % 4 == 0), so a row arrives as
    // NQ buffer_load_dwordx4 ... lds lanes; the halo column x0 - 1 sits at float 3 of the row
    // DW = 1 (narrow boxes, OX = 8: the padded rows would not fit 4 ring stages): rows hold exactly [x0 - 1, x0 + OX + 1)
    // and arrive as single dwords (buffer_load_dword ... lds), 64 halo elements per instruction
    static constexpr int DW = DW_;
    static constexpr int NQ = DW ? OX + 2 : (OX + 8) / 4, RX = DW ? OX + 2 : NQ * 4, X0 = DW ? 0 : 3;
    static constexpr int GROUPS = HZ * HY * NQ;                              // DMA lanes (16-byte groups / dwords) per channel
    static constexpr int NCH = (GROUPS + 63) / 64;                           // DMA instructions per channel
    static constexpr int CS_RAW = DW ? NCH * 64 : GROUPS * 4;                // DW: whole pieces (surplus lanes land in the pad)
    // channel stride (floats) == 49 (mod 64): the 4 channel lanes (0, 49, 34, 19 mod 64) x 16 tile lanes (stride 2) of a
    // ds_read_b32 then cover the 64 banks almost exactly once.  An odd stride is fine for the DMA: buffer_load_dwordx4 ...
    // lds only needs a 4-byte aligned LDS address (measured: same results; 8x8x8 boxes 133 -> 115 us, the others unchanged)
    static constexpr int CS = CS_RAW + (49 - CS_RAW % 64 + 64) % 64;
    static constexpr int IN_FLOATS = 4 * CS;
    static constexpr int W_FLOATS = COB * 4096;
    static constexpr int STAGE = IN_FLOATS + W_FLOATS;
    static constexpr int MAX_COUT = 384;
    static constexpr int LDS_BYTES = NBUF * STAGE * 4 + 512 + MAX_COUT * 4 + NCH * 256;   // + statistics scratch, bias, DMA geometry
    static constexpr int WPW = COB * 4;                                     // 1 KiB filter pieces per wave
    static constexpr int P = NCH + WPW;                                     // DMA instructions per wave and stage
    static_assert(FLAT ? (GZ * GY * GX <= 64 && WZ * WY * WX == 1 && COB == 1 && DW == 1) : (GZ * GY * GX == 16), "16 tiles per wave");
    static_assert(FLAT || WZ * WY * WX * COB == 4, "4 waves");
    static_assert(NBUF == 3 || NBUF == 4, "ring depth");
    static_assert(2 * P <= 63 && P <= 21, "vmcnt range (two stages in flight), DMA slots");
    static_assert(LDS_BYTES <= 160 * 1024, "LDS");
};

#ifndef MIS_WINO_DBG_CT
#define MIS_WINO_DBG_CT 0
#endif
#ifndef MIS_WINO_NV
#define MIS_WINO_NV 16      // transform points whose accumulators live in VGPRs (the MFMA takes either): 64 fewer v_accvgpr_read per
                            // box; 16 is what fits beside the patch registers without the allocator parking VGPRs in AGPRs (24) or
                            // spilling (32); serial config-3 step 19.94 -> 19.84 ms
#endif
constexpr int NVP = MIS_WINO_NV;
// Ablation builds (scripts/wino_variants.sh; results are WRONG with any bit set, timing only -- DESIGN.md quotes them):
// 1 no stores, 2 no epilogue (the MFMAs become dead code), 4 no cursor recompute, 8 no DMA, 16 no transform, 32 no filter
// reads, 64 no barrier, 128 no filter DMA, 256 no input DMA, 512 transform interleaved with the MFMA slots, 1024 no wait
// for the DMA stage, 2048 every box fetches the first box's input
constexpr int DBG = MIS_WINO_DBG_CT;

extern __shared__ __attribute__((aligned(16))) float mis_wino_lds[];

// Everything a wave needs to issue the DMAs of one (box, chunk) stage.  voff / wvoff are per lane, the rest is uniform.
template <class C>
struct Issue {
    unsigned voff[C::NCH];      // this lane's 16-byte group of piece p (byte offset in the image, OOB = padding)
    unsigned wvoff[C::WPW];     // filter piece, relative to the (channel group, chunk) base
    i32x4 rx, rw;
    unsigned st;                // LDS byte address of the stage buffer being filled
    unsigned cbase;             // byte offset of this wave's input channel of the chunk
    unsigned wbase;             // byte offset of the (channel group, chunk) filter block

    template <int I>
    __device__ __forceinline__ void piece(int wave) const {
        if constexpr (I >= 0 && I < C::NCH && !(DBG & 256)) {
            if constexpr (C::DW) {
                dma_dword_s(st + (unsigned)(wave * C::CS + I * 64) * 4u, voff[I], cbase, rx);
            } else if (I * 64 + 64 <= C::GROUPS || I * 64 + (int)(threadIdx.x & 63) < C::GROUPS) {   // ragged last piece
                dma_dwordx4_s(st + (unsigned)(wave * C::CS + I * 256) * 4u, voff[I], cbase, rx);
            }
        } else if constexpr (I >= C::NCH && I < C::P && !(DBG & 128)) {
            dma_dwordx4_s(st + (unsigned)(C::IN_FLOATS + (wave + 4 * (I - C::NCH)) * 256) * 4u, wvoff[I - C::NCH], wbase, rw);
        }
    }
    template <int I, int END>
    __device__ __forceinline__ void pieces(int wave) const {
        if constexpr (I < END) { piece<I>(wave); pieces<I + 1, END>(wave); }
    }
};

// LDS address of the lane's patch, one base per z plane (pinned: the ds_read2_b32 offsets, 8 bits of dwords, then reach
// every row of the plane and the compiler does not re-derive a base per load)
typedef const __attribute__((address_space(3))) float* lds_ptr;
struct RawPlanes { lds_ptr z[4]; };

// One chunk = 64 slots, one MFMA each (transform point xi = K of `cur`), in program order (a scheduling barrier closes
// every slot: a wave alone on its SIMD issues one instruction per 4 cycles, so at most 7 others fit under an MFMA):
//   K % 4 == 0   ds_read_b128 of the filter points of group K/4 + 2 (the last two: groups 0, 1 of the next stage)
//   K < 16       two ds_read2_b32 of the next chunk's patch
//   2, 7, 12 ..  one DMA of the stage NBUF-1 ahead (spread out: the CU's one texture-address unit serves all 4 waves)
//   (the transform of the next chunk follows the run, see iter)
template <class C, bool FIRST, int K>
__device__ __forceinline__ void slots(const f32x4* __restrict__ wl, const f32x4* __restrict__ wl_next,
                                      const RawPlanes& raw, const f32x2 (&cur)[32], f32x2 (&nxt)[32],
                                      f32x4 (&acc)[64], f32x4 (&ar)[4], const Issue<C>& is, int wave) {
    if constexpr (K < 64) {
        constexpr int G = K / 4;
        if constexpr (K % 4 == 0 && !(DBG & 32)) ar[(G + 2) % 4] = G < 14 ? wl[(G + 2) * 64] : wl_next[(G - 14) * 64];
        {
            // asm with the accumulator tied to an AGPR tuple: through the builtin hipcc placed 14 of the first chunk's
            // (C = 0) results in VGPRs and copied them into AGPRs in the second chunk (112 v_accvgpr moves per box)
            const float av = ar[G % 4][K % 4], bv = cur[K / 2][K % 2];
            if constexpr (K < NVP) {
                if constexpr (FIRST) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, 0" : "=v"(acc[K]) : "v"(av), "v"(bv));
                else asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[K]) : "v"(av), "v"(bv));
            } else {
                if constexpr (FIRST) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, 0" : "=a"(acc[K]) : "v"(av), "v"(bv));
                else asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(acc[K]) : "v"(av), "v"(bv));
            }
        }
        if constexpr (K < 16 && !(DBG & 16)) {
#pragma unroll
            for (int j = 2 * K; j < 2 * K + 2; ++j) {
                const int z = j / 8, y = (j / 2) % 4, xp = j % 2;
                nxt[j] = f32x2{raw.z[z][y * C::RX + 2 * xp], raw.z[z][y * C::RX + 2 * xp + 1]};
            }
        }
        constexpr int SP = C::P <= 12 ? 5 : 3;      // DMA spacing in slots
        if constexpr (K >= 2 && (K - 2) % SP == 0 && (K - 2) / SP < C::P && !(DBG & 8)) is.template piece<(K - 2) / SP>(wave);
        if constexpr (K >= 16 && K < 40 && !(DBG & 16) && (DBG & 512)) in_unit<K - 16>(nxt);      // development: interleaved
        __builtin_amdgcn_sched_barrier(0);
        slots<C, FIRST, K + 1>(wl, wl_next, raw, cur, nxt, acc, ar, is, wave);
    }
}

// NB = true: the launch is the data gradient dL/da of a convolution whose input a = ReLU(InstanceNorm(xn)) has no other
// consumer.  The epilogue then also forms the two sums the normalisation's backward needs per (n, channel),
//   s1 = sum dz,  t2 = sum dz * xn,   dz = da * (xn > mean ? 1 : slope)
// (the backward's second stage turns t2 into sum dz * xhat = rstd * (t2 - mean * s1)), and writes them per run of boxes
// into `stat` exactly as the forward writes its (sum, sum of squares): mis_norm_act_bwd's partial-sum pass over da and
// xn -- 906 MB at 96^3 -- is not run.  The 16 float2 loads of xn a lane needs are issued in front of the LAST chunk's
// DMAs, so that chunk's closing vmcnt wait covers them (vmcnt retires in order: loads issued in the epilogue itself
// would wait for every DMA in flight).  The thresholds of all (n, c) sit in the LDS bias table (a data gradient has
// no bias): N * Cout <= MAX_COUT.
template <class C, bool NB, bool SPLIT>
__device__ __forceinline__ void wino_fwd_body(const WinoArgs& a) {
    float* const lds = mis_wino_lds;
    // persistent: XCD x owns the boxes [x * per, (x + 1) * per); its workgroups walk them 32 apart, so the 32 resident
    // workgroups of an XCD always work on neighbouring boxes (shared halos and filter points in one L2)
    const int xcd = blockIdx.x % MIS_NUM_XCD, slot = blockIdx.x / MIS_NUM_XCD, nslot = gridDim.x / MIS_NUM_XCD;
    const unsigned per = a.n_blocks_padded / MIS_NUM_XCD;
    const unsigned box_end = (xcd + 1) * per < a.n_blocks ? (xcd + 1) * per : a.n_blocks;
    unsigned box = xcd * per + slot;
    if (box >= box_end) return;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lk = lane >> 4, lj = lane & 15;
    const int cb = wave % C::COB, grp = wave / C::COB;                       // this wave's channel block and group
    int tx, ty, tz;
    bool lane_live = true;                 // FLAT: lanes past the box's last tile compute tile 0 again and store nothing
    if constexpr (C::FLAT) {
        int t = grp * 16 + lj;
        lane_live = t < C::BZ * C::BY * C::BX;
        t = lane_live ? t : 0;
        tx = t % C::BX; ty = (t / C::BX) % C::BY; tz = t / (C::BX * C::BY);
    } else {
        const int gx = grp % C::WX, gy = (grp / C::WX) % C::WY, gz = grp / (C::WX * C::WY);
        tx = gx * C::GX + lj % C::GX; ty = gy * C::GY + (lj / C::GX) % C::GY; tz = gz * C::GZ + lj / (C::GX * C::GY);
    }
    const long long S = (long long)a.D * a.H * a.W;
    const unsigned s_bytes = (unsigned)S * 4u;
    const unsigned lds0 = lds_addr(lds);
    const int nst = SPLIT ? a.nci4 / a.ks : a.nci4;      // chunks per entry of the walk

    // box coordinates (channel group fastest, then [SPLIT: contraction slice,] x, y, z, image).  Decoded by division once; the walk L += nslot then
    // advances them by the decoded stride with carries (a scalar division is a ~100-cycle dependent chain, and a lone
    // wave per SIMD cannot hide it: 4 divisions per box were 0.5 us of the 2.9 us a box costs outside its MFMAs)
    struct Box { int cg, kz, bx, by, bz, n, z0, y0, x0; long long idx; };
    auto finish = [&](Box& b) {
        b.z0 = b.bz * C::OZ; b.y0 = b.by * C::OY; b.x0 = b.bx * C::OX;
        b.idx = ((long long)b.bz * a.boxes_y + b.by) * a.boxes_x + b.bx;
    };
    auto decode = [&](unsigned L) {
        Box b;
        unsigned t = L;
        b.cg = t % a.co_groups; t /= a.co_groups;
        b.kz = 0;
        if constexpr (SPLIT) { b.kz = t % a.ks; t /= a.ks; }
        b.bx = t % a.boxes_x;   t /= a.boxes_x;
        b.by = t % a.boxes_y;   t /= a.boxes_y;
        b.bz = t % a.boxes_z;   t /= a.boxes_z;
        b.n = t;
        finish(b);
        return b;
    };
    const Box stride = decode((unsigned)nslot);
    auto advance = [&](Box b) {
        int c;
        b.cg += stride.cg;     c = b.cg >= a.co_groups; b.cg -= c ? a.co_groups : 0;
        if constexpr (SPLIT) { b.kz += stride.kz + c; c = b.kz >= a.ks; b.kz -= c ? a.ks : 0; }
        b.bx += stride.bx + c; c = b.bx >= a.boxes_x;   b.bx -= c ? a.boxes_x : 0;
        b.by += stride.by + c; c = b.by >= a.boxes_y;   b.by -= c ? a.boxes_y : 0;
        b.bz += stride.bz + c; c = b.bz >= a.boxes_z;   b.bz -= c ? a.boxes_z : 0;
        b.n += stride.n + c;
        finish(b);
        return b;
    };

    // ---- DMA issue cursor: (box, chunk) of the next stage to issue; runs NBUF-1 stages ahead of the compute ----
    Issue<C> is;
    is.rw = make_rsrc(a.wt, (unsigned)a.co_groups * C::COB * (unsigned)a.nci4 * 16384u);
#pragma unroll
    for (int i = 0; i < C::WPW; ++i) {
        const int j = wave + 4 * i, b = j / 16, pp = j % 16;
        is.wvoff[i] = (unsigned)((b * a.nci4) * 4096 + pp * 256 + lane * 4) * 4u;
    }
    int icg = 0, ikz = 0;        // channel group (and contraction slice) of the cursor's box
    // per-lane geometry of the DMA pieces, packed (hz | hy << 4 | q << 9 | valid << 14), kept in LDS: as registers
    // hipcc spills these box-invariant values to scratch, and a scratch reload waits on vmcnt(0) -- on the DMAs in flight
    unsigned* const s_geo = reinterpret_cast<unsigned*>(lds + C::NBUF * C::STAGE + 128 + C::MAX_COUT);
    if (tid < 64) {
#pragma unroll
        for (int p = 0; p < C::NCH; ++p) {
            const int e = p * 64 + tid;
            const int row = e / C::NQ, q = e - row * C::NQ;
            const int hz = row / C::HY, hy = row - hz * C::HY;
            s_geo[p * 64 + tid] = (unsigned)(hz | hy << 4 | q << 9 | (e < C::GROUPS ? 1 << 14 : 0));
        }
    }
    __syncthreads();
    auto cursor_box = [&](const Box& b_in, bool live) {     // per-lane offsets and descriptors of box b (all OOB past the end)
        Box b = b_in;
        if (DBG & 2048) { b.n = 0; b.z0 = 0; b.y0 = 0; b.x0 = 0; }      // ablation: every box reads the first box's input (L2-hot)
        icg = b.cg; ikz = b.kz;
        is.rx = make_rsrc(a.x + (long long)b.n * a.x_bs, live ? (unsigned)a.Cin * s_bytes : 0u);
        unsigned geo[C::NCH];
#pragma unroll
        for (int p = 0; p < C::NCH; ++p) geo[p] = s_geo[p * 64 + lane];
        __builtin_amdgcn_sched_barrier(0);      // all the LDS reads in flight together, not one round trip per piece
#pragma unroll
        for (int p = 0; p < C::NCH; ++p) {
            const unsigned g = geo[p];
            const int qz = b.z0 - 1 + (int)(g & 15u), qy = b.y0 - 1 + (int)((g >> 4) & 31u);
            const int qx = C::DW ? b.x0 - 1 + (int)((g >> 9) & 31u) : b.x0 - 4 + 4 * (int)((g >> 9) & 31u);
            const bool ok = (g >> 14) && (unsigned)qz < (unsigned)a.D && (unsigned)qy < (unsigned)a.H &&
                            (unsigned)qx < (unsigned)a.W;
            is.voff[p] = ok ? (unsigned)((qz * a.H + qy) * a.W + qx) * 4u : OOB;
        }
    };
    auto cursor_set = [&](unsigned gs_issue, int istage) {      // uniform parts of the stage about to be issued
        is.st = lds0 + (gs_issue % C::NBUF) * (unsigned)(C::STAGE * 4);
        if constexpr (SPLIT) istage += ikz * nst;                        // the slice's first chunk
        is.cbase = (unsigned)(istage * 4 + wave) * s_bytes;              // channel >= Cin: beyond the descriptor
        is.wbase = (unsigned)(icg * C::COB * a.nci4 + istage) * 16384u;
    };

    // this lane's patch: tile (tz, ty, tx) of the box, channel lk of the chunk
    const int poff = ((2 * tz) * C::HY + 2 * ty) * C::RX + C::X0 + 2 * tx + lk * C::CS;
    auto raw_of = [&](unsigned gs) { return lds + (gs % C::NBUF) * C::STAGE + poff; };
    auto wl_of = [&](unsigned gs) {
        return reinterpret_cast<const f32x4*>(lds + (gs % C::NBUF) * C::STAGE + C::IN_FLOATS + cb * 4096) + lane;
    };

    float* const s_bias = lds + C::NBUF * C::STAGE + 128;
    if constexpr (NB) {
        for (int i = tid; i < a.N * a.Cout; i += 256) s_bias[i] = a.thr[i];             // thresholds [n][c]
    } else {
        for (int i = tid; i < a.Cout; i += 256) s_bias[i] = a.bias ? a.bias[i] : 0.f;      // published by the barrier below
    }
    constexpr int A = C::NBUF - 1;
    const unsigned nslot_u = (unsigned)nslot;
    Box bb = decode(box), nb = bb;       // the box being computed, the box under the DMA cursor
    cursor_box(bb, true);
#pragma unroll
    for (int s = 0; s < A; ++s) { cursor_set((unsigned)s, s); is.template pieces<0, C::P>(wave); }   // nst > A
    static_assert(A == 3, "the waits below are written for a ring of 4");
    vmwait<2 * C::P>::go();                // stage 0 has landed when only the later prologue stages are in flight
    __syncthreads();
    f32x2 ua[32], ub[32];
    f32x4 acc[64];
    f32x4 ar[4];
    {
        const float* __restrict__ raw = raw_of(0);
#pragma unroll
        for (int j = 0; j < 32; ++j) {
            const int z = j / 8, y = (j / 2) % 4, xp = j % 2;
            ua[j] = f32x2{raw[(z * C::HY + y) * C::RX + 2 * xp], raw[(z * C::HY + y) * C::RX + 2 * xp + 1]};
        }
    }
    ar[0] = wl_of(0)[0]; ar[1] = wl_of(0)[64];
    in_units<0, 24>(ua);                   // chunk 0 of the first box: not overlapped
    vmwait<C::P>::go();                    // stage 1
    __syncthreads();

    unsigned gs = 0;                       // global stage counter of the compute
    const int sw = nst - A;                // from this step on the cursor issues the NEXT box's first stages
    // On entry stage gs+1 has landed for every wave and everyone is done with stage gs-1 (the barrier at the end of the
    // previous iteration).  The wait for stage gs+2 sits at the END of the iteration, in front of the epilogue's stores
    // (vmcnt counts stores too: a wait behind them would wait for their write acknowledgements).
    auto iter = [&](auto first, f32x2 (&cur)[32], f32x2 (&nxt)[32], int s, bool relaxed) {
        cursor_set(gs + A, s + A < nst ? s + A : s + A - nst);
        RawPlanes rp;
        {
            const float* r0 = raw_of(gs + 1);
#pragma unroll
            for (int z = 0; z < 4; ++z) {
                rp.z[z] = (lds_ptr)r0 + z * C::HY * C::RX;
                asm volatile("" : "+v"(rp.z[z]));
            }
        }
        slots<C, decltype(first)::value, 0>(wl_of(gs), wl_of(gs + 1), rp, cur, nxt, acc, ar, is, wave);
        // the transform of the next chunk, after the run: VALU work does not overlap this wave's MFMAs wherever it is placed
        // (scripts/ubench/mfma_overlap.hip), and in one block it costs 4 % less than spread over the slots
        if (!(DBG & 16) && !(DBG & 512)) in_units<0, 24>(nxt);
        ++gs;
        // stage gs+1 landed (mine) ...  (1024: ablation, do not wait for it).  `extra`: younger plain loads that may stay
        // in flight (NB: the norm input issued in front of this chunk's DMAs; vmcnt retires in order, so everything
        // older than them -- the stage this wait is for -- has landed all the same)
        // (a uniform branch around the wait only: two copies of the MFMA run behind a branch make hipcc move the
        // accumulators through VGPRs -- 593 spills)
        if (NB && relaxed) vmwait<C::P + 16>::go();
        else vmwait<(DBG & 1024) ? 2 * C::P : C::P>::go();
        if (!(DBG & 64)) __syncthreads();  // ... and everyone's; everyone is done with stage gs-1
    };

    // statistics: per-lane running sums over the boxes of a run (consecutive boxes of this workgroup in the same image and
    // channel group); only the last box of a run pays the cross-lane reduction, the others store zeros in their slot
    float run1[4] = {0.f, 0.f, 0.f, 0.f}, run2[4] = {0.f, 0.f, 0.f, 0.f};
    f32x2 xv[NB ? 16 : 1];             // NB: the norm input at this lane's 4 channels x 2 x 2 (z, y) x-pairs
    for (; box < box_end; box += nslot_u, bb = nb) {
        const bool nlive = box + nslot_u < box_end;
        nb = advance(bb);
        if (sw == 0 && !(DBG & 4)) cursor_box(nb, nlive);
        iter(std::true_type{}, ua, ub, 0, false);
        if (sw == 1 && !(DBG & 4)) cursor_box(nb, nlive);
        iter(std::false_type{}, ub, ua, 1, false);
        for (int s = 2; s < nst; s += 2) {      // nst is even (Cin % 8 == 0)
            if (sw == s && !(DBG & 4)) cursor_box(nb, nlive);
            if constexpr (NB) {
                // in front of the DMAs of the second-to-last chunk: that chunk's closing wait lets these 16 loads fly
                // (vmcnt(P + 16)), the last chunk's vmcnt(P) retires them -- two chunks of latency, as the DMAs have
                if (s + 2 >= nst && !(DBG & 4096)) {
                    const int oz = bb.z0 + 2 * tz, oy = bb.y0 + 2 * ty, ox = bb.x0 + 2 * tx;
                    const bool okx = lane_live && oz < a.D && oy < a.H && ox < a.W;
                    const int c0 = (bb.cg * C::COB + cb) * 16;
                    // inline asm: hipcc's own waitcnt insertion does not see the asm DMAs and would guard a builtin
                    // load with vmcnt(0) at its first use -- a wait for every DMA in flight (+1.5 us per box measured)
                    const i32x4 rxn = make_rsrc(a.xn + (long long)bb.n * a.xn_bs + (long long)c0 * S, 16u * s_bytes);
                    const unsigned vx = okx ? (unsigned)(lk * 4) * s_bytes + (unsigned)((oz * a.H + oy) * a.W + ox) * 4u : OOB;
                    unsigned vzy[4];
#pragma unroll
                    for (int zy = 0; zy < 4; ++zy) vzy[zy] = vx + (unsigned)(((zy >> 1) * a.H + (zy & 1)) * a.W) * 4u;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const unsigned so = (unsigned)r * s_bytes;
#pragma unroll
                        for (int zy = 0; zy < 4; ++zy)
                            asm volatile("buffer_load_dwordx2 %0, %1, %2, %3 offen"
                                         : "=v"(xv[r * 4 + zy]) : "v"(vzy[zy]), "s"(rxn), "s"(so) : "memory");
                    }
                }
            }
            iter(std::false_type{}, ua, ub, s, NB && s + 2 >= nst);
            if (sw == s + 1 && !(DBG & 4)) cursor_box(nb, nlive);
            iter(std::false_type{}, ub, ua, s + 1, false);
        }

        if constexpr (NB) {     // the last chunk's vmcnt(P) retired the xn loads: from here on their registers hold data
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("" : "+v"(xv[i]));
        }
        // ---- epilogue: inverse transform (64 -> 2x2x2 per (channel, tile)), bias, store, optional statistics ----
        if (DBG & 2) continue;
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");        // the last (asm) MFMAs have left the pipe before their AGPRs are read
        const int oz = bb.z0 + 2 * tz, oy = bb.y0 + 2 * ty, ox = bb.x0 + 2 * tx;
        const bool ok = lane_live && oz < a.D && oy < a.H && ox < a.W;
        const int co0 = (bb.cg * C::COB + cb) * 16;
        const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(
            (void*)(a.y + (SPLIT ? (long long)bb.kz * a.y_ks : 0LL) + (long long)bb.n * a.y_bs + (long long)co0 * S), 0,
            (int)(16u * s_bytes), 0x00020000);
        const unsigned vo = ok && !(DBG & 1) ? (unsigned)(lk * 4) * s_bytes + (unsigned)((oz * a.H + oy) * a.W + ox) * 4u : OOB;
        // Two of the lane's four channel rows at a time (register pairs: packed adds), one (x, y) column of points at a
        // time, folded into the y and x sums as it is read.  Few live registers (the next box's first chunk is already
        // waiting in 64 of them) and no spills: a scratch reload here would wait on vmcnt(0), i.e. on the DMAs in
        // flight and on the write acknowledgements of the stores.
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            f32x2 o[2][2][2];
#pragma unroll
            for (int x = 0; x < 4; ++x) {
                f32x2 py[2][2];
#pragma unroll
                for (int y = 0; y < 4; ++y) {
                    // read in place, here: left to itself hipcc copies all 256 accumulators to VGPRs first
                    const f32x2 m0 = (y * 4 + x < NVP) ? f32x2{acc[y * 4 + x][2 * h], acc[y * 4 + x][2 * h + 1]} : acc_pair(acc[y * 4 + x], h),
                                m1 = acc_pair(acc[16 + y * 4 + x], h),
                                m2 = acc_pair(acc[32 + y * 4 + x], h), m3 = acc_pair(acc[48 + y * 4 + x], h);
                    const f32x2 pz0 = m0 + m1 + m2, pz1 = m1 - m2 - m3;
                    if (y == 0) { py[0][0] = pz0; py[1][0] = pz1; }
                    if (y == 1) { py[0][0] += pz0; py[1][0] += pz1; py[0][1] = pz0; py[1][1] = pz1; }
                    if (y == 2) { py[0][0] += pz0; py[1][0] += pz1; py[0][1] -= pz0; py[1][1] -= pz1; }
                    if (y == 3) { py[0][1] -= pz0; py[1][1] -= pz1; }
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int zz = 0; zz < 2; ++zz)
#pragma unroll
                    for (int yy = 0; yy < 2; ++yy) {
                        if (x == 0) { o[zz][yy][0] = py[zz][yy]; }
                        if (x == 1) { o[zz][yy][0] += py[zz][yy]; o[zz][yy][1] = py[zz][yy]; }
                        if (x == 2) { o[zz][yy][0] += py[zz][yy]; o[zz][yy][1] -= py[zz][yy]; }
                        if (x == 3) { o[zz][yy][1] -= py[zz][yy]; }
                    }
                __builtin_amdgcn_sched_barrier(0);
            }
            f32x2 bv = *reinterpret_cast<const f32x2*>(s_bias + (NB ? bb.n * a.Cout : 0) + co0 + lk * 4 + 2 * h);     // LDS: lgkmcnt, not vmcnt
            const f32x2 th = bv;                 // NB: the two channels' thresholds (a data gradient has no bias)
            if constexpr (NB) bv = f32x2{0.f, 0.f};
            f32x2 s1 = {0.f, 0.f}, s2 = s1;
#pragma unroll
            for (int zz = 0; zz < 2; ++zz)
#pragma unroll
                for (int yy = 0; yy < 2; ++yy) {
                    const f32x2 v0 = o[zz][yy][0] + bv, v1 = o[zz][yy][1] + bv;
                    if constexpr (NB && !(DBG & 8192)) {
#pragma unroll
                        for (int rr = 0; rr < 2; ++rr) {
                            const f32x2 x = xv[(2 * h + rr) * 4 + zz * 2 + yy];      // (x, x + 1) of channel 2h + rr
                            const float d0 = x[0] > th[rr] ? v0[rr] : v0[rr] * a.slope;
                            const float d1 = x[1] > th[rr] ? v1[rr] : v1[rr] * a.slope;
                            s1[rr] += d0 + d1;
                            s2[rr] = fmaf(d0, x[0], fmaf(d1, x[1], s2[rr]));
                        }
                    } else {
                        s1 += v0 + v1;
                        s2 += v0 * v0 + v1 * v1;
                    }
#pragma unroll
                    for (int rr = 0; rr < 2; ++rr) {
                        typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
                        const f32x2 v = {v0[rr], v1[rr]};
                        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, v), ry,
                                                              (int)(vo + (unsigned)((zz * a.H + yy) * a.W) * 4u),
                                                              (int)((unsigned)(2 * h + rr) * s_bytes), 0);
                    }
                }
            if constexpr (C::DW) {      // this variant also runs partly filled boxes: tiles outside the volume do not count
                if (!ok) { s1 = f32x2{0.f, 0.f}; s2 = s1; }
            }
            run1[2 * h] += s1[0]; run1[2 * h + 1] += s1[1]; run2[2 * h] += s2[0]; run2[2 * h + 1] += s2[1];
            __builtin_amdgcn_sched_barrier(0);
        }
        const bool flush = !nlive || nb.n != bb.n || nb.cg != bb.cg;      // uniform
        if (a.stat && !flush) {
            if (tid < C::COB * 16)
                a.stat[(long long)(bb.cg * C::COB * 16 + tid) * a.stat_sc + (long long)bb.n * a.stat_sn + bb.idx] = make_float2(0.f, 0.f);
        } else if (a.stat) {
            // per-channel (sum, sum of squares) of the run that ends with this box.
            // The scratch is a piece of LDS beyond the ring (the ring is live: later stages are in flight).
            float st1[4], st2[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) { st1[r] = run1[r]; st2[r] = run2[r]; run1[r] = 0.f; run2[r] = 0.f; }
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int o = 1; o < 16; o <<= 1) {
                    st1[r] += __shfl_xor(st1[r], o, 64);
                    st2[r] += __shfl_xor(st2[r], o, 64);
                }
            float2* red = reinterpret_cast<float2*>(lds + C::NBUF * C::STAGE);
            if (lj == 0) {
#pragma unroll
                for (int r = 0; r < 4; ++r) red[wave * 16 + lk * 4 + r] = make_float2(st1[r], st2[r]);
            }
            __syncthreads();
            if (tid < C::COB * 16) {
                const int b = tid / 16, c = tid % 16;
                float sx = 0.f, sq = 0.f;
#pragma unroll
                for (int g = 0; g < 4 / C::COB; ++g) {      // waves g * COB + b share the channel block
                    const float2 p = red[(g * C::COB + b) * 16 + c];
                    sx += p.x; sq += p.y;
                }
                a.stat[(long long)((bb.cg * C::COB + b) * 16 + c) * a.stat_sc + (long long)bb.n * a.stat_sn + bb.idx] =
                    make_float2(sx, sq);
            }
            // the next write of `red` is a whole box (>= 2 barriers) away
        }
    }
}

template <class C, bool NB = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void wino_fwd_kernel(const WinoArgs a) {
    wino_fwd_body<C, NB, false>(a);
}
// (box, contraction slice) entries, partial outputs: see WinoArgs::ks
template <class C>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void wino_fwd_split_kernel(const WinoArgs a) {
    wino_fwd_body<C, false, true>(a);
}

template <class C, bool NB = false, bool SPLIT = false>
int launch_wino(WinoArgs a, hipStream_t stream) {
    a.boxes_z = (int)mis_cdiv(a.D, C::OZ);
    a.boxes_y = (int)mis_cdiv(a.H, C::OY);
    a.boxes_x = (int)mis_cdiv(a.W, C::OX);
    a.co_groups = a.Cout / (16 * C::COB);
    const long long nb = (long long)a.N * a.boxes_z * a.boxes_y * a.boxes_x * a.co_groups * (SPLIT ? a.ks : 1);
    if (nb <= 0 || nb > 0x7fffffffLL) return MIS_ERR_ARG;
    a.n_blocks = (unsigned)nb;
    a.n_blocks_padded = (unsigned)(mis_cdiv(nb, MIS_NUM_XCD) * MIS_NUM_XCD);
    static std::atomic<unsigned long long> attr_done{0};
    const unsigned grid = a.n_blocks_padded < 256u ? a.n_blocks_padded : 256u;      // persistent: one workgroup per CU
    if constexpr (SPLIT) {
        if (mis_set_lds_attr(reinterpret_cast<const void*>(&wino_fwd_split_kernel<C>), C::LDS_BYTES, attr_done) != MIS_OK)
            return MIS_ERR_LAUNCH;
        hipLaunchKernelGGL((wino_fwd_split_kernel<C>), dim3(grid), dim3(256), C::LDS_BYTES, stream, a);
    } else {
        if (mis_set_lds_attr(reinterpret_cast<const void*>(&wino_fwd_kernel<C, NB>), C::LDS_BYTES, attr_done) != MIS_OK)
            return MIS_ERR_LAUNCH;
        hipLaunchKernelGGL((wino_fwd_kernel<C, NB>), dim3(grid), dim3(256), C::LDS_BYTES, stream, a);
    }
    return mis_launch_status();
}

using WinoV2 = WinoCfg<1, 4, 4, 4, 1, 1, 1, 4, 1>;
using WinoV3 = WinoCfg<3, 3, 6, 1, 1, 1, 1, 4, 1, 1>;

// Contraction slices for a launch with too few (box, channel group) entries for the chip (the 6^3 level: 8 ... 128 entries of
// 32 ... 64 chunks each; half batches at 12^3): the smallest count that brings the entries to >= 192, slices of an even
// number >= 8 of chunks.  Variants 2 / 3 only (the levels where it happens); 1 = no split.  Chosen on the entry count alone.
int wino_splits(int N, int Cin, int Cout, int D, int H, int W, int variant) {
    static const bool on = [] { const char* e = getenv("MIS_WINO_SPLIT"); return !(e && e[0] == '0'); }();
    if (!on || (variant != 2 && variant != 3)) return 1;
    const long long boxes = variant == 2 ? mis_cdiv(D, 8) * mis_cdiv(H, 8) * mis_cdiv(W, 8) : (long long)(D / 6) * (H / 6) * (W / 12);
    const long long entries = (long long)N * boxes * (Cout / 16);
    const int nst = (Cin + 3) / 4;
    int ks = 1;
    while (entries * ks < 192 && ks < 8 && nst % (2 * ks) == 0 && nst / (2 * ks) >= 8 && (nst / (2 * ks)) % 2 == 0) ks *= 2;
    // Round-6 prototype (MIS_WINO_SPLIT_QUANT=1, off by default: measured, profiles/r06_wino_quant.txt): a launch whose entries
    // fill the 256 persistent workgroups badly -- the 24^3 level of config 3: 864 entries = 3.375 -> 4 rounds, 18.5 % idle --
    // cut into two contraction slices: 1728 half entries = 6.75 -> 7 half rounds = 3.5 rounds, for a reduction launch
    static const bool quant = [] { const char* e = getenv("MIS_WINO_SPLIT_QUANT"); return e && e[0] == '1'; }();
    if (quant && ks == 1 && entries > 256 && nst % 2 == 0 && nst / 2 >= 8 && (nst / 2) % 2 == 0) {
        const long long r1 = mis_cdiv(entries, 256) * 2, r2 = mis_cdiv(entries * 2, 256);      // in half rounds
        if (r2 * 100 <= r1 * 90) ks = 2;
    }
    return ks;
}

// y[n][c][v] = bias[c] + sum_k part[k][n][c][v] (fixed order), and the (sum, sum of squares) of y per (n, c) into the first
// statistics tile of the image (zeros into the others): what the unsplit launch's epilogue leaves.  One workgroup per (n, c).
__global__ __launch_bounds__(256) void wino_split_reduce_kernel(const float* __restrict__ part, long long part_ks, long long part_bs,
                                                                const float* __restrict__ bias, float* __restrict__ y,
                                                                long long y_bs, int Cout, int S, int ks, float2* stat,
                                                                long long stat_sc, long long stat_sn, int tiles) {
    __shared__ float2 red[256];
    const int n = blockIdx.x / Cout, c = blockIdx.x - n * Cout;
    const float* __restrict__ p = part + (long long)n * part_bs + (long long)c * S;
    float* __restrict__ o = y + (long long)n * y_bs + (long long)c * S;
    const float b = bias ? bias[c] : 0.f;
    float s1 = 0.f, s2 = 0.f;
    for (int v = threadIdx.x; v < S; v += 256) {
        float acc = p[v];
        for (int k = 1; k < ks; ++k) acc += p[(long long)k * part_ks + v];
        acc += b;
        o[v] = acc;
        s1 += acc; s2 = fmaf(acc, acc, s2);
    }
    if (!stat) return;
    red[threadIdx.x] = make_float2(s1, s2);
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) { red[threadIdx.x].x += red[threadIdx.x + w].x; red[threadIdx.x].y += red[threadIdx.x + w].y; }
        __syncthreads();
    }
    for (int t = threadIdx.x; t < tiles; t += 256)
        stat[(long long)c * stat_sc + (long long)n * stat_sn + t] = t == 0 ? red[0] : make_float2(0.f, 0.f);
}

}  // namespace

// Which Winograd variant serves this 3x3x3 'same' convolution, or -1 (use the direct kernel, mis_conv_fwd):
//   0: boxes of 4 x 4 x 32 outputs (W a multiple of 32: the 96^3 level),  1: 4 x 8 x 16 (W a multiple of 16: 48^3),
//   2: 8 x 8 x 8 (24^3; halo rows as single dwords; also partly filled boxes: the 6^3 level),
//   3: 6 x 6 x 12 voxels = 3 x 3 x 6 tiles taken 16 at a time (W == 12: the 12^3 level, 54 of 64 tile lanes busy).
// Needs Cin % 8 == 0 (two 4-channel chunks per loop trip), Cin >= 16 (the DMA ring runs 3 chunks ahead),
// Cout % 16 == 0 (MFMA rows), Cout <= 384 (bias table in LDS), even D / H / W (2x2x2 tiles).
extern "C" int mis_conv3d_wino_select(int N, int Cin, int Cout, int D, int H, int W) {
    if (N <= 0 || Cin < 16 || Cin % 8 || Cout <= 0 || Cout % 16 || Cout > 384 || D <= 0 || H <= 0 || W <= 0) return -1;
    if (D % 2 || H % 2 || W % 2) return -1;
    if (((long long)Cin + 32) * D * H * W * 4 >= (1LL << 30)) return -1;
    if (W % 32 == 0 && D % 4 == 0 && H % 4 == 0) return 0;      // 16-byte halo rows: W % 4 == 0
    if (W % 16 == 0 && D % 4 == 0 && H % 8 == 0) return 1;
    if (W % 8 == 0 && D % 8 == 0 && H % 8 == 0 && Cin >= 32) return 2;      // the 24^3 level: boxes of 8 x 8 x 8
    // the 12^3 level: 4 boxes of 54 tiles per 12^3 volume (when the launch has enough boxes for the serial walk over the
    // input channels to pay: the direct kernel spreads a small problem over more workgroups)
    // (entries of the workgroups' walk: with few boxes mis_conv3d_wino_fwd_ws cuts the contraction into slices)
    if (W == 12 && D % 6 == 0 && H % 6 == 0 && Cin >= 32 &&
        (long long)N * (D / 6) * (H / 6) * (Cout / 16) * wino_splits(N, Cin, Cout, D, H, W, 3) >= 64) return 3;
    if (Cin >= 32) {
        // partly filled 8 x 8 x 8 boxes (the 12^3 level: 42 % of the box volume is output) still beat the direct kernel
        // 1.3 - 1.8x when there are enough boxes for the 256 CUs; out-of-range tiles are not stored and not counted
        const long long boxes = mis_cdiv(D, 8) * mis_cdiv(H, 8) * mis_cdiv(W, 8);
        if ((long long)D * H * W * 5 >= boxes * 512 * 2 && (long long)N * boxes * (Cout / 16) * wino_splits(N, Cin, Cout, D, H, W, 2) >= 128)
            return 2;
    }
    return -1;
}

namespace {
template <class C>
long long boxes_per_image(int D, int H, int W) {
    return mis_cdiv(D, C::OZ) * mis_cdiv(H, C::OY) * mis_cdiv(W, C::OX);
}
}  // namespace

// partial-statistics entries per image that mis_conv3d_wino_fwd writes with `stat` (one per box), as
// mis_conv_fwd_stat_tiles does for the direct kernel
extern "C" long long mis_conv3d_wino_stat_tiles(int D, int H, int W, int variant) {
    if (D <= 0 || H <= 0 || W <= 0) return MIS_ERR_ARG;
    if (variant == 0) return boxes_per_image<WinoCfg<1, 1, 16, 2, 2, 1, 1, 4>>(D, H, W);
    if (variant == 1) return boxes_per_image<WinoCfg<1, 2, 8, 2, 2, 1, 1, 4>>(D, H, W);
    if (variant == 2) return boxes_per_image<WinoCfg<1, 4, 4, 4, 1, 1, 1, 4, 1>>(D, H, W);
    if (variant == 3) return boxes_per_image<WinoCfg<3, 3, 6, 1, 1, 1, 1, 4, 1, 1>>(D, H, W);
    return MIS_ERR_UNSUPPORTED;
}

// kernel name as rocprofv3 prints it (minus the anonymous-namespace prefix), for bench.py's attribution
extern "C" int mis_conv3d_wino_kernel_name(int variant, char* name, int name_len) {
    if (!name || name_len <= 0) return MIS_ERR_ARG;
    if (variant == 0) snprintf(name, name_len, "wino_fwd_kernel<WinoCfg<1, 1, 16, 2, 2, 1, 1, 4, 0, 0>, false>");
    else if (variant == 1) snprintf(name, name_len, "wino_fwd_kernel<WinoCfg<1, 2, 8, 2, 2, 1, 1, 4, 0, 0>, false>");
    else if (variant == 2) snprintf(name, name_len, "wino_fwd_kernel<WinoCfg<1, 4, 4, 4, 1, 1, 1, 4, 1, 0>, false>");
    else if (variant == 3) snprintf(name, name_len, "wino_fwd_kernel<WinoCfg<3, 3, 6, 1, 1, 1, 1, 4, 1, 1>, false>");
    else return MIS_ERR_UNSUPPORTED;
    return MIS_OK;
}

// y = conv3d(x, w, k = 3, 'same') + bias from the transformed filter; Cout % 16 == 0, even D, H, W.
// variant: 0 = 4 x 4 x 32 boxes (1 channel block), 1 = 4 x 4 x 16 boxes (2 channel blocks)
extern "C" int mis_conv3d_wino_fwd(const float* x, long long x_bs, const float* wt, const float* bias, float* y,
                                   long long y_bs, int N, int Cin, int Cout, int D, int H, int W, float* stat,
                                   long long stat_sc, long long stat_sn, int variant, hipStream_t stream) {
    if (!x || !wt || !y || N <= 0 || Cin <= 0 || Cout <= 0 || D <= 0 || H <= 0 || W <= 0) return MIS_ERR_ARG;
    const long long S = (long long)D * H * W;
    if (x_bs < (long long)Cin * S || y_bs < (long long)Cout * S) return MIS_ERR_ARG;
    if (Cout % 16 || Cout > 384 || Cin % 8 || Cin < 16 || (W % 4 && variant < 2) || D % 2 || H % 2 || W % 2 || y_bs % 2 || ((uintptr_t)y & 7) || ((uintptr_t)wt & 15))
        return MIS_ERR_UNSUPPORTED;
    if (((long long)Cin + 32) * S * 4 >= (1LL << 30)) return MIS_ERR_UNSUPPORTED;
    WinoArgs a{};
    a.x = x; a.x_bs = x_bs; a.wt = wt; a.bias = bias; a.y = y; a.y_bs = y_bs;
    a.N = N; a.Cin = Cin; a.Cout = Cout; a.D = D; a.H = H; a.W = W;
    a.nci4 = (Cin + 3) / 4;
    a.stat = reinterpret_cast<float2*>(stat); a.stat_sc = stat_sc; a.stat_sn = stat_sn;
    if (variant == 0) return launch_wino<WinoCfg<1, 1, 16, 2, 2, 1, 1, 4>>(a, stream);
    if (variant == 1) return launch_wino<WinoCfg<1, 2, 8, 2, 2, 1, 1, 4>>(a, stream);
    if (variant == 2) return launch_wino<WinoCfg<1, 4, 4, 4, 1, 1, 1, 4, 1>>(a, stream);
    if (variant == 3) {
        if (W != 12 || D % 6 || H % 6) return MIS_ERR_UNSUPPORTED;
        return launch_wino<WinoCfg<3, 3, 6, 1, 1, 1, 1, 4, 1, 1>>(a, stream);
    }
    return MIS_ERR_UNSUPPORTED;
}

// Contraction slices mis_conv3d_wino_fwd_ws cuts this launch into (1: none) and the floats of workspace it then needs.
extern "C" int mis_conv3d_wino_fwd_splits(int N, int Cin, int Cout, int D, int H, int W, int variant) {
    if (N <= 0 || Cin <= 0 || Cout <= 0 || D <= 0 || H <= 0 || W <= 0 || variant < 0 || variant > 3) return MIS_ERR_ARG;
    return wino_splits(N, Cin, Cout, D, H, W, variant);
}
extern "C" long long mis_conv3d_wino_fwd_workspace_bytes(int N, int Cin, int Cout, int D, int H, int W, int variant) {
    const int ks = mis_conv3d_wino_fwd_splits(N, Cin, Cout, D, H, W, variant);
    if (ks < 0) return ks;
    return ks > 1 ? (long long)ks * N * Cout * D * H * W * 4 : 0;
}

// mis_conv3d_wino_fwd with the contraction cut into slices when the launch has too few boxes for the chip (the 6^3 level of
// unet_3D / V-Net, half batches at 12^3): (box, slice) entries fill the CUs, the slices' partial outputs go to `workspace`
// and a second launch sums them in a fixed order, adds the bias and leaves the same statistics partials.  Same result as
// the unsplit launch up to the association of the sum over input channels.
extern "C" int mis_conv3d_wino_fwd_ws(const float* x, long long x_bs, const float* wt, const float* bias, float* y,
                                      long long y_bs, int N, int Cin, int Cout, int D, int H, int W, float* stat,
                                      long long stat_sc, long long stat_sn, int variant, float* workspace,
                                      long long workspace_bytes, hipStream_t stream) {
    const int ks = mis_conv3d_wino_fwd_splits(N, Cin, Cout, D, H, W, variant);
    if (ks < 0) return ks;
    if (ks == 1)
        return mis_conv3d_wino_fwd(x, x_bs, wt, bias, y, y_bs, N, Cin, Cout, D, H, W, stat, stat_sc, stat_sn, variant, stream);
    if (!x || !wt || !y || !workspace) return MIS_ERR_ARG;
    const long long S = (long long)D * H * W;
    if (x_bs < (long long)Cin * S || y_bs < (long long)Cout * S) return MIS_ERR_ARG;
    if (Cout % 16 || Cout > 384 || Cin % 8 || Cin < 16 || D % 2 || H % 2 || W % 2 || ((uintptr_t)workspace & 15) || ((uintptr_t)wt & 15))
        return MIS_ERR_UNSUPPORTED;
    if (variant == 3 && (W != 12 || D % 6 || H % 6)) return MIS_ERR_UNSUPPORTED;
    if (((long long)Cin + 32) * S * 4 >= (1LL << 30)) return MIS_ERR_UNSUPPORTED;
    if (workspace_bytes < (long long)ks * N * Cout * S * 4) return MIS_ERR_WORKSPACE;
    WinoArgs a{};
    a.x = x; a.x_bs = x_bs; a.wt = wt; a.bias = nullptr; a.y = workspace; a.y_bs = (long long)Cout * S;
    a.N = N; a.Cin = Cin; a.Cout = Cout; a.D = D; a.H = H; a.W = W;
    a.nci4 = (Cin + 3) / 4;
    a.ks = ks; a.y_ks = (long long)N * Cout * S;
    const int st = variant == 2 ? launch_wino<WinoV2, false, true>(a, stream) : launch_wino<WinoV3, false, true>(a, stream);
    if (st) return st;
    const int tiles = (int)mis_conv3d_wino_stat_tiles(D, H, W, variant);
    hipLaunchKernelGGL(wino_split_reduce_kernel, dim3((unsigned)(N * Cout)), dim3(256), 0, stream, workspace, a.y_ks, a.y_bs, bias, y,
                       y_bs, Cout, (int)S, ks, reinterpret_cast<float2*>(stat), stat_sc, stat_sn, tiles);
    return mis_launch_status();
}

// The data gradient of mis_conv3d_wino_fwd (dy -> da, transformed filter of pack mode 5) for a convolution whose input
// a = act(InstanceNorm(xn)) (no affine, negative slope `slope`, no dropout) is read by nothing else: besides da the launch
// leaves, per (n, channel) and run of boxes, part = (sum dz, sum dz * xn) with dz = da * (xn > mean ? 1 : slope) -- the
// first stage of that normalisation's backward (reference: autograd of nn.InstanceNorm3d + nn.ReLU in UnetConv3,
// code/networks/utils.py:105-109) -- in the layout of the forward's statistics partials (part_sc / part_sn strides in
// float2, mis_conv3d_wino_stat_tiles entries per image).  mis_norm_act_bwd_tiles finishes the backward from them.
// `mean`: [N][Cin_of_the_forward] = [N][Cout here]; N * Cout <= 384.  Variants 0 and 1 (the 96^3 and 48^3 levels).
extern "C" int mis_conv3d_wino_dgrad_norm(const float* dy, long long dy_bs, const float* wt, float* da, long long da_bs,
                                          int N, int Cin, int Cout, int D, int H, int W, const float* xn, long long xn_bs,
                                          const float* mean, float slope, float* part, long long part_sc,
                                          long long part_sn, int variant, hipStream_t stream) {
    if (!dy || !wt || !da || !xn || !mean || !part || N <= 0 || Cin <= 0 || Cout <= 0 || D <= 0 || H <= 0 || W <= 0)
        return MIS_ERR_ARG;
    const long long S = (long long)D * H * W;
    if (dy_bs < (long long)Cin * S || da_bs < (long long)Cout * S || xn_bs < (long long)Cout * S) return MIS_ERR_ARG;
    if (variant < 0 || variant > 1 || (long long)N * Cout > 384) return MIS_ERR_UNSUPPORTED;
    if (Cout % 16 || Cin % 8 || Cin < 16 || W % 4 || D % 2 || H % 2 || da_bs % 2 || xn_bs % 2 || ((uintptr_t)da & 7) ||
        ((uintptr_t)xn & 7) || ((uintptr_t)wt & 15))
        return MIS_ERR_UNSUPPORTED;
    if (mis_conv3d_wino_select(N, Cin, Cout, D, H, W) != variant) return MIS_ERR_UNSUPPORTED;
    if (((long long)Cin + 32) * S * 4 >= (1LL << 30) || 16 * S * 4 >= (1LL << 31)) return MIS_ERR_UNSUPPORTED;
    WinoArgs a{};
    a.x = dy; a.x_bs = dy_bs; a.wt = wt; a.bias = nullptr; a.y = da; a.y_bs = da_bs;
    a.N = N; a.Cin = Cin; a.Cout = Cout; a.D = D; a.H = H; a.W = W;
    a.nci4 = (Cin + 3) / 4;
    a.stat = reinterpret_cast<float2*>(part); a.stat_sc = part_sc; a.stat_sn = part_sn;
    a.xn = xn; a.xn_bs = xn_bs; a.thr = mean; a.slope = slope;
    if (variant == 0) return launch_wino<WinoCfg<1, 1, 16, 2, 2, 1, 1, 4>, true>(a, stream);
    return launch_wino<WinoCfg<1, 2, 8, 2, 2, 1, 1, 4>, true>(a, stream);
}
