// Winograd F(2x2x2, 3x3x3) weight gradient of the stride-1 'same' 3x3x3 convolutions of unet_3D / V-Net.
//
// Replaces the autograd weight gradient of nn.Conv3d(k=3, pad=1) in UnetConv3 / UnetUp3_CT / ConvBlock
// (reference code/networks/utils.py:99-123, code/networks/unet_3D.py:28-57, code/networks/vnet.py:15-22):
//     dw[co][ci][tap] = sum_{n, voxel} dy[n][co][voxel] * x[n][ci][voxel + tap - 1]
//
// The bilinear form of conv_wino.hip, differentiated with respect to the filter:
//     dW = G^T [ sum_tiles (A dy A^T) (.) (B^T d B) ] G          per dimension
//     A dy  = (y0, y0 + y1, y0 - y1, -y1)            the 2 outputs of a tile
//     B^T d = (d0 - d2, d1 + d2, d2 - d1, d1 - d3)   its 4 inputs
//     G^T m = (m0 + (m1 + m2) / 2, (m1 - m2) / 2, (m1 + m2) / 2 + m3)
// 64 multiplies per tile and (co, ci) instead of 216; fp32 end to end (rounding differs from the direct form).
//
// Mapping (gfx950): a workgroup owns one block of 16 output x 16 input channels and a run of STAGES (boxes of 32 tiles
// of one image).  For each transform point xi one v_mfma_f32_16x16x4_f32 contracts 4 tiles:
//   A[i = lane&15][k = lane>>4] = V_xi[co0 + i][tile k]     the lane transforms the dy patch of its (co, tile)
//   B[k = lane>>4][j = lane&15] = U_xi[ci0 + j][tile k]     ... and the x patch of its (ci, tile)
//   D[row = (lane>>4)*4 + r][col = lane&15]                 64 x 4 accumulators (AGPRs), kept for the whole run
// A wave takes 2 of the stage's 8 chunks of 4 tiles.  Both operands of a stage (haloed x of 16 channels, dy of 16
// channels) arrive by LDS-DMA (buffer_load_dwordx4 ... lds) into a double buffer, one stage ahead; the LDS image is
// linear in (row, channel, 16-byte group), so every DMA instruction is a full 64-lane piece and the patch of a lane is
// contiguous in x.  VALU work does not overlap a wave's own MFMAs on this hardware (scripts/ubench/mfma_overlap.hip),
// so the transforms sit between the MFMA runs; LDS reads and DMA issue sit inside them.
// At the end: G^T . G in registers, the 4 waves summed through LDS, one partial per workgroup; a second kernel sums the
// partials in a fixed order (deterministic) into dw[Cout][Cin][27].
#include "common.h"
#include "wino.h"
#include <stdio.h>
#include <stdlib.h>

namespace {

using namespace mis_dma;
using namespace mis_wino;

struct WgArgs {
    const float* x; long long x_bs;
    const float* dy; long long dy_bs;
    float* ws;                          // [task][27][16 co][16 ci]
    int N, Cin, Cout, D, H, W;
    int sz, sy, sx, n_stage;            // stages per image along z, y, x; N * sz * sy * sx
    int ci_blocks, co_blocks, splits;   // tasks = co_blocks * ci_blocks * splits, splits = 8 * nt (nt per XCD)
    int seg, nseg;                      // z-ring kernel: a unit = `seg` consecutive stages along z, nseg units per column
    unsigned long long* prof;           // MIS_WR_PROF builds only: per (workgroup, wave) cycle sums of the loop's phases
};

// stage = TZ x TY x TX tiles (32), TX a multiple of 4: 8 chunks of 4 x-adjacent tiles
template <int TZ_, int TY_, int TX_, int DPAD_ = 1>
struct WgCfg {
    static constexpr int TZ = TZ_, TY = TY_, TX = TX_;
    static constexpr int OZ = 2 * TZ, OY = 2 * TY, OX = 2 * TX;
    static constexpr int HZ = OZ + 2, HY = OY + 2;
    static constexpr int NQ = (OX + 8) / 4, RX = NQ * 4;          // x rows hold [x0 - 4, x0 + OX + 4)
    static constexpr int XROWS = HZ * HY, XG = XROWS * 16 * NQ, XF = XG * 4;
    static constexpr int DQ = OX / 4 + DPAD_, DRX = DQ * 4;       // dy rows: OX floats (+ one pad group: bank spread)
    static constexpr int DROWS = OZ * OY, DG = DROWS * 16 * DQ, DF = DG * 4;
    static constexpr int STAGE = XF + DF;
    static constexpr int XP = XG / 64, DP = DG / 64, P = XP + DP, PW = (P + 3) / 4;
    static constexpr int CX = TX / 4;                             // chunks along x
    static constexpr int LDS_BYTES = 2 * STAGE * 4;
    static_assert(TZ * TY * TX == 32 && TX % 4 == 0, "32 tiles, chunks of 4 along x");
    static_assert(XG % 64 == 0 && DG % 64 == 0, "whole DMA pieces");
    static_assert(LDS_BYTES <= 160 * 1024, "LDS");
    static_assert(4 * 27 * 256 * 4 <= LDS_BYTES, "the final cross-wave sum reuses the stage buffers");
    static_assert(PW <= 20, "class bits: 5 pieces per register, 4 registers");
};

#ifndef MIS_WGW_DBG_CT
#define MIS_WGW_DBG_CT 0
#endif
#ifndef MIS_WR_ABL
#define MIS_WR_ABL 0           // ablation builds of the ring kernel (timing only, results wrong): 1 no DMA in the loop, 2 no
#endif                         // transforms, 4 no patch loads, 8 no barrier, 16 no MFMAs
#ifndef MIS_WR_PROF
#define MIS_WR_PROF 0          // 1: the ring kernel times its loop phases with s_memtime (development builds, scripts/wgrad_prof.sh)
#endif
// ablation builds (timing only, results wrong): 1 no DMA issue in the loop, 2 no transforms, 4 no patch loads, 8 no barrier
constexpr int WDBG = MIS_WGW_DBG_CT;

extern __shared__ __attribute__((aligned(16))) float mis_wgw_lds[];

// V = A dy A^T in z and y of the 2x2x2 patch r[z][y] (pairs over x): 16 pairs vzy[z'][y'], 18 packed adds
__device__ __forceinline__ void vzy_transform(const f32x2 (&r)[4], f32x2 (&v)[16], f32x2 zero) {
    f32x2 a[4][2];      // z' x y
#pragma unroll
    for (int y = 0; y < 2; ++y) {
        a[0][y] = r[y];
        a[1][y] = pk_add(r[y], r[2 + y]);
        a[2][y] = pk_sub(r[y], r[2 + y]);
        a[3][y] = pk_sub(zero, r[2 + y]);
    }
#pragma unroll
    for (int z = 0; z < 4; ++z) {
        v[z * 4 + 0] = a[z][0];
        v[z * 4 + 1] = pk_add(a[z][0], a[z][1]);
        v[z * 4 + 2] = pk_sub(a[z][0], a[z][1]);
        v[z * 4 + 3] = pk_sub(zero, a[z][1]);
    }
}

__device__ __forceinline__ float f_add(float a, float b) { float r; asm volatile("v_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float f_sub(float a, float b) { float r; asm volatile("v_sub_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }

template <class C>
struct WgIssue {
    unsigned rel[C::PW];        // per lane: byte offset of this lane's 16-byte group of piece i (biased, >= 0)
    unsigned cls[4];            // 6 class bits per piece, 5 pieces per register: which faces of the box the group lies beyond
    i32x4 rx, rd;               // x / dy of the image under the cursor
    unsigned st;                // LDS byte address of the stage buffer
    unsigned soff;              // byte offset of the box origin
    unsigned flags;             // faces of the volume the box touches

    template <int I>
    __device__ __forceinline__ void piece(int wave) const {
        if constexpr (I < C::PW) {
            // uniform; the 4 * PW - P surplus slots repeat the last piece (same data to the same place: no branch
            // in the MFMA run -- a branch there makes hipcc rename the accumulators through VGPR copies)
            const int p = wave + 4 * I < C::P ? wave + 4 * I : C::P - 1;
            // which operand piece I belongs to is known at compile time when the x pieces split evenly over the waves
            // (no descriptor select, and the dy pieces need no class test: they have no halo)
            constexpr bool all_x = 4 * I + 3 < C::XP, all_d = 4 * I >= C::XP;
            if constexpr (all_d) {
                dma_dwordx4_s(st + (unsigned)p * 1024u, rel[I], soff, rd);
            } else {
                const unsigned c = (cls[I / 5] >> ((I % 5) * 6)) & 63u;
                const unsigned vo = (c & flags) ? OOB : rel[I];
                if constexpr (all_x) dma_dwordx4_s(st + (unsigned)p * 1024u, vo, soff, rx);
                else dma_dwordx4_s(st + (unsigned)p * 1024u, vo, soff, p < C::XP ? rx : rd);
            }
        }
    }
};

// One chunk: 64 MFMAs (point xi = K).  The A operand of a group of 4 points is generated from the pair vzy[K/4] by 3
// VALU instructions; inside the run: the LDS reads of the next chunk's patches, and (second chunk of a stage) the DMAs of
// the next stage.
template <class C, bool ISSUE, int K>
__device__ __forceinline__ void wg_slots(f32x2 (&u)[32], const f32x2 (&v)[16], f32x4 (&acc)[64],
                                         f32x2 (&rn)[4], const float* __restrict__ xsrc, const float* __restrict__ dsrc,
                                         const WgIssue<C>& is, int wave, float (&av)[4]) {
    if constexpr (K < 64) {
        if constexpr (K % 4 == 0) {
            const f32x2 p = v[K / 4];
            // (p0 + p1, p0 - p1) in one packed add (op_sel picks p0 for both halves of src0 and p1 for both of src1)
            f32x2 pm;
            asm volatile("v_pk_add_f32 %0, %1, %1 op_sel:[0,1] op_sel_hi:[0,1] neg_hi:[0,1]" : "=v"(pm) : "v"(p));
            av[0] = p[0]; av[1] = pm[0]; av[2] = pm[1]; av[3] = f_sub(0.f, p[1]);
        }
        acc[K] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[K % 4], u[K / 2][K % 2], acc[K], 0, 0, 0);
        if constexpr (!(WDBG & 4)) {   // next chunk's x patch, float K of the 4x4x4 patch (z, y, x) = (K / 16, (K / 4) % 4, K % 4): into the register
            // the MFMA above has just consumed (one patch buffer instead of two: 64 registers)
            constexpr int z = K / 16, y = (K / 4) % 4, xx = K % 4;
            // volatile: one ds_read_b32 with an immediate offset per float.  Left alone hipcc pairs them into ds_read2_b32
            // and pays a v_add_u32 per pair for the base (VALU time is the scarce resource here, LDS issue is free)
            u[K / 2][K % 2] = ((const volatile __attribute__((address_space(3))) float*)xsrc)[(z * C::HY + y) * 16 * C::RX + xx];
        }
        if constexpr (K < 4 && !(WDBG & 4)) {   // ... and its dy patch: (z, y) = (K / 2, K % 2), both x
            rn[K] = *reinterpret_cast<const f32x2*>(dsrc + ((K / 2) * C::OY + (K % 2)) * 16 * C::DRX);
        }
        if constexpr (ISSUE && K % 3 == 1 && !(WDBG & 1)) is.template piece<K / 3>(wave);
        __builtin_amdgcn_sched_barrier(0);
        wg_slots<C, ISSUE, K + 1>(u, v, acc, rn, xsrc, dsrc, is, wave, av);
    }
}

// ---- G^T . G: 64 points -> 27 taps for the lane's 4 (co, ci) pairs, two accumulator rows at a time; the 4 waves are
// summed through LDS (the stage buffers are free by now) and the workgroup's partial goes to out[27][16 co][16 ci] ----
__device__ __forceinline__ void wg_finish(f32x4 (&acc)[64], float* lds, float* __restrict__ out, int tid, int wave, int lt, int lc) {
    float* const red = lds + wave * (27 * 256);
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");          // the last MFMAs have left the pipe
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        f32x2 gz[3][16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const f32x2 m0 = acc_pair(acc[i], h), m1 = acc_pair(acc[16 + i], h), m2 = acc_pair(acc[32 + i], h),
                        m3 = acc_pair(acc[48 + i], h);
            const f32x2 t = (m1 + m2) * 0.5f;
            gz[0][i] = m0 + t; gz[1][i] = (m1 - m2) * 0.5f; gz[2][i] = t + m3;
        }
#pragma unroll
        for (int z = 0; z < 3; ++z) {
            f32x2 gy[3][4];
#pragma unroll
            for (int x = 0; x < 4; ++x) {
                const f32x2 t = (gz[z][4 + x] + gz[z][8 + x]) * 0.5f;
                gy[0][x] = gz[z][x] + t; gy[1][x] = (gz[z][4 + x] - gz[z][8 + x]) * 0.5f; gy[2][x] = t + gz[z][12 + x];
            }
#pragma unroll
            for (int y = 0; y < 3; ++y) {
                const f32x2 t = (gy[y][1] + gy[y][2]) * 0.5f;
                const f32x2 w0 = gy[y][0] + t, w1 = (gy[y][1] - gy[y][2]) * 0.5f, w2 = t + gy[y][3];
                const int tap = (z * 3 + y) * 3;
#pragma unroll
                for (int rr = 0; rr < 2; ++rr) {
                    const int e = (lt * 4 + 2 * h + rr) * 16 + lc;          // co * 16 + ci
                    red[(tap + 0) * 256 + e] = w0[rr];
                    red[(tap + 1) * 256 + e] = w1[rr];
                    red[(tap + 2) * 256 + e] = w2[rr];
                }
            }
        }
    }
    __syncthreads();
    for (int e = tid; e < 27 * 256; e += 256)
        out[e] = (lds[e] + lds[27 * 256 + e]) + (lds[2 * 27 * 256 + e] + lds[3 * 27 * 256 + e]);
}

template <class C>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void wino_wgrad_kernel(const WgArgs a) {
    float* const lds = mis_wgw_lds;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lt = lane >> 4, lc = lane & 15;
    const long long S = (long long)a.D * a.H * a.W;
    const unsigned s_bytes = (unsigned)S * 4u;

    // Workgroup b runs on XCD b % 8 (private L2).  XCD x owns the stages [x * per8, (x + 1) * per8); its workgroups are
    // (pair, j): all channel-block pairs of the SAME stages sit on one XCD (they share x or dy), and the nt workgroups
    // of a pair walk the range interleaved (j, j + nt, ...), so that at any time the XCD works on neighbouring boxes and
    // their halos meet in its L2 (PMC, 16->16 .. 48->16 at 96^3: 4.2 GB of HBM reads per launch with one contiguous range per workgroup)
    const int pairs = a.ci_blocks * a.co_blocks, nt = a.splits / MIS_NUM_XCD;
    const int xcd = blockIdx.x % MIS_NUM_XCD, local = blockIdx.x / MIS_NUM_XCD;
    const int pair = local % pairs, j = local / pairs;
    const int cib = pair % a.ci_blocks, cob = pair / a.ci_blocks;
    const int task = pair * a.splits + xcd * nt + j;                   // partial index: (pair, split)
    const int per8 = (a.n_stage + MIS_NUM_XCD - 1) / MIS_NUM_XCD;
    const int s_begin = xcd * per8 + j, s_lim = (xcd + 1) * per8 < a.n_stage ? (xcd + 1) * per8 : a.n_stage;
    const int s_end = s_lim;        // stages s_begin, s_begin + nt, ... < s_end

    // ---- per-lane DMA geometry (stage-invariant) ----
    const int BIAS = (a.H * a.W + a.W + 4) * 4;             // keeps the halo's negative offsets >= 0 (see rx below)
    WgIssue<C> is;
#pragma unroll
    for (int i = 0; i < 4; ++i) is.cls[i] = 0;
#pragma unroll
    for (int i = 0; i < C::PW; ++i) {
        const int p = wave + 4 * i < C::P ? wave + 4 * i : C::P - 1;
        unsigned rel = OOB, cls = 0;
        if (p < C::XP) {
            const int g = p * 64 + lane;
            const int row = g / (16 * C::NQ), rem = g - row * (16 * C::NQ), ci = rem / C::NQ, q = rem - ci * C::NQ;
            const int hz = row / C::HY, hy = row - hz * C::HY;
            rel = (unsigned)((((hz - 1) * a.H + (hy - 1)) * a.W + 4 * q - 4) * 4 + BIAS) + (unsigned)ci * s_bytes;
            cls = (hz == 0 ? 1 : 0) | (hz == C::HZ - 1 ? 2 : 0) | (hy == 0 ? 4 : 0) | (hy == C::HY - 1 ? 8 : 0) |
                  (q == 0 ? 16 : 0) | (q == C::NQ - 1 ? 32 : 0);
            if (cib * 16 + ci >= a.Cin) rel = OOB;          // channel padding of the last block
        } else {
            const int g = (p - C::XP) * 64 + lane;
            const int row = g / (16 * C::DQ), rem = g - row * (16 * C::DQ), co = rem / C::DQ, q = rem - co * C::DQ;
            const int oz = row / C::OY, oy = row - oz * C::OY;
            if (q < C::OX / 4 && cob * 16 + co < a.Cout) {
                rel = (unsigned)(((oz * a.H + oy) * a.W + 4 * q) * 4) + (unsigned)co * s_bytes;
            }
        }
        is.rel[i] = rel;
        is.cls[i / 5] |= cls << ((i % 5) * 6);
    }
    const unsigned lds0 = lds_addr(lds);

    struct Box { int n, bz, by, bx; };
    auto decode = [&](int s) {
        Box b;
        int t = s;
        b.bx = t % a.sx; t /= a.sx;
        b.by = t % a.sy; t /= a.sy;
        b.bz = t % a.sz; t /= a.sz;
        b.n = t;
        return b;
    };
    auto cursor = [&](int s, int buf) {       // uniform parts of stage s
        const Box b = decode(s < s_end ? s : s_begin);
        const bool live = s < s_end;
        // x: the descriptor starts BIAS bytes before the channel block, so that halo offsets are never negative; the
        // groups that would fall before the tensor are exactly the ones the class bits turn into padding
        is.rx = make_rsrc(reinterpret_cast<const char*>(a.x + (long long)b.n * a.x_bs + (long long)cib * 16 * S) - BIAS,
                          live ? 17u * s_bytes + (unsigned)BIAS : 0u);     // covers voffset + soffset

        is.rd = make_rsrc(a.dy + (long long)b.n * a.dy_bs + (long long)cob * 16 * S, live ? 17u * s_bytes : 0u);
        is.soff = (unsigned)(((b.bz * C::OZ) * a.H + b.by * C::OY) * a.W + b.bx * C::OX) * 4u;
        is.flags = (b.bz == 0 ? 1u : 0u) | (b.bz == a.sz - 1 ? 2u : 0u) | (b.by == 0 ? 4u : 0u) | (b.by == a.sy - 1 ? 8u : 0u) |
                   (b.bx == 0 ? 16u : 0u) | (b.bx == a.sx - 1 ? 32u : 0u);
        is.st = lds0 + (unsigned)buf * (C::STAGE * 4);
    };
    auto issue_all = [&]() {
        is.template piece<0>(wave); is.template piece<1>(wave); is.template piece<2>(wave); is.template piece<3>(wave);
        is.template piece<4>(wave); is.template piece<5>(wave); is.template piece<6>(wave); is.template piece<7>(wave);
        is.template piece<8>(wave); is.template piece<9>(wave); is.template piece<10>(wave); is.template piece<11>(wave);
        is.template piece<12>(wave); is.template piece<13>(wave); is.template piece<14>(wave); is.template piece<15>(wave);
        is.template piece<16>(wave); is.template piece<17>(wave); is.template piece<18>(wave); is.template piece<19>(wave);
    };

    // ---- this wave's two chunks (c = wave, wave + 4) and this lane's patch inside them ----
    int xoff[2], doff[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int c = wave + 4 * h;
        const int cx = c % C::CX, cy = (c / C::CX) % C::TY, cz = c / (C::CX * C::TY);
        const int tx = 4 * cx + lt;
        xoff[h] = ((2 * cz) * C::HY + 2 * cy) * 16 * C::RX + lc * C::RX + 3 + 2 * tx;
        doff[h] = ((2 * cz) * C::OY + 2 * cy) * 16 * C::DRX + lc * C::DRX + 2 * tx;
    }

    // accumulators zeroed by an MFMA (0 * 0 + 0): a tuple defined by v_accvgpr_write x 4 does not coalesce with the loop's
    // MFMA results and hipcc then copies half of the accumulators through VGPRs on every trip
    f32x4 acc[64];
    const f32x2 zero = {0.f, 0.f};
    {
        float z0 = 0.f;
        asm volatile("" : "+v"(z0));
#pragma unroll
        for (int i = 0; i < 64; ++i)
            acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(z0, z0, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
    }
    f32x2 u[32], v[16], rn[4];
    float av[4];

    if (s_begin < s_end) {
        cursor(s_begin, 0);
        issue_all();
        cursor(s_begin + nt, 1);
        issue_all();
        vmwait<0>::go();
        __syncthreads();
        // first chunk of the first stage: loaded and transformed without overlap
        {
            const float* __restrict__ xs = lds + xoff[0];
            const float* __restrict__ ds = lds + C::XF + doff[0];
#pragma unroll
            for (int k = 0; k < 64; ++k) {
                const int z = k / 16, y = (k / 4) % 4, xx = k % 4;
                u[k / 2][k % 2] = xs[(z * C::HY + y) * 16 * C::RX + xx];
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) rn[k] = *reinterpret_cast<const f32x2*>(ds + ((k / 2) * C::OY + (k % 2)) * 16 * C::DRX);
            in_units<0, 24>(u);
            vzy_transform(rn, v, zero);
        }
        for (int s = s_begin, it = 0; s < s_end; s += nt, ++it) {
            const int buf = it & 1;
            const float* __restrict__ sb = lds + buf * C::STAGE;
            // chunk A; LDS reads of chunk B of the same stage
            wg_slots<C, false, 0>(u, v, acc, rn, sb + xoff[1], sb + C::XF + doff[1], is, wave, av);
            if (!(WDBG & 2)) { in_units<0, 24>(u); vzy_transform(rn, v, zero); }
            // every wave has read stage s completely; stage s+1 (issued one stage ago) has landed
            vmwait<0>::go();
            if (!(WDBG & 8)) __syncthreads();
            cursor(s + 2 * nt, buf);         // refill this buffer with the stage after next while chunk B runs
            const float* __restrict__ nb = lds + (buf ^ 1) * C::STAGE;
            // chunk B; LDS reads of chunk A of stage s+1 (zeros after the last stage: unused)
            wg_slots<C, true, 0>(u, v, acc, rn, nb + xoff[0], nb + C::XF + doff[0], is, wave, av);
            if (!(WDBG & 2)) { in_units<0, 24>(u); vzy_transform(rn, v, zero); }
        }
    }
    vmwait<0>::go();
    __syncthreads();

    wg_finish(acc, lds, a.ws + (long long)task * (27 * 256), tid, wave, lt, lc);
}


// =====================================================================================================================
// z-ring form (round 4).  A stage is ONE pair of output planes (2 x OY x OX voxels, 32 tiles).  A workgroup owns a
// CONTIGUOUS range of the launch's stage sequence (image, y, x column, then z fastest) and walks it down z: consecutive
// stages share two of their four input planes, so the haloed x planes live in a RING of 8 plane slots in LDS and a stage
// fetches only its two NEW planes (1.9x the useful input instead of 3.75x / 3.4x for the boxes of the kernel above).  The
// new planes of stage s + 2 go to ring slots nobody reads during stage s, so they are issued in the FIRST chunk's MFMA
// run; only dy (double buffered) has to wait for the mid-stage barrier and goes out in the second run.
//
// What this kernel is really built around (measured this round): with one wave per SIMD the wave issues at
// most one instruction every ~4 cycles, of ANY kind.  A 32-cycle MFMA hides ~6 other instructions; the box kernel's DMA
// slots carried ~16 (M0 save / restore, s_nop, per-piece selects and class tests) and its per-stage cursor ~110 scalar
// instructions in one block -- that, not the bytes, was its "DMA issue cost" (halving the bytes alone gained 4 %).  Here
//   * a DMA slot is ds_read + buffer_load ... lds; M0 is set one slot earlier by a single s_add (nothing else in the
//     kernel uses M0; no save / restore, no s_nop: the MFMA in between provides the wait state);
//   * the lane's DMA offset already contains the y / x face test (recomputed once per column, not per piece), the z faces
//     are whole planes and handled by the descriptor's range (one scalar select per plane and stage);
//   * the per-stage scalar bookkeeping (~15 instructions) sits inside the MFMA runs, the part used by one run is computed
//     during the other;
//   * no stage kinds: inside a column every stage issues one plane pair + dy.  A change of column drains the pipeline and
//     re-runs the prologue (~1 stage of 48 or 24; workgroups cross 1-3 columns per launch).
template <int TY_, int TX_, int DPAD_ = 1>
struct WrCfg {
    static constexpr int TY = TY_, TX = TX_;
    static constexpr int OY = 2 * TY, OX = 2 * TX, HY = OY + 2;
    static constexpr int NQ = (OX + 8) / 4, RX = NQ * 4;                  // x rows hold [x0 - 4, x0 + OX + 4)
    static constexpr int PLG = HY * 16 * NQ, PLF = PLG * 4, PLB = PLF * 4;  // one z plane of 16 channels: groups / floats / bytes
    static constexpr int PP = PLG / 64;                                   // DMA pieces per plane
    static constexpr int PW = (PP + 3) / 4;                               // ... per wave (pieces w, w + 4, ...; surplus: the last)
    static constexpr int R = 8;                                           // ring slots
    static constexpr int DQ = OX / 4 + DPAD_, DRX = DQ * 4;
    static constexpr int DROWS = 2 * OY, DG = DROWS * 16 * DQ, DF = DG * 4, DP = DG / 64;
    static constexpr int DW = (DP + 3) / 4;                               // dy pieces per wave
    static constexpr int LDS_BYTES = R * PLB + 2 * DF * 4;
    static constexpr int CX = TX / 4;
    static_assert(TY * TX == 32 && TX % 4 == 0, "32 tiles, chunks of 4 along x");
    static_assert(PLG % 64 == 0 && DG % 64 == 0, "whole DMA pieces");
    static_assert(LDS_BYTES <= 160 * 1024, "LDS");
    static_assert(4 * 27 * 256 * 4 <= LDS_BYTES, "the final cross-wave sum reuses the ring");
    static_assert(PW == 4 && DW <= 5, "slot maps of the two runs: 2 x 4 plane pieces, 5 dy pieces");
};

// M0 := a + b (LDS byte address of the next DMA piece).  M0 is reserved for hipcc but unused by anything it generates here
// (gfx9 LDS instructions do not need it); checked in the ISA: the only writers are these statements.
// (s_add writes SCC: declared, or hipcc keeps a compare result alive across the statement)
__device__ __forceinline__ void set_m0(unsigned a, unsigned b) { asm volatile("s_add_i32 m0, %0, %1" ::"s"(a), "s"(b) : "scc"); }
__device__ __forceinline__ void dma_x4_m0(unsigned voff, unsigned soff, i32x4 rsrc) {
    asm volatile("buffer_load_dwordx4 %0, %1, %2 offen lds" ::"v"(voff), "s"(rsrc), "s"(soff) : "memory");
}

template <class C>
struct WrState {
    // per lane
    unsigned e_vo[C::PW];       // DMA byte offset of this lane's 16-byte group inside a plane (piece slot e), y / x faces applied
    unsigned d_rel[C::DW];
    // per wave
    unsigned e_off[C::PW], d_off[C::DW];            // LDS byte offset of the slot's piece inside its plane / the dy buffer
    // per column
    i32x4 rxa, rxb, rd;         // x descriptor for the first / second new plane (word 2 = range, 0: plane beyond the volume), dy
    unsigned full;              // range of a live x descriptor
    // the fill cursor (stage f + 2 while stage f computes)
    unsigned dst_a;             // LDS address of the first new plane's ring slot; the second follows it
    unsigned qa;                // its slot index (even)
    unsigned soff_a;            // byte offset of the first new plane (plane 2 of the stage being filled)
    unsigned dsoff, ddst;       // dy of the stage being filled: byte offset, LDS address
    int f_left;                 // stages of this column segment still to be filled (<= 0: dead fills)
    int f_bz;                   // z index of the stage being filled
    unsigned hw_bytes, lds0, dbase;
    int sz;

    // early plane pair of the NEXT fill: run in the second chunk (its DMAs use only dy state)
    __device__ __forceinline__ void prep_planes() {
        ++f_bz; --f_left;
        soff_a += 2u * hw_bytes;
        qa = (qa + 2u) & 7u;
        dst_a = lds0 + qa * (unsigned)C::PLB;
        rxa[2] = (int)(f_left > 0 ? full : 0u);
        rxb[2] = (int)((f_left > 0 && f_bz < sz - 1) ? full : 0u);       // plane 3 of the column's last stage: below the volume
    }
    // dy of the CURRENT fill: run in the first chunk (its DMAs use only the plane state)
    __device__ __forceinline__ void prep_dy(unsigned parity) {
        dsoff = soff_a - 2u * hw_bytes;                                  // plane 2 of a stage is its first output plane + 1 ... see soff_a
        ddst = dbase + parity * (unsigned)(C::DF * 4);
        rd[2] = (int)(f_left > 0 ? full : 0u);
    }
};

// One chunk of the ring kernel: 64 MFMAs; the next chunk's patch comes from four plane pointers (ring slots).
// SECOND = false: the stage's first chunk -- 8 plane pieces (M0 at K = 8 e + 3, DMA at 8 e + 4), dy bookkeeping at K = 62;
// SECOND = true: 5 dy pieces (M0 at K = 12 l + 5, DMA at 12 l + 6), plane bookkeeping of the next fill at K = 62.
template <class C, bool SECOND, int K>
__device__ __forceinline__ void wr_slots(f32x2 (&u)[32], const f32x2 (&v)[16], f32x4 (&acc)[64], f32x2 (&rn)[4],
                                         const float* __restrict__ x0, const float* __restrict__ x1,
                                         const float* __restrict__ x2, const float* __restrict__ x3,
                                         const float* __restrict__ dsrc, WrState<C>& st, unsigned parity, float (&av)[4]) {
    if constexpr (K < 64) {
        if constexpr (K % 4 == 0) {
            const f32x2 p = v[K / 4];
            f32x2 pm;
            asm volatile("v_pk_add_f32 %0, %1, %1 op_sel:[0,1] op_sel_hi:[0,1] neg_hi:[0,1]" : "=v"(pm) : "v"(p));
            av[0] = p[0]; av[1] = pm[0]; av[2] = pm[1]; av[3] = f_sub(0.f, p[1]);
        }
        if constexpr (!(MIS_WR_ABL & 16))
            acc[K] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[K % 4], u[K / 2][K % 2], acc[K], 0, 0, 0);
        else asm volatile("" :: "v"(av[K % 4]), "v"(u[K / 2][K % 2]));
        if constexpr (!(MIS_WR_ABL & 4)) {
            constexpr int z = K / 16, y = (K / 4) % 4, xx = K % 4;
            const float* __restrict__ xs = z == 0 ? x0 : z == 1 ? x1 : z == 2 ? x2 : x3;
            u[K / 2][K % 2] = ((const volatile __attribute__((address_space(3))) float*)xs)[y * 16 * C::RX + xx];
        }
        if constexpr (K < 4 && !(MIS_WR_ABL & 4)) rn[K] = *reinterpret_cast<const f32x2*>(dsrc + ((K / 2) * C::OY + (K % 2)) * 16 * C::DRX);
        if constexpr (MIS_WR_ABL & 1) {
            if constexpr (K == 62) { if constexpr (SECOND) st.prep_planes(); else st.prep_dy(parity); }
        } else if constexpr (!SECOND) {
            if constexpr (K % 8 == 3) {
                constexpr int E = K / 8;                                   // slots 0..3: first new plane, 4..7: second
                set_m0(E < 4 ? st.dst_a : st.dst_a + (unsigned)C::PLB, st.e_off[E % 4]);
            }
            if constexpr (K % 8 == 4) {
                constexpr int E = K / 8;
                dma_x4_m0(st.e_vo[E % 4], E < 4 ? st.soff_a : st.soff_a + st.hw_bytes, E < 4 ? st.rxa : st.rxb);
            }
            if constexpr (K == 62) st.prep_dy(parity);
        } else {
            if constexpr (K % 12 == 5 && K / 12 < C::DW) set_m0(st.ddst, st.d_off[K / 12]);
            if constexpr (K % 12 == 6 && K / 12 < C::DW) dma_x4_m0(st.d_rel[K / 12], st.dsoff, st.rd);
            if constexpr (K == 62) st.prep_planes();
        }
        __builtin_amdgcn_sched_barrier(0);
        wr_slots<C, SECOND, K + 1>(u, v, acc, rn, x0, x1, x2, x3, dsrc, st, parity, av);
    }
}

template <class C>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void wino_wgrad_ring_kernel(const WgArgs a) {
    float* const lds = mis_wgw_lds;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lt = lane >> 4, lc = lane & 15;
    const long long S = (long long)a.D * a.H * a.W;
    const unsigned s_bytes = (unsigned)S * 4u;
    const int HW = a.H * a.W;

    // workgroup -> (channel-block pair, XCD, j).  A UNIT is a segment of `seg` stages of one (image, y, x) column; units are
    // numbered (image, z segment, y, x) with x fastest.  Workgroup b runs on XCD b % 8: XCD x owns the units [x per8, (x + 1)
    // per8) and its nt workgroups per pair walk them interleaved (j, j + nt, ...), so at any time they sit on NEIGHBOURING columns
    // at the same depth, going down z in lock-step: the y / x halo rows a workgroup fetches are in the XCD's L2 for its
    // neighbours (PMC: with one contiguous stage range per workgroup -- no bubbles, perfect balance -- the 96^3 launches read
    // 3.07 GB from HBM instead of 1.7: every halo came from HBM again), and all channel-block pairs of a unit share x or dy there
    const int pairs = a.ci_blocks * a.co_blocks, nt = a.splits / MIS_NUM_XCD;
    const int xcd = blockIdx.x % MIS_NUM_XCD, local = blockIdx.x / MIS_NUM_XCD;
    // a.seg == 0 (the smaller levels: tensors that the 256 MB infinity cache holds, short columns): one CONTIGUOUS range of the
    // (image, y, x, z) stage sequence per workgroup instead -- perfect balance, a prologue only where the range crosses a
    // column end (48^3 level: 178 us against 188 us with units; at 96^3 the other way round, 367 against 386 us).  There
    // a.splits is ANY count (the host picks the one that fills whole rounds of 256 workgroups): the (range, pair) list, pair
    // fastest, is cut into 8 contiguous pieces, one per XCD -- all channel-block pairs of a range share x or dy in one L2
    // (one pair per XCD instead: 32 -> 32 at 48^3 171 -> 195 us)
    const bool by_units = a.seg > 0;
    int pair, j = 0, chunk;
    if (by_units) {
        pair = local % pairs; j = local / pairs; chunk = xcd * nt + j;
    } else {
        const int total = pairs * a.splits, per = (total + MIS_NUM_XCD - 1) / MIS_NUM_XCD, t = xcd * per + local;
        if (local >= per || t >= total) return;
        chunk = t / pairs; pair = t - chunk * pairs;
    }
    const int cib = pair % a.ci_blocks, cob = pair / a.ci_blocks;
    const int task = pair * a.splits + chunk;
    const int per8 = (a.n_stage + MIS_NUM_XCD - 1) / MIS_NUM_XCD;      // a.n_stage: UNITS (by_units) or stages of the launch
    const int u_begin = xcd * per8 + j, u_lim = (xcd + 1) * per8 < a.n_stage ? (xcd + 1) * per8 : a.n_stage;
    long long g = (long long)a.n_stage * chunk / a.splits;
    const long long g_end = (long long)a.n_stage * (chunk + 1) / a.splits;
    int unit = u_begin;

    WrState<C> st;
    st.hw_bytes = (unsigned)HW * 4u;
    st.sz = a.sz;
    st.lds0 = lds_addr(lds);
    st.dbase = st.lds0 + (unsigned)(C::R * C::PLB);
    const unsigned x_bias = (unsigned)(HW + a.W + 4) * 4u;
    st.full = 17u * s_bytes + x_bias;

    // ---- per-lane DMA geometry ----
    // x: the descriptor starts (HW + W + 4) floats before the channel block; a lane's offset inside a plane is
    // (hy W + 4 q) floats (>= 0), the plane index adds t HW through the scalar offset
    unsigned e_rel[C::PW], e_cls = 0;
#pragma unroll
    for (int e = 0; e < C::PW; ++e) {
        const int pin = wave + 4 * e < C::PP ? wave + 4 * e : C::PP - 1;       // surplus slot: the plane's last piece again
        const int g = pin * 64 + lane;
        const int hy = g / (16 * C::NQ), rem = g - hy * (16 * C::NQ), ci = rem / C::NQ, q = rem - ci * C::NQ;
        unsigned rel = (unsigned)((hy * a.W + 4 * q) * 4) + (unsigned)ci * s_bytes;
        if (cib * 16 + ci >= a.Cin) rel = OOB;
        e_rel[e] = rel;
        e_cls |= ((hy == 0 ? 1u : 0u) | (hy == C::HY - 1 ? 2u : 0u) | (q == 0 ? 4u : 0u) | (q == C::NQ - 1 ? 8u : 0u)) << (4 * e);
        st.e_off[e] = (unsigned)pin * 1024u;
    }
#pragma unroll
    for (int l = 0; l < C::DW; ++l) {
        const int p = wave + 4 * l < C::DP ? wave + 4 * l : C::DP - 1;
        const int g = p * 64 + lane;
        const int row = g / (16 * C::DQ), rem = g - row * (16 * C::DQ), co = rem / C::DQ, q = rem - co * C::DQ;
        const int oz = row / C::OY, oy = row - oz * C::OY;
        unsigned rel = OOB;
        if (q < C::OX / 4 && cob * 16 + co < a.Cout) rel = (unsigned)(((oz * a.H + oy) * a.W + 4 * q) * 4) + (unsigned)co * s_bytes;
        st.d_rel[l] = rel;
        st.d_off[l] = (unsigned)p * 1024u;
    }

    // ---- this wave's two chunks (c = wave, wave + 4) and this lane's patch inside them ----
    int xoff[2], doff[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int c = wave + 4 * h;
        const int cx = c % C::CX, cy = c / C::CX;
        const int tx = 4 * cx + lt;
        xoff[h] = (2 * cy) * 16 * C::RX + lc * C::RX + 3 + 2 * tx;
        doff[h] = (2 * cy) * 16 * C::DRX + lc * C::DRX + 2 * tx;
    }

    f32x4 acc[64];
    const f32x2 zero = {0.f, 0.f};
    {
        float z0 = 0.f;
        asm volatile("" : "+v"(z0));
#pragma unroll
        for (int i = 0; i < 64; ++i)
            acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(z0, z0, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
    }
    f32x2 u[32], v[16], rn[4];
    float av[4];
    float* const dbuf = lds + C::R * C::PLF;
#if MIS_WR_PROF
    const unsigned long long prof_c0 = __builtin_amdgcn_s_memtime(), prof_w0 = __builtin_amdgcn_s_memrealtime();
#endif

    while (by_units ? unit < u_lim : g < g_end) {
        // ---- a column segment: stages bz0 .. bz0 + cnt - 1 of column (n, by, bx) ----
        int bx, by, n, bz0, cnt;
        if (by_units) {
            int t = unit;
            bx = t % a.sx; t /= a.sx;
            by = t % a.sy; t /= a.sy;
            const int zs = t % a.nseg;
            n = t / a.nseg; bz0 = zs * a.seg; cnt = a.seg;
            unit += nt;
        } else {
            const int col = (int)(g / a.sz);
            bz0 = (int)(g - (long long)col * a.sz);
            cnt = (int)((long long)(a.sz - bz0) < g_end - g ? (long long)(a.sz - bz0) : g_end - g);
            g += cnt;
            int t = col;
            bx = t % a.sx; t /= a.sx;
            by = t % a.sy; n = t / a.sy;
        }
        const unsigned flags = (by == 0 ? 1u : 0u) | (by == a.sy - 1 ? 2u : 0u) | (bx == 0 ? 4u : 0u) | (bx == a.sx - 1 ? 8u : 0u);
#pragma unroll
        for (int e = 0; e < C::PW; ++e) st.e_vo[e] = (((e_cls >> (4 * e)) & 15u) & flags) ? OOB : e_rel[e];
        st.rxa = make_rsrc(reinterpret_cast<const char*>(a.x + (long long)n * a.x_bs + (long long)cib * 16 * S) - x_bias, st.full);
        st.rxb = st.rxa;
        st.rd = make_rsrc(a.dy + (long long)n * a.dy_bs + (long long)cob * 16 * S, st.full);
        // byte offset of plane 0 (input plane 2 bz0 - 1, biased by the descriptor base) of the segment's first stage
        const unsigned soff0 = (unsigned)(((2 * bz0) * a.H + by * C::OY) * a.W + bx * C::OX) * 4u;

        // ---- prologue (not overlapped): stage bz0 = four planes + dy, stage bz0 + 1 = its two new planes + dy ----
        // (second and later segments) the previous segment's last -- dead -- DMAs of every wave have landed and its last
        // LDS reads are done before anything is written to the ring again
        vmwait<0>::go();
        __syncthreads();
        {
            i32x4 r = st.rxa;
#pragma unroll
            for (int tt = 0; tt < 6; ++tt) {               // ring slots 0..5: planes 0..3 of the first stage, 2..3 of the second
                const int zp = 2 * bz0 - 1 + tt;
                r[2] = (int)((zp >= 0 && zp < a.D && (tt < 4 || cnt > 1)) ? st.full : 0u);
#pragma unroll
                for (int e = 0; e < C::PW; ++e) {
                    set_m0(st.lds0 + (unsigned)(tt * C::PLB), st.e_off[e]);
                    asm volatile("s_nop 0");
                    dma_x4_m0(st.e_vo[e], soff0 + (unsigned)tt * st.hw_bytes, r);
                }
            }
            i32x4 rdd = st.rd;
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                rdd[2] = (int)((k == 0 || cnt > 1) ? st.full : 0u);
#pragma unroll
                for (int l = 0; l < C::DW; ++l) {
                    set_m0(st.dbase + (unsigned)(k * C::DF * 4), st.d_off[l]);
                    asm volatile("s_nop 0");
                    dma_x4_m0(st.d_rel[l], soff0 + (unsigned)k * 2u * st.hw_bytes, rdd);
                }
            }
        }
        // fill cursor: the next fill is stage bz0 + 2 (its planes 2, 3 -> ring slots 6, 7)
        st.f_bz = bz0 + 2; st.f_left = cnt - 2;
        st.qa = 6u; st.dst_a = st.lds0 + 6u * (unsigned)C::PLB;
        st.soff_a = soff0 + (2u * 2u + 2u) * st.hw_bytes;                 // stage bz0 + 2, plane 2
        st.rxa[2] = (int)(st.f_left > 0 ? st.full : 0u);
        st.rxb[2] = (int)((st.f_left > 0 && st.f_bz < a.sz - 1) ? st.full : 0u);
        st.prep_dy(0u);
        vmwait<0>::go();
        __syncthreads();

        unsigned r_jb = 0;               // ring slot of plane 0 of the stage being computed
        auto plane_ptr = [&](unsigned jb, int z, int h) -> const float* {
            return lds + (((jb + (unsigned)z) & 7u) * (unsigned)C::PLF) + xoff[h];
        };
        {
#pragma unroll
            for (int k = 0; k < 64; ++k) {
                const int z = k / 16, y = (k / 4) % 4, xx = k % 4;
                u[k / 2][k % 2] = plane_ptr(0, z, 0)[y * 16 * C::RX + xx];
            }
#pragma unroll
            for (int k = 0; k < 4; ++k)
                rn[k] = *reinterpret_cast<const f32x2*>(dbuf + doff[0] + ((k / 2) * C::OY + (k % 2)) * 16 * C::DRX);
            in_units<0, 24>(u);
            vzy_transform(rn, v, zero);
        }
        for (int s = 0; s < cnt; ++s) {
            const unsigned par = (unsigned)(s & 1);
            // first chunk; LDS reads of the second chunk of the same stage; the new plane pair of stage s + 2
            wr_slots<C, false, 0>(u, v, acc, rn, plane_ptr(r_jb, 0, 1), plane_ptr(r_jb, 1, 1), plane_ptr(r_jb, 2, 1),
                                  plane_ptr(r_jb, 3, 1), dbuf + par * C::DF + doff[1], st, par, av);
            if (!(MIS_WR_ABL & 2)) { in_units<0, 24>(u); vzy_transform(rn, v, zero); }
            if (!(MIS_WR_ABL & 1)) vmwait<2 * C::PW>::go();   // everything but the 8 pieces just issued: stage s + 1 has landed
            if (!(MIS_WR_ABL & 8)) __syncthreads();   // every wave has read stage s completely
            // second chunk; LDS reads of the first chunk of stage s + 1 (stale data after the last stage: unused); dy of s + 2
            r_jb = (r_jb + 2u) & 7u;
            wr_slots<C, true, 0>(u, v, acc, rn, plane_ptr(r_jb, 0, 0), plane_ptr(r_jb, 1, 0), plane_ptr(r_jb, 2, 0),
                                 plane_ptr(r_jb, 3, 0), dbuf + (par ^ 1u) * C::DF + doff[0], st, par, av);
            if (!(MIS_WR_ABL & 2)) { in_units<0, 24>(u); vzy_transform(rn, v, zero); }
        }
    }
#if MIS_WR_PROF
    if (a.prof && lane == 0) {
        unsigned long long* o = a.prof + ((long long)blockIdx.x * 4 + wave) * 8;
        o[4] = __builtin_amdgcn_s_memtime() - prof_c0;
        o[0] = __builtin_amdgcn_s_memrealtime() - prof_w0;
        o[5] = by_units ? (unsigned long long)(u_begin < u_lim ? (u_lim - u_begin + nt - 1) / nt : 0) * a.seg
                        : (unsigned long long)((long long)a.n_stage * (chunk + 1) / a.splits - (long long)a.n_stage * chunk / a.splits);
    }
#endif
    vmwait<0>::go();
    __syncthreads();

    // ---- G^T . G: 64 points -> 27 taps for the lane's 4 (co, ci) pairs, two accumulator rows at a time ----
    wg_finish(acc, lds, a.ws + (long long)task * (27 * 256), tid, wave, lt, lc);
}

// =====================================================================================================================
// z-ring form for rows of 24 voxels (the 24^3 level: conv3 / up_concat3 of unet_3D, block_three / block_seven of V-Net;
// round 4).  24 does not split into the 16- or 32-voxel columns of the kernels above, and a column of 8 x 24 voxels is 48
// tiles = 12 chunks: THREE MFMA runs per wave and stage instead of two.  What changes with it:
//   * planes of 10 rows x 16 channels x 36 floats (32 + one pad group: a channel stride of 32 floats would put the 16 channel
//     lanes of a patch read on two LDS banks) = 23 KB; a ring of 6 slots (138 KB) -- the new plane pair of stage s + 1 is
//     issued in the first run of stage s into the slots stage s - 1 freed, one stage (three runs) ahead of its first read;
//   * dy does not pass through LDS (no room for a double buffer): the lane loads the 2 x 2 x 2 patch of its (co, tile) with four
//     buffer_load_dwordx2 in the first slots of the run BEFORE the one that consumes it, as the flat kernel does -- inline
//     asm, so that the vmcnt bookkeeping stays in one place (hipcc does not see the asm DMAs and would wait for vmcnt(0));
//   * one barrier per stage (after the second run: every wave has read stage s, every wave's new planes have landed).
template <int TY_, int TX_>
struct Wr3Cfg {
    static constexpr int TY = TY_, TX = TX_;
    static constexpr int OY = 2 * TY, OX = 2 * TX, HY = OY + 2;
    static constexpr int NQ = (OX + 8) / 4 + 1, RX = NQ * 4;              // x rows hold [x0 - 4, x0 + OX + 4) + one pad group
    static constexpr int PLG = HY * 16 * NQ;                              // 16-byte groups of one z plane of 16 channels
    static constexpr int PP = (PLG + 63) / 64, PLF = PP * 256, PLB = PLF * 4;   // whole DMA pieces: the slot is padded
    static constexpr int PW = (PP + 3) / 4;
    static constexpr int R = 6;
    static constexpr int NRUN = TY * TX / 16;
    static constexpr int CX = TX / 4;
    static constexpr int LDS_BYTES = R * PLB;
    static_assert(NRUN == 3 && TX % 4 == 0, "48 tiles: three chunks per wave");
    static_assert(LDS_BYTES <= 160 * 1024, "LDS");
    static_assert(4 * 27 * 256 * 4 <= LDS_BYTES, "the final cross-wave sum reuses the ring");
    static_assert(4 * (2 * PW) + 2 < 64, "DMA slots of the first run");
};

template <class C>
struct Wr3State {
    unsigned e_vo[C::PW];           // per lane: DMA byte offset of the lane's group in a plane (piece slot e), faces applied
    unsigned e_off[C::PW];          // LDS byte offset of the slot's piece inside its plane
    unsigned d_vo[C::NRUN];         // per lane: byte offset of the dy patch of (co = lane & 15, tile lane >> 4 of chunk h)
    i32x4 rxa, rxb, rd, rdn;        // x of the first / second new plane of the fill; dy of this stage / of the next (0: beyond)
    unsigned full, hw_bytes, w_bytes, lds0;
    unsigned dst_a, qa, soff_a;     // fill cursor: LDS address / ring slot / byte offset of the first new plane
    unsigned dsoff;                 // dy byte offset of the stage being computed
    int f, f_bz, cnt, sz;           // the stage the NEXT fill is for (segment-relative), its z index

    __device__ __forceinline__ void next_fill() {
        ++f; ++f_bz;
        soff_a += 2u * hw_bytes;
        qa = qa + 2u >= 6u ? qa + 2u - 6u : qa + 2u;
        dst_a = lds0 + qa * (unsigned)C::PLB;
        rxa[2] = (int)(f < cnt ? full : 0u);
        rxb[2] = (int)((f < cnt && f_bz < sz - 1) ? full : 0u);       // plane 3 of the column's last stage: below the volume
    }
};

// the lane's four dy pairs (z, y) of one chunk: buffer_load_dwordx2, scalar offsets for the plane / row steps
template <class C, int K>
__device__ __forceinline__ void wr3_dy(f32x2& r, unsigned vo, const i32x4& rsrc, unsigned base, const Wr3State<C>& st) {
    const unsigned so = base + (K / 2 ? st.hw_bytes : 0u) + (K % 2 ? st.w_bytes : 0u);
    asm volatile("buffer_load_dwordx2 %0, %1, %2, %3 offen" : "=v"(r) : "v"(vo), "s"(rsrc), "s"(so) : "memory");
}

// One run: 64 MFMAs of the current chunk; the LDS reads of the next chunk's x patch (planes x0..x3); in the first four slots
// the dy loads of the next chunk.  RUN = 0: the 2 PW plane pieces of the next stage (M0 at K = 4 e + 1, DMA at 4 e + 2);
// RUN = 1: the fill cursor moves on (K = 62).
template <class C, int RUN, int K>
__device__ __forceinline__ void wr3_slots(f32x2 (&u)[32], const f32x2 (&v)[16], f32x4 (&acc)[64], f32x2 (&rn)[4],
                                          const float* __restrict__ x0, const float* __restrict__ x1,
                                          const float* __restrict__ x2, const float* __restrict__ x3,
                                          unsigned dvo, const i32x4& drs, unsigned dbase, Wr3State<C>& st, float (&av)[4]) {
    if constexpr (K < 64) {
        if constexpr (K % 4 == 0) {
            const f32x2 p = v[K / 4];
            f32x2 pm;
            asm volatile("v_pk_add_f32 %0, %1, %1 op_sel:[0,1] op_sel_hi:[0,1] neg_hi:[0,1]" : "=v"(pm) : "v"(p));
            av[0] = p[0]; av[1] = pm[0]; av[2] = pm[1]; av[3] = f_sub(0.f, p[1]);
        }
        // asm with the accumulator tied to an AGPR tuple: with three unrolled runs hipcc's allocator otherwise parks
        // accumulator tuples in VGPRs and moves them back and forth (1200 v_accvgpr_* in the loop)
        asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(acc[K]) : "v"(av[K % 4]), "v"(u[K / 2][K % 2]));
        {
            constexpr int z = K / 16, y = (K / 4) % 4, xx = K % 4;
            const float* __restrict__ xs = z == 0 ? x0 : z == 1 ? x1 : z == 2 ? x2 : x3;
            u[K / 2][K % 2] = ((const volatile __attribute__((address_space(3))) float*)xs)[y * 16 * C::RX + xx];
        }
        if constexpr (K < 4) wr3_dy<C, K>(rn[K], dvo, drs, dbase, st);
        if constexpr (RUN == 0) {
            constexpr int E = (K - 1) / 4;                                 // slots 0 .. PW-1: first new plane, PW .. 2 PW - 1: second
            if constexpr (K >= 5 && K % 4 == 1 && E - 1 < 2 * C::PW)
                set_m0((E - 1) < C::PW ? st.dst_a : st.dst_a + (unsigned)C::PLB, st.e_off[(E - 1) % C::PW]);
            if constexpr (K >= 6 && K % 4 == 2 && E - 1 < 2 * C::PW)
                dma_x4_m0(st.e_vo[(E - 1) % C::PW], (E - 1) < C::PW ? st.soff_a : st.soff_a + st.hw_bytes,
                          (E - 1) < C::PW ? st.rxa : st.rxb);
        } else if constexpr (RUN == 1) {
            if constexpr (K == 62) st.next_fill();
        }
        __builtin_amdgcn_sched_barrier(0);
        wr3_slots<C, RUN, K + 1>(u, v, acc, rn, x0, x1, x2, x3, dvo, drs, dbase, st, av);
    }
}

template <class C>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void wino_wgrad_ring3_kernel(const WgArgs a) {
    float* const lds = mis_wgw_lds;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lt = lane >> 4, lc = lane & 15;
    const long long S = (long long)a.D * a.H * a.W;
    const unsigned s_bytes = (unsigned)S * 4u;
    const int HW = a.H * a.W;

    // workgroup -> (channel-block pair, XCD, j) and its units / stage range: as wino_wgrad_ring_kernel
    const int pairs = a.ci_blocks * a.co_blocks, nt = a.splits / MIS_NUM_XCD;
    const int xcd = blockIdx.x % MIS_NUM_XCD, local = blockIdx.x / MIS_NUM_XCD;
    const bool by_units = a.seg > 0;
    int pair, j = 0, chunk;
    if (by_units) {
        pair = local % pairs; j = local / pairs; chunk = xcd * nt + j;
    } else {
        const int total = pairs * a.splits, per = (total + MIS_NUM_XCD - 1) / MIS_NUM_XCD, t = xcd * per + local;
        if (local >= per || t >= total) return;
        chunk = t / pairs; pair = t - chunk * pairs;
    }
    const int cib = pair % a.ci_blocks, cob = pair / a.ci_blocks;
    const int task = pair * a.splits + chunk;
    const int per8 = (a.n_stage + MIS_NUM_XCD - 1) / MIS_NUM_XCD;
    const int u_begin = xcd * per8 + j, u_lim = (xcd + 1) * per8 < a.n_stage ? (xcd + 1) * per8 : a.n_stage;
    long long g = (long long)a.n_stage * chunk / a.splits;
    const long long g_end = (long long)a.n_stage * (chunk + 1) / a.splits;
    int unit = u_begin;

    Wr3State<C> st;
    st.hw_bytes = (unsigned)HW * 4u;
    st.w_bytes = (unsigned)a.W * 4u;
    st.sz = a.sz;
    st.lds0 = lds_addr(lds);
    const unsigned x_bias = (unsigned)(HW + a.W + 4) * 4u;
    st.full = 17u * s_bytes + x_bias;

    // ---- per-lane DMA geometry of a plane (see the ring kernel above): group g = (row, channel, 16-byte group) ----
    unsigned e_rel[C::PW], e_cls = 0;
#pragma unroll
    for (int e = 0; e < C::PW; ++e) {
        const int pin = wave + 4 * e < C::PP ? wave + 4 * e : C::PP - 1;       // surplus slot: the plane's last piece again
        const int gg = pin * 64 + lane;
        const int hy = gg / (16 * C::NQ), rem = gg - hy * (16 * C::NQ), ci = rem / C::NQ, q = rem - ci * C::NQ;
        unsigned rel = (unsigned)((hy * a.W + 4 * q) * 4) + (unsigned)ci * s_bytes;
        if (gg >= C::PLG || q == C::NQ - 1 || cib * 16 + ci >= a.Cin) rel = OOB;       // slot padding, pad group, channels past Cin
        e_rel[e] = rel;
        e_cls |= ((hy == 0 ? 1u : 0u) | (hy == C::HY - 1 ? 2u : 0u) | (q == 0 ? 4u : 0u) | (q == C::NQ - 2 ? 8u : 0u)) << (4 * e);
        st.e_off[e] = (unsigned)pin * 1024u;
    }

    // ---- this wave's three chunks (c = wave, wave + 4, wave + 8) and this lane's patches inside them ----
    int xoff[C::NRUN];
#pragma unroll
    for (int h = 0; h < C::NRUN; ++h) {
        const int c = wave + 4 * h;
        const int cx = c % C::CX, cy = c / C::CX;
        const int tx = 4 * cx + lt;
        xoff[h] = (2 * cy) * 16 * C::RX + lc * C::RX + 3 + 2 * tx;
        st.d_vo[h] = cob * 16 + lc < a.Cout ? (unsigned)lc * s_bytes + (unsigned)(((2 * cy) * a.W + 2 * tx) * 4) : OOB;
    }

    f32x4 acc[64];
    const f32x2 zero = {0.f, 0.f};
    {
        float z0 = 0.f;
        asm volatile("" : "+v"(z0));
#pragma unroll
        for (int i = 0; i < 64; ++i)
            acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(z0, z0, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
    }
    f32x2 u[32], v[16], rn[4];
    float av[4];
    auto land = [&]() {             // the asm loads' destinations hold data from here on
#pragma unroll
        for (int i = 0; i < 4; ++i) asm volatile("" : "+v"(rn[i]));
    };

    while (by_units ? unit < u_lim : g < g_end) {
        // ---- a column segment: stages bz0 .. bz0 + cnt - 1 of column (n, by, bx) ----
        int bx, by, n, bz0, cnt;
        if (by_units) {
            int t = unit;
            bx = t % a.sx; t /= a.sx;
            by = t % a.sy; t /= a.sy;
            const int zs = t % a.nseg;
            n = t / a.nseg; bz0 = zs * a.seg; cnt = a.seg;
            unit += nt;
        } else {
            const int col = (int)(g / a.sz);
            bz0 = (int)(g - (long long)col * a.sz);
            cnt = (int)((long long)(a.sz - bz0) < g_end - g ? (long long)(a.sz - bz0) : g_end - g);
            g += cnt;
            int t = col;
            bx = t % a.sx; t /= a.sx;
            by = t % a.sy; n = t / a.sy;
        }
        const unsigned flags = (by == 0 ? 1u : 0u) | (by == a.sy - 1 ? 2u : 0u) | (bx == 0 ? 4u : 0u) | (bx == a.sx - 1 ? 8u : 0u);
#pragma unroll
        for (int e = 0; e < C::PW; ++e) st.e_vo[e] = (((e_cls >> (4 * e)) & 15u) & flags) ? OOB : e_rel[e];
        st.rxa = make_rsrc(reinterpret_cast<const char*>(a.x + (long long)n * a.x_bs + (long long)cib * 16 * S) - x_bias, st.full);
        st.rxb = st.rxa;
        st.rd = make_rsrc(a.dy + (long long)n * a.dy_bs + (long long)cob * 16 * S, 16u * s_bytes);
        st.rdn = st.rd;
        st.cnt = cnt;
        // byte offset of the column origin at the segment's first stage: x plane 0 (input plane 2 bz0 - 1, biased by the
        // descriptor base) and the first dy plane
        const unsigned soff0 = (unsigned)(((2 * bz0) * a.H + by * C::OY) * a.W + bx * C::OX) * 4u;
        st.dsoff = soff0;

        // ---- prologue (not overlapped): the four planes of stage bz0 and the dy patch of its first chunk ----
        vmwait<0>::go();
        __syncthreads();            // the previous segment's last LDS reads are done before the ring is written again
        {
            i32x4 r = st.rxa;
#pragma unroll
            for (int tt = 0; tt < 4; ++tt) {
                const int zp = 2 * bz0 - 1 + tt;
                r[2] = (int)((zp >= 0 && zp < a.D) ? st.full : 0u);
#pragma unroll
                for (int e = 0; e < C::PW; ++e) {
                    set_m0(st.lds0 + (unsigned)(tt * C::PLB), st.e_off[e]);
                    asm volatile("s_nop 0");
                    dma_x4_m0(st.e_vo[e], soff0 + (unsigned)tt * st.hw_bytes, r);
                }
            }
            wr3_dy<C, 0>(rn[0], st.d_vo[0], st.rd, st.dsoff, st);
            wr3_dy<C, 1>(rn[1], st.d_vo[0], st.rd, st.dsoff, st);
            wr3_dy<C, 2>(rn[2], st.d_vo[0], st.rd, st.dsoff, st);
            wr3_dy<C, 3>(rn[3], st.d_vo[0], st.rd, st.dsoff, st);
        }
        // fill cursor: the first fill is stage bz0 + 1 (its planes 2, 3 -> ring slots 4, 5)
        st.f = 1; st.f_bz = bz0 + 1;
        st.qa = 4u; st.dst_a = st.lds0 + 4u * (unsigned)C::PLB;
        st.soff_a = soff0 + 4u * st.hw_bytes;
        st.rxa[2] = (int)(st.f < cnt ? st.full : 0u);
        st.rxb[2] = (int)((st.f < cnt && st.f_bz < a.sz - 1) ? st.full : 0u);
        vmwait<0>::go();
        land();
        __syncthreads();

        unsigned jb = 0;                // ring slot of plane 0 of the stage being computed
        auto plane_ptr = [&](unsigned slot, int z, int h) -> const float* {
            unsigned q = slot + (unsigned)z;
            q = q >= 6u ? q - 6u : q;
            return lds + q * (unsigned)C::PLF + xoff[h];
        };
        {
#pragma unroll
            for (int k = 0; k < 64; ++k) {
                const int z = k / 16, y = (k / 4) % 4, xx = k % 4;
                u[k / 2][k % 2] = plane_ptr(0, z, 0)[y * 16 * C::RX + xx];
            }
            in_units<0, 24>(u);
            vzy_transform(rn, v, zero);
        }
        for (int s = 0; s < cnt; ++s) {
            st.rdn[2] = (int)(s + 1 < cnt ? 16u * s_bytes : 0u);
            // first run: chunk 0; x patch and dy of chunk 1; the new plane pair of stage s + 1
            wr3_slots<C, 0, 0>(u, v, acc, rn, plane_ptr(jb, 0, 1), plane_ptr(jb, 1, 1), plane_ptr(jb, 2, 1), plane_ptr(jb, 3, 1),
                               st.d_vo[1], st.rd, st.dsoff, st, av);
            in_units<0, 24>(u);
            vmwait<2 * C::PW>::go();    // dy of chunk 1 (older than the plane pieces just issued)
            land();
            vzy_transform(rn, v, zero);
            // second run: chunk 1; x patch and dy of chunk 2; the fill cursor moves on
            wr3_slots<C, 1, 0>(u, v, acc, rn, plane_ptr(jb, 0, 2), plane_ptr(jb, 1, 2), plane_ptr(jb, 2, 2), plane_ptr(jb, 3, 2),
                               st.d_vo[2], st.rd, st.dsoff, st, av);
            in_units<0, 24>(u);
            vmwait<0>::go();            // dy of chunk 2; the planes of stage s + 1 (mine) ...
            land();
            vzy_transform(rn, v, zero);
            __syncthreads();            // ... and everyone's; every wave has read stage s completely
            // third run: chunk 2; x patch and dy of chunk 0 of stage s + 1 (after the last stage: stale planes, zero dy -- unused)
            jb = jb + 2u >= 6u ? jb + 2u - 6u : jb + 2u;
            wr3_slots<C, 2, 0>(u, v, acc, rn, plane_ptr(jb, 0, 0), plane_ptr(jb, 1, 0), plane_ptr(jb, 2, 0), plane_ptr(jb, 3, 0),
                               st.d_vo[0], st.rdn, st.dsoff + 2u * st.hw_bytes, st, av);
            in_units<0, 24>(u);
            vmwait<0>::go();
            land();
            vzy_transform(rn, v, zero);
            st.dsoff += 2u * st.hw_bytes;
        }
    }
    vmwait<0>::go();
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");       // the last (asm) MFMAs have left the pipe before their AGPRs are read
    __syncthreads();

    // ---- G^T . G: 64 points -> 27 taps for the lane's 4 (co, ci) pairs, two accumulator rows at a time ----
    wg_finish(acc, lds, a.ws + (long long)task * (27 * 256), tid, wave, lt, lc);
}

// dw[co][ci][tap] (+)= sum over the splits of the task partials [pair][split][27][16 co][16 ci].  A workgroup owns 64
// consecutive elements of a pair's 27 x 256 partial; thread (e = tid & 63, g = tid >> 6) sums the splits k = g, g + 4, ... in
// ascending order -- every load of a wave is 256 contiguous bytes -- and the four groups are combined in a fixed order through
// LDS (deterministic).  (Round 3's version gave 16 lanes to one output, each lane walking partials 27 KB apart: 64 cache lines
// per load instruction, 0.7 GB of HBM / L2 reads per config-3 step for 30 MB of partials.)
__global__ __launch_bounds__(256) void wino_wgrad_reduce_kernel(const float* __restrict__ ws, float* __restrict__ dw,
                                                                int Cout, int Cin, int ci_blocks, int splits,
                                                                int accumulate) {
    __shared__ float red[4][64];
    const int per_pair = 27 * 256 / 64;                       // 108 workgroups per channel-block pair
    const int pair = blockIdx.x / per_pair, e = (blockIdx.x - pair * per_pair) * 64 + (threadIdx.x & 63), g = threadIdx.x >> 6;
    const float* p = ws + (long long)pair * splits * (27 * 256) + e;
    float s0 = 0.f, s1 = 0.f;
    int k = g;
    for (; k + 4 < splits; k += 8) {                          // two independent chains per thread: loads in flight together
        s0 += p[(long long)k * (27 * 256)];
        s1 += p[(long long)(k + 4) * (27 * 256)];
    }
    if (k < splits) s0 += p[(long long)k * (27 * 256)];
    red[g][threadIdx.x & 63] = s0 + s1;
    __syncthreads();
    if (g == 0) {
        const int l = threadIdx.x;
        const float s = (red[0][l] + red[1][l]) + (red[2][l] + red[3][l]);
        const int tap = e / 256, co = (pair / ci_blocks) * 16 + (e % 256) / 16, ci = (pair % ci_blocks) * 16 + e % 16;
        if (co < Cout && ci < Cin) {
            const long long o = ((long long)co * Cin + ci) * 27 + tap;
            dw[o] = accumulate ? dw[o] + s : s;
        }
    }
}

// =====================================================================================================================
// Flat form for the 12^3 level (round 4).  Their tensors live in L2 (128 channels x 12^3 x 8 volumes =
// 7 MB), their 6 / 3 tiles per row do not split into the 4-tile chunks of the staged kernels above, and the channel-block
// pairs (32 ... 256 of them) already fill the chip -- so there is no LDS stage, no DMA and no barrier: a WAVE walks its own
// run of chunks, a chunk is ANY 4 tiles of the flattened (image, z, y, x) tile list (a table in LDS gives each tile's byte
// offsets) and a lane fetches its own 4 x 4 x 4 patch of x and 2 x 2 x 2 patch of dy straight into registers, one chunk
// ahead of the MFMAs that consume them (two patch buffers, ping-pong).  x comes from a re-laid, zero-padded copy
//     xp[n][ci block][z + 1][y + 1][(x + 1) / 2][ci % 16][(x + 1) % 2]
// made by one small launch: no face tests, and the 16 channel lanes of a load read 128 contiguous bytes (an x pair per
// channel) -- with the plain NCDHW layout every load instruction touched 16 channel planes, and the texture-address unit, not
// the MFMAs, set the pace (3.3 us per chunk; first version of this kernel).  64 MFMAs + ~165 vector instructions per chunk as
// in the staged kernels; all 36 loads of the next chunk are issued in the first 20 slots of the run.
struct FlatArgs {
    const float* xp;                    // packed padded x (see above); image stride = ci_blocks * PB floats
    const float* dy; long long dy_bs;
    float* ws;                          // [task][27][16 co][16 ci]
    int N, Cin, Cout, D, H, W;
    int nchunks, ci_blocks, co_blocks, splits;
};

template <int HD, int WD>
struct FlatGeoC {
    static constexpr int HP = HD + 2, WPH = (WD + 2) / 2;               // padded rows, x pairs per row
    static constexpr int ROW = WPH * 32;                                // floats per padded row of a 16-channel block
    static constexpr int row_bytes(int z, int y) { return ((z * HP + y) * ROW) * 4; }
};

template <int HD, int WD, int K>
__device__ __forceinline__ void fl_slots(const f32x2 (&u)[32], f32x2 (&un)[32], const f32x2 (&v)[16], f32x4 (&acc)[64],
                                         f32x2 (&rnn)[4], __amdgpu_buffer_rsrc_t rx, __amdgpu_buffer_rsrc_t rd, int vx, int vd,
                                         float (&av)[4]) {
    if constexpr (K < 64) {
        using G = FlatGeoC<HD, WD>;
        if constexpr (K % 4 == 0) {
            const f32x2 p = v[K / 4];
            f32x2 pm;
            asm volatile("v_pk_add_f32 %0, %1, %1 op_sel:[0,1] op_sel_hi:[0,1] neg_hi:[0,1]" : "=v"(pm) : "v"(p));
            av[0] = p[0]; av[1] = pm[0]; av[2] = pm[1]; av[3] = f_sub(0.f, p[1]);
        }
        acc[K] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[K % 4], u[K / 2][K % 2], acc[K], 0, 0, 0);
        if constexpr (K < 16) {           // rows (z, y) = (K / 4, K % 4) of the next chunk's x patch: both x pairs
            constexpr int z = K / 4, y = K % 4;
            un[2 * K] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(rx, vx, G::row_bytes(z, y), 0));
            un[2 * K + 1] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(rx, vx + 128, G::row_bytes(z, y), 0));
        }
        if constexpr (K >= 16 && K < 20) {           // ... and its dy patch: (z, y) = (j / 2, j % 2)
            constexpr int j = K - 16;
            rnn[j] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(rd, vd, (((j / 2) * HD + (j % 2)) * WD) * 4, 0));
        }
        __builtin_amdgcn_sched_barrier(0);
        fl_slots<HD, WD, K + 1>(u, un, v, acc, rnn, rx, rd, vx, vd, av);
    }
}

template <int HD, int WD>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void wino_wgrad_flat_kernel(const FlatArgs a) {
    using G = FlatGeoC<HD, WD>;
    float* const lds = mis_wgw_lds;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lt = lane >> 4, lc = lane & 15;
    const int pairs = a.ci_blocks * a.co_blocks;
    const int pair = blockIdx.x % pairs, split = blockIdx.x / pairs;
    const int cib = pair % a.ci_blocks, cob = pair / a.ci_blocks;
    const int task = pair * a.splits + split;
    const long long S = (long long)a.D * HD * WD;
    const long long PB = (long long)(a.D + 2) * G::HP * G::ROW;          // floats of one (image, channel block) of xp
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(a.xp) + (long long)cib * PB, 0, (int)((((long long)(a.N - 1) * a.ci_blocks + 1) * PB) * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(a.dy) + (long long)cob * 16 * S, 0, (int)(((long long)(a.N - 1) * a.dy_bs + ((long long)a.Cout - cob * 16) * S) * 4), 0x00020000);
    const bool d_live = cob * 16 + lc < a.Cout;
    const unsigned xc = (unsigned)lc * 8u, dc = (unsigned)(lc * S) * 4u;

    // tile table in LDS (behind the reduction area): byte offsets of tile t = (n, tz, ty, tx) in xp and in dy
    int2* const tab = reinterpret_cast<int2*>(lds + 4 * 27 * 256);
    {
        const int TX = WD / 2, TY = HD / 2, T = TX * TY * (a.D / 2), total = a.N * T;
        for (int t = tid; t < 4 * a.nchunks; t += 256) {
            int2 e = make_int2((int)OOB, (int)OOB);
            if (t < total) {
                const int n = t / T, r = t - n * T, tx = r % TX, ty = (r / TX) % TY, tz = r / (TX * TY);
                e.x = (int)(((long long)n * a.ci_blocks * PB + ((long long)(2 * tz) * G::HP + 2 * ty) * G::ROW + tx * 32) * 4);
                e.y = (int)(((long long)n * a.dy_bs + ((long long)(2 * tz) * HD + 2 * ty) * WD + 2 * tx) * 4);
            }
            tab[t] = e;
        }
    }
    __syncthreads();

    // this wave's run of chunks
    const int nw = a.splits * 4, gw = split * 4 + wave;
    const int c_begin = (int)((long long)a.nchunks * gw / nw), c_end = (int)((long long)a.nchunks * (gw + 1) / nw);

    f32x4 acc[64];
    const f32x2 zero = {0.f, 0.f};
    {
        float z0 = 0.f;
        asm volatile("" : "+v"(z0));
#pragma unroll
        for (int i = 0; i < 64; ++i)
            acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(z0, z0, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
    }
    f32x2 ua[32], ub[32], v[16], rn[4];
    float av[4];
    auto offsets = [&](int c, int& vx, int& vd) {         // a chunk past the run: everything out of range (zeros)
        int2 t = make_int2((int)OOB, (int)OOB);
        if (c < c_end) t = tab[4 * c + lt];
        vx = (unsigned)t.x != OOB ? (int)((unsigned)t.x + xc) : (int)OOB;
        vd = (d_live && (unsigned)t.y != OOB) ? (int)((unsigned)t.y + dc) : (int)OOB;
    };
    if (c_begin < c_end) {
        int vx, vd, vxn, vdn;
        offsets(c_begin, vx, vd);
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            ua[2 * k] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(rx, vx, G::row_bytes(k / 4, k % 4), 0));
            ua[2 * k + 1] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(rx, vx + 128, G::row_bytes(k / 4, k % 4), 0));
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
            rn[j] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(rd, vd, (((j / 2) * HD + (j % 2)) * WD) * 4, 0));
        offsets(c_begin + 1, vxn, vdn);
        in_units<0, 24>(ua);
        vzy_transform(rn, v, zero);
        for (int c = c_begin; c < c_end; c += 2) {
            // chunk c from ua while chunk c + 1 arrives in ub; then the roles swap (c + 1 past the run: zeros, no effect)
            fl_slots<HD, WD, 0>(ua, ub, v, acc, rn, rx, rd, vxn, vdn, av);
            offsets(c + 2, vx, vd);
            in_units<0, 24>(ub); vzy_transform(rn, v, zero);
            fl_slots<HD, WD, 0>(ub, ua, v, acc, rn, rx, rd, vx, vd, av);
            offsets(c + 3, vxn, vdn);
            in_units<0, 24>(ua); vzy_transform(rn, v, zero);
        }
    }
    __syncthreads();
    wg_finish(acc, lds, a.ws + (long long)task * (27 * 256), tid, wave, lt, lc);
}

// xp[n][cb][pz][py][pp][c][e] = x[n][cb * 16 + c][pz - 1][py - 1][2 pp + e - 1], zero outside the volume and for channels
// >= C (one thread per element of xp: the 32 lanes of an (x pair, channel block) write 128 contiguous bytes)
template <int HD, int WD>
__global__ __launch_bounds__(256) void wino_flat_pack_kernel(const float* __restrict__ x, long long x_bs, float* __restrict__ xp,
                                                             int N, int C, int CB, int D) {
    using G = FlatGeoC<HD, WD>;
    const long long total = (long long)N * CB * (D + 2) * G::HP * G::ROW;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    long long t = i;
    const int e = (int)(t & 1), c = (int)((t >> 1) & 15); t >>= 5;
    const int pp = (int)(t % G::WPH); t /= G::WPH;
    const int py = (int)(t % G::HP); t /= G::HP;
    const int pz = (int)(t % (D + 2)); t /= (D + 2);
    const int cb = (int)(t % CB), n = (int)(t / CB);
    const int ch = cb * 16 + c, xx = 2 * pp + e - 1, yy = py - 1, zz = pz - 1;
    float v = 0.f;
    if (ch < C && (unsigned)xx < (unsigned)WD && (unsigned)yy < (unsigned)HD && (unsigned)zz < (unsigned)D)
        v = x[(long long)n * x_bs + ((long long)ch * D + zz) * HD * WD + (long long)yy * WD + xx];
    xp[i] = v;
}

struct FlatGeo { int nchunks, ci_blocks, co_blocks, splits; long long part_bytes, pad_bytes; };

FlatGeo flat_geometry(int N, int Cin, int Cout, int D, int H, int W) {
    FlatGeo g;
    const int T = (D / 2) * (H / 2) * (W / 2);
    g.nchunks = (N * T + 3) / 4;
    g.ci_blocks = (Cin + 15) / 16; g.co_blocks = (Cout + 15) / 16;
    const int pairs = g.ci_blocks * g.co_blocks;
    // one resident workgroup per CU (512 registers per wave); a wave wants >= ~8 chunks to pay for the 27-tap reduction
    int splits = 256 / pairs;
    const int cap = g.nchunks / (4 * 8);
    if (splits > cap) splits = cap;
    if (splits < 1) splits = 1;
    g.splits = splits;
    g.part_bytes = ((long long)pairs * splits * 27 * 256 * 4 + 255) / 256 * 256;
    g.pad_bytes = (long long)N * g.ci_blocks * (D + 2) * (H + 2) * ((W + 2) / 2) * 32 * 4;
    return g;
}

bool flat_fits(int N, int Cin, int Cout, int D, int H, int W) {
    // 12 x 12 planes only: at 6^3 (54 chunks per pair for 8 volumes) the pack / reduce launches and the 27-tap epilogue cost
    // what the Winograd form saves (measured 71 / 108 us against the direct kernel's 62 / 104 us)
    if (!(H == 12 && W == 12) || D % 2 || D < 2 || D > 16) return false;
    const FlatGeo g = flat_geometry(N, Cin, Cout, D, H, W);
    return g.pad_bytes < (1LL << 30) && (long long)N * Cout * D * H * W * 4 < (1LL << 30) &&
           4 * 27 * 256 * 4 + (long long)g.nchunks * 4 * 8 <= 160 * 1024;
}

template <int HD, int WD>
int launch_flat(const float* x, long long x_bs, const float* dy, long long dy_bs, float* dw, float* workspace, int N, int Cin,
                int Cout, int D, int accumulate, hipStream_t stream) {
    const FlatGeo g = flat_geometry(N, Cin, Cout, D, HD, WD);
    float* xp = reinterpret_cast<float*>(reinterpret_cast<char*>(workspace) + g.part_bytes);
    const long long padded = g.pad_bytes / 4;
    hipLaunchKernelGGL((wino_flat_pack_kernel<HD, WD>), dim3((unsigned)((padded + 255) / 256)), dim3(256), 0, stream, x, x_bs, xp, N,
                       Cin, g.ci_blocks, D);
    FlatArgs a{};
    a.xp = xp; a.dy = dy; a.dy_bs = dy_bs; a.ws = workspace;
    a.N = N; a.Cin = Cin; a.Cout = Cout; a.D = D; a.H = HD; a.W = WD;
    a.nchunks = g.nchunks; a.ci_blocks = g.ci_blocks; a.co_blocks = g.co_blocks; a.splits = g.splits;
    const int ldsb = 4 * 27 * 256 * 4 + g.nchunks * 4 * 8;
    static std::atomic<unsigned long long> attr_done{0};
    if (mis_set_lds_attr(reinterpret_cast<const void*>(&wino_wgrad_flat_kernel<HD, WD>), 160 * 1024, attr_done) != MIS_OK) return MIS_ERR_LAUNCH;
    hipLaunchKernelGGL((wino_wgrad_flat_kernel<HD, WD>), dim3(g.ci_blocks * g.co_blocks * g.splits), dim3(256), ldsb, stream, a);

    hipLaunchKernelGGL(wino_wgrad_reduce_kernel, dim3(g.ci_blocks * g.co_blocks * 108), dim3(256), 0, stream, workspace, dw, Cout, Cin,
                       g.ci_blocks, g.splits, accumulate);
    return mis_launch_status();
}

template <class C>
void geometry(WgArgs& a) {
    a.sz = a.D / C::OZ; a.sy = a.H / C::OY; a.sx = a.W / C::OX;
    a.n_stage = a.N * a.sz * a.sy * a.sx;
    a.ci_blocks = (a.Cin + 15) / 16; a.co_blocks = (a.Cout + 15) / 16;
    const int pairs = a.ci_blocks * a.co_blocks;
    // nt workgroups per pair and XCD: one resident workgroup per CU (160 KB of LDS), so fill one round of the 256 CUs,
    // or two when one round would leave more than a tenth of them idle
    int nt1 = 256 / (MIS_NUM_XCD * pairs), nt2 = 512 / (MIS_NUM_XCD * pairs);
    int nt = (nt1 >= 1 && MIS_NUM_XCD * pairs * nt1 * 10 >= 256 * 9) ? nt1 : (nt2 >= 1 ? nt2 : 1);
    const int per8 = (a.n_stage + MIS_NUM_XCD - 1) / MIS_NUM_XCD;
    if (nt > per8) nt = per8;
    a.splits = MIS_NUM_XCD * nt;
}

template <class C>
int launch_wg(WgArgs a, float* dw, int accumulate, hipStream_t stream) {
    geometry<C>(a);
    static std::atomic<unsigned long long> attr_done{0};
    if (mis_set_lds_attr(reinterpret_cast<const void*>(&wino_wgrad_kernel<C>), C::LDS_BYTES, attr_done) != MIS_OK)
        return MIS_ERR_LAUNCH;
    const int tasks = a.ci_blocks * a.co_blocks * a.splits;
    hipLaunchKernelGGL(wino_wgrad_kernel<C>, dim3(tasks), dim3(256), C::LDS_BYTES, stream, a);

    hipLaunchKernelGGL(wino_wgrad_reduce_kernel, dim3(a.ci_blocks * a.co_blocks * 108), dim3(256), 0, stream, a.ws, dw, a.Cout, a.Cin,
                       a.ci_blocks, a.splits, accumulate);
    return mis_launch_status();
}

using WgV0 = WgCfg<1, 2, 16>;     // stages of 2 x 4 x 32 voxels (W a multiple of 32)
using WgV1 = WgCfg<2, 2, 8>;      // stages of 4 x 4 x 16 voxels (W a multiple of 16)
using WgV2 = WgCfg<4, 2, 4, 0>;   // stages of 8 x 4 x 8 voxels (W a multiple of 8: the 24^3 level); no dy pad group: LDS
using WrV3 = WrCfg<2, 16>;        // z-ring, stages of 2 x 4 x 32 voxels (W a multiple of 32, H of 4): the 96^3 level
using WrV4 = WrCfg<4, 8>;         // z-ring, stages of 2 x 8 x 16 voxels (W a multiple of 16, H of 8): the 48^3 level
using Wr3V = Wr3Cfg<4, 12>;       // z-ring of three runs, stages of 2 x 8 x 24 voxels (W a multiple of 24, H of 8): the 24^3 level

// z-ring geometry: units = (image, z segment, y, x) with x fastest, `seg` stages each.  The segment count and the workgroups
// per (pair, XCD) come from a small model of the busiest workgroup: units per workgroup x (seg + the cost of a unit start --
// drain, prologue of six planes, first patch load: ~1 stage), times the residency rounds (one workgroup per CU).
template <class C>
void ring_geometry(WgArgs& a) {
    a.sz = a.D / 2; a.sy = a.H / C::OY; a.sx = a.W / C::OX;
    a.ci_blocks = (a.Cin + 15) / 16; a.co_blocks = (a.Cout + 15) / 16;
    const int pairs = a.ci_blocks * a.co_blocks;
    const int cols = a.N * a.sy * a.sx;
    // long columns over tensors the infinity cache cannot hold (the 96^3 level): units.  Otherwise contiguous stage ranges
    const long long bytes = ((long long)a.Cin + a.Cout) * a.N * a.D * a.H * a.W * 4;
    // MIS_WGRAD_RING_UNITS = 1 / 2 forces units / contiguous ranges (tests run the small shapes through both)
    static const int forced = getenv("MIS_WGRAD_RING_UNITS") ? atoi(getenv("MIS_WGRAD_RING_UNITS")) : 0;
    if (forced == 2 || (forced != 1 && (a.sz < 32 || bytes <= (256LL << 20)))) {
        a.n_stage = cols * a.sz;
        // ranges per pair: the count that minimises rounds of 256 resident workgroups x (stages per range + ~3 stages of
        // prologue / final transform); e.g. 48 pairs (192 -> 64 at 24^3): 5 ranges = 240 workgroups in one round, where 8 ranges
        // = 384 workgroups took two rounds with the second half empty (305 -> 262 us)
        double best = 1e300;
        int best_s = 1;
        for (int sp = 1; sp <= 128 && sp * 4 <= (a.n_stage > 4 ? a.n_stage : 4); ++sp) {
            const long long wgs = (long long)pairs * sp;
            const double t = (double)((wgs + 255) / 256) * ((a.n_stage + sp - 1) / sp + 3.0);
            if (t < best - 1e-9) { best = t; best_s = sp; }
        }
        a.splits = best_s;
        a.seg = a.nseg = 0;
        return;
    }
    double best = 1e300;
    int best_nseg = 1, best_nt = 1;
    for (int nseg = 1; nseg <= a.sz; ++nseg) {
        if (a.sz % nseg) continue;
        const int seg = a.sz / nseg, U = cols * nseg, per8 = (U + MIS_NUM_XCD - 1) / MIS_NUM_XCD;
        for (int nt = 1; nt <= 64 && nt <= per8; ++nt) {
            const int wgs = MIS_NUM_XCD * pairs * nt;
            if (wgs > 512 && nt > 1) break;
            const int rounds = (wgs + 255) / 256;
            const int upw = (per8 + nt - 1) / nt;                       // units of the busiest workgroup
            const double t = (double)rounds * (upw * (seg + 1.0) + 0.5);
            if (t < best - 1e-9) { best = t; best_nseg = nseg; best_nt = nt; }
        }
    }
    a.nseg = best_nseg; a.seg = a.sz / best_nseg;
    a.n_stage = cols * a.nseg;                                         // UNITS (the ring kernel's a.n_stage)
    a.splits = MIS_NUM_XCD * best_nt;
}

template <class C>
int launch_ring(WgArgs a, float* dw, int accumulate, hipStream_t stream) {
    ring_geometry<C>(a);
    static std::atomic<unsigned long long> attr_done{0};
    if (mis_set_lds_attr(reinterpret_cast<const void*>(&wino_wgrad_ring_kernel<C>), C::LDS_BYTES, attr_done) != MIS_OK)
        return MIS_ERR_LAUNCH;
    const int tasks = a.ci_blocks * a.co_blocks * a.splits;
    const int grid = a.seg > 0 ? tasks : (int)(mis_cdiv(tasks, MIS_NUM_XCD) * MIS_NUM_XCD);      // contiguous ranges: 8 equal pieces
    hipLaunchKernelGGL(wino_wgrad_ring_kernel<C>, dim3(grid), dim3(256), C::LDS_BYTES, stream, a);

    hipLaunchKernelGGL(wino_wgrad_reduce_kernel, dim3(a.ci_blocks * a.co_blocks * 108), dim3(256), 0, stream, a.ws, dw, a.Cout, a.Cin,
                       a.ci_blocks, a.splits, accumulate);
    return mis_launch_status();
}

template <class C>
int launch_ring3(WgArgs a, float* dw, int accumulate, hipStream_t stream) {
    ring_geometry<C>(a);
    static std::atomic<unsigned long long> attr_done{0};
    if (mis_set_lds_attr(reinterpret_cast<const void*>(&wino_wgrad_ring3_kernel<C>), C::LDS_BYTES, attr_done) != MIS_OK)
        return MIS_ERR_LAUNCH;
    const int tasks = a.ci_blocks * a.co_blocks * a.splits;
    const int grid = a.seg > 0 ? tasks : (int)(mis_cdiv(tasks, MIS_NUM_XCD) * MIS_NUM_XCD);
    hipLaunchKernelGGL(wino_wgrad_ring3_kernel<C>, dim3(grid), dim3(256), C::LDS_BYTES, stream, a);
    hipLaunchKernelGGL(wino_wgrad_reduce_kernel, dim3(a.ci_blocks * a.co_blocks * 108), dim3(256), 0, stream, a.ws, dw, a.Cout, a.Cin,
                       a.ci_blocks, a.splits, accumulate);
    return mis_launch_status();
}

unsigned long long* g_wr_prof = nullptr;

// MIS_WGRAD_RING=0: the box kernels of round 2 for every level (A/B switch)
bool ring_enabled() {
    static const bool on = [] { const char* e = getenv("MIS_WGRAD_RING"); return !(e && e[0] == '0'); }();
    return on;
}

bool variant_fits(int variant, int D, int H, int W) {
    switch (variant) {
        case 0: return W % 32 == 0 && H % 4 == 0 && D % 2 == 0;
        case 1: return W % 16 == 0 && H % 4 == 0 && D % 4 == 0;
        case 2: return W % 8 == 0 && H % 4 == 0 && D % 8 == 0;
        case 3: return W % 32 == 0 && H % 4 == 0 && D % 2 == 0;
        case 4: return W % 16 == 0 && H % 8 == 0 && D % 2 == 0;
        case 6: return W % 24 == 0 && H % 8 == 0 && D % 2 == 0;
    }
    return false;
}

bool flat_enabled() {
    static const bool on = [] { const char* e = getenv("MIS_WGRAD_FLAT"); return !(e && e[0] == '0'); }();
    return on;
}

}  // namespace

// kernel name of a weight-gradient variant as rocprofv3 prints it (minus the anonymous-namespace prefix), for bench.py
extern "C" int mis_conv3d_wino_wgrad_kernel_name(int variant, char* name, int name_len) {
    if (!name || name_len <= 0) return MIS_ERR_ARG;
    static const char* const names[7] = {"wino_wgrad_kernel<WgCfg<1, 2, 16, 1> >", "wino_wgrad_kernel<WgCfg<2, 2, 8, 1> >",
                                         "wino_wgrad_kernel<WgCfg<4, 2, 4, 0> >", "wino_wgrad_ring_kernel<WrCfg<2, 16, 1> >",
                                         "wino_wgrad_ring_kernel<WrCfg<4, 8, 1> >", "wino_wgrad_flat_kernel<12, 12>",
                                         "wino_wgrad_ring3_kernel<Wr3Cfg<4, 12> >"};
    if (variant < 0 || variant > 6) return MIS_ERR_UNSUPPORTED;
    snprintf(name, name_len, "%s", names[variant]);
    return MIS_OK;
}

// Which variant serves the weight gradient of this 3x3x3 'same' convolution, or -1 (use mis_conv_wgrad).
// 3 / 4 / 6: the z-ring kernels (96^3 / 48^3 / 24^3 levels), 0 / 1 / 2: the box kernels, 5: the flat form (12^3)
extern "C" int mis_conv3d_wino_wgrad_select(int N, int Cin, int Cout, int D, int H, int W) {
    if (N <= 0 || Cin < 8 || Cout < 8 || D <= 0 || H <= 0 || W <= 0) return -1;
    if (((long long)17 * D * H * W + (long long)H * W + W + 64) * 4 >= (1LL << 31)) return -1;
    if (ring_enabled()) {
        if (variant_fits(3, D, H, W)) return 3;
        if (variant_fits(4, D, H, W)) return 4;
        if (variant_fits(6, D, H, W)) return 6;
    }
    for (int v = 0; v < 3; ++v)
        if (variant_fits(v, D, H, W)) return v;
    if (flat_enabled() && flat_fits(N, Cin, Cout, D, H, W)) return 5;       // 12^3: the flat form
    return -1;
}

extern "C" long long mis_conv3d_wino_wgrad_workspace_bytes(int N, int Cin, int Cout, int D, int H, int W, int variant) {
    if (N <= 0 || Cin <= 0 || Cout <= 0 || D <= 0 || H <= 0 || W <= 0) return MIS_ERR_ARG;
    if (variant == 5) {
        if (!flat_fits(N, Cin, Cout, D, H, W)) return MIS_ERR_UNSUPPORTED;
        const FlatGeo g = flat_geometry(N, Cin, Cout, D, H, W);
        return g.part_bytes + g.pad_bytes;
    }
    if (variant < 0 || variant > 6 || !variant_fits(variant, D, H, W)) return MIS_ERR_UNSUPPORTED;
    WgArgs a{};
    a.N = N; a.Cin = Cin; a.Cout = Cout; a.D = D; a.H = H; a.W = W;
    if (variant == 0) geometry<WgV0>(a); else if (variant == 1) geometry<WgV1>(a); else if (variant == 2) geometry<WgV2>(a);
    else if (variant == 3) ring_geometry<WrV3>(a); else if (variant == 4) ring_geometry<WrV4>(a); else ring_geometry<Wr3V>(a);
    return (long long)a.ci_blocks * a.co_blocks * a.splits * 27 * 256 * 4;
}

// dw[Cout][Cin][27] (+)= the weight gradient; workspace: mis_conv3d_wino_wgrad_workspace_bytes
extern "C" int mis_conv3d_wino_wgrad(const float* x, long long x_bs, const float* dy, long long dy_bs, float* dw,
                                     float* workspace, long long workspace_bytes, int N, int Cin, int Cout, int D, int H,
                                     int W, int accumulate, int variant, hipStream_t stream) {
    if (!x || !dy || !dw || !workspace || N <= 0 || Cin <= 0 || Cout <= 0 || D <= 0 || H <= 0 || W <= 0) return MIS_ERR_ARG;
    const long long S = (long long)D * H * W;
    if (x_bs < (long long)Cin * S || dy_bs < (long long)Cout * S) return MIS_ERR_ARG;
    if (Cin < 8 || Cout < 8 || ((long long)17 * S + (long long)H * W + W + 64) * 4 >= (1LL << 31)) return MIS_ERR_UNSUPPORTED;
    if (variant == 5) {
        if (!flat_fits(N, Cin, Cout, D, H, W)) return MIS_ERR_UNSUPPORTED;
        if (((uintptr_t)dy & 7) || dy_bs % 2 || ((uintptr_t)workspace & 255)) return MIS_ERR_UNSUPPORTED;
        if (workspace_bytes < mis_conv3d_wino_wgrad_workspace_bytes(N, Cin, Cout, D, H, W, 5)) return MIS_ERR_WORKSPACE;
        return launch_flat<12, 12>(x, x_bs, dy, dy_bs, dw, workspace, N, Cin, Cout, D, accumulate, stream);
    }
    if (variant < 0 || variant > 6 || !variant_fits(variant, D, H, W)) return MIS_ERR_UNSUPPORTED;
    if (variant == 6 && (((uintptr_t)dy & 7) || dy_bs % 2)) return MIS_ERR_UNSUPPORTED;
    if (((uintptr_t)x & 15) || ((uintptr_t)dy & 15) || x_bs % 4 || dy_bs % 4 || W % 4) return MIS_ERR_UNSUPPORTED;
    if (workspace_bytes < mis_conv3d_wino_wgrad_workspace_bytes(N, Cin, Cout, D, H, W, variant)) return MIS_ERR_WORKSPACE;
    WgArgs a{};
    a.x = x; a.x_bs = x_bs; a.dy = dy; a.dy_bs = dy_bs; a.ws = workspace;
    a.N = N; a.Cin = Cin; a.Cout = Cout; a.D = D; a.H = H; a.W = W;
    if (variant == 0) return launch_wg<WgV0>(a, dw, accumulate, stream);
    if (variant == 1) return launch_wg<WgV1>(a, dw, accumulate, stream);
    if (variant == 2) return launch_wg<WgV2>(a, dw, accumulate, stream);
    a.prof = g_wr_prof;
    if (variant == 3) return launch_ring<WrV3>(a, dw, accumulate, stream);
    if (variant == 6) return launch_ring3<Wr3V>(a, dw, accumulate, stream);
    return launch_ring<WrV4>(a, dw, accumulate, stream);
}

// Development hook (MIS_WR_PROF builds): device buffer of 8 x uint64 per (workgroup, wave) that the z-ring kernel fills
// with the cycles it spent in [set-up, first chunk + transforms, DMA wait, barrier, second chunk + transforms] and its
// stage count.  NULL (the default) switches it off; the product build ignores it.
extern "C" int mis_debug_wgrad_prof(unsigned long long* buf) {
    g_wr_prof = buf;
    return MIS_WR_PROF ? MIS_OK : MIS_ERR_UNSUPPORTED;
}
