// Winograd F(2x2, 3x3) weight gradient of the stride-1 'same' 3x3 convolutions of the 2-D UNet.
//
// Replaces the autograd weight gradient of nn.Conv2d(k=3, padding=1) in ConvBlock (reference code/networks/unet.py:30-45):
//     dw[co][ci][tap] = sum_{n, pixel} dy[n][co][pixel] * x[n][ci][pixel + tap - 1]
// as  dW = G^T [ sum_tiles (A dy A^T) (.) (B^T d B) ] G  per dimension (see conv_wino_wgrad.hip for the 3-D form): 16
// multiplies per 2x2 tile and (co, ci) instead of 36; fp32 end to end.
//
// Mapping (gfx950): a workgroup owns COB blocks of 16 output channels x one block of 16 input channels and a run of
// stages (boxes of 8 x 16 pixels = 32 tiles of one image).  One v_mfma_f32_16x16x4_f32 per transform point (16) and
// output block contracts 4 tiles:
//   A[i = lane&15][k = lane>>4] = V_xi[co0 + i][tile k]   (the lane transforms the dy patch of its (co, tile))
//   B[k = lane>>4][j = lane&15] = U_xi[ci0 + j][tile k]   (... and the x patch of its (ci, tile): shared by the COB blocks)
// 64 x COB accumulator registers per lane, kept for the whole run; two workgroups share a CU, so one wave's transforms
// run under another's MFMAs.  x (haloed, 16 channels) and dy stages are double-buffered by LDS-DMA; the LDS image is
// linear in (row, channel, 16-byte group).  XCD-aware interleaved stage order as in the 3-D kernel.  At the end G^T . G
// (16 -> 9) in registers, waves summed through LDS, partials summed in a fixed order by a second kernel (deterministic).
#include <cstdio>
#include "common.h"
#include "wino.h"
#include <stdlib.h>

namespace {

using namespace mis_dma;
using namespace mis_wino;

struct Wg2Args {
    const float* x; long long x_bs;
    const float* dy; long long dy_bs;
    float* ws;                          // [task][9][COB * 16 co][16 ci]
    int N, Cin, Cout, H, W;
    int sy, sx, n_stage;                // stages per image along y, x; N * sy * sx
    int ci_blocks, co_groups, splits;   // tasks = co_groups * ci_blocks * splits, splits = 8 * nt
};

template <int COB_, int TY_ = 4, int TX_ = 8>
struct Wg2Cfg {
    static constexpr int COB = COB_, TY = TY_, TX = TX_;             // stage = 4 x 8 (or 2 x 16) tiles: 8 chunks of 4 x-adjacent tiles
    static constexpr int OY = 2 * TY, OX = 2 * TX, HY = OY + 2;
    static constexpr int NQ = (OX + 8) / 4, RX = NQ * 4;             // x rows hold [x0 - 4, x0 + OX + 4)
    static constexpr int XG = HY * 16 * NQ, XF = XG * 4;
    static constexpr int DQ = OX / 4 + 1, DRX = DQ * 4;              // dy rows: OX floats + one pad group (bank spread)
    static constexpr int DC = COB * 16;                              // dy channels per stage
    static constexpr int DG = OY * DC * DQ, DF = DG * 4;
    static constexpr int STAGE = XF + DF;
    static constexpr int XP = XG / 64, DP = DG / 64, P = XP + DP, PW = (P + 3) / 4;
    static constexpr int CX = TX / 4;
    static constexpr int RED = 9 * DC * 16;                          // floats of one wave's transformed partial
    static constexpr int LDS_FLOATS = 2 * STAGE > 4 * RED ? 2 * STAGE : 4 * RED;
    static constexpr int LDS_BYTES = LDS_FLOATS * 4;
    static_assert(TY * TX == 32 && TX % 4 == 0, "32 tiles per stage");
    static_assert(XG % 64 == 0 && DG % 64 == 0, "whole DMA pieces");
    static_assert(LDS_BYTES <= 80 * 1024, "two workgroups per CU");
    static_assert(PW <= 10 && PW <= (COB == 1 ? 8 : 10), "class bits: 5 pieces per register, 2 registers; DMA slots");
};

extern __shared__ __attribute__((aligned(16))) float mis_wg2_lds[];

__device__ __forceinline__ float f_add(float a, float b) { float r; asm volatile("v_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float f_sub(float a, float b) { float r; asm volatile("v_sub_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }

template <class C>
struct Wg2Issue {
    unsigned rel[C::PW];        // per lane: byte offset of this lane's 16-byte group of piece i (biased, >= 0)
    unsigned cls[2];            // 4 class bits per piece (y low / high, x low / high face), 5 pieces per register
    i32x4 rx, rd;
    unsigned st, soff, flags;

    template <int I>
    __device__ __forceinline__ void piece(int wave) const {
        if constexpr (I < C::PW) {
            const int p = wave + 4 * I < C::P ? wave + 4 * I : C::P - 1;       // surplus slots repeat the last piece
            const unsigned c = (cls[I / 5] >> ((I % 5) * 4)) & 15u;
            const unsigned vo = (c & flags) ? OOB : rel[I];
            dma_dwordx4_s(st + (unsigned)p * 1024u, vo, soff, p < C::XP ? rx : rd);
        }
    }
};

// One chunk: 16 x COB MFMAs.  Slot K = b * 16 + xi.  The A operands of 4 points come from the pair vy[b][xi / 4] by 3 VALU
// instructions.  The next chunk's x patch is loaded into the register the LAST block's MFMA has just consumed.
template <class C, bool ISSUE, int K>
__device__ __forceinline__ void wg2_slots(f32x2 (&u)[8], const f32x2 (&vy)[C::COB][4], f32x4 (&acc)[C::COB][16],
                                          f32x2 (&rn)[C::COB][2], const float* __restrict__ xsrc,
                                          const float* __restrict__ dsrc, const Wg2Issue<C>& is, int wave, float (&av)[4]) {
    if constexpr (K < 16 * C::COB) {
        constexpr int b = K / 16, xi = K % 16;
        if constexpr (xi % 4 == 0) {
            const f32x2 p = vy[b][xi / 4];
            av[0] = p[0]; av[1] = f_add(p[0], p[1]); av[2] = f_sub(p[0], p[1]); av[3] = f_sub(0.f, p[1]);
        }
        acc[b][xi] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[xi % 4], u[xi / 2][xi % 2], acc[b][xi], 0, 0, 0);
        if constexpr (b == C::COB - 1) {
            constexpr int y = xi / 4, xx = xi % 4;
            // volatile LDS pointer: one ds_read_b32 with an immediate offset (see conv_wino_wgrad.hip)
            u[xi / 2][xi % 2] = ((const volatile __attribute__((address_space(3))) float*)xsrc)[y * 16 * C::RX + xx];
        }
        if constexpr (xi < 2) rn[b][xi] = *reinterpret_cast<const f32x2*>(dsrc + (xi * C::DC + b * 16) * C::DRX);
        constexpr int SP = C::COB == 1 ? 2 : 3;                              // DMA spacing: PW pieces fit the run
        if constexpr (ISSUE && K % SP == 1) is.template piece<K / SP>(wave);
        __builtin_amdgcn_sched_barrier(0);
        wg2_slots<C, ISSUE, K + 1>(u, vy, acc, rn, xsrc, dsrc, is, wave, av);
    }
}

template <class C>
__global__ __launch_bounds__(256, 2) void wino2d_wgrad_kernel(const Wg2Args a) {
    float* const lds = mis_wg2_lds;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lt = lane >> 4, lc = lane & 15;
    const long long S = (long long)a.H * a.W;
    const unsigned s_bytes = (unsigned)S * 4u;

    const int pairs = a.ci_blocks * a.co_groups, nt = a.splits / MIS_NUM_XCD;
    const int xcd = blockIdx.x % MIS_NUM_XCD, local = blockIdx.x / MIS_NUM_XCD;
    const int pair = local % pairs, j = local / pairs;
    const int cib = pair % a.ci_blocks, cog = pair / a.ci_blocks;
    const int task = pair * a.splits + xcd * nt + j;
    const int per8 = (a.n_stage + MIS_NUM_XCD - 1) / MIS_NUM_XCD;
    const int s_begin = xcd * per8 + j;
    const int s_end = (xcd + 1) * per8 < a.n_stage ? (xcd + 1) * per8 : a.n_stage;

    const int BIAS = (a.W + 4) * 4;
    Wg2Issue<C> is;
    is.cls[0] = is.cls[1] = 0;
#pragma unroll
    for (int i = 0; i < C::PW; ++i) {
        const int p = wave + 4 * i < C::P ? wave + 4 * i : C::P - 1;
        unsigned rel = OOB, cls = 0;
        if (p < C::XP) {
            const int g = p * 64 + lane;
            const int hy = g / (16 * C::NQ), rem = g - hy * (16 * C::NQ), ci = rem / C::NQ, q = rem - ci * C::NQ;
            rel = (unsigned)(((hy - 1) * a.W + 4 * q - 4) * 4 + BIAS) + (unsigned)ci * s_bytes;
            cls = (hy == 0 ? 1 : 0) | (hy == C::HY - 1 ? 2 : 0) | (q == 0 ? 4 : 0) | (q == C::NQ - 1 ? 8 : 0);
            if (cib * 16 + ci >= a.Cin) rel = OOB;
        } else {
            const int g = (p - C::XP) * 64 + lane;
            const int oy = g / (C::DC * C::DQ), rem = g - oy * (C::DC * C::DQ), co = rem / C::DQ, q = rem - co * C::DQ;
            if (q < C::OX / 4 && cog * C::DC + co < a.Cout) rel = (unsigned)((oy * a.W + 4 * q) * 4) + (unsigned)co * s_bytes;
        }
        is.rel[i] = rel;
        is.cls[i / 5] |= cls << ((i % 5) * 4);
    }
    const unsigned lds0 = lds_addr(lds);
    auto cursor = [&](int s, int buf) {
        const bool live = s < s_end;
        int t = live ? s : s_begin;
        const int bx = t % a.sx; t /= a.sx;
        const int by = t % a.sy; t /= a.sy;
        const int n = t;
        is.rx = make_rsrc(reinterpret_cast<const char*>(a.x + (long long)n * a.x_bs + (long long)cib * 16 * S) - BIAS,
                          live ? 17u * s_bytes + (unsigned)BIAS : 0u);
        is.rd = make_rsrc(a.dy + (long long)n * a.dy_bs + (long long)cog * C::DC * S, live ? (unsigned)(C::DC + 1) * s_bytes : 0u);
        is.soff = (unsigned)((by * C::OY) * a.W + bx * C::OX) * 4u;
        is.flags = (by == 0 ? 1u : 0u) | (by == a.sy - 1 ? 2u : 0u) | (bx == 0 ? 4u : 0u) | (bx == a.sx - 1 ? 8u : 0u);
        is.st = lds0 + (unsigned)buf * (C::STAGE * 4);
    };
    auto issue_all = [&]() {
        is.template piece<0>(wave); is.template piece<1>(wave); is.template piece<2>(wave); is.template piece<3>(wave);
        is.template piece<4>(wave); is.template piece<5>(wave); is.template piece<6>(wave); is.template piece<7>(wave);
        is.template piece<8>(wave); is.template piece<9>(wave);
    };

    // this wave's two chunks (c = wave, wave + 4: tile rows c / 2, x halves c % 2) and this lane's patch inside them
    int xoff[2], doff[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int c = wave + 4 * h, cx = c % C::CX, cy = c / C::CX, tx = 4 * cx + lt;
        xoff[h] = (2 * cy) * 16 * C::RX + lc * C::RX + 3 + 2 * tx;
        doff[h] = (2 * cy) * C::DC * C::DRX + lc * C::DRX + 2 * tx;
    }

    f32x4 acc[C::COB][16];
    {
        float z0 = 0.f;
        asm volatile("" : "+v"(z0));
#pragma unroll
        for (int b = 0; b < C::COB; ++b)
#pragma unroll
            for (int i = 0; i < 16; ++i)
                acc[b][i] = __builtin_amdgcn_mfma_f32_16x16x4f32(z0, z0, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
    }
    const f32x2 zero = {0.f, 0.f};
    f32x2 u[8], vy[C::COB][4], rn[C::COB][2];
    float av[4];
    auto transforms = [&]() {
#pragma unroll
        for (int r = 0; r < 4; ++r) bt4_inner(u[r * 2], u[r * 2 + 1]);
        bt4(u[0], u[2], u[4], u[6]);
        bt4(u[1], u[3], u[5], u[7]);
#pragma unroll
        for (int b = 0; b < C::COB; ++b) {
            vy[b][0] = rn[b][0];
            vy[b][1] = pk_add(rn[b][0], rn[b][1]);
            vy[b][2] = pk_sub(rn[b][0], rn[b][1]);
            vy[b][3] = pk_sub(zero, rn[b][1]);
        }
        // The packed adds above are inline asm: hipcc does not know they are VALU writes and inserts no wait states between one
        // of them and an MFMA that reads its result.  Left alone it hoisted the run's first MFMA to TWO instructions behind the
        // v_pk_add_f32 that produces its B operand (Wg2Cfg<1, 2, 16>): lanes 48 - 63 -- the last pass of the wave -- then read the
        // register before it was written, timing-dependent (found as errors in tap (0, 0) only, i.e. transform point 0).  Every
        // operand of the run passes through this statement, which sits behind all the adds and carries the wait states.
        asm volatile("s_nop 3" : "+v"(u[0]), "+v"(u[1]), "+v"(u[2]), "+v"(u[3]), "+v"(u[4]), "+v"(u[5]), "+v"(u[6]), "+v"(u[7]));
#pragma unroll
        for (int b = 0; b < C::COB; ++b) asm volatile("" : "+v"(vy[b][0]), "+v"(vy[b][1]), "+v"(vy[b][2]), "+v"(vy[b][3]));
    };

    if (s_begin < s_end) {
        cursor(s_begin, 0);
        issue_all();
        cursor(s_begin + nt, 1);
        issue_all();
        vmwait<0>::go();
        __syncthreads();
        {
            const float* __restrict__ xs = lds + xoff[0];
            const float* __restrict__ ds = lds + C::XF + doff[0];
#pragma unroll
            for (int k = 0; k < 16; ++k) u[k / 2][k % 2] = xs[(k / 4) * 16 * C::RX + k % 4];
#pragma unroll
            for (int b = 0; b < C::COB; ++b)
#pragma unroll
                for (int y = 0; y < 2; ++y) rn[b][y] = *reinterpret_cast<const f32x2*>(ds + (y * C::DC + b * 16) * C::DRX);
            transforms();
        }
        for (int s = s_begin, it = 0; s < s_end; s += nt, ++it) {
            const int buf = it & 1;
            const float* __restrict__ sb = lds + buf * C::STAGE;
            wg2_slots<C, false, 0>(u, vy, acc, rn, sb + xoff[1], sb + C::XF + doff[1], is, wave, av);
            transforms();
            vmwait<0>::go();                 // stage s + nt (issued one stage ago) has landed ...
            __syncthreads();                 // ... for everyone, and everyone has read stage s completely
            cursor(s + 2 * nt, buf);
            const float* __restrict__ nb = lds + (buf ^ 1) * C::STAGE;
            wg2_slots<C, true, 0>(u, vy, acc, rn, nb + xoff[0], nb + C::XF + doff[0], is, wave, av);
            transforms();
        }
    }
    vmwait<0>::go();
    __syncthreads();

    // ---- G^T . G: 16 points -> 9 taps for the lane's 4 x COB (co, ci) pairs ----
    float* const red = lds + wave * C::RED;
#pragma unroll
    for (int b = 0; b < C::COB; ++b) {
        f32x4 gy[3][4];
#pragma unroll
        for (int x = 0; x < 4; ++x) {
            const f32x4 m0 = acc[b][x], m1 = acc[b][4 + x], m2 = acc[b][8 + x], m3 = acc[b][12 + x];
            const f32x4 t = (m1 + m2) * 0.5f;
            gy[0][x] = m0 + t; gy[1][x] = (m1 - m2) * 0.5f; gy[2][x] = t + m3;
        }
#pragma unroll
        for (int y = 0; y < 3; ++y) {
            const f32x4 t = (gy[y][1] + gy[y][2]) * 0.5f;
            const f32x4 w0 = gy[y][0] + t, w1 = (gy[y][1] - gy[y][2]) * 0.5f, w2 = t + gy[y][3];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int e = (b * 16 + lt * 4 + r) * 16 + lc;         // co * 16 + ci
                red[(y * 3 + 0) * (C::DC * 16) + e] = w0[r];
                red[(y * 3 + 1) * (C::DC * 16) + e] = w1[r];
                red[(y * 3 + 2) * (C::DC * 16) + e] = w2[r];
            }
        }
    }
    __syncthreads();
    float* __restrict__ out = a.ws + (long long)task * C::RED;
    for (int e = tid; e < C::RED; e += 256)
        out[e] = (lds[e] + lds[C::RED + e]) + (lds[2 * C::RED + e] + lds[3 * C::RED + e]);
}

// dw[co][ci][tap] (+)= sum over the splits of the task partials [pair][split][9][dc co][16 ci].  As in conv_wino_wgrad.hip
// (round 4): a workgroup owns 64 consecutive elements of a pair's partial, thread (e = tid & 63, g = tid >> 6) sums the splits
// k = g, g + 4, ... in ascending order -- 256 contiguous bytes per load of a wave -- and the four groups are combined in a fixed
// order through LDS.  (Before: 16 lanes per output walking partials one whole partial apart, 64 cache lines per load.)
__global__ __launch_bounds__(256) void wino2d_wgrad_reduce_kernel(const float* __restrict__ ws, float* __restrict__ dw,
                                                                  int Cout, int Cin, int ci_blocks, int splits, int dc,
                                                                  int accumulate) {
    __shared__ float red_s[4][64];
    const int red = 9 * dc * 16;                              // elements of one partial (a multiple of 64: dc = 16 / 32)
    const int per_pair = red / 64;
    const int pair = blockIdx.x / per_pair, e = (blockIdx.x - pair * per_pair) * 64 + (threadIdx.x & 63), g = threadIdx.x >> 6;
    const float* p = ws + (long long)pair * splits * red + e;
    float s0 = 0.f, s1 = 0.f;
    int k = g;
    for (; k + 4 < splits; k += 8) {
        s0 += p[(long long)k * red];
        s1 += p[(long long)(k + 4) * red];
    }
    if (k < splits) s0 += p[(long long)k * red];
    red_s[g][threadIdx.x & 63] = s0 + s1;
    __syncthreads();
    if (g == 0) {
        const int l = threadIdx.x;
        const float s = (red_s[0][l] + red_s[1][l]) + (red_s[2][l] + red_s[3][l]);
        const int tap = e / (dc * 16), r = e - tap * (dc * 16);
        const int co = (pair / ci_blocks) * dc + r / 16, ci = (pair % ci_blocks) * 16 + r % 16;
        if (co < Cout && ci < Cin) {
            const long long o = ((long long)co * Cin + ci) * 9 + tap;
            dw[o] = accumulate ? dw[o] + s : s;
        }
    }
}

template <class C>
void geometry2(Wg2Args& a) {
    a.sy = a.H / C::OY; a.sx = a.W / C::OX;
    a.n_stage = a.N * a.sy * a.sx;
    a.ci_blocks = (a.Cin + 15) / 16; a.co_groups = (a.Cout + C::DC - 1) / C::DC;
    const int pairs = a.ci_blocks * a.co_groups;
    // two resident workgroups per CU: fill one round of the 512 slots, or two when one would leave > 10 % idle
    int nt1 = 512 / (MIS_NUM_XCD * pairs), nt2 = 1024 / (MIS_NUM_XCD * pairs);
    int nt = (nt1 >= 1 && MIS_NUM_XCD * pairs * nt1 * 10 >= 512 * 9) ? nt1 : (nt2 >= 1 ? nt2 : 1);
    const int per8 = (a.n_stage + MIS_NUM_XCD - 1) / MIS_NUM_XCD;
    if (nt > per8) nt = per8;
    a.splits = MIS_NUM_XCD * nt;
}

template <class C>
int launch_wg2(Wg2Args a, float* dw, int accumulate, hipStream_t stream) {
    geometry2<C>(a);
    static std::atomic<unsigned long long> attr_done{0};
    if (mis_set_lds_attr(reinterpret_cast<const void*>(&wino2d_wgrad_kernel<C>), C::LDS_BYTES, attr_done) != MIS_OK)
        return MIS_ERR_LAUNCH;
    const int tasks = a.ci_blocks * a.co_groups * a.splits;
    hipLaunchKernelGGL(wino2d_wgrad_kernel<C>, dim3(tasks), dim3(256), C::LDS_BYTES, stream, a);
    static_assert((9 * C::DC * 16) % 64 == 0, "whole 64-element pieces of a partial");
    hipLaunchKernelGGL(wino2d_wgrad_reduce_kernel, dim3(a.ci_blocks * a.co_groups * (9 * C::DC * 16 / 64)), dim3(256), 0, stream, a.ws,
                       dw, a.Cout, a.Cin, a.ci_blocks, a.splits, C::DC, accumulate);
    return mis_launch_status();
}

// wide stages (round 4): 2 x 16 tiles = 4 x 32 pixels: dy rows of 128 bytes, x rows of 160 instead of 64 / 96 (the same halo
// ratio and LDS; scripts/ubench/hbm_pieces.hip: HBM gives 64-byte pieces 3.1 TB/s, 128-byte pieces 4.4)
using Wg2W1 = Wg2Cfg<1, 2, 16>;
using Wg2W2 = Wg2Cfg<2, 2, 16>;

}  // namespace

// variant serving the weight gradient of this 3x3 'same' convolution (D = 1): 0 = one block of 16 output channels per
// workgroup, 1 = two (stages of 8 x 16 pixels: H % 8, W % 16); 2 / 3 = the same with stages of 4 x 32 pixels (H % 4, W % 32);
// -1 = use mis_conv_wgrad.  Cin, Cout >= 8.
extern "C" int mis_conv2d_wino_wgrad_select(int N, int Cin, int Cout, int H, int W) {
    if (N <= 0 || Cin < 8 || Cout < 8 || H <= 0 || W <= 0) return -1;
    if (((long long)33 * H * W + W + 64) * 4 >= (1LL << 31)) return -1;
    static const bool wide = [] { const char* e = getenv("MIS_W2_WIDE"); return !(e && e[0] == '0'); }();
    if (wide && H % 4 == 0 && W % 32 == 0) return Cout > 16 ? 3 : 2;
    if (H % 8 || W % 16) return -1;
    return Cout > 16 ? 1 : 0;
}

// kernel a variant launches as rocprofv3 prints it (minus the anonymous-namespace prefix): bench.py keys its flop
// attribution (the 2.25x Winograd factor) on this string
extern "C" int mis_conv2d_wino_wgrad_kernel_name(int variant, char* name, int name_len) {
    if (!name || name_len <= 0) return MIS_ERR_ARG;
    static const char* const names[4] = {"wino2d_wgrad_kernel<Wg2Cfg<1, 4, 8> >", "wino2d_wgrad_kernel<Wg2Cfg<2, 4, 8> >",
                                         "wino2d_wgrad_kernel<Wg2Cfg<1, 2, 16> >", "wino2d_wgrad_kernel<Wg2Cfg<2, 2, 16> >"};
    if (variant < 0 || variant > 3) return MIS_ERR_UNSUPPORTED;
    snprintf(name, name_len, "%s", names[variant]);
    return MIS_OK;
}

extern "C" long long mis_conv2d_wino_wgrad_workspace_bytes(int N, int Cin, int Cout, int H, int W, int variant) {
    if (N <= 0 || Cin <= 0 || Cout <= 0 || H <= 0 || W <= 0) return MIS_ERR_ARG;
    Wg2Args a{};
    a.N = N; a.Cin = Cin; a.Cout = Cout; a.H = H; a.W = W;
    if (variant == 0) { geometry2<Wg2Cfg<1>>(a); return (long long)a.ci_blocks * a.co_groups * a.splits * Wg2Cfg<1>::RED * 4; }
    if (variant == 1) { geometry2<Wg2Cfg<2>>(a); return (long long)a.ci_blocks * a.co_groups * a.splits * Wg2Cfg<2>::RED * 4; }
    if (variant == 2) { geometry2<Wg2W1>(a); return (long long)a.ci_blocks * a.co_groups * a.splits * Wg2W1::RED * 4; }
    if (variant == 3) { geometry2<Wg2W2>(a); return (long long)a.ci_blocks * a.co_groups * a.splits * Wg2W2::RED * 4; }
    return MIS_ERR_UNSUPPORTED;
}

// dw[Cout][Cin][9] (+)= the weight gradient; workspace: mis_conv2d_wino_wgrad_workspace_bytes
extern "C" int mis_conv2d_wino_wgrad(const float* x, long long x_bs, const float* dy, long long dy_bs, float* dw,
                                     float* workspace, long long workspace_bytes, int N, int Cin, int Cout, int H, int W,
                                     int accumulate, int variant, hipStream_t stream) {
    if (!x || !dy || !dw || !workspace || N <= 0 || Cin <= 0 || Cout <= 0 || H <= 0 || W <= 0) return MIS_ERR_ARG;
    const long long S = (long long)H * W;
    if (x_bs < (long long)Cin * S || dy_bs < (long long)Cout * S) return MIS_ERR_ARG;
    if (mis_conv2d_wino_wgrad_select(N, Cin, Cout, H, W) < 0 || variant < 0 || variant > 3) return MIS_ERR_UNSUPPORTED;
    if (variant >= 2 ? (H % 4 || W % 32) : (H % 8 || W % 16)) return MIS_ERR_UNSUPPORTED;
    if (((uintptr_t)x & 15) || ((uintptr_t)dy & 15) || x_bs % 4 || dy_bs % 4) return MIS_ERR_UNSUPPORTED;
    if (workspace_bytes < mis_conv2d_wino_wgrad_workspace_bytes(N, Cin, Cout, H, W, variant)) return MIS_ERR_WORKSPACE;
    Wg2Args a{};
    a.x = x; a.x_bs = x_bs; a.dy = dy; a.dy_bs = dy_bs; a.ws = workspace;
    a.N = N; a.Cin = Cin; a.Cout = Cout; a.H = H; a.W = W;
    if (variant == 2) return launch_wg2<Wg2W1>(a, dw, accumulate, stream);
    if (variant == 3) return launch_wg2<Wg2W2>(a, dw, accumulate, stream);
    return variant == 0 ? launch_wg2<Wg2Cfg<1>>(a, dw, accumulate, stream) : launch_wg2<Wg2Cfg<2>>(a, dw, accumulate, stream);
}
