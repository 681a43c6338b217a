// Winograd F(2x2, 3x3) convolution forward (and, with the flipped / transposed transformed filter, the data gradient)
// for the stride-1 'same' 3x3 convolutions of the 2-D UNet.
//
// Replaces: nn.Conv2d(k=3, padding=1) of ConvBlock (reference code/networks/unet.py:30-45: conv - BN - LeakyReLU -
// Dropout - conv - BN - LeakyReLU) in the encoder / decoder blocks (unet.py:64-98, 100-160).
//
// Same construction as conv_wino.hip in two dimensions: 16 multiplies per 2x2 outputs instead of 36 (2.25x fewer
// matrix-pipe flops), fp32 end to end.
//   Y = A^T [ sum_ci (G g G^T) (.) (B^T d B) ] A,   B^T d = (d0-d2, d1+d2, d2-d1, d1-d3),  A^T m = (m0+m1+m2, m1-m2-m3)
// A wave owns 16 tiles (2x2 pixels each) x COB blocks of 16 output channels: 16 transform points x COB x 4 accumulator
// registers (64 or 128), so two workgroups (8 waves) share a CU and one wave's transforms / epilogue run under the
// other's MFMAs (VALU work does not overlap the issuing wave's own MFMAs, scripts/ubench/mfma_overlap.hip).  Per
// chunk of 4 input channels: the lane transforms its own 4x4 patch (16 v_pk_add_f32), 16 x COB MFMAs
//   A[i = lane&15][k = lane>>4] = W_xi[co0 + i][ci0 + k]  (LDS, ds_read_b128 = 4 points),
//   B[k = lane>>4][j = lane&15] = U_xi[ci0 + k][tile j]   (registers).
// The haloed box of the workgroup (8 x 8 tiles = 16 x 16 pixels; rows as whole 16-byte groups [x0-4, x0+20)) and the
// filter points of the chunk arrive by LDS-DMA into a ring of NBUF stages.
#include "common.h"
#include "wino.h"
#include <stdio.h>
#include <stdlib.h>

// Prototype of review item 5 (round 4), off in the product build: -DMIS_W2_NORM=1 applies BatchNorm scale / shift +
// LeakyReLU on the LDS -> register read of the patch (the raw convolution output of the previous layer would then be the only
// activation kept).  scripts/w2_norm_proto.sh builds a side library with it and measures the cost; DESIGN.md s.7 has the result.
#ifndef MIS_W2_NORM
#define MIS_W2_NORM 0
#endif

namespace {

using namespace mis_dma;
using namespace mis_wino;

struct W2Args {
    const float* x; long long x_bs;
    const float* wt;      // [co_blocks][Cin_pad/4][4][64 lanes][4]  (pack mode 6 / 7)
    const float* bias;
    float* y; long long y_bs;
    int N, Cin, Cout, H, W;
    int nci4;
    int boxes_y, boxes_x, co_groups;
    unsigned n_blocks, n_blocks_padded;
    float2* stat; long long stat_sc, stat_sn;
    const float2* nrm; float slope;      // MIS_W2_NORM: (scale, shift) per input channel, or null
};

#if MIS_W2_NORM
const float2* g_w2_nrm = nullptr;
float g_w2_slope = 0.f;
#endif

template <int BY_, int BX_, int COB_, int NBUF_>
struct W2Cfg {
    static constexpr int BY = BY_, BX = BX_, COB = COB_, NBUF = NBUF_;
    static constexpr int OY = 2 * BY, OX = 2 * BX, HY = OY + 2;
    static constexpr int NQ = (OX + 8) / 4, RX = NQ * 4;                    // rows hold [x0 - 4, x0 + OX + 4)
    static constexpr int CG = HY * NQ;                                      // 16-byte groups per channel
    static constexpr int NCH = (4 * CG + 63) / 64;                          // input DMA pieces per stage (4 channels, linear)
    static constexpr int IN_FLOATS = NCH * 256;
    static constexpr int W_FLOATS = COB * 1024;
    static constexpr int STAGE = IN_FLOATS + W_FLOATS;
    static constexpr int PT = NCH + COB * 4;                                // pieces per stage
    static constexpr int PW = (PT + 3) / 4;                                 // ... per wave (surplus slots repeat the last one)
    static constexpr int LDS_BYTES = NBUF * STAGE * 4 + 512 + (MIS_W2_NORM ? 4096 : 0);   // + statistics scratch (+ norm table)
    static_assert(BY * BX == 64, "4 waves x 16 tiles");
    static_assert(NBUF >= 3 && (NBUF - 1) * PW <= 63, "ring depth / vmcnt range");
    static_assert(LDS_BYTES <= 80 * 1024, "two workgroups per CU");
};

extern __shared__ __attribute__((aligned(16))) float mis_w2_lds[];

template <class C>
__global__ __launch_bounds__(256, 2) void wino2d_fwd_kernel(const W2Args a) {
    float* const lds = mis_w2_lds;
    const unsigned L = mis_xcd_remap(blockIdx.x, a.n_blocks_padded);
    if (L >= a.n_blocks) return;
    unsigned t = L;
    const int cg = t % a.co_groups; t /= a.co_groups;
    const int bx = t % a.boxes_x;   t /= a.boxes_x;
    const int by = t % a.boxes_y;   t /= a.boxes_y;
    const int n = t;
    const int y0 = by * C::OY, x0 = bx * C::OX;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lk = lane >> 4, lj = lane & 15;
    const int tile = wave * 16 + lj, ty = tile / C::BX, tx = tile % C::BX;
    const long long S = (long long)a.H * a.W;
    const unsigned s_bytes = (unsigned)S * 4u;
    const i32x4 rx = make_rsrc(a.x + (long long)n * a.x_bs, (unsigned)a.Cin * s_bytes);
    const i32x4 rw = make_rsrc(a.wt, (unsigned)a.co_groups * C::COB * (unsigned)a.nci4 * 4096u);
    const unsigned lds0 = lds_addr(lds);
    const int nst = a.nci4;

    // ---- DMA pieces of this wave: piece p = wave + 4 i of the stage (input pieces first, then filter pieces) ----
    unsigned voff[C::PW];
#pragma unroll
    for (int i = 0; i < C::PW; ++i) {
        const int p = wave + 4 * i < C::PT ? wave + 4 * i : C::PT - 1;
        if (p < C::NCH) {
            const int g = p * 64 + lane;                                   // linear (channel, row, group)
            const int ci = g / C::CG, r2 = g - ci * C::CG, hy = r2 / C::NQ, q = r2 - hy * C::NQ;
            const int qy = y0 + hy - 1, qx = x0 - 4 + 4 * q;
            const bool ok = ci < 4 && (unsigned)qy < (unsigned)a.H && (unsigned)qx < (unsigned)a.W;
            voff[i] = ok ? (unsigned)(qy * a.W + qx) * 4u + (unsigned)ci * s_bytes : OOB;
        } else {
            const int j = p - C::NCH, b = j / 4, pp = j % 4;
            voff[i] = (unsigned)(((cg * C::COB + b) * nst) * 1024 + pp * 256 + lane * 4) * 4u;
        }
    }
    auto issue = [&](int s) {       // stage s: channels 4 s .. 4 s + 3 and their filter points
        const unsigned st = lds0 + (unsigned)(s % C::NBUF) * (C::STAGE * 4);
        const unsigned cbase = (unsigned)(s * 4) * s_bytes, wbase = (unsigned)s * 4096u;
#pragma unroll
        for (int i = 0; i < C::PW; ++i) {
            const int p = wave + 4 * i < C::PT ? wave + 4 * i : C::PT - 1;      // uniform
            const bool in = p < C::NCH;
            dma_dwordx4_s(st + (unsigned)(in ? p * 1024 : C::IN_FLOATS * 4 + (p - C::NCH) * 1024), voff[i], in ? cbase : wbase,
                          in ? rx : rw);
        }
    };

    const int poff = lk * (C::CG * 4) + (2 * ty) * C::RX + 3 + 2 * tx;
    f32x4 acc[C::COB][16];
#pragma unroll
    for (int b = 0; b < C::COB; ++b)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[b][i] = f32x4{0.f, 0.f, 0.f, 0.f};

#if MIS_W2_NORM
    float2* const s_nrm = reinterpret_cast<float2*>(lds + C::NBUF * C::STAGE + 128);
    const bool norm = a.nrm != nullptr;                      // uniform
    if (norm) for (int i = tid; i < 4 * nst; i += 256) s_nrm[i] = i < a.Cin ? a.nrm[i] : make_float2(0.f, 0.f);
    // a box on the image border has patch elements in the zero padding: they must stay zero after the activation
    const bool border = y0 == 0 || x0 == 0 || y0 + C::OY >= a.H || x0 + C::OX >= a.W;
    bool rok[4], cok[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        rok[r] = (unsigned)(y0 + 2 * ty - 1 + r) < (unsigned)a.H;
        cok[r] = (unsigned)(x0 + 2 * tx - 1 + r) < (unsigned)a.W;
    }
#endif
    constexpr int A = C::NBUF - 1;
#pragma unroll
    for (int s = 0; s < A; ++s) issue(s);                    // stages past the last chunk deliver zeros (never read)
    for (int s = 0; s < nst; ++s) {
        vmwait<(A - 1) * C::PW>::go();                       // stage s has landed (mine) ...
        __syncthreads();                                     // ... and everyone's; everyone is done with stage s-1
        issue(s + A);                                        // into the buffer of stage s-1
        const float* __restrict__ sb = lds + (s % C::NBUF) * C::STAGE;
        f32x2 u[8];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            u[r * 2] = f32x2{sb[poff + r * C::RX], sb[poff + r * C::RX + 1]};
            u[r * 2 + 1] = f32x2{sb[poff + r * C::RX + 2], sb[poff + r * C::RX + 3]};
        }
#if MIS_W2_NORM
        if (norm) {
            const float2 st = s_nrm[4 * s + lk];
            const f32x2 sc = {st.x, st.x}, sh = {st.y, st.y}, sl = {a.slope, a.slope};
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const f32x2 v = u[i] * sc + sh, w = v * sl;
                u[i] = f32x2{fmaxf(v[0], w[0]), fmaxf(v[1], w[1])};
            }
            if (border) {
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int r = i / 2, c = (i % 2) * 2;
                    u[i] = f32x2{rok[r] && cok[c] ? u[i][0] : 0.f, rok[r] && cok[c + 1] ? u[i][1] : 0.f};
                }
            }
        }
#endif
#pragma unroll
        for (int r = 0; r < 4; ++r) bt4_inner(u[r * 2], u[r * 2 + 1]);
        bt4(u[0], u[2], u[4], u[6]);
        bt4(u[1], u[3], u[5], u[7]);
        const f32x4* __restrict__ wl = reinterpret_cast<const f32x4*>(sb + C::IN_FLOATS) + lane;
#pragma unroll
        for (int b = 0; b < C::COB; ++b)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 a4 = wl[b * 256 + g * 64];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int xi = g * 4 + i;
                    acc[b][xi] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[i], u[xi / 2][xi % 2], acc[b][xi], 0, 0, 0);
                }
            }
    }
    vmwait<0>::go();

    // ---- epilogue: inverse transform (16 -> 2x2 per (channel, tile)), bias, store, optional statistics ----
    const int oy = y0 + 2 * ty, ox = x0 + 2 * tx;
    const bool ok = oy < a.H && ox < a.W;
    float st1[C::COB][4], st2[C::COB][4];
#pragma unroll
    for (int b = 0; b < C::COB; ++b) {
        const int co0 = (cg * C::COB + b) * 16;
        const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(
            (void*)(a.y + (long long)n * a.y_bs + (long long)co0 * S), 0, (int)(16u * s_bytes), 0x00020000);
        const unsigned vo = ok ? (unsigned)(lk * 4) * s_bytes + (unsigned)(oy * a.W + ox) * 4u : OOB;
        f32x4 py[2][4];
#pragma unroll
        for (int x = 0; x < 4; ++x) {
            py[0][x] = acc[b][x] + acc[b][4 + x] + acc[b][8 + x];
            py[1][x] = acc[b][4 + x] - acc[b][8 + x] - acc[b][12 + x];
        }
        f32x4 bv = {0.f, 0.f, 0.f, 0.f};
        if (a.bias) bv = *reinterpret_cast<const f32x4*>(a.bias + co0 + lk * 4);
        f32x4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = s1;
#pragma unroll
        for (int yy = 0; yy < 2; ++yy) {
            const f32x4 v0 = py[yy][0] + py[yy][1] + py[yy][2] + bv, v1 = py[yy][1] - py[yy][2] - py[yy][3] + bv;
            s1 += v0 + v1;
            s2 += v0 * v0 + v1 * v1;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
                const f32x2 v = {v0[r], v1[r]};
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, v), ry, (int)(vo + (unsigned)(yy * a.W) * 4u),
                                                      (int)((unsigned)r * s_bytes), 0);
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) { st1[b][r] = s1[r]; st2[b][r] = s2[r]; }
    }
    if (a.stat) {
        // every box is full here (the host only passes `stat` then): per-channel (sum, sum of squares) of the box
#pragma unroll
        for (int b = 0; b < C::COB; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int o = 1; o < 16; o <<= 1) {
                    st1[b][r] += __shfl_xor(st1[b][r], o, 64);
                    st2[b][r] += __shfl_xor(st2[b][r], o, 64);
                }
        __syncthreads();                                     // the ring is idle now
        float2* red = reinterpret_cast<float2*>(lds);
        if (lj == 0) {
#pragma unroll
            for (int b = 0; b < C::COB; ++b)
#pragma unroll
                for (int r = 0; r < 4; ++r) red[(wave * C::COB + b) * 16 + lk * 4 + r] = make_float2(st1[b][r], st2[b][r]);
        }
        __syncthreads();
        if (tid < C::COB * 16) {
            const int b = tid / 16, c = tid % 16;
            float sx = 0.f, sq = 0.f;
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                const float2 p = red[(w * C::COB + b) * 16 + c];
                sx += p.x; sq += p.y;
            }
            const long long box = (long long)by * a.boxes_x + bx;
            a.stat[(long long)((cg * C::COB + b) * 16 + c) * a.stat_sc + (long long)n * a.stat_sn + box] = make_float2(sx, sq);
        }
    }
}

template <class C>
int launch_w2(W2Args a, hipStream_t stream) {
    a.boxes_y = (int)mis_cdiv(a.H, C::OY);
    a.boxes_x = (int)mis_cdiv(a.W, C::OX);
    a.co_groups = a.Cout / (16 * C::COB);
    const long long nb = (long long)a.N * a.boxes_y * a.boxes_x * a.co_groups;
    if (nb <= 0 || nb > 0x7fffffffLL) return MIS_ERR_ARG;
    a.n_blocks = (unsigned)nb;
    a.n_blocks_padded = (unsigned)(mis_cdiv(nb, MIS_NUM_XCD) * MIS_NUM_XCD);
    static std::atomic<unsigned long long> attr_done{0};
    if (mis_set_lds_attr(reinterpret_cast<const void*>(&wino2d_fwd_kernel<C>), C::LDS_BYTES, attr_done) != MIS_OK)
        return MIS_ERR_LAUNCH;
    hipLaunchKernelGGL(wino2d_fwd_kernel<C>, dim3(a.n_blocks_padded), dim3(256), C::LDS_BYTES, stream, a);
    return mis_launch_status();
}

#ifndef MIS_W2V0_NBUF
#define MIS_W2V0_NBUF 3        // ring of 3: 34 KB of LDS, four workgroups per CU (the register limit).  These launches (16 output
#endif                         // channels: the 256^2 level) are bound by the bytes in flight, not by the pipe: 150 -> 142 us (16 -> 16,
                               // 48 slices), 197 -> 188 us (32 -> 16) against a ring of 4 (scripts/w2_nbuf_ab.sh)
using W2V0 = W2Cfg<8, 8, 1, MIS_W2V0_NBUF>;      // 16 x 16 pixel boxes, one block of 16 output channels
using W2V1 = W2Cfg<8, 8, 2, 3>;      // ... two blocks (Cout a multiple of 32): the transform is shared.  Ring of 3: 47 KB
                                     // of LDS, three workgroups per CU (a ring of 4 is 10 % slower)

// Wide boxes (round 4): 4 x 16 tiles = 8 x 32 pixels instead of 16 x 16.  Same tile count, same LDS, but a box row is 128 bytes
// of output (one store instruction of a wave = 16 lanes x 8 contiguous bytes) and 160 bytes of input instead of 64 / 96:
// scripts/ubench/hbm_pieces.hip -- a copy in 64-byte pieces gets 3.1 TB/s from HBM, in 128-byte pieces 4.4, in streams 5.1 --
// and the 256^2 level of the 2-D UNet, whose launches move 400 MB for 41 us of matrix-pipe work, ran at exactly that 2.85 TB/s.
using W2V2 = W2Cfg<4, 16, 1, MIS_W2V0_NBUF>;
using W2V3 = W2Cfg<4, 16, 2, 3>;

}  // namespace

// Which variant serves this 3x3 'same' convolution (D = 1), or -1 (use the direct kernel, mis_conv_fwd):
//   0: one block of 16 output channels per wave, 1: two.  Needs Cin % 4 == 0, Cin >= 8, Cout % 16 == 0, even H,
//   W % 4 == 0, and whole 16 x 16 boxes (the fused statistics need them, and ragged boxes waste the matrix pipe).
extern "C" int mis_conv2d_wino_select(int N, int Cin, int Cout, int H, int W) {
    if (N <= 0 || Cin < 8 || Cin % 4 || Cout <= 0 || Cout % 16 || H <= 0 || W <= 0) return -1;
    if (((long long)Cin + 32) * H * W * 4 >= (1LL << 30)) return -1;
    static const bool wide = [] { const char* e = getenv("MIS_W2_WIDE"); return !(e && e[0] == '0'); }();
    if (wide && H % 8 == 0 && W % 32 == 0) return Cout % 32 == 0 ? 3 : 2;      // 8 x 32-pixel boxes: 128-byte output rows
    if (H % 16 || W % 16) return -1;
    return Cout % 32 == 0 ? 1 : 0;
}

extern "C" long long mis_conv2d_wino_stat_tiles(int H, int W, int variant) {
    if (H <= 0 || W <= 0 || variant < 0 || variant > 3) return MIS_ERR_ARG;
    return variant >= 2 ? mis_cdiv(H, 8) * mis_cdiv(W, 32) : mis_cdiv(H, 16) * mis_cdiv(W, 16);
}

extern "C" int mis_conv2d_wino_kernel_name(int variant, char* name, int name_len) {
    if (!name || name_len <= 0) return MIS_ERR_ARG;
    if (variant == 0) snprintf(name, name_len, "wino2d_fwd_kernel<W2Cfg<8, 8, 1, %d>>", MIS_W2V0_NBUF);
    else if (variant == 1) snprintf(name, name_len, "wino2d_fwd_kernel<W2Cfg<8, 8, 2, 3>>");
    else if (variant == 2) snprintf(name, name_len, "wino2d_fwd_kernel<W2Cfg<4, 16, 1, %d>>", MIS_W2V0_NBUF);
    else if (variant == 3) snprintf(name, name_len, "wino2d_fwd_kernel<W2Cfg<4, 16, 2, 3>>");
    else return MIS_ERR_UNSUPPORTED;
    return MIS_OK;
}

// y = conv2d(x, w, k = 3, 'same') + bias from the transformed filter (pack mode 6 forward / 7 data gradient)
extern "C" int mis_conv2d_wino_fwd(const float* x, long long x_bs, const float* wt, const float* bias, float* y,
                                   long long y_bs, int N, int Cin, int Cout, int H, int W, float* stat, long long stat_sc,
                                   long long stat_sn, int variant, hipStream_t stream) {
    if (!x || !wt || !y || N <= 0 || Cin <= 0 || Cout <= 0 || H <= 0 || W <= 0) return MIS_ERR_ARG;
    const long long S = (long long)H * W;
    if (x_bs < (long long)Cin * S || y_bs < (long long)Cout * S) return MIS_ERR_ARG;
    if (mis_conv2d_wino_select(N, Cin, Cout, H, W) < 0 || ((variant & 1) && Cout % 32) || variant < 0 || variant > 3)
        return MIS_ERR_UNSUPPORTED;
    if (variant >= 2 ? (H % 8 || W % 32) : (H % 16 || W % 16)) return MIS_ERR_UNSUPPORTED;
    if (y_bs % 2 || ((uintptr_t)y & 7) || ((uintptr_t)wt & 15) || ((uintptr_t)x & 15) || x_bs % 4) return MIS_ERR_UNSUPPORTED;
    W2Args a{};
    a.x = x; a.x_bs = x_bs; a.wt = wt; a.bias = bias; a.y = y; a.y_bs = y_bs;
    a.N = N; a.Cin = Cin; a.Cout = Cout; a.H = H; a.W = W;
    a.nci4 = (Cin + 3) / 4;
    a.stat = reinterpret_cast<float2*>(stat); a.stat_sc = stat_sc; a.stat_sn = stat_sn;
#if MIS_W2_NORM
    a.nrm = g_w2_nrm; a.slope = g_w2_slope;
    if (a.nrm && Cin > 512) return MIS_ERR_UNSUPPORTED;
#endif
    if (variant == 2) return launch_w2<W2V2>(a, stream);
    if (variant == 3) return launch_w2<W2V3>(a, stream);
    return variant == 0 ? launch_w2<W2V0>(a, stream) : launch_w2<W2V1>(a, stream);
}

#if MIS_W2_NORM
// prototype hook: the next mis_conv2d_wino_fwd launches read their input through LeakyReLU(scale * x + shift)
extern "C" int mis_debug_w2_norm(const float* scale_shift, float slope) {
    g_w2_nrm = reinterpret_cast<const float2*>(scale_shift);
    g_w2_slope = slope;
    return MIS_OK;
}
#endif
