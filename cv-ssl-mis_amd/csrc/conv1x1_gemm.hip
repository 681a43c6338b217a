// 1x1x1 convolution of channel-major (NCDHW) volumes with FEW voxels and MANY channels as a batched GEMM on the matrix pipe.
//
// Replaces, on V-Net's deep levels, the contraction inside nn.Conv3d(C, 2C, 2, stride=2) / nn.ConvTranspose3d(2C, C, 2, stride=2)
// (reference code/networks/vnet.py:73, :100) once the 2x2x2 taps are folded into channels (space-to-depth, pool_upsample.hip):
//     y[n][m][s] = bias[m] + sum_k W[k][m] x[n][k][s]          s over the D*H*W voxels of the COARSE volume
// with K = 8 Cin = 512 ... 1024, M = 128 ... 256 (down) or K = 128 ... 256, M = 8 Cout = 512 ... 1024 (up) and S = 12^3 or 6^3.
// The generic direct kernel (conv_fwd.hip, Cfg<1,1,1,4,8,16,...>) tiles space as 4 x 8 x 16 voxels: 21 % (6^3) to 56 % (12^3) of
// a tile is volume, a launch has 64 ... 192 workgroups each walking all K channels, and it ran the pipe at 0.09 (123 us for
// 1.8 GFLOP).  A 1x1x1 convolution has no spatial structure: per image it is C[M][S] = W^T[K][M]^T . X[K][S], both operands
// contraction-major -- the TN form of gemm.hip.  This file is that kernel with (a) a batch dimension (the images: B and C move
// by a batch stride, A is shared), (b) the bias indexed by ROW, (c) split-K over the channels with a fixed-order reduction,
// chosen so that (image, slice, tile) entries fill the chip.
#include "common.h"

namespace {

constexpr int BK1 = 32;

struct C1Args {
    const float* A; long long lda;              // [K][M]: the weights, contraction-major
    const float* B; long long ldb, b_bs;        // [batch][K][N]
    float* C; long long ldc, c_bs;              // [batch][M][N]
    const float* bias;                          // [M] or null
    float* ws;                                  // split-K partials [batch][KS][M][N]
    int M, N, K, batch, KS, kchunk, accumulate;
    int tiles_m, tiles_n;
    unsigned n_blocks, n_blocks_padded;
};

template <int BT>
__global__ __launch_bounds__(256) void conv1x1_tn_kernel(const C1Args a) {
    constexpr int LD = BT + 16;        // = 16 mod 32: the four k rows of an operand read land on disjoint bank groups
    constexpr int NI = BT / 32;        // 16-wide MFMA tiles per wave and dimension (wave = BT/2 x BT/2)
    constexpr int Q = BT / 4;          // float4 per staged row
    __shared__ __attribute__((aligned(16))) float sA[BK1 * LD];
    __shared__ __attribute__((aligned(16))) float sB[BK1 * LD];

    const unsigned L = mis_xcd_remap(blockIdx.x, a.n_blocks_padded);
    if (L >= a.n_blocks) return;
    const unsigned tiles = (unsigned)(a.tiles_n * a.tiles_m);
    const unsigned per_img = tiles * (unsigned)a.KS;
    const int b = L / per_img;
    const unsigned r0 = L - b * per_img;
    const int kz = r0 / tiles;
    const unsigned tl = r0 - kz * tiles;
    const int tn = tl % a.tiles_n, tm = tl / a.tiles_n;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lk = lane >> 4, lj = lane & 15;
    const int wm = (wave >> 1) * (BT / 2), wn = (wave & 1) * (BT / 2);
    const int m0 = tm * BT, n0 = tn * BT;
    const int kbeg = kz * a.kchunk;
    const int kend = kbeg + a.kchunk < a.K ? kbeg + a.kchunk : a.K;
    const float* __restrict__ Bp = a.B + (long long)b * a.b_bs;

    f32x4 acc[NI][NI];
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // rows = k (contraction), BT floats = Q float4 per row; zero fill outside the matrices.  The tiles of k-step s + 1 are
    // loaded into registers before the MFMAs of k-step s and stored to LDS after them
    constexpr int NLD = (BK1 * Q + 255) / 256;
    float4 ra[NLD], rb[NLD];
    auto fetch = [&](int k0) {
#pragma unroll
        for (int it = 0; it < NLD; ++it) {
            const int e = tid + it * 256;
            const int r = e / Q, q = e - r * Q;
            const int k = k0 + r;
            const bool in = e < BK1 * Q;
            {
                const bool ok = in && k < kend && m0 + q * 4 < a.M;
                const long long off = ok ? (long long)k * a.lda + m0 + q * 4 : 0;
                ra[it] = *reinterpret_cast<const float4*>(a.A + off);
                if (!ok) ra[it] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
            {
                const bool ok = in && k < kend && n0 + q * 4 < a.N;
                const long long off = ok ? (long long)k * a.ldb + n0 + q * 4 : 0;
                rb[it] = *reinterpret_cast<const float4*>(Bp + off);
                if (!ok) rb[it] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
    };
    fetch(kbeg);
    for (int k0 = kbeg; k0 < kend; k0 += BK1) {
        __syncthreads();
#pragma unroll
        for (int it = 0; it < NLD; ++it) {
            const int e = tid + it * 256;
            const int r = e / Q, q = e - r * Q;
            if (e < BK1 * Q) {
                *reinterpret_cast<float4*>(sA + r * LD + q * 4) = ra[it];
                *reinterpret_cast<float4*>(sB + r * LD + q * 4) = rb[it];
            }
        }
        if (k0 + BK1 < kend) fetch(k0 + BK1);
        __syncthreads();
#pragma unroll
        for (int s = 0; s < BK1 / 4; ++s) {
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

    // D row = lk * 4 + r -> m, column = lj -> n (16 lanes = 64 contiguous bytes of a row)
    const bool direct = a.KS == 1;
    float* __restrict__ out = direct ? a.C + (long long)b * a.c_bs : a.ws + ((long long)b * a.KS + kz) * a.M * a.N;
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
                        if (a.bias) v += a.bias[m];
                        if (a.accumulate) v += *p;
                    }
                    *p = v;
                }
            }
        }
}

// C[b][m][n] (+)= bias[m] + sum_kz ws[b][kz][m][n], slices in ascending order (deterministic); one float4 per thread
__global__ __launch_bounds__(256) void conv1x1_reduce_kernel(const C1Args a) {
    const long long q = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long per = (long long)a.M * a.N / 4;
    if (q >= per * a.batch) return;
    const int b = (int)(q / per);
    const long long e = (q - (long long)b * per) * 4;
    const int m = (int)(e / a.N), n = (int)(e - (long long)m * a.N);
    const float* __restrict__ w = a.ws + (long long)b * a.KS * a.M * a.N + e;
    float4 s = *reinterpret_cast<const float4*>(w);
    for (int k = 1; k < a.KS; ++k) {
        const float4 v = *reinterpret_cast<const float4*>(w + (long long)k * a.M * a.N);
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    if (a.bias) { const float bv = a.bias[m]; s.x += bv; s.y += bv; s.z += bv; s.w += bv; }
    float* p = a.C + (long long)b * a.c_bs + (long long)m * a.ldc + n;
    if (a.accumulate) { const float4 o = *reinterpret_cast<const float4*>(p); s.x += o.x; s.y += o.y; s.z += o.z; s.w += o.w; }
    *reinterpret_cast<float4*>(p) = s;
}

// ------------------------------------------------------------------------------------------------ weight gradient
// dw[m][n] (+)= sum_b sum_s A[b][m][s] B[b][n][s]: the weight gradient of the same layers (A = dy or x, B = the other one), an NT
// product whose contraction runs over images x voxels.  Both operands are contraction-MINOR here (a row = one channel's voxels),
// so the tiles are transposed on their way into LDS: thread (row r fastest, k quad q) loads A[b][m0 + r][s0 + 4 q ..] as a float4
// and stores its four floats to sA[4 q + j][r] -- lanes of a wave hold consecutive rows, i.e. consecutive banks (the loads
// themselves touch one 16-byte piece per row: these tensors live in L2 / the infinity cache).  Slices = (image, voxel chunk).
struct W1Args {
    const float* A; long long a_bs;            // [batch][M][S]
    const float* B; long long b_bs;            // [batch][N][S]
    float* C; long long ldc;                   // [M][N]
    float* ws;                                 // partials [slice][M][N]
    int M, N, S, batch, SJ, schunk, accumulate;
    int tiles_m, tiles_n;
    unsigned n_blocks, n_blocks_padded;
};

template <int BT>
__global__ __launch_bounds__(256) void conv1x1_wgrad_kernel(const W1Args a) {
    constexpr int LD = BT + 16;
    constexpr int NI = BT / 32;
    constexpr int QK = BK1 / 4;        // float4 per row and k-step
    __shared__ __attribute__((aligned(16))) float sA[BK1 * LD];
    __shared__ __attribute__((aligned(16))) float sB[BK1 * LD];

    const unsigned L = mis_xcd_remap(blockIdx.x, a.n_blocks_padded);
    if (L >= a.n_blocks) return;
    const unsigned tiles = (unsigned)(a.tiles_n * a.tiles_m);
    const int slice = L / tiles;
    const unsigned tl = L - slice * tiles;
    const int tn = tl % a.tiles_n, tm = tl / a.tiles_n;
    const int b = slice / a.SJ, sj = slice - b * a.SJ;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lk = lane >> 4, lj = lane & 15;
    const int wm = (wave >> 1) * (BT / 2), wn = (wave & 1) * (BT / 2);
    const int m0 = tm * BT, n0 = tn * BT;
    const int sbeg = sj * a.schunk;
    const int send = sbeg + a.schunk < a.S ? sbeg + a.schunk : a.S;
    const float* __restrict__ Ap = a.A + (long long)b * a.a_bs;
    const float* __restrict__ Bp = a.B + (long long)b * a.b_bs;

    f32x4 acc[NI][NI];
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    constexpr int NLD = BT * QK / 256;
    float4 ra[NLD], rb[NLD];
    auto fetch = [&](int s0) {
#pragma unroll
        for (int it = 0; it < NLD; ++it) {
            const int e = tid + it * 256;
            const int r = e % BT, q = e / BT;
            const int sp = s0 + 4 * q;
            {
                const bool ok = sp < send && m0 + r < a.M;
                ra[it] = *reinterpret_cast<const float4*>(Ap + (ok ? (long long)(m0 + r) * a.S + sp : 0));
                if (!ok) ra[it] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
            {
                const bool ok = sp < send && n0 + r < a.N;
                rb[it] = *reinterpret_cast<const float4*>(Bp + (ok ? (long long)(n0 + r) * a.S + sp : 0));
                if (!ok) rb[it] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
    };
    fetch(sbeg);
    for (int s0 = sbeg; s0 < send; s0 += BK1) {
        __syncthreads();
#pragma unroll
        for (int it = 0; it < NLD; ++it) {
            const int e = tid + it * 256;
            const int r = e % BT, q = e / BT;
            sA[(4 * q + 0) * LD + r] = ra[it].x; sA[(4 * q + 1) * LD + r] = ra[it].y;
            sA[(4 * q + 2) * LD + r] = ra[it].z; sA[(4 * q + 3) * LD + r] = ra[it].w;
            sB[(4 * q + 0) * LD + r] = rb[it].x; sB[(4 * q + 1) * LD + r] = rb[it].y;
            sB[(4 * q + 2) * LD + r] = rb[it].z; sB[(4 * q + 3) * LD + r] = rb[it].w;
        }
        if (s0 + BK1 < send) fetch(s0 + BK1);
        __syncthreads();
#pragma unroll
        for (int s = 0; s < BK1 / 4; ++s) {
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
    float* __restrict__ out = a.ws + (long long)slice * a.M * a.N;
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const int n = n0 + wn + j * 16 + lj;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + wm + i * 16 + lk * 4 + r;
                if (m < a.M && n < a.N) out[(long long)m * a.N + n] = acc[i][j][r];
            }
        }
}

// C[m][n] (+)= sum over the slices, ascending (deterministic)
__global__ __launch_bounds__(256) void conv1x1_wgrad_reduce_kernel(const W1Args a) {
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long total = (long long)a.M * a.N;
    if (e >= total) return;
    const int slices = a.batch * a.SJ;
    float s0 = 0.f, s1 = 0.f;
    int k = 0;
    for (; k + 1 < slices; k += 2) { s0 += a.ws[(long long)k * total + e]; s1 += a.ws[(long long)(k + 1) * total + e]; }
    if (k < slices) s0 += a.ws[(long long)k * total + e];
    const int m = (int)(e / a.N), n = (int)(e - (long long)m * a.N);
    float* p = a.C + (long long)m * a.ldc + n;
    const float s = s0 + s1;
    *p = a.accumulate ? *p + s : s;
}

void wplan(W1Args& a) {
    a.tiles_m = (int)mis_cdiv(a.M, 128);
    a.tiles_n = (int)mis_cdiv(a.N, 128);
    const long long tiles = (long long)a.tiles_m * a.tiles_n;
    // (image, voxel chunk) slices: ~ 1.5 resident workgroups per CU, chunks of >= 4 k-steps
    long long sj = 384 / (tiles * a.batch);
    const long long sjmax = mis_cdiv(a.S, 4 * BK1);
    if (sj > sjmax) sj = sjmax;
    if (sj < 1) sj = 1;
    a.schunk = (int)(mis_cdiv(mis_cdiv(a.S, sj), BK1) * BK1);
    a.SJ = (int)mis_cdiv(a.S, a.schunk);
    const long long nb = tiles * a.batch * a.SJ;
    a.n_blocks = (unsigned)nb;
    a.n_blocks_padded = (unsigned)(mis_cdiv(nb, MIS_NUM_XCD) * MIS_NUM_XCD);
}

constexpr int BT1 = 128;

// slices over the channels: (image, slice, tile) entries ~ 1.5 resident workgroups per CU, >= 4 k-steps per slice
void plan(C1Args& a) {
    a.tiles_m = (int)mis_cdiv(a.M, BT1);
    a.tiles_n = (int)mis_cdiv(a.N, BT1);
    const long long tiles = (long long)a.tiles_m * a.tiles_n * a.batch;
    long long ks = 384 / tiles;
    const long long kmax = a.K / (4 * BK1);
    if (ks > kmax) ks = kmax;
    if (ks < 1) ks = 1;
    a.kchunk = (int)(mis_cdiv(mis_cdiv(a.K, ks), BK1) * BK1);
    a.KS = (int)mis_cdiv(a.K, a.kchunk);
    const long long nb = tiles * a.KS;
    a.n_blocks = (unsigned)nb;
    a.n_blocks_padded = (unsigned)(mis_cdiv(nb, MIS_NUM_XCD) * MIS_NUM_XCD);
}

bool a16(const void* p) { return ((uintptr_t)p & 15) == 0; }

}  // namespace

// Workspace (bytes) of mis_conv1x1_gemm for these sizes (0: the launch is not split).
extern "C" long long mis_conv1x1_gemm_workspace_bytes(int N, int Cin, int Cout, long long S) {
    if (N <= 0 || Cin <= 0 || Cout <= 0 || S <= 0 || S > 0x7fffffffLL) return MIS_ERR_ARG;
    C1Args a{};
    a.M = Cout; a.N = (int)S; a.K = Cin; a.batch = N;
    plan(a);
    return a.KS > 1 ? (long long)N * a.KS * Cout * S * 4 : 0;
}

// y[n][co][s] (+)= bias[co] + sum_ci wt[ci][co] x[n][ci][s]: a 1x1x1 convolution of N channel-major volumes of S voxels with
// the weights given CONTRACTION-major ([Cin][Cout], row stride ldw).  x / y: channel stride S, batch strides x_bs / y_bs
// (views into wider buffers are fine).  S, Cout, the strides multiples of 4; 16-byte aligned pointers.  Deterministic.
extern "C" int mis_conv1x1_gemm(const float* x, long long x_bs, const float* wt, long long ldw, const float* bias, float* y,
                                long long y_bs, int N, int Cin, int Cout, long long S, int accumulate, float* workspace,
                                long long workspace_bytes, hipStream_t stream) {
    if (!x || !wt || !y || N <= 0 || Cin <= 0 || Cout <= 0 || S <= 0) return MIS_ERR_ARG;
    if (x_bs < (long long)Cin * S || y_bs < (long long)Cout * S || ldw < Cout) return MIS_ERR_ARG;
    if (S % 4 || Cout % 4 || ldw % 4 || x_bs % 4 || y_bs % 4 || !a16(x) || !a16(wt) || !a16(y) || S > 0x7fffffffLL)
        return MIS_ERR_UNSUPPORTED;
    C1Args a{};
    a.A = wt; a.lda = ldw; a.B = x; a.ldb = S; a.b_bs = x_bs; a.C = y; a.ldc = S; a.c_bs = y_bs; a.bias = bias;
    a.M = Cout; a.N = (int)S; a.K = Cin; a.batch = N; a.accumulate = accumulate;
    plan(a);
    if ((long long)a.tiles_m * a.tiles_n * a.batch * a.KS > 0x7fffffffLL) return MIS_ERR_UNSUPPORTED;
    if (a.KS > 1) {
        if (!workspace || !a16(workspace) || workspace_bytes < (long long)N * a.KS * Cout * S * 4) return MIS_ERR_WORKSPACE;
        a.ws = workspace;
    }
    hipLaunchKernelGGL(conv1x1_tn_kernel<BT1>, dim3(a.n_blocks_padded), dim3(256), 0, stream, a);
    if (a.KS > 1) {
        const long long q = (long long)N * Cout * S / 4;
        hipLaunchKernelGGL(conv1x1_reduce_kernel, dim3((unsigned)mis_cdiv(q, 256)), dim3(256), 0, stream, a);
    }
    return mis_launch_status();
}

// Workspace (bytes) of mis_conv1x1_wgrad: the per-slice partials.
extern "C" long long mis_conv1x1_wgrad_workspace_bytes(int N, int M, int Nc, long long S) {
    if (N <= 0 || M <= 0 || Nc <= 0 || S <= 0 || S > 0x7fffffffLL) return MIS_ERR_ARG;
    W1Args a{};
    a.M = M; a.N = Nc; a.S = (int)S; a.batch = N;
    wplan(a);
    return (long long)N * a.SJ * M * Nc * 4;
}

// dw[m][n] (+)= sum over the N images and the S voxels of a[img][m][s] * b[img][n][s]: the weight gradient of the 1x1x1
// convolution above (a = dy, b = x gives [Cout][Cin]; swapped operands give the transposed parameter layout directly).
// a / b: channel-major views (channel stride S, batch strides a_bs / b_bs); dw: row stride ldw.  S and the batch strides
// multiples of 4, 16-byte aligned a / b.  Deterministic (fixed slices, fixed-order reduction).
extern "C" int mis_conv1x1_wgrad(const float* av, long long a_bs, const float* bv, long long b_bs, float* dw, long long ldw, int N,
                                 int M, int Nc, long long S, int accumulate, float* workspace, long long workspace_bytes,
                                 hipStream_t stream) {
    if (!av || !bv || !dw || !workspace || N <= 0 || M <= 0 || Nc <= 0 || S <= 0) return MIS_ERR_ARG;
    if (a_bs < (long long)M * S || b_bs < (long long)Nc * S || ldw < Nc) return MIS_ERR_ARG;
    if (S % 4 || a_bs % 4 || b_bs % 4 || !a16(av) || !a16(bv) || S > 0x7fffffffLL) return MIS_ERR_UNSUPPORTED;
    W1Args a{};
    a.A = av; a.a_bs = a_bs; a.B = bv; a.b_bs = b_bs; a.C = dw; a.ldc = ldw; a.ws = workspace;
    a.M = M; a.N = Nc; a.S = (int)S; a.batch = N; a.accumulate = accumulate;
    wplan(a);
    if (workspace_bytes < (long long)N * a.SJ * M * Nc * 4) return MIS_ERR_WORKSPACE;
    hipLaunchKernelGGL(conv1x1_wgrad_kernel<128>, dim3(a.n_blocks_padded), dim3(256), 0, stream, a);
    hipLaunchKernelGGL(conv1x1_wgrad_reduce_kernel, dim3((unsigned)mis_cdiv((long long)M * Nc, 256)), dim3(256), 0, stream, a);
    return mis_launch_status();
}
