// Weight re-layout for the MFMA convolution kernels (conv_fwd.hip).
//
// The reference keeps conv weights as [Cout][Cin][kd][kh][kw] (torch layout;
// state_dict keys in SURVEY.md s.8b).  The kernels want the A operand of
// v_mfma_f32_16x16x4_f32 contiguous over 16 output channels, so once per step
// (weights change every SGD update) each layer is repacked to
//   mode 0 (forward):        wp[ci][tap][co]           = w[co][ci][tap]
//   mode 1 (data gradient):  wp[co][tap][ci]           = w[co][ci][TAPS-1-tap]
// zero-padded to (roundup(K-channels,4), TAPS, roundup(M-channels,16)).
// Mode 1 is the spatially flipped, channel-transposed filter, so conv_fwd on dy
// with it is exactly the autograd input-gradient of the reference conv.
//   mode 4 / 5 (3x3x3 only): the Winograd F(2x2x2, 3x3x3) filter transform G g G^T of the forward / data-gradient
//   filter in the operand layout of conv_wino.hip:  wt[M/16][K/4][16][64 lanes][4]  with transform point
//   xi = 4 * (third index) + (last index), lane = (k % 4) * 16 + m % 16  (M, K = output, input channels of the launch).
//   mode 6 / 7 (3x3 only): the same for F(2x2, 3x3), 16 points:  wt[M/16][K/4][4][64 lanes][4]  (conv_wino2d.hip).
#include "common.h"

namespace {

// The filter transform G g G^T (G^T) runs in DOUBLE and is rounded to fp32 once per point (round 4): three nested fp32 passes
// put three roundings on every transformed value, and a filter error is a systematic error of the whole layer -- it was the
// largest term of the Winograd forward's excess over the direct kernel in the float64 gradient gate (tests/test_parity_gpu.py).
// The pass is HBM-bound (27 -> 64 floats per (m, k) pair), the fp64 adds are free.
typedef double wt_t;

// G g of one dimension: point xi of (g0, g1, g2)
__device__ __forceinline__ wt_t wino_g(int xi, wt_t g0, wt_t g1, wt_t g2) {
    return xi == 0 ? g0 : (xi == 3 ? g2 : 0.5 * (xi == 1 ? (g0 + g1 + g2) : (g0 - g1 + g2)));
}

// element i of the transformed filter of one layer (mode 4: forward, mode 5: data gradient)
__device__ __forceinline__ float wino_element(const float* __restrict__ w, int Cout, int Cin, int mode, int Kp, unsigned i) {
    const int e4 = i & 3, lane = (i >> 2) & 63, x4 = (i >> 8) & 15;
    const unsigned blk = i >> 12;                         // (m / 16) * (Kp / 4) + k / 4
    const int k4 = blk % (unsigned)(Kp / 4), mb = blk / (unsigned)(Kp / 4);
    const int xi = x4 * 4 + e4, m = mb * 16 + (lane & 15), k = k4 * 4 + (lane >> 4);
    const int M = mode == 4 ? Cout : Cin, K = mode == 4 ? Cin : Cout;
    if (m >= M || k >= K) return 0.f;
    const float* g = mode == 4 ? w + ((long long)m * Cin + k) * 27 : w + ((long long)k * Cin + m) * 27;
    const int xz = xi >> 4, xy = (xi >> 2) & 3, xx = xi & 3;
    wt_t pz[3];
#pragma unroll
    for (int z = 0; z < 3; ++z) {
        wt_t py[3];
#pragma unroll
        for (int y = 0; y < 3; ++y) {
            const int t = (z * 3 + y) * 3;
            // data gradient: the spatially flipped filter (tap 26 - t)
            py[y] = mode == 4 ? wino_g(xx, g[t], g[t + 1], g[t + 2]) : wino_g(xx, g[26 - t], g[25 - t], g[24 - t]);
        }
        pz[z] = wino_g(xy, py[0], py[1], py[2]);
    }
    return (float)wino_g(xz, pz[0], pz[1], pz[2]);
}

// ... and of the 2-D transform F(2x2, 3x3) (mode 6: forward, mode 7: data gradient): wt[M/16][K/4][4][64 lanes][4]
__device__ __forceinline__ float wino2_element(const float* __restrict__ w, int Cout, int Cin, int mode, int Kp, unsigned i) {
    const int e4 = i & 3, lane = (i >> 2) & 63, x4 = (i >> 8) & 3;
    const unsigned blk = i >> 10;
    const int k4 = blk % (unsigned)(Kp / 4), mb = blk / (unsigned)(Kp / 4);
    const int xi = x4 * 4 + e4, m = mb * 16 + (lane & 15), k = k4 * 4 + (lane >> 4);
    const int M = mode == 6 ? Cout : Cin, K = mode == 6 ? Cin : Cout;
    if (m >= M || k >= K) return 0.f;
    const float* g = mode == 6 ? w + ((long long)m * Cin + k) * 9 : w + ((long long)k * Cin + m) * 9;
    const int xy = xi >> 2, xx = xi & 3;
    wt_t py[3];
#pragma unroll
    for (int y = 0; y < 3; ++y) {
        const int t = y * 3;
        py[y] = mode == 6 ? wino_g(xx, g[t], g[t + 1], g[t + 2]) : wino_g(xx, g[8 - t], g[7 - t], g[6 - t]);
    }
    return (float)wino_g(xy, py[0], py[1], py[2]);
}

__global__ __launch_bounds__(256) void pack_kernel(const float* __restrict__ w, float* __restrict__ wp,
                                                   int Cout, int Cin, int taps, int mode, int Kp, int Mp) {
    const long long total = (long long)Kp * (mode >= 6 ? 16 : mode >= 4 ? 64 : taps) * Mp;
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int m = (int)(i % Mp);
        const int t = (int)((i / Mp) % taps);
        const int k = (int)(i / ((long long)Mp * taps));
        float v = 0.f;
        if (mode >= 6) {
            v = wino2_element(w, Cout, Cin, mode, Kp, (unsigned)i);
        } else if (mode >= 4) {
            v = wino_element(w, Cout, Cin, mode, Kp, (unsigned)i);
        } else if (mode == 0) {
            if (k < Cin && m < Cout) v = w[((long long)m * Cin + k) * taps + t];
        } else if (mode == 1) {
            if (k < Cout && m < Cin) v = w[((long long)k * Cin + m) * taps + (taps - 1 - t)];
        } else if (mode == 2) {   // 1x1 weight stored input-major [Cin][Cout] (ConvTranspose3d view): forward
            if (k < Cin && m < Cout) v = w[(long long)k * Cout + m];
        } else {                  // mode 3: same storage, data-gradient pack
            if (k < Cout && m < Cin) v = w[(long long)m * Cout + k];
        }
        wp[i] = v;
    }
}

// Batched form: one launch repacks every conv layer of a network (a UNet step otherwise spends ~68 launches
// of ~5 us on this).  `descs` is a device array of n jobs ordered by `start` (prefix sum of the jobs' packed
// sizes); each thread finds its job by binary search.
struct PackJob {
    const float* w; float* wp;
    int Cout, Cin, taps, mode, Kp, Mp;
    long long start;
};

constexpr int MAX_JOBS = 256;

// G g of one dimension, all 4 points (the expressions of wino_g: the batched and the single-layer pack agree bit for bit)
__device__ __forceinline__ void wino_g4(wt_t g0, wt_t g1, wt_t g2, wt_t (&o)[4]) {
    o[0] = g0; o[1] = 0.5 * (g0 + g1 + g2); o[2] = 0.5 * (g0 - g1 + g2); o[3] = g2;
}

// One thread = one (m, k) pair of a Winograd job: reads its 27 (9) taps once, applies G along x, y, z and writes the 64
// (16) points as float4s -- consecutive lanes write consecutive 16 bytes, every store instruction of a wave is 1 KB
// contiguous.  (One thread per output element re-read the taps 64 times through 16 scattered lines per wave and load:
// 200 us per launch for V-Net's 83 MB of transformed filters.)
__device__ __forceinline__ void wino_pair3(const PackJob& j, unsigned item) {
    const int lane = item & 63;
    const unsigned blk = item >> 6;
    const int k4 = blk % (unsigned)(j.Kp / 4), mb = blk / (unsigned)(j.Kp / 4);
    const int m = mb * 16 + (lane & 15), k = k4 * 4 + (lane >> 4);
    const bool fwd = j.mode == 4;
    const int M = fwd ? j.Cout : j.Cin, K = fwd ? j.Cin : j.Cout;
    float4* __restrict__ out = reinterpret_cast<float4*>(j.wp) + (long long)blk * 1024 + lane;     // + x4 * 64
    if (m >= M || k >= K) {
#pragma unroll
        for (int x4 = 0; x4 < 16; ++x4) out[x4 * 64] = make_float4(0.f, 0.f, 0.f, 0.f);
        return;
    }
    const float* __restrict__ src = fwd ? j.w + ((long long)m * j.Cin + k) * 27 : j.w + ((long long)k * j.Cin + m) * 27;
    wt_t g[27];
#pragma unroll
    for (int t = 0; t < 27; ++t) g[t] = fwd ? src[t] : src[26 - t];       // data gradient: the spatially flipped filter
    wt_t a[9][4];                       // x pass: rows (z, y)
#pragma unroll
    for (int r = 0; r < 9; ++r) wino_g4(g[3 * r], g[3 * r + 1], g[3 * r + 2], a[r]);
    wt_t b[3][4][4];                    // y pass: [z][xy][xx]
#pragma unroll
    for (int z = 0; z < 3; ++z)
#pragma unroll
        for (int xx = 0; xx < 4; ++xx) {
            wt_t o[4];
            wino_g4(a[z * 3][xx], a[z * 3 + 1][xx], a[z * 3 + 2][xx], o);
#pragma unroll
            for (int xy = 0; xy < 4; ++xy) b[z][xy][xx] = o[xy];
        }
#pragma unroll
    for (int xy = 0; xy < 4; ++xy) {    // z pass, point xi = xz * 16 + xy * 4 + xx: float4 number xz * 4 + xy
        wt_t c[4][4];                   // [xx][xz]
#pragma unroll
        for (int xx = 0; xx < 4; ++xx) wino_g4(b[0][xy][xx], b[1][xy][xx], b[2][xy][xx], c[xx]);
#pragma unroll
        for (int xz = 0; xz < 4; ++xz)
            out[(xz * 4 + xy) * 64] = make_float4((float)c[0][xz], (float)c[1][xz], (float)c[2][xz], (float)c[3][xz]);
    }
}

__device__ __forceinline__ void wino_pair2(const PackJob& j, unsigned item) {
    const int lane = item & 63;
    const unsigned blk = item >> 6;
    const int k4 = blk % (unsigned)(j.Kp / 4), mb = blk / (unsigned)(j.Kp / 4);
    const int m = mb * 16 + (lane & 15), k = k4 * 4 + (lane >> 4);
    const bool fwd = j.mode == 6;
    const int M = fwd ? j.Cout : j.Cin, K = fwd ? j.Cin : j.Cout;
    float4* __restrict__ out = reinterpret_cast<float4*>(j.wp) + (long long)blk * 256 + lane;      // + x4 * 64
    if (m >= M || k >= K) {
#pragma unroll
        for (int x4 = 0; x4 < 4; ++x4) out[x4 * 64] = make_float4(0.f, 0.f, 0.f, 0.f);
        return;
    }
    const float* __restrict__ src = fwd ? j.w + ((long long)m * j.Cin + k) * 9 : j.w + ((long long)k * j.Cin + m) * 9;
    wt_t g[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) g[t] = fwd ? src[t] : src[8 - t];
    wt_t a[3][4];
#pragma unroll
    for (int r = 0; r < 3; ++r) wino_g4(g[3 * r], g[3 * r + 1], g[3 * r + 2], a[r]);
    wt_t c[4][4];                       // [xx][xy]
#pragma unroll
    for (int xx = 0; xx < 4; ++xx) wino_g4(a[0][xx], a[1][xx], a[2][xx], c[xx]);
#pragma unroll
    for (int xy = 0; xy < 4; ++xy) out[xy * 64] = make_float4((float)c[0][xy], (float)c[1][xy], (float)c[2][xy], (float)c[3][xy]);
}

// work items of a job: (m, k) pairs for the Winograd transforms (64 / 16 outputs each), single floats otherwise
__device__ __forceinline__ long long pack_items(const PackJob& j) {
    const long long floats = (long long)j.Kp * (j.mode >= 6 ? 16 : j.mode >= 4 ? 64 : j.taps) * j.Mp;
    return j.mode >= 6 ? floats / 16 : j.mode >= 4 ? floats / 64 : floats;
}

__global__ __launch_bounds__(256) void pack_batch_kernel(const PackJob* __restrict__ jobs, int n, long long total) {
    __shared__ PackJob s_jobs[MAX_JOBS];     // 12 KiB: the whole table (the search and the job fields stay in LDS)
    __shared__ long long s_first[MAX_JOBS + 1];     // prefix sum of the jobs' work items
    for (int i = threadIdx.x; i < n; i += 256) s_jobs[i] = jobs[i];
    __syncthreads();
    if (threadIdx.x == 0) {
        long long acc = 0;
        for (int i = 0; i < n; ++i) { s_first[i] = acc; acc += pack_items(s_jobs[i]); }
        s_first[n] = acc;
    }
    __syncthreads();
    const long long items = s_first[n];
    for (long long g = blockIdx.x * 256LL + threadIdx.x; g < items; g += (long long)gridDim.x * 256) {
        int lo = 0, hi = n - 1;
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (s_first[mid] <= g) lo = mid; else hi = mid - 1;
        }
        const PackJob& j = s_jobs[lo];
        const unsigned i = (unsigned)(g - s_first[lo]);      // one layer's pack is far below 2^32 floats
        if (j.mode >= 6) { wino_pair2(j, i); continue; }
        if (j.mode >= 4) { wino_pair3(j, i); continue; }
        const unsigned mt = (unsigned)j.Mp * (unsigned)j.taps;
        const int k = (int)(i / mt);
        const unsigned r = i - (unsigned)k * mt;
        const int t = (int)(r / (unsigned)j.Mp);
        const int m = (int)(r - (unsigned)t * (unsigned)j.Mp);
        float v = 0.f;
        if (j.mode == 0) {
            if (k < j.Cin && m < j.Cout) v = j.w[((long long)m * j.Cin + k) * j.taps + t];
        } else {
            if (k < j.Cout && m < j.Cin) v = j.w[((long long)k * j.Cin + m) * j.taps + (j.taps - 1 - t)];
        }
        j.wp[i] = v;
    }
}

}  // namespace

// Host helper: fills one job of a host-side table (the caller copies the table to the device once per plan).
// Returns the job's packed size in floats (the increment of `start` for the next job), or a negative error.
extern "C" long long mis_conv_pack_job(void* job_out, const float* w, float* wp, int Cout, int Cin, int taps, int mode,
                                       long long start) {
    if (!job_out || !w || !wp || Cout <= 0 || Cin <= 0 || taps <= 0 || mode < 0 || mode > 7 || mode == 2 || mode == 3)
        return MIS_ERR_ARG;
    if (((mode == 4 || mode == 5) && taps != 27) || (mode >= 6 && taps != 9)) return MIS_ERR_ARG;
    const bool fwd = mode == 0 || mode == 4 || mode == 6;
    const int K = fwd ? Cin : Cout, M = fwd ? Cout : Cin;
    PackJob j{w, wp, Cout, Cin, taps, mode, (K + 3) / 4 * 4, (M + 15) / 16 * 16, start};
    *reinterpret_cast<PackJob*>(job_out) = j;
    return (long long)j.Kp * (mode >= 6 ? 16 : mode >= 4 ? 64 : taps) * j.Mp;
}

extern "C" int mis_conv_pack_job_bytes() { return (int)sizeof(PackJob); }

extern "C" int mis_conv_pack_batch(const void* jobs_device, int n, long long total_floats, hipStream_t stream) {
    if (!jobs_device || n <= 0 || total_floats <= 0) return MIS_ERR_ARG;
    if (n > MAX_JOBS) return MIS_ERR_UNSUPPORTED;
    // a thread writes 64 / 16 floats of a Winograd job, one of a plain one (grid-stride): every block first copies the
    // job table into LDS, so no more blocks than the work needs
    long long blocks = mis_cdiv(total_floats, 256 * 32);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(pack_batch_kernel, dim3((unsigned)blocks), dim3(256), 0, stream,
                       reinterpret_cast<const PackJob*>(jobs_device), n, total_floats);
    return mis_launch_status();
}

extern "C" long long mis_conv_packed_floats(int Cout, int Cin, int taps, int mode) {
    if (Cout <= 0 || Cin <= 0 || taps <= 0 || mode < 0 || mode > 7) return MIS_ERR_ARG;
    if (((mode == 4 || mode == 5) && taps != 27) || (mode >= 6 && taps != 9)) return MIS_ERR_ARG;
    const bool fwd = mode == 0 || mode == 2 || mode == 4 || mode == 6;
    const int K = fwd ? Cin : Cout, M = fwd ? Cout : Cin;
    return (long long)((K + 3) / 4 * 4) * (mode >= 6 ? 16 : mode >= 4 ? 64 : taps) * ((M + 15) / 16 * 16);
}

extern "C" int mis_conv_pack_weights(const float* w, float* wp, int Cout, int Cin, int taps, int mode,
                                     hipStream_t stream) {
    if (!w || !wp || Cout <= 0 || Cin <= 0 || taps <= 0 || mode < 0 || mode > 7) return MIS_ERR_ARG;
    if ((mode == 2 || mode == 3) && taps != 1) return MIS_ERR_ARG;   // input-major storage is only defined for 1x1 weights
    if (((mode == 4 || mode == 5) && taps != 27) || (mode >= 6 && taps != 9)) return MIS_ERR_ARG;   // Winograd: 3x3x3 / 3x3
    const bool fwd = mode == 0 || mode == 2 || mode == 4 || mode == 6;
    const int K = fwd ? Cin : Cout, M = fwd ? Cout : Cin;
    const int Kp = (K + 3) / 4 * 4, Mp = (M + 15) / 16 * 16;
    const long long total = (long long)Kp * (mode >= 6 ? 16 : mode >= 4 ? 64 : taps) * Mp;
    long long blocks = mis_cdiv(total, 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(pack_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, w, wp, Cout, Cin, taps, mode,
                       Kp, Mp);
    return mis_launch_status();
}
