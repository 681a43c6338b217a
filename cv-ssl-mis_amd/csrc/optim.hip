// Fused SGD(momentum, weight-decay) + EMA teacher update over flat parameter
// buffers, teacher input noise, per-step schedule state and pseudo-label argmax.
//
// Replaces (reference, per step):
//   optimizer.step()  with optim.SGD(lr, momentum=0.9, weight_decay=1e-4)   train_mean_teacher_2D.py:189-190,232
//   update_ema_variables(model, ema_model, alpha, iter_num)                 train_mean_teacher_2D.py:124-128,233
//       (a Python loop of 2 launches per parameter tensor)
//   lr_ = base_lr * (1 - iter_num / max_iterations) ** 0.9                  train_mean_teacher_2D.py:234-236
//   consistency_weight = consistency * sigmoid_rampup(iter_num // 150, rampup)   :119-121, utils/ramps.py:20-27
//   noise = clamp(randn_like(x) * 0.1, -0.2, 0.2); ema_inputs = x + noise    :208-210
//   torch.argmax(softmax(outputs), dim=1)  (cross-teaching pseudo labels)    train_cross_teaching...py:234-237
//
// All parameters of a model live in ONE flat fp32 buffer (student params,
// grads, momentum, teacher params share the same layout), so the whole update
// is a single HBM-bound stream: 5 reads + 3 writes of 4 B per parameter.
#include "common.h"

namespace {

__global__ __launch_bounds__(256) void sgd_ema_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                      float* __restrict__ m, float* __restrict__ ema,
                                                      long long n, float lr_arg, float momentum, float wd,
                                                      float alpha_arg, float grad_scale,
                                                      const MisStepState* __restrict__ st) {
    const float lr = st ? st->lr : lr_arg;
    const float alpha = st ? st->ema_alpha : alpha_arg;
    const float one_m_alpha = 1.f - alpha;
    const long long units = n >> 2;
    for (long long u = blockIdx.x * 256LL + threadIdx.x; u < units; u += (long long)gridDim.x * 256) {
        float4 pv = reinterpret_cast<float4*>(p)[u];
        const float4 gv = reinterpret_cast<const float4*>(g)[u];
        float4 mv = reinterpret_cast<float4*>(m)[u];
        float* pp = &pv.x; const float* gp = &gv.x; float* mp = &mv.x;
        float4 ev;
        if (ema) ev = reinterpret_cast<float4*>(ema)[u];
        float* ep = &ev.x;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float d = gp[j] * grad_scale + wd * pp[j];
            mp[j] = momentum * mp[j] + d;
            pp[j] = pp[j] - lr * mp[j];
            if (ema) ep[j] = ep[j] * alpha + one_m_alpha * pp[j];
        }
        reinterpret_cast<float4*>(p)[u] = pv;
        reinterpret_cast<float4*>(m)[u] = mv;
        if (ema) reinterpret_cast<float4*>(ema)[u] = ev;
    }
    // tail (n not a multiple of 4)
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const long long i = (units << 2) + threadIdx.x;
        const float d = g[i] * grad_scale + wd * p[i];
        m[i] = momentum * m[i] + d;
        p[i] = p[i] - lr * m[i];
        if (ema) ema[i] = ema[i] * alpha + one_m_alpha * p[i];
    }
}

__global__ __launch_bounds__(256) void noise_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                    long long n, float sigma, float clampv, unsigned salt,
                                                    const MisStepState* __restrict__ st) {
    const unsigned long long seed = st->seed, off = st->offset;
    const long long units = (n + 3) >> 2;
    for (long long u = blockIdx.x * 256LL + threadIdx.x; u < units; u += (long long)gridDim.x * 256) {
        uint32_t r[4];
        mis_philox4((uint32_t)u, (uint32_t)((unsigned long long)u >> 32), salt, (uint32_t)off, (uint32_t)seed,
                    (uint32_t)(seed >> 32) ^ (uint32_t)(off >> 32), r);
        // two Box-Muller pairs -> 4 standard normals
        float z[4];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const float u1 = 1.f - mis_u01(r[2 * h]);  // (0,1]
            const float u2 = mis_u01(r[2 * h + 1]);
            const float rad = sqrtf(-2.f * logf(u1));
            float sn, cs;
            sincosf(6.283185307179586f * u2, &sn, &cs);
            z[2 * h] = rad * cs; z[2 * h + 1] = rad * sn;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const long long i = u * 4 + j;
            if (i < n) {
                float e = z[j] * sigma;
                e = fminf(fmaxf(e, -clampv), clampv);
                y[i] = x[i] + e;
            }
        }
    }
}

struct SchedArgs {
    double base_lr, max_iterations, ema_decay, consistency, rampup;
    long long ramp_div;         // iter_num // ramp_div feeds the ramp (150 in the reference)
    long long cons_start_iter;  // consistency forced to 0.0 below this iter (1000 for 2D/ViT, 0 for 3D)
    int lr_post_increment;      // 0: MT scripts (lr from iter before ++), 1: cross-teaching
};

// fills lr / ema_alpha / cons_weight / cons_gate for step `iter_num`
__device__ void fill_schedule(MisStepState* st, const SchedArgs& a) {
    const long long k = st->iter_num;
    // lr used BY step k was set after step k-1 (lr_0 = base_lr)
    double lr = a.base_lr;
    if (k >= 1) {
        const double it = a.lr_post_increment ? (double)k : (double)(k - 1);
        double f = 1.0 - it / a.max_iterations;
        if (f < 0.0) f = 0.0;
        lr = a.base_lr * pow(f, 0.9);
    }
    st->lr = (float)lr;
    double alpha = 1.0 - 1.0 / ((double)k + 1.0);
    if (alpha > a.ema_decay) alpha = a.ema_decay;
    st->ema_alpha = (float)alpha;
    double w;
    if (a.rampup == 0.0) {
        w = a.consistency;
    } else {
        double cur = (double)(k / a.ramp_div);
        if (cur < 0.0) cur = 0.0;
        if (cur > a.rampup) cur = a.rampup;
        const double ph = 1.0 - cur / a.rampup;
        w = a.consistency * exp(-5.0 * ph * ph);
    }
    st->cons_weight = (float)w;
    st->cons_gate = k >= a.cons_start_iter ? 1.f : 0.f;
}

__global__ void step_init_kernel(MisStepState* st, unsigned long long seed, long long iter_num, SchedArgs a) {
    st->seed = seed; st->offset = (unsigned long long)iter_num; st->iter_num = iter_num;
    fill_schedule(st, a);
}

__global__ void step_advance_kernel(MisStepState* st, SchedArgs a) {
    st->iter_num += 1; st->offset += 1;
    fill_schedule(st, a);
}

template <int C>
__global__ __launch_bounds__(256) void argmax_kernel(const float* __restrict__ x, long long x_bs,
                                                     unsigned char* __restrict__ out, int B, long long S) {
    const long long total = (long long)B * S;
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int b = (int)(i / S);
        const long long s = i - (long long)b * S;
        const float* __restrict__ xb = x + (long long)b * x_bs + s;
        float best = xb[0];
        int bi = 0;
#pragma unroll
        for (int c = 1; c < C; ++c) {
            const float v = xb[(long long)c * S];
            if (v > best) { best = v; bi = c; }  // first maximum wins, like torch.argmax
        }
        out[i] = (unsigned char)bi;
    }
}

unsigned stream_grid(long long units) {
    long long b = mis_cdiv(units, 256);
    if (b > 2048) b = 2048;
    if (b < 1) b = 1;
    return (unsigned)b;
}

}  // namespace

extern "C" int mis_sgd_ema_step(float* param, const float* grad, float* momentum_buf, float* ema_param,
                                long long n, float lr, float momentum, float weight_decay, float ema_alpha,
                                float grad_scale, const MisStepState* state, hipStream_t stream) {
    if (!param || !grad || !momentum_buf || n <= 0) return MIS_ERR_ARG;
    if (((uintptr_t)param | (uintptr_t)grad | (uintptr_t)momentum_buf | (uintptr_t)ema_param) & 15)
        return MIS_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(sgd_ema_kernel, dim3(stream_grid(n >> 2)), dim3(256), 0, stream, param, grad, momentum_buf,
                       ema_param, n, lr, momentum, weight_decay, ema_alpha, grad_scale, state);
    return mis_launch_status();
}

extern "C" int mis_teacher_noise(const float* x, float* y, long long n, float sigma, float clamp_abs,
                                 unsigned salt, const MisStepState* state, hipStream_t stream) {
    if (!x || !y || !state || n <= 0) return MIS_ERR_ARG;
    hipLaunchKernelGGL(noise_kernel, dim3(stream_grid((n + 3) >> 2)), dim3(256), 0, stream, x, y, n, sigma,
                       clamp_abs, salt, state);
    return mis_launch_status();
}

extern "C" int mis_step_init(MisStepState* state, unsigned long long seed, long long iter_num, double base_lr,
                             double max_iterations, double ema_decay, double consistency, double rampup,
                             long long ramp_div, long long cons_start_iter, int lr_post_increment,
                             hipStream_t stream) {
    if (!state || max_iterations <= 0 || ramp_div <= 0) return MIS_ERR_ARG;
    SchedArgs a{base_lr, max_iterations, ema_decay, consistency, rampup, ramp_div, cons_start_iter,
                lr_post_increment};
    hipLaunchKernelGGL(step_init_kernel, dim3(1), dim3(1), 0, stream, state, seed, iter_num, a);
    return mis_launch_status();
}

extern "C" int mis_step_advance(MisStepState* state, double base_lr, double max_iterations, double ema_decay,
                                double consistency, double rampup, long long ramp_div,
                                long long cons_start_iter, int lr_post_increment, hipStream_t stream) {
    if (!state || max_iterations <= 0 || ramp_div <= 0) return MIS_ERR_ARG;
    SchedArgs a{base_lr, max_iterations, ema_decay, consistency, rampup, ramp_div, cons_start_iter,
                lr_post_increment};
    hipLaunchKernelGGL(step_advance_kernel, dim3(1), dim3(1), 0, stream, state, a);
    return mis_launch_status();
}

extern "C" int mis_argmax_channels(const float* x, long long x_bs, unsigned char* out, int B, int C, long long S,
                                   hipStream_t stream) {
    if (!x || !out || B <= 0 || C <= 0 || S <= 0) return MIS_ERR_ARG;
    const unsigned grid = stream_grid((long long)B * S);
    switch (C) {
        case 2: hipLaunchKernelGGL(argmax_kernel<2>, dim3(grid), dim3(256), 0, stream, x, x_bs, out, B, S); break;
        case 3: hipLaunchKernelGGL(argmax_kernel<3>, dim3(grid), dim3(256), 0, stream, x, x_bs, out, B, S); break;
        case 4: hipLaunchKernelGGL(argmax_kernel<4>, dim3(grid), dim3(256), 0, stream, x, x_bs, out, B, S); break;
        default: return MIS_ERR_UNSUPPORTED;
    }
    return mis_launch_status();
}

extern "C" int mis_abi_version(void) { return 1; }

// Test support: fill the LDS of every CU with NaNs.  LDS keeps what the previous kernel left there, so a kernel that reads a
// cell it never wrote (typically under a zero weight: 0 * NaN) is correct or not depending on what ran before it; the GPU
// tests call this in front of every test to make such reads fail deterministically.  Not used by the product path.
namespace {
extern __shared__ __attribute__((aligned(16))) float mis_poison_lds[];
__global__ __launch_bounds__(1024) void poison_lds_kernel(int floats, float* sink) {
    for (int i = threadIdx.x; i < floats; i += 1024) mis_poison_lds[i] = __builtin_nanf("");
    __syncthreads();
    if (sink && mis_poison_lds[threadIdx.x % floats] == 1.f) sink[0] = 1.f;      // keeps the stores
}
}  // namespace

namespace {
// A register-light, LDS-free wave that keeps ONE execution pipe of its SIMD busy: the co-residency probe of
// scripts/interference.py (does a foreign wave on the same SIMD change a kernel's results?).  kind 1: v_mfma_f32_16x16x32_bf16,
// 2: v_mfma_f32_16x16x4_f32, 3: 32-bit integer / unpacked fp32 VALU (v_perm / v_and / v_sub), 4: v_pk_add_f32.
typedef __bf16 dbg_bf16x8 __attribute__((ext_vector_type(8)));
template <int KIND>
__global__ __launch_bounds__(256) void debug_spin_kernel(int iters, float* sink) {
    f32x4 acc[4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    mis_u32x4 ua = {threadIdx.x, 1u, 2u, 3u}, ub = {5u, 6u, threadIdx.x, 8u};
    float a = (float)threadIdx.x, b = 1.5f;
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    f32x2 v0 = {a, b}, v1 = {b, a};
    unsigned w0 = threadIdx.x, w1 = 77u;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            if constexpr (KIND == 1) acc[m] = mis_bf3_mfma1(ua, ub, acc[m]);
            if constexpr (KIND == 2) acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[m], 0, 0, 0);
            if constexpr (KIND == 3) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(w0) : "v"(w1), "v"(0x07060302u));
                    asm volatile("v_and_b32 %0, %0, %1" : "+v"(w1) : "v"(w0));
                    asm volatile("v_sub_f32 %0, %0, %1" : "+v"(a) : "v"(b));
                }
            }
            if constexpr (KIND == 4) {
#pragma unroll
                for (int j = 0; j < 8; ++j) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(v0) : "v"(v1));
            }
        }
    }
    const float s = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3] + a + v0[0] + v0[1] + __uint_as_float(w0 ^ w1);
    if (s == 12345.678f) sink[threadIdx.x] = s;
}
}  // namespace

// diagnostics: `blocks` workgroups of 4 such waves on `stream` (see debug_spin_kernel); sink: any device float buffer >= 256
extern "C" int mis_debug_spin(int kind, int blocks, int iters, float* sink, hipStream_t stream) {
    if (!sink || blocks <= 0 || iters <= 0) return MIS_ERR_ARG;
    switch (kind) {
        case 1: hipLaunchKernelGGL(debug_spin_kernel<1>, dim3(blocks), dim3(256), 0, stream, iters, sink); break;
        case 2: hipLaunchKernelGGL(debug_spin_kernel<2>, dim3(blocks), dim3(256), 0, stream, iters, sink); break;
        case 3: hipLaunchKernelGGL(debug_spin_kernel<3>, dim3(blocks), dim3(256), 0, stream, iters, sink); break;
        case 4: hipLaunchKernelGGL(debug_spin_kernel<4>, dim3(blocks), dim3(256), 0, stream, iters, sink); break;
        default: return MIS_ERR_UNSUPPORTED;
    }
    return mis_launch_status();
}

extern "C" int mis_debug_poison_lds(float* sink, hipStream_t stream) {
    constexpr int BYTES = 160 * 1024;
    static std::atomic<unsigned long long> done{0};
    if (mis_set_lds_attr(reinterpret_cast<const void*>(&poison_lds_kernel), BYTES, done) != MIS_OK) return MIS_ERR_LAUNCH;
    // one workgroup holds a CU's whole LDS: 2048 of them pass over all 256 CUs several times
    hipLaunchKernelGGL(poison_lds_kernel, dim3(2048), dim3(1024), BYTES, stream, BYTES / 4, sink);
    return mis_launch_status();
}
