/*
 * mis_hip.h -- C ABI of libmis_hip.so: hand-written gfx950 (MI355X / CDNA4) kernels for the
 * Mean-Teacher semi-supervised segmentation training step of ziyangwang007/CV-SSL-MIS.
 *
 * The reference has no plugin/FFI layer: every device instruction it runs comes from stock
 * torch.nn modules.  This header is therefore the interface a maintainer binds *underneath*
 * the reference's Python operator surface (net_factory / net_factory_3d / losses.DiceLoss /
 * update_ema_variables / optim.SGD); each entry point cites the reference code it replaces.
 * INTEGRATION.md shows the ctypes stub.
 *
 * Conventions
 *  - every pointer is a raw DEVICE pointer into caller-owned memory (e.g. torch tensors); the
 *    library never allocates, never retains pointers, keeps no global/thread-local state and is
 *    callable from any host thread;
 *  - activations are fp32 NCDHW (2-D images: D == 1); channel and spatial dims are dense, the
 *    batch stride (`*_bs`, in elements) is explicit, so channel slices of a concatenated skip
 *    buffer are valid operands (torch.cat of the reference never materialises);
 *  - every call is asynchronous on `stream` (hipStream_t passed as void*);
 *  - return value: 0, or MIS_ERR_* (< 0); `*_workspace_bytes` return a byte count or MIS_ERR_*;
 *  - results are run-to-run deterministic (fixed reduction trees, no floating-point atomics).
 */
#ifndef MIS_HIP_H
#define MIS_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

#define MIS_OK 0
#define MIS_ERR_ARG (-1)
#define MIS_ERR_UNSUPPORTED (-2)
#define MIS_ERR_LAUNCH (-3)
#define MIS_ERR_WORKSPACE (-4)

typedef void* mis_stream_t; /* hipStream_t */

/* Device-resident per-step state (40 bytes): Philox seed/offset, iter_num and the schedule values
 * of the current step.  Lets one captured hipGraph replay with fresh dropout masks, learning rate,
 * EMA alpha and consistency weight.  Layout: u64 seed, u64 offset, i64 iter_num, f32 lr,
 * f32 ema_alpha, f32 cons_weight, f32 cons_gate. */
typedef struct MisStepState MisStepState;

int mis_abi_version(void);

/* ---- convolution: nn.Conv2d / nn.Conv3d, kernel 1 or 3, stride 1, zero 'same' padding ----------
 * reference: code/networks/unet.py:37,41 (3x3), :73 (1x1), :138 (out_conv);
 *            code/networks/utils.py:104,107 (3x3x3); code/networks/unet_3D.py:59 (1x1x1).
 * Contraction runs on v_mfma_f32_16x16x4_f32 (fp32 in / fp32 accumulate == an fmaf chain). */
int mis_conv_cin_pad(int cin);   /* K-channel padding of packed weights (multiple of 4)  */
int mis_conv_cout_pad(int cout); /* M-channel padding of packed weights (multiple of 16) */
/* floats of the packed buffer; mode 0 = forward, 1 = data-gradient, 4 / 5 = the Winograd-transformed forward /
 * data-gradient filter of a 3x3x3 conv (taps = 27; consumed by mis_conv3d_wino_fwd), 6 / 7 = of a 3x3 conv (taps = 9;
 * mis_conv2d_wino_fwd) */
long long mis_conv_packed_floats(int Cout, int Cin, int taps, int mode);
/* w: [Cout][Cin][taps] (torch layout) -> wp.  mode 0: wp[ci][tap][co]; mode 1: wp[co][tap][ci] with
 * the taps reversed (the flipped, transposed filter of the autograd input-gradient). */
int mis_conv_pack_weights(const float* w, float* wp, int Cout, int Cin, int taps, int mode, mis_stream_t stream);
/* Batched re-layout: every conv layer of a network (modes 0 / 1) in ONE launch.  The caller builds a table of jobs
 * with mis_conv_pack_job (host side; `start` = running sum of the returned sizes), copies it to the device once per
 * network, and calls mis_conv_pack_batch once per step (weights change with every SGD update). */
int mis_conv_pack_job_bytes(void);
long long mis_conv_pack_job(void* job_out, const float* w, float* wp, int Cout, int Cin, int taps, int mode,
                            long long start);
int mis_conv_pack_batch(const void* jobs_device, int n, long long total_floats, mis_stream_t stream);
/* y[n][co] = bias[co] + sum_{ci,tap} x[n][ci][p + tap - pad] * w.  bias may be NULL.
 * With a mode-1 pack, `x` = dL/dy, Cin/Cout swapped, bias NULL: y = dL/dx.
 * Tiles are fetched by LDS-DMA through 32-bit buffer descriptors: one image must satisfy
 * (Cin + 32) * D*H*W * 4 < 2^30 bytes and wp must be 16-byte aligned, else MIS_ERR_UNSUPPORTED. */
int mis_conv_fwd(const float* x, long long x_bs, const float* wp, const float* bias, float* y, long long y_bs,
                 int N, int Cin, int Cout, int D, int H, int W, int kd, int kh, int kw, mis_stream_t stream);
/* Convolution + the statistics of the normalisation that follows it (nn.Conv -> nn.BatchNorm / nn.InstanceNorm of
 * the reference's ConvBlock / UnetConv3, unet.py:37-38, networks/utils.py:104-105) in one kernel: per (channel, image,
 * tile) the epilogue also writes (sum, sum of squares) of the output it just produced:
 *   stat[(co * stat_sc + n * stat_sn + tile) * 2 + {0,1}],  tile < T = mis_conv_fwd_stat_tiles(...)
 * (T = 0: geometry not eligible, use mis_conv_fwd + mis_norm_stats).  mis_norm_stats_finalize turns the partials
 * into (mean, rstd) and updates the running statistics exactly like mis_norm_stats, without reading the activation:
 * BatchNorm (per_sample = 0): stat_sc = N*T, stat_sn = T; InstanceNorm (per_sample = 1): stat_sc = T, stat_sn = Cout*T. */
long long mis_conv_fwd_stat_tiles(int N, int Cin, int Cout, int D, int H, int W, int kd, int kh, int kw);
int mis_conv_fwd_stats(const float* x, long long x_bs, const float* wp, const float* bias, float* y, long long y_bs,
                       int N, int Cin, int Cout, int D, int H, int W, int kd, int kh, int kw, float* stat,
                       long long stat_sc, long long stat_sn, mis_stream_t stream);
int mis_norm_stats_finalize(const float* part, int N, int C, long long S, int tiles, int per_sample, float eps,
                            float* mean, float* rstd, float* running_mean, float* running_var,
                            long long* num_batches_tracked, float momentum, mis_stream_t stream);
/* name of the kernel instantiation mis_conv_fwd launches for this geometry (profiling attribution) */
int mis_conv_fwd_kernel_name(int N, int Cin, int Cout, int D, int H, int W, int kd, int kh, int kw, char* name,
                             int name_len);
/* ---- Winograd F(2x2x2, 3x3x3) form of the stride-1 'same' 3x3x3 convolution (csrc/conv_wino.hip) -----------
 * reference: nn.Conv3d(k=3, pad=1) of UnetConv3 / UnetUp3_CT (code/networks/utils.py:99-123), unet_3D.py:28-57,
 * vnet.py:15-22.  Same contract as mis_conv_fwd / mis_conv_fwd_stats (x, y NCDHW fp32, optional bias, optional per-box
 * (sum, sumsq) partials for the normalisation that follows), 3.375x fewer matrix-pipe flops; fp32 end to end, the result
 * differs from the direct form by rounding only.  The filter comes transformed (mis_conv_pack_weights / pack jobs,
 * mode 4 = forward, 5 = data gradient: the same launch on dy with Cin and Cout swapped).
 * mis_conv3d_wino_select: variant serving this geometry, or -1 (use mis_conv_fwd): 0 = boxes of 4x4x32 outputs (W % 32),
 *   1 = 4x8x16 (W % 16), 2 = 8x8x8 (also partly filled boxes -- 6^3 -- when >= 40 % of the box volume is output
 *   and the launch has >= 128 boxes; tiles outside the volume are neither stored nor counted in the statistics),
 *   3 = 6x6x12 (W == 12, D % 6 == 0, H % 6 == 0: the 12^3 level; 54 tiles per box taken 16 at a time).
 * mis_conv3d_wino_stat_tiles: partial-statistics entries per image (boxes) of that variant. */
int mis_conv3d_wino_select(int N, int Cin, int Cout, int D, int H, int W);
long long mis_conv3d_wino_stat_tiles(int D, int H, int W, int variant);
int mis_conv3d_wino_kernel_name(int variant, char* name, int name_len);
int mis_conv3d_wino_fwd(const float* x, long long x_bs, const float* wt, const float* bias, float* y, long long y_bs,
                        int N, int Cin, int Cout, int D, int H, int W, float* stat, long long stat_sc,
                        long long stat_sn, int variant, mis_stream_t stream);
/* The same convolution with the contraction over the input channels cut into slices when the launch has too few boxes to
 * fill the chip (the 6^3 level, half batches at 12^3; variants 2 / 3): (box, slice) entries instead of boxes, partial
 * outputs in `workspace` (mis_conv3d_wino_fwd_workspace_bytes; 0 = this launch is not split and needs none), summed in a
 * fixed order by a second launch that also adds the bias and leaves the statistics partials.  _splits: the slice count. */
int mis_conv3d_wino_fwd_splits(int N, int Cin, int Cout, int D, int H, int W, int variant);
long long mis_conv3d_wino_fwd_workspace_bytes(int N, int Cin, int Cout, int D, int H, int W, int variant);
int mis_conv3d_wino_fwd_ws(const float* x, long long x_bs, const float* wt, const float* bias, float* y, long long y_bs,
                           int N, int Cin, int Cout, int D, int H, int W, float* stat, long long stat_sc, long long stat_sn,
                           int variant, float* workspace, long long workspace_bytes, mis_stream_t stream);
/* The data gradient (dy -> da, filter of pack mode 5) of a conv whose input a = act(InstanceNorm(xn)) -- no affine, no
 * dropout, read by nothing else -- with the first stage of that normalisation's backward in its epilogue: per (n, channel)
 * and run of boxes part = (sum dz, sum dz * xn), dz = da * (xn > mean ? 1 : slope), in the layout of the forward's
 * statistics partials (mis_conv3d_wino_stat_tiles entries per image).  Replaces the partial-sum pass of autograd's
 * nn.InstanceNorm3d + nn.ReLU backward in UnetConv3 (code/networks/utils.py:105-109); mis_norm_act_bwd_tiles finishes
 * it.  Variants 0 / 1 only, N * Cout <= 384 (Cout = channels of da). */
int mis_conv3d_wino_dgrad_norm(const float* dy, long long dy_bs, const float* wt, float* da, long long da_bs, int N,
                               int Cin, int Cout, int D, int H, int W, const float* xn, long long xn_bs,
                               const float* mean, float slope, float* part, long long part_sc, long long part_sn,
                               int variant, mis_stream_t stream);
/* ---- Winograd F(2x2, 3x3) form of the stride-1 'same' 3x3 convolution of the 2-D UNet (csrc/conv_wino2d.hip) ----
 * reference: nn.Conv2d(k=3, padding=1) of ConvBlock (code/networks/unet.py:30-45).  Same contract as the 3-D entry
 * points above with D = 1; filter transformed by pack modes 6 (forward) / 7 (data gradient).
 * _select: 0 / 1 = boxes of 16 x 16 pixels with one / two blocks of 16 output channels per wave (H, W multiples of 16),
 * 2 / 3 = the same with boxes of 8 x 32 pixels (H % 8 == 0, W % 32 == 0: 128-byte output rows; preferred), -1 = use mis_conv_fwd.
 * _stat_tiles: statistics partials per image (= boxes) of that variant. */
int mis_conv2d_wino_select(int N, int Cin, int Cout, int H, int W);
long long mis_conv2d_wino_stat_tiles(int H, int W, int variant);
int mis_conv2d_wino_kernel_name(int variant, char* name, int name_len);
int mis_conv2d_wino_fwd(const float* x, long long x_bs, const float* wt, const float* bias, float* y, long long y_bs,
                        int N, int Cin, int Cout, int H, int W, float* stat, long long stat_sc, long long stat_sn,
                        int variant, mis_stream_t stream);
/* ... and its weight gradient (csrc/conv_wino2d_wgrad.hip): dw[Cout][Cin][9], deterministic; _select: 0 / 1 = stages of
 * 8 x 16 pixels (H % 8, W % 16), 2 / 3 = stages of 4 x 32 pixels (H % 4, W % 32; preferred), odd = two blocks of 16 output
 * channels per workgroup, -1 = use mis_conv_wgrad */
int mis_conv2d_wino_wgrad_select(int N, int Cin, int Cout, int H, int W);
/* the kernel a variant launches, as a profiler names it (bench.py's flop attribution keys on it) */
int mis_conv2d_wino_wgrad_kernel_name(int variant, char* name, int name_len);
long long mis_conv2d_wino_wgrad_workspace_bytes(int N, int Cin, int Cout, int H, int W, int variant);
int mis_conv2d_wino_wgrad(const float* x, long long x_bs, const float* dy, long long dy_bs, float* dw, float* workspace,
                          long long workspace_bytes, int N, int Cin, int Cout, int H, int W, int accumulate, int variant,
                          mis_stream_t stream);
/* Winograd F(2x2x2, 3x3x3) weight gradient (csrc/conv_wino_wgrad.hip): the same result as mis_conv_wgrad for a 3x3x3
 * 'same' convolution up to fp32 rounding, deterministic (fixed summation order).  _select: variant or -1 (use
 * mis_conv_wgrad); needs 16-byte aligned x / dy, batch strides and W multiples of 4 floats. */
int mis_conv3d_wino_wgrad_select(int N, int Cin, int Cout, int D, int H, int W);
/* the kernel a variant launches, as a profiler names it */
int mis_conv3d_wino_wgrad_kernel_name(int variant, char* name, int name_len);
long long mis_conv3d_wino_wgrad_workspace_bytes(int N, int Cin, int Cout, int D, int H, int W, int variant);
int mis_conv3d_wino_wgrad(const float* x, long long x_bs, const float* dy, long long dy_bs, float* dw, float* workspace,
                          long long workspace_bytes, int N, int Cin, int Cout, int D, int H, int W, int accumulate,
                          int variant, mis_stream_t stream);
long long mis_conv_wgrad_workspace_bytes(int N, int Cin, int Cout, int D, int H, int W, int kd, int kh, int kw);
/* the kernel mis_conv_wgrad launches for this geometry, as a profiler names it */
int mis_conv_wgrad_kernel_name(int N, int Cin, int Cout, int D, int H, int W, int kd, int kh, int kw, char* name,
                               int name_len);
/* dw[Cout][Cin][taps] (+)= sum_{n,p} dy[n][co][p] * x[n][ci][p + tap - pad]  (autograd weight gradient) */
int mis_conv_wgrad(const float* x, long long x_bs, const float* dy, long long dy_bs, float* dw, float* workspace,
                   long long workspace_bytes, int N, int Cin, int Cout, int D, int H, int W, int kd, int kh,
                   int kw, int accumulate, mis_stream_t stream);
/* out[c] (+)= sum_{n,s} x[n][c][s]: bias gradient of convs NOT followed by a normalisation */
int mis_channel_sum(const float* x, long long x_bs, int N, int C, long long S, float* out, int accumulate,
                    void* workspace, long long workspace_bytes, mis_stream_t stream);

/* ---- normalisation + activation + dropout -------------------------------------------------------
 * reference 2-D: nn.BatchNorm2d -> nn.LeakyReLU(0.01) -> nn.Dropout(p)      code/networks/unet.py:38-43
 *           3-D: nn.InstanceNorm3d -> nn.ReLU [-> nn.Dropout(0.3)]  code/networks/utils.py:105-109,
 *                                                                   code/networks/unet_3D.py:61-62,85,90
 * per_sample = 0: statistics per channel over (N,S) (BatchNorm, biased variance; running stats updated
 * with `momentum` and the unbiased variance when running_* != NULL); per_sample = 1: per (n,c) over S
 * (InstanceNorm).  mean/rstd: G floats, G = C or N*C.  slope: 0.01 LeakyReLU, 0 ReLU.
 * Dropout mask = Philox(seed, offset, drop_salt, logical element index) from `state`, or an explicit
 * scale mask (0 or 1/(1-p), contiguous [N][C][S]) for parity tests; never stored. */
long long mis_norm_workspace_bytes(int N, int C, long long S, int per_sample);
int mis_norm_stats(const float* x, long long x_bs, int N, int C, long long S, int per_sample, float eps,
                   float* mean, float* rstd, float* running_mean, float* running_var,
                   long long* num_batches_tracked, float momentum, void* workspace, long long workspace_bytes,
                   mis_stream_t stream);
int mis_norm_stats_from_running(const float* running_mean, const float* running_var, float eps, float* mean,
                                float* rstd, int C, mis_stream_t stream); /* eval-mode BatchNorm */
int mis_norm_act_fwd(const float* x, long long x_bs, float* y, long long y_bs, int N, int C, long long S,
                     int per_sample, const float* mean, const float* rstd, const float* gamma, const float* beta,
                     float slope, float drop_p, unsigned drop_salt, const MisStepState* state,
                     const float* drop_mask, mis_stream_t stream);
/* dx = d(loss)/dx given da = d(loss)/d(output); dgamma/dbeta (BatchNorm affine) may be NULL */
int mis_norm_act_bwd(const float* x, long long x_bs, const float* da, long long da_bs, float* dx, long long dx_bs,
                     int N, int C, long long S, int per_sample, const float* mean, const float* rstd,
                     const float* gamma, const float* beta, float slope, float drop_p, unsigned drop_salt,
                     const MisStepState* state, const float* drop_mask, float* dgamma, float* dbeta,
                     int accumulate_affine, void* workspace, long long workspace_bytes, mis_stream_t stream);
/* Group-wise form: with per_sample != 0, `cg` consecutive channels share one (mean, rstd) (mean / rstd hold
 * N*C/cg floats): cg = 1 is InstanceNorm, cg = C/16 is nn.GroupNorm(16, C) -- the conv+GN+ReLU block of
 * reference code/networks/vnet.py:19-20,48,76,103 -- with the per-channel affine in gamma / beta.  Statistics of a
 * GroupNorm: mis_norm_stats on the same memory viewed as [N, C/cg, cg*S] with per_sample = 1, or
 * mis_norm_stats_finalize(part, N, C/cg, cg*S, cg*tiles, 1, ...) on the conv epilogue's partials.
 * no_norm != 0 (backward): activation + dropout only (normalization='none' blocks, vnet.py:22,84,110); mean must
 * hold zeros and rstd ones, gamma / beta NULL.  Workspace: mis_norm_workspace_bytes(N, C, S, per_sample). */
int mis_norm_act_fwd_g(const float* x, long long x_bs, float* y, long long y_bs, int N, int C, long long S,
                       int per_sample, int cg, const float* mean, const float* rstd, const float* gamma,
                       const float* beta, float slope, float drop_p, unsigned drop_salt, const MisStepState* state,
                       const float* drop_mask, mis_stream_t stream);
int mis_norm_act_bwd_g(const float* x, long long x_bs, const float* da, long long da_bs, float* dx, long long dx_bs,
                       int N, int C, long long S, int per_sample, int cg, int no_norm, const float* mean,
                       const float* rstd, const float* gamma, const float* beta, float slope, float drop_p,
                       unsigned drop_salt, const MisStepState* state, const float* drop_mask, float* dgamma,
                       float* dbeta, int accumulate_affine, void* workspace, long long workspace_bytes,
                       mis_stream_t stream);

/* First layer of the 3-D networks, Conv3d(1 -> 16, k = 3, pad = 1) -> BatchNorm / InstanceNorm -> (Leaky)ReLU
 * (reference code/networks/unet_3D.py:28 conv1, networks/utils.py:99-107; vnet.py:123 block_one): its input needs no
 * gradient, so the gradient at the conv output is only read by the weight gradient.  mis_norm_act_bwd_sums runs the
 * reduction half of mis_norm_act_bwd (sums: 2 floats per group; dgamma / dbeta), mis_conv_wgrad_cin1_norm forms the
 * gradient on its load path from da (gradient at the activation), y (conv output) and those sums: the apply pass and the
 * re-read of its result are gone.  No dropout; InstanceNorm only without affine. */
int mis_norm_act_bwd_sums(const float* x, long long x_bs, const float* da, long long da_bs, int N, int C, long long S,
                          int per_sample, const float* mean, const float* rstd, const float* gamma, const float* beta,
                          float slope, float* sums, float* dgamma, float* dbeta, int accumulate_affine, void* workspace,
                          long long workspace_bytes, mis_stream_t stream);
/* ... from the partials of mis_conv3d_wino_dgrad_norm: sums[n * C + c] = (mean dz, mean dz * xhat), and dx (may be NULL:
 * the consumer forms it from the sums) = the gradient at the normalisation's input.  InstanceNorm without affine. */
int mis_norm_act_bwd_tiles(const float* x, long long x_bs, const float* da, long long da_bs, float* dx, long long dx_bs,
                           int N, int C, long long S, const float* mean, const float* rstd, float slope,
                           const float* part, int tiles, float* sums, mis_stream_t stream);
/* y = act(norm(x) + res): the tail of MONAI's UnetResBlock as UNETR / SwinUNETR use it (conv - IN - lrelu - conv - IN,
 * + shortcut, lrelu; reference code/networks/unetr.py UnetrBasicBlock(res_block=True), net_factory_3d.py:37-38) in one
 * pass over x and res instead of normalise / add / activate.  Backward: dz = dy * act'(.), dres (+)= dz, dx = the
 * normalisation's backward of dz (res is re-read on the load path of both passes).  BatchNorm (optional affine) or
 * InstanceNorm without affine; no dropout; S % 4 == 0.  Workspace: mis_norm_workspace_bytes.
 * add_after_act != 0: y = act(norm(x)) + res instead -- V-Net's x_up + skip (code/networks/vnet.py:210-222); then
 * dres (+)= dy and the activation's derivative does not involve res. */
int mis_norm_res_act_fwd(const float* x, long long x_bs, const float* res, long long res_bs, float* y, long long y_bs,
                         int N, int C, long long S, int per_sample, const float* mean, const float* rstd,
                         const float* gamma, const float* beta, float slope, int add_after_act, mis_stream_t stream);
int mis_norm_res_act_bwd(const float* x, long long x_bs, const float* res, long long res_bs, const float* dy,
                         long long dy_bs, float* dx, long long dx_bs, float* dres, long long dres_bs, int accumulate_dres,
                         int N, int C, long long S, int per_sample, const float* mean, const float* rstd,
                         const float* gamma, const float* beta, float slope, float* dgamma, float* dbeta,
                         int accumulate_affine, int add_after_act, void* workspace, long long workspace_bytes,
                         mis_stream_t stream);
int mis_conv_wgrad_cin1_norm_eligible(int N, int Cout, int D, int H, int W);
int mis_conv_wgrad_cin1_norm(const float* x, long long x_bs, const float* da, long long da_bs, const float* y,
                             long long y_bs, int N, int D, int H, int W, int per_sample, const float* mean,
                             const float* rstd, const float* gamma, const float* beta, const float* sums, float slope,
                             float* dw, float* workspace, long long workspace_bytes, int accumulate,
                             mis_stream_t stream);

/* Last block of the 3-D networks fused with the 1x1x1 classifier (reference code/networks/unet_3D.py: up_concat1 ->
 * dropout2 -> final = nn.Conv3d(16, n_classes, 1); vnet.py:180-181; unetr.py out): logits[N][K][S] =
 * W[K][C] . drop(act(norm(x))) + b, i.e. mis_norm_act_fwd followed by a 1x1x1 mis_conv_fwd, without storing the
 * activation; the backward recomputes it from x and yields dx, dgamma / dbeta (BatchNorm) and the classifier's
 * dw[K][C], db[K] (NULL: none) from dlogits in two passes.  mis_norm_head_eligible: C == 16, 2 <= K <= 4; per_sample != 0
 * (InstanceNorm) only without affine.  Workspace: mis_norm_head_workspace_bytes. */
int mis_norm_head_eligible(int C, int K);
long long mis_norm_head_workspace_bytes(int N, int C, long long S, int per_sample, int K);
int mis_norm_head_fwd(const float* x, long long x_bs, int N, int C, long long S, int per_sample, const float* mean,
                      const float* rstd, const float* gamma, const float* beta, float slope, float drop_p,
                      unsigned drop_salt, const MisStepState* state, const float* drop_mask, const float* w,
                      const float* b, int K, float* logits, long long l_bs, mis_stream_t stream);
int mis_norm_head_bwd(const float* x, long long x_bs, const float* dlogits, long long dl_bs, float* dx, long long dx_bs,
                      int N, int C, long long S, int per_sample, const float* mean, const float* rstd,
                      const float* gamma, const float* beta, float slope, float drop_p, unsigned drop_salt,
                      const MisStepState* state, const float* drop_mask, const float* w, int K, float* dgamma,
                      float* dbeta, int accumulate_affine, float* dw, float* db, int accumulate_w, void* workspace,
                      long long workspace_bytes, mis_stream_t stream);

/* mis_norm_act_fwd_g + mis_maxpool2_fwd in one pass (reference unet_3D.py:35-47 conv_k -> maxpool_k; unet.py:56): writes
 * the activation y AND its 2x max-pool `pooled` [N][C][D/2 (D > 1)][H/2][W/2] with the argmax codes idx [N*C][So] (the
 * same values mis_maxpool2_fwd produces).  W % 8 == 0, H even, D even or 1. */
int mis_norm_act_fwd_pool(const float* x, long long x_bs, float* y, long long y_bs, float* pooled, long long p_bs,
                          unsigned char* idx, int N, int C, int D, int H, int W, int per_sample, int cg,
                          const float* mean, const float* rstd, const float* gamma, const float* beta, float slope,
                          float drop_p, unsigned drop_salt, const MisStepState* state, const float* drop_mask,
                          mis_stream_t stream);
/* mis_norm_act_bwd_g for an activation [N][C][D][H][W] that also feeds a 2x max-pool (reference unet_3D.py:35-47 conv_k ->
 * maxpool_k with the skip connection to the decoder; unet.py:56 DownBlock): the incoming gradient is da (the other
 * consumers; NULL: none) plus the backward of the pool -- dpool [N][C][D/2 (D > 1)][H/2][W/2] scattered by
 * mis_maxpool2_fwd's argmax codes idx -- added on the load path of both passes; mis_maxpool2_bwd's read-modify-write of
 * the full-resolution gradient is not needed.  W % 4 == 0, H even, D even or 1. */
int mis_norm_act_bwd_pool(const float* x, long long x_bs, const float* da, long long da_bs, const float* dpool,
                          long long dp_bs, const unsigned char* idx, float* dx, long long dx_bs, int N, int C, int D,
                          int H, int W, int per_sample, int cg, const float* mean, const float* rstd,
                          const float* gamma, const float* beta, float slope, float drop_p, unsigned drop_salt,
                          const MisStepState* state, const float* drop_mask, float* dgamma, float* dbeta,
                          int accumulate_affine, void* workspace, long long workspace_bytes, mis_stream_t stream);

/* ---- 2x max-pool / 2x linear up-sampling ----------------------------------------------------------
 * reference: nn.MaxPool2d(2) unet.py:56; nn.MaxPool3d(2) unet_3D.py:35-47;
 *            nn.Upsample(bilinear, align_corners=True) unet.py:74-75;
 *            nn.Upsample(trilinear, align_corners=False) networks/utils.py:264.
 * D == 1 selects the 2-D form.  idx: one byte per output element (argmax inside the window). */
int mis_maxpool2_fwd(const float* x, long long x_bs, float* y, long long y_bs, unsigned char* idx, int N, int C,
                     int D, int H, int W, mis_stream_t stream);
int mis_maxpool2_bwd(const float* dy, long long dy_bs, const unsigned char* idx, float* dx, long long dx_bs,
                     int N, int C, int D, int H, int W, int accumulate, mis_stream_t stream);
int mis_upsample2_fwd(const float* x, long long x_bs, float* y, long long y_bs, int N, int C, int D, int H,
                      int W, int align_corners, mis_stream_t stream);
int mis_upsample2_bwd(const float* dy, long long dy_bs, float* dx, long long dx_bs, int N, int C, int D, int H,
                      int W, int align_corners, int accumulate, mis_stream_t stream);

/* ---- fused loss tail ------------------------------------------------------------------------------
 * reference: code/train_mean_teacher_2D.py:213-229 / _3D.py:142-158, DiceLoss code/utils/losses.py:165-201.
 * student [B][C][S] logits (first L samples labeled), teacher [B-L][C][S] logits, label [L][S] u8 or i64.
 * out (device, >= 5 + C floats): loss, loss_ce, loss_dice, consistency_loss, consistency_weight,
 * class-wise dice...;  dlogits (may be NULL) = d(loss * loss_scale)/d(student logits).
 * Consistency weight/gate come from `state` when non-NULL, else from `cons_weight` (gate on). */
long long mis_loss_tail_workspace_bytes(int B, int C, long long S);
int mis_loss_tail(const float* student, long long s_bs, const float* teacher, long long t_bs, const void* label,
                  int label_bytes, int B, int L, int C, long long S, float cons_weight,
                  const MisStepState* state, float loss_scale, float* out, float* dlogits, long long d_bs,
                  void* workspace, long long workspace_bytes, mis_stream_t stream);

/* Cross-teaching loss tail (code/train_cross_teaching_between_cnn_transformer_2D.py:221-245):
 * loss_m = 0.5*(CE + Dice)(own[:L], label) + w * Dice(softmax(own[L:]), argmax(other[L:]));
 * out (>= 5 floats): loss_m, loss_ce, loss_dice, pseudo_supervision, consistency_weight. */
long long mis_cross_teaching_tail_workspace_bytes(int B, int C, long long S);
/* same, with the pseudo-supervision term selectable: pseudo_ce = 1 -> CE(own[L:], argmax(other[L:])) as in
 * cross pseudo supervision (code/train_cross_pseudo_supervision_3D.py:168-175, _2D.py:187-194); 0 -> Dice. */
int mis_cross_pseudo_tail(const float* own, long long s_bs, const float* other, long long o_bs, const void* label,
                          int label_bytes, int B, int L, int C, long long S, float cons_weight,
                          const MisStepState* state, int pseudo_ce, float* out, float* dlogits, long long d_bs,
                          void* workspace, long long workspace_bytes, mis_stream_t stream);
/* same plus a Mean-Teacher consistency term against an EMA teacher's logits [B-L][C][S] on the unlabeled half
 * (code/train_cnn_meet_vit_2D.py:300-337): + mt_weight * mean((softmax(own[L:]) - softmax(teacher))^2); the
 * script's factor 7 on the pseudo-supervision and its `iter_num < 1000` gate are folded into cons_weight /
 * mt_weight by the caller.  teacher == NULL: identical to mis_cross_pseudo_tail.  out (>= 7 floats): the five above,
 * consistency (MSE) loss, mt_weight.  MIS_ERR_UNSUPPORTED for teacher != NULL with pseudo_ce. */
int mis_cross_pseudo_mt_tail(const float* own, long long s_bs, const float* other, long long o_bs,
                             const float* teacher, long long t_bs, const void* label, int label_bytes, int B, int L,
                             int C, long long S, float cons_weight, float mt_weight, const MisStepState* state,
                             int pseudo_ce, float* out, float* dlogits, long long d_bs, void* workspace,
                             long long workspace_bytes, mis_stream_t stream);
int mis_cross_teaching_tail(const float* own, long long s_bs, const float* other, long long o_bs, const void* label,
                            int label_bytes, int B, int L, int C, long long S, float cons_weight,
                            const MisStepState* state, float* out, float* dlogits, long long d_bs, void* workspace,
                            long long workspace_bytes, mis_stream_t stream);

/* UA-MT (code/train_uncertainty_aware_mean_teacher_3D.py:148-179, _2D.py:161-191).
 * mis_softmax_mean_accumulate: acc[u] = (first ? 0 : acc[u]) + scale * sum_{r<R} softmax(logits[r*U+u]) -- folds one
 *   MC-dropout teacher pass on the batch repeat(unlabeled, R) into the running mean prediction (:153-163).
 * mis_uamt_tail: 0.5*(CE+Dice)(student[:L], label) + w * sum(mask*(softmax(student[L:]) - softmax(teacher))^2)
 *   / (2*sum(mask) + 1e-16), mask = -sum_c pm*log(pm+1e-6) < (0.75 + 0.25*sigmoid_rampup(iter, max_iterations))*ln2
 *   (iter from `state` when given, else `iter_num`).  out (>= 7+C floats): loss, loss_ce, loss_dice,
 *   consistency_loss, consistency_weight, C class-wise dice, #unmasked voxels, threshold. */
int mis_softmax_mean_accumulate(const float* logits, long long l_bs, float* acc, long long a_bs, int U, int R, int C,
                                long long S, float scale, int first, mis_stream_t stream);
long long mis_uamt_tail_workspace_bytes(int B, int C, long long S);
int mis_uamt_tail(const float* student, long long s_bs, const float* teacher, long long t_bs, const float* mean_probs,
                  long long mp_bs, const void* label, int label_bytes, int B, int L, int C, long long S,
                  float cons_weight, const MisStepState* state, long long iter_num, double max_iterations,
                  float loss_scale, float* out, float* dlogits, long long d_bs, void* workspace,
                  long long workspace_bytes, mis_stream_t stream);

/* ---- stand-alone loss operators (drop-in utils.losses surface) ----------------------------------------
 * reference: losses.DiceLoss code/utils/losses.py:165-201; losses.softmax_mse_loss :74-91.
 * mis_dice_loss_fwd: probs [B][C][S], label [B][S]; out[0] = loss, out[1+c] = class-wise dice; the
 * workspace keeps the coefficients mis_dice_loss_bwd needs (dprobs = grad_out[0] * dLoss/dprobs).
 * mis_softmax_mse: backward == 0: out = (softmax(input) - softmax(target))^2 (un-reduced);
 *                  backward == 1: out = d/d(input_logits) given the elementwise upstream grad_out. */
long long mis_dice_workspace_bytes(int B, int C, long long S);
int mis_dice_loss_fwd(const float* probs, long long p_bs, const void* label, int label_bytes, int B, int C,
                      long long S, const float* weight, float* out, void* workspace, long long workspace_bytes,
                      mis_stream_t stream);
int mis_dice_loss_bwd(const float* probs, long long p_bs, const void* label, int label_bytes, int B, int C,
                      long long S, const void* workspace, const float* grad_out, float* dprobs, long long d_bs,
                      mis_stream_t stream);
int mis_softmax_mse(const float* input_logits, long long a_bs, const float* target_logits, long long b_bs,
                    const float* grad_out, long long g_bs, float* out, long long o_bs, int B, int C, long long S,
                    int backward, mis_stream_t stream);
/* ema = alpha*ema + (1-alpha)*param over a flat buffer (update_ema_variables, train_mean_teacher_2D.py:124-128) */
int mis_ema_update(float* ema_param, const float* param, long long n, float alpha, mis_stream_t stream);

/* ---- optimizer, EMA, schedules, noise, pseudo-labels ---------------------------------------------
 * reference: optim.SGD(momentum 0.9, wd 1e-4) train_mean_teacher_2D.py:189-190,232;
 *            update_ema_variables :124-128,233; poly LR :234-236; consistency ramp :119-121 + utils/ramps.py:20-27;
 *            teacher noise :208-210; argmax pseudo labels train_cross_teaching...py:234-237.
 * All n parameters of a model live in one flat fp32 buffer (same layout for param/grad/momentum/EMA).
 * lr / ema_alpha come from `state` when non-NULL.  grad_scale folds the 1/world of data-parallel averaging. */
int mis_sgd_ema_step(float* param, const float* grad, float* momentum_buf, float* ema_param, long long n,
                     float lr, float momentum, float weight_decay, float ema_alpha, float grad_scale,
                     const MisStepState* state, mis_stream_t stream);
int mis_teacher_noise(const float* x, float* y, long long n, float sigma, float clamp_abs, unsigned salt,
                      const MisStepState* state, mis_stream_t stream);
int mis_step_init(MisStepState* state, unsigned long long seed, long long iter_num, double base_lr,
                  double max_iterations, double ema_decay, double consistency, double rampup, long long ramp_div,
                  long long cons_start_iter, int lr_post_increment, mis_stream_t stream);
int mis_step_advance(MisStepState* state, double base_lr, double max_iterations, double ema_decay,
                     double consistency, double rampup, long long ramp_div, long long cons_start_iter,
                     int lr_post_increment, mis_stream_t stream);
int mis_argmax_channels(const float* x, long long x_bs, unsigned char* out, int B, int C, long long S,
                        mis_stream_t stream);

/* Conv3d(k = 2, stride = 2) and ConvTranspose3d(k = 2, stride = 2) of V-Net (reference code/networks/vnet.py:73, :100)
 * read / written in place on the fine volume -- no space_to_depth re-layout -- for the channel pairs
 * mis_conv_k2s2_eligible names (V-Net's two largest levels); forward and data gradient:
 *   mis_conv_k2s2_down: y[N][Cout][Do][Ho][Wo] (+)= bias + sum w[co][ci*8 + tap] x[ci][2v + tap]      x fine, w [Cout][Cin*8]
 *                       (= dX of ConvTranspose3d with w = its parameter [Cin_t][Cout_t*8], Cout = Cin_t, Cin = Cout_t)
 *   mis_conv_k2s2_up:   y[N][Cout][2Do][2Ho][2Wo] (+)= bias + sum w[ci][co*8 + tap] x[ci][v]          x coarse, w [Cin][Cout*8]
 *                       (= dX of Conv3d(k2s2) with w = its parameter [Cout_c][Cin_c*8], Cin = Cout_c, Cout = Cin_c)
 * Do x Ho x Wo is always the COARSE geometry; tap = dz*4 + dy*2 + dx; bias may be NULL; accumulate != 0 adds to y. */
int mis_conv_k2s2_eligible(int Cin, int Cout, int Do, int Ho, int Wo, int up);
int mis_conv_k2s2_down(const float* x, long long x_bs, const float* w, const float* bias, float* y, long long y_bs, int N,
                       int Cin, int Cout, int Do, int Ho, int Wo, int accumulate, mis_stream_t stream);
int mis_conv_k2s2_up(const float* x, long long x_bs, const float* w, const float* bias, float* y, long long y_bs, int N,
                     int Cin, int Cout, int Do, int Ho, int Wo, int accumulate, mis_stream_t stream);
/* ... and their weight gradient from the tensors as they lie (no space_to_depth view, no transpose of the result):
 *   dw[cc][cf*8 + tap] (+)= sum_{n, v} coarse[n][cc][v] * fine[n][cf][2v + tap]
 * Conv3d(k2s2): coarse = dy (CC = Cout), fine = x (CF = Cin); ConvTranspose3d(k2s2): coarse = x (CC = Cin), fine = dy
 * (CF = Cout) -- dw is the parameter's own layout in both cases.  Deterministic (one partial per workgroup, summed in
 * order).  workspace >= mis_conv_k2s2_wgrad_workspace_bytes(CF, CC). */
int mis_conv_k2s2_wgrad_eligible(int CF, int CC, int Do, int Ho, int Wo);
long long mis_conv_k2s2_wgrad_workspace_bytes(int CF, int CC);
int mis_conv_k2s2_wgrad(const float* coarse, long long c_bs, const float* fine, long long f_bs, float* dw, int N, int CF,
                        int CC, int Do, int Ho, int Wo, int accumulate, float* workspace, long long workspace_bytes,
                        mis_stream_t stream);
/* ---- V-Net data movement (code/networks/vnet.py) -------------------------------------------------------
 * A kernel-2/stride-2 Conv3d (:73) is space_to_depth + the 1x1x1 MFMA conv with the weight viewed as
 * [Cout][8*Cin]; ConvTranspose3d k2 s2 (:100) is the 1x1x1 conv to 8*Cout channels (weight stored input-major,
 * pack modes 2/3 of mis_conv_pack_weights) + depth_to_space (+ bias).  N,C,D,H,W describe the FINE tensor;
 * coarse channel = c*8 + kz*4 + ky*2 + kx.  mis_add: out = a (+ b) -- the additive skips (:210-222). */
int mis_space_to_depth2(const float* src, long long src_bs, float* dst, long long dst_bs, const float* bias, int N,
                        int C, int D, int H, int W, int to_depth, int accumulate, mis_stream_t stream);
/* 2-D twin: fine [N][C][H][W] <-> coarse [N][4C][H/2][W/2] (coarse channel = c*4 + ky*2 + kx) around a 1x1 convolution =
 * nn.ConvTranspose2d(C1, C2, kernel_size=2, stride=2) of UpBlock(bilinear=False) (reference code/networks/unet.py:76-78) */
int mis_space_to_depth2d(const float* src, long long src_bs, float* dst, long long dst_bs, const float* bias, int N, int C,
                         int H, int W, int to_depth, int accumulate, mis_stream_t stream);
int mis_add(const float* a, long long a_bs, const float* b, long long b_bs, float* out, long long o_bs, int N, int C,
            long long S, mis_stream_t stream);

/* ---- token-major operators of SwinUnet (rows x C with an explicit row stride `ld`) ----------------
 * reference: code/networks/swin_transformer_unet_skip_expand_decoder_sys.py (line numbers below).
 * mis_gemm: fp32 MFMA GEMM.  trans = 0: C[M,N] (+)= A[M,K] . B[N,K]^T (+ bias[N])  -- nn.Linear forward
 *           (:14,16,107,109,320,361,390,690) and, with B = weight^T, its input gradient;
 *           trans = 1: C[M,N] (+)= A[K,M]^T . B[K,N]  -- weight gradient dY^T . X (split over K, deterministic). */
long long mis_gemm_workspace_bytes(int M, int N, int K, int trans);

/* Weight AND bias gradient of nn.Linear in one pass over dy (replaces autograd of nn.Linear at
 * code/networks/swin_transformer_unet_skip_expand_decoder_sys.py:14,16,107,109; code/networks/unetr.py's ViT Linears):
 *   dW[M,N] (+)= dy[K,M]^T . x[K,N]     db[M] (+)= sum_k dy[k][m]
 * mis_gemm's trans = 1 form; the workgroups of the first tile column also sum the dy tile they staged, so the separate
 * mis_colsum launches are not needed.  Deterministic (fixed k slices, fixed-order sums).  M % 4 == N % 4 == 0, 16-byte
 * aligned operands; workspace >= mis_gemm_dw_workspace_bytes(M, N, K) (0: none needed). */
long long mis_gemm_dw_workspace_bytes(int M, int N, int K);
int mis_gemm_dw(const float* dy, long long lddy, const float* x, long long ldx, float* dW, long long lddw, float* db, int M,
                int N, int K, int accumulate, float* workspace, long long workspace_bytes, hipStream_t stream);
/* mis_gemm_dw without its finishing launch (db may be NULL: the weight gradient of a bias-free Linear, :361-362, :390).
 * *slices > 0: the contraction was split and `workspace` -- the caller's own until the sums have run -- holds *slices partial
 * matrices [M][N] followed by *slices partial rows [M] of the bias gradient; dW / db are untouched and `accumulate` is the sum's
 * business (two mis_colsum_job records: stride M*N -> dW, stride M -> db).  *slices == 0: dW / db are complete. */
int mis_gemm_dw_parts(const float* dy, long long lddy, const float* x, long long ldx, float* dW, long long lddw, float* db,
                      int M, int N, int K, int accumulate, float* workspace, long long workspace_bytes, int* slices,
                      hipStream_t stream);
/* the NT kernel instantiation mis_gemm / mis_gemm_ex run this shape with (aligned operands), as a profiler names it */
int mis_gemm_nt_kernel_name(int M, int N, int K, int epilogue, char* name, int name_len);
/* Arithmetic of the nn.Linear GEMMs (mis_gemm / mis_gemm_ex / mis_gemm_dw / mis_gemm_expand): bit 0 forward + dX, bit 1 dW as
 * "bf16x3" products -- each fp32 operand is cut EXACTLY into three bf16 pieces (x = h + m + l) and the product is the sum of the
 * six piece products hh + hm + mh + hl + lh + mm on v_mfma_f32_16x16x32_bf16 with fp32 accumulation (the dropped ml + lm + ll
 * are < 2^-24 |a||b|): the result is as close to exact arithmetic as the fp32 fmaf chain of v_mfma_f32_16x16x4_f32, on a
 * pipe that is 2.4x faster per product.  mask 0 = fp32 MFMA everywhere; default 3 (environment MIS_GEMM_BF3 at load time).
 * Returns the previous mask; mask < 0 only queries. */
int mis_gemm_set_split_precision(int mask);
/* the TN kernel mis_gemm(trans = 1) / mis_gemm_dw run this shape and these operands with, as a profiler names it */
int mis_gemm_tn_kernel_name(const float* A, long long lda, const float* B, long long ldb, const float* C, long long ldc,
                            int M, int N, int K, char* name, int name_len);
/* mis_gemm_expand for FinalPatchExpand_X4 (c = 96) with its LayerNorm and the bias-free 1 x 1 output convolution in the GEMM's
 * epilogue (:401-409 `x = self.expand(x)`, `rearrange`, `x = self.norm(x)`; :749-752 `self.output(x)`): a 96-wide tile is one
 * (p1, p2), its rows are whole tokens of the shuffled tensor.  Writes mean / rstd per shuffled token (for mis_ln_head_bwd*),
 * logits [B][NC][H P][W P] (batch stride logits_bs floats) and, when `out` is not NULL, the shuffled tokens (dense
 * [B H P W P][96]: what the backward reads; NULL for a forward nobody differentiates).  Same bits as mis_gemm_expand +
 * mis_ln_head_fwd.  MIS_ERR_UNSUPPORTED: c != 96, NC outside 2..4, or a shape mis_gemm_expand refuses */
int mis_gemm_expand_ln_head(const float* x, long long lda, const float* W, long long ldb, float* out, int B, int H, int Wd,
                            int K, int P, int c, const float* gamma, const float* beta, const float* head_w, int NC,
                            float eps, float* mean, float* rstd, float* logits, long long logits_bs, mis_stream_t stream);
/* ... on natural-order planes of the expand weight (mis_gemm_split_*_layout, natural = 1; mis_gemm_nt_split_natural(B H W, P P c, K)
 * must say 1 and K <= 96): the persistent resident-panel register-A kernel -- the (p1, p2) column panels stay in LDS, a wave's
 * accumulators are token rows of the shuffled tensor, LayerNorm and the head run in registers. */
int mis_gemm_expand_ln_head_split(const float* x, long long lda, const void* B3, float* out, int B, int H, int W, int K, int P,
                                  int c, const float* gamma, const float* beta, const float* head_w, int NC, float eps,
                                  float* mean, float* rstd, float* logits, long long logits_bs, hipStream_t stream);
/* The NT form with a pre-split B operand.  The bf16x3 kernels cut every fp32 operand into three bf16 pieces; for B = an
 * nn.Linear weight (forward: F.linear(x, W), swin_transformer_unet_skip_expand_decoder_sys.py:14,16,107,109; data gradient:
 * dy . W, i.e. B = W^T) that cut is the same for every tile of the launch and every launch of the step, so it is done once:
 * mis_gemm_split_b writes the three piece planes (bf16 [N][K rounded up to 32], in the lane order the kernels consume) into
 * B3 (mis_gemm_split_bytes(N, K) bytes, 16-byte aligned); mis_gemm_split_job / _batch do it for every weight of a network in one
 * launch (records filled on the host, as mis_transpose_job / mis_transpose_batch).  mis_gemm_nt_split is then
 *   epilogue 0, ex_P 0   mis_gemm(trans = 0)            C (+)= A . B^T + bias
 *   epilogue 1 .. 3      mis_gemm_ex                    the fused GELU / residual epilogues, E1 / C2 / rowscale as there
 *   ex_P > 0             mis_gemm_expand                C = the pixel-shuffled product, M = B ex_H ex_W rows, N = P P ex_c
 * with the same results bit for bit as those entry points under mis_gemm_set_split_precision bit 0 (same pieces, same products,
 * same order) and a third of their vector instructions.  workspace >= mis_gemm_nt_split_workspace_bytes(M, N, K).
 * MIS_ERR_UNSUPPORTED (alignment, or a contraction the tile rule leaves to the short-contraction kernel: K <= 96): use the
 * fp32-B entry points. */
long long mis_gemm_split_bytes(int N, int K);
int mis_gemm_split_b(const float* B, long long ldb, int N, int K, void* B3, mis_stream_t stream);
long long mis_gemm_split_job_bytes(void);
long long mis_gemm_split_job(void* job, const float* B, long long ldb, int N, int K, void* B3, long long first);
int mis_gemm_split_batch(const void* jobs, int n, long long units, mis_stream_t stream);
long long mis_gemm_nt_split_workspace_bytes(int M, int N, int K);
int mis_gemm_nt_split(const float* A, long long lda, const void* B3, float* C, long long ldc, const float* bias, int M, int N,
                      int K, int accumulate, int epilogue, const float* E1, long long lde1, float* C2, long long ldc2,
                      const float* rowscale, long long rows_per_scale, int ex_H, int ex_W, int ex_P, int ex_c,
                      float* workspace, long long workspace_bytes, mis_stream_t stream);
/* Round 6: the register-A form of mis_gemm_nt_split.  For many token rows the A operand of the forward / data-gradient GEMMs
 * (nn.Linear at :14,16,107,109 and their autograd) goes from HBM straight into the v_mfma_f32_16x16x32_bf16 operand registers
 * (lane (row, g) = 8 consecutive contraction elements = 32 contiguous bytes of a row-major activation row); only the pre-split
 * weight planes pass through LDS.  That kernel reads the planes in the NATURAL element order inside a K = 32 block:
 * mis_gemm_nt_split_natural(M, N, K) = 1 says mis_gemm_nt_split_layout(.., layout = 1, ..) serves the shape and that its planes
 * must be cut with mis_gemm_split_b_layout / mis_gemm_split_job_layout (natural = 1); layout / natural = 0 are the entry points
 * above.  Same split products, same k-block order: results agree with the staged kernels to fp32 rounding. */
int mis_gemm_nt_split_natural(int M, int N, int K);
int mis_gemm_split_b_layout(const float* B, long long ldb, int N, int K, void* B3, int natural, hipStream_t stream);
long long mis_gemm_split_job_layout(void* job, const float* B, long long ldb, int N, int K, void* B3, long long first, int natural);
int mis_gemm_nt_split_layout(const float* A, long long lda, const void* B3, float* C, long long ldc, const float* bias, int M,
                             int N, int K, int accumulate, int epilogue, const float* E1, long long lde1, float* C2,
                             long long ldc2, const float* rowscale, long long rows_per_scale, int ex_H, int ex_W, int ex_P,
                             int ex_c, float* workspace, long long workspace_bytes, int layout, hipStream_t stream);
int mis_gemm_nt_split_layout_kernel_name(int M, int N, int K, int epilogue, int layout, char* name, int name_len);
/* proj / fc2 of a 96-channel Swin block with everything up to the next LayerNorm in the register-A kernels' epilogue (planes in
 * the natural order): X = E1 + rowscale[m / rows_per_scale] * (A . B^T + bias) (the residual stream, :276 / :281) and
 * Y = LayerNorm(X) * gamma + beta with mean / rstd [M] kept for the backward (`self.norm2(x)`, :280, or the next block's `norm1`,
 * :249) -- the LayerNorm launch and its read of X are gone.  N must be 96.  MIS_ERR_UNSUPPORTED: mis_gemm_nt_split_natural(M, 96, K)
 * is 0, or an operand is not float4-addressable. */
int mis_gemm_nt_residual_ln(const float* A, long long lda, const void* B3, const float* bias, int M, int N, int K,
                            const float* E1, long long lde1, const float* rowscale, long long rows_per_scale, float* X,
                            long long ldx, const float* gamma, const float* beta, float eps, float* Y, long long ldy,
                            float* mean, float* rstd, hipStream_t stream);
int mis_gemm_nt_split_kernel_name(int M, int N, int K, int epilogue, char* name, int name_len);
/* nn.Linear of PatchExpand / FinalPatchExpand_X4 fused with their pixel shuffle
 * 'b h w (p1 p2 c) -> b (h p1) (w p2) c' (swin_transformer_unet_skip_expand_decoder_sys.py:373-380, :401-408):
 * x [B*H*W][K] (row stride lda), W [P*P*c][K] (row stride ldb), out [B*H*P*W*P][c] dense.  No bias.
 * MIS_ERR_UNSUPPORTED when c % 16 != 0 or when mis_gemm would split K for this shape (use mis_gemm +
 * mis_token_rearrange then). */
int mis_gemm_expand(const float* x, long long lda, const float* W, long long ldb, float* out, int B, int H, int Wd,
                    int K, int P, int c, mis_stream_t stream);
int mis_gemm(const float* A, long long lda, const float* B, long long ldb, float* C, long long ldc,
             const float* bias, int M, int N, int K, int trans, int accumulate, float* workspace,
             long long workspace_bytes, mis_stream_t stream);
/* NT form with a fused epilogue, v = A.B^T + bias (reference ...sys.py: Mlp :9-25, SwinTransformerBlock.forward
 * :244-288):  1: C = v and C2 = gelu(v) (fc1 + GELU; the pre-activation stays for the backward);
 *             2: C = v * gelu'(E1) (the dX of fc2 lands directly in the gradient of fc1's output);
 *             3: C = E1 + rowscale[m / rows_per_scale] * v (proj / fc2 + DropPath + residual add; rowscale NULL = 1).
 * E1, C2: [M][N] views with their own row strides.  Workspace as mis_gemm (trans = 0). */
int mis_gemm_ex(const float* A, long long lda, const float* B, long long ldb, float* C, long long ldc,
                const float* bias, int M, int N, int K, int epilogue, const float* E1, long long lde1, float* C2,
                long long ldc2, const float* rowscale, long long rows_per_scale, float* workspace,
                long long workspace_bytes, mis_stream_t stream);
/* table[site*B + b] = DropPath scale of sample b at residual site `site` (what mis_residual_droppath derives on the
 * fly from (state, salt[site], b)); p, salt: device arrays of nsites entries */
int mis_droppath_table(float* table, const float* p, const unsigned* salt, int nsites, int B,
                       const MisStepState* state, mis_stream_t stream);
/* out[c][r] = in[r][c]: weight^T for the input-gradient GEMM (packed once per step) */
int mis_transpose(const float* in, long long ldi, float* out, long long ldo, int rows, int cols, mis_stream_t stream);
/* every Linear weight of a network in ONE launch (the dX GEMM of nn.Linear wants W^T; reference: autograd of nn.Linear in
 * swin_transformer_unet_skip_expand_decoder_sys.py:17-31,115-150): mis_transpose_job fills one record of a HOST table
 * (mis_transpose_job_bytes each; dense row-major in[rows][cols] -> out[cols][rows]; `first` = tiles of the jobs before it)
 * and returns the job's tile count; mis_transpose_batch runs a DEVICE copy of the table. */
long long mis_transpose_job_bytes(void);
long long mis_transpose_job(void* job, const float* in, float* out, int rows, int cols, long long first);
int mis_transpose_batch(const void* jobs, int n, long long tiles, mis_stream_t stream);
/* nn.LayerNorm over the last dim (:204,211,323,365,393,716-717); mean/rstd: M floats saved for backward */
int mis_layernorm_fwd(const float* x, long long ldx, float* y, long long ldy, const float* gamma, const float* beta,
                      float* mean, float* rstd, long long M, int C, float eps, mis_stream_t stream);
long long mis_colreduce_workspace_bytes(long long M, int C);
int mis_layernorm_bwd(const float* x, long long ldx, const float* dy, long long lddy, float* dx, long long lddx,
                      const float* gamma, const float* mean, const float* rstd, float* dgamma, float* dbeta,
                      long long M, int C, int accumulate_dx, int accumulate_affine, void* workspace,
                      long long workspace_bytes, mis_stream_t stream);
/* mis_layernorm_bwd in two halves (same results bit for bit): _parts = dx and the per-slab partials of the affine gradients in
 * `workspace` (mis_colreduce_workspace_bytes), _final = dgamma / dbeta from them.  The second half may run on another stream
 * ordered behind the first -- autograd's LayerNorm backward (:204,211) gives the affine gradients to nobody downstream, so the
 * token plans keep it off the data-gradient chain (the workspace must then stay untouched until it ran) */
int mis_layernorm_bwd_parts(const float* x, long long ldx, const float* dy, long long lddy, float* dx, long long lddx,
                            const float* gamma, const float* mean, const float* rstd, long long M, int C,
                            int accumulate_dx, void* workspace, long long workspace_bytes, mis_stream_t stream);
int mis_layernorm_bwd_final(const void* workspace, long long workspace_bytes, long long M, int C, float* dgamma,
                            float* dbeta, int accumulate_affine, mis_stream_t stream);
/* Every finishing column sum of a backward pass in one launch.  The parameter gradients that autograd hands to nobody downstream
 * -- LayerNorm weight / bias (:204,211,323,365,393), the split-K partials of the nn.Linear weight / bias gradients
 * (:14,16,107,109), relative_position_bias_table (:99-102) -- are all  out[c] (+)= sum_s part[s][c]  over per-slab / per-slice /
 * per-window partial rows that the producing kernels (mis_layernorm_bwd_parts, mis_gemm_dw_parts,
 * mis_window_attention_bwd_parts_ws) left in caller-owned workspaces.  mis_colsum_job fills one record of a HOST table
 * (mis_colsum_job_bytes each): part = float [slabs][stride] (pairs = 0, -> out_a), float2 [slabs][stride] (pairs = 1: .x ->
 * out_a, .y -> out_b, either may be NULL) or the k-slices of a split GEMM (pairs = 2: float, C % 4 == 0, 16-byte aligned,
 * summed in fp32 in the order of the GEMMs' own reduction -- the bits of mis_gemm_dw), C <= stride columns, `first` = workgroups
 * of the jobs before it; it returns the job's workgroup count.  mis_colsum_batch runs the table (uploaded by the caller) as one
 * launch: fixed order (pairs 0 / 1: double accumulators, the bits of mis_layernorm_bwd_final / mis_window_attention_dtable_ws),
 * a job's result independent of its batch.  mis_colreduce_slabs = the partial rows mis_layernorm_bwd_parts leaves for M tokens. */
long long mis_colsum_job_bytes(void);
long long mis_colreduce_slabs(long long M);
long long mis_colsum_job(void* job, const void* part, long long stride, long long slabs, int C, int pairs, float* out_a,
                         float* out_b, int accumulate, long long first);
int mis_colsum_batch(const void* jobs, int n, long long blocks, mis_stream_t stream);
/* mis_layernorm_bwd_parts and, in the same pass, the backward of the residual add that produced the LayerNorm's input
 * (`x = shortcut + self.drop_path(x)` followed by `self.norm2(x)` / the next block's `norm1`, :276-281): with
 * total = gin + LayerNorm'(dy) (gin, may be NULL: what the input's other readers already left in its gradient),
 * d_shortcut (+)= total and d_branch = rowscale[row / rows_per_scale] * total (rowscale NULL: 1, no DropPath).  The input's
 * own gradient is not written: one write and one read of a token tensor less than mis_layernorm_bwd +
 * mis_residual_droppath(backward).  C <= 1536; affine partials as mis_layernorm_bwd_parts (finish with _final) */
int mis_layernorm_bwd_residual_parts(const float* x, long long ldx, const float* dy, long long lddy, const float* gin,
                                     long long ldgin, float* d_shortcut, long long ldds, int accumulate_shortcut,
                                     float* d_branch, long long lddb, const float* rowscale, long long rows_per_scale,
                                     const float* gamma, const float* mean, const float* rstd, long long M, int C,
                                     void* workspace, long long workspace_bytes, mis_stream_t stream);
/* out[c] (+)= sum_rows x[row][c]: nn.Linear bias gradient */
int mis_colsum(const float* x, long long ldx, long long M, int C, float* out, int accumulate, void* workspace,
               long long workspace_bytes, mis_stream_t stream);
/* nn.GELU, exact erf form (:10,15): backward == 0: out = gelu(x); 1: out = dy * gelu'(x) */
int mis_gelu(const float* x, const float* dy, float* out, long long n, int backward, mis_stream_t stream);
/* x = shortcut + drop_path(branch) (:285-286; timm DropPath = per-sample Bernoulli / (1-p)).
 * forward: out = a + s_b*y; backward (y NULL, a = d out): out (optional) = a, out2 = s_b*a. */
int mis_residual_droppath(const float* a, long long lda, const float* y, long long ldy, float* out, long long ldo,
                          float* out2, long long ldo2, long long M, int C, long long rows_per_sample, float drop_p,
                          unsigned salt, const MisStepState* state, const float* scale_override, int backward,
                          mis_stream_t stream);
/* mode 0: PatchMerging 2x2 gather (:336-344); mode 1: PatchExpand / FinalPatchExpand_X4 pixel shuffle
 * 'b h w (p1 p2 c) -> b (h p1) (w p2) c' (:377-380,405-408); inverse = 1: the backward scatter */
int mis_token_rearrange(const float* src, long long lds, float* dst, long long ldd, int B, int H, int W, int C, int P,
                        int mode, int inverse, mis_stream_t stream);
/* PatchEmbed 4x4/stride-4 conv as im2col rows [B*H/4*W/4][in_chans*16], with the 1 -> in_chans channel
 * repeat of vision_transformer.py:49-50 folded in (:573-588) */
int mis_patch_im2col(const float* x, long long x_bs, float* out, int B, int H, int W, int in_chans,
                     mis_stream_t stream);
/* same with an explicit source channel count: src_chans == in_chans reads x[b][c] (3-channel inputs, the other branch
 * of vision_transformer.py:48-50), src_chans == 1 repeats the single channel */
int mis_patch_im2col_c(const float* x, long long x_bs, float* out, int B, int H, int W, int in_chans, int src_chans,
                       mis_stream_t stream);
/* up_x4 tail (:775-786): token-major [B*S][K] -> NCHW logits[B][NC][S] through the bias-free 1x1 conv */
int mis_head_fwd(const float* x, long long ldx, const float* w, float* logits, long long y_bs, int B, long long S,
                 int K, int NC, mis_stream_t stream);
long long mis_head_workspace_bytes(int K, int NC);
int mis_head_bwd(const float* x, long long ldx, const float* w, const float* dlogits, long long dy_bs, float* dx,
                 long long lddx, float* dw, int accumulate_dw, int B, long long S, int K, int NC, void* workspace,
                 long long workspace_bytes, mis_stream_t stream);
/* LayerNorm(C) + the bias-free 1x1 output convolution of SwinUnet's tail in one pass (FinalPatchExpand_X4.norm + output:
 * swin_transformer_unet_skip_expand_decoder_sys.py:390-409, :671, :749-752): the normalised 16x expanded token tensor is
 * never written.  x [B*S][C] token-major, w [NC][C], logits [B][NC][S]; C % 4 == 0, C <= 128, NC in 2..4; mean / rstd
 * [B*S] are kept for the backward, which writes dx (+)= and dgamma / dbeta / dw (+)= from one more read of x.
 * Deterministic.  workspace >= mis_ln_head_workspace_bytes(B*S, C, NC). */
int mis_ln_head_fwd(const float* x, long long ldx, const float* gamma, const float* beta, const float* w, float* mean,
                    float* rstd, float* logits, long long y_bs, int B, long long S, int C, int NC, float eps,
                    mis_stream_t stream);
long long mis_ln_head_workspace_bytes(long long M, int C, int NC);
int mis_ln_head_bwd(const float* x, long long ldx, const float* gamma, const float* beta, const float* w, const float* mean,
                    const float* rstd, const float* dlogits, long long dl_bs, float* dx, long long lddx, int accumulate_dx,
                    float* dgamma, float* dbeta, float* dw, int accumulate_params, int B, long long S, int C, int NC,
                    void* workspace, long long workspace_bytes, mis_stream_t stream);
/* ... with dx stored through the INVERSE pixel shuffle of FinalPatchExpand_X4 (reference
 * swin_transformer_unet_skip_expand_decoder_sys.py:401-408): x rows are the tokens of the (H P) x (W P) grid, dx is the gradient of the
 * expand Linear's output [B H W][P P C] -- the separate un-shuffle pass over the network's largest tensor disappears */
int mis_ln_head_bwd_unshuffle(const float* x, long long ldx, const float* gamma, const float* beta, const float* w,
                              const float* mean, const float* rstd, const float* dlogits, long long dl_bs, float* dx,
                              long long lddx, int accumulate_dx, float* dgamma, float* dbeta, float* dw, int accumulate_params,
                              int B, int H, int W, int P, int C, int NC, void* workspace, long long workspace_bytes,
                              mis_stream_t stream);
/* (shifted-)window attention core (:115-150 with the roll/partition/reverse of :244-288 folded into the
 * token addressing): qkv [B*H*W][3*nH*32] in natural token order -> out [B*H*W][nH*32]; window 7x7,
 * head_dim 32; bias_table = relative_position_bias_table [169][nH]; shift in {0,3}. */
int mis_window_attention_fwd(const float* qkv, long long ldq, float* out, long long ldo, const float* bias_table,
                             int B, int H, int W, int nH, int shift, float scale, mis_stream_t stream);
long long mis_window_attention_workspace_bytes(int B, int H, int W, int nH);
int mis_window_attention_bwd(const float* qkv, long long ldq, const float* dout, long long ldo, float* dqkv,
                             long long lddq, const float* bias_table, float* dbias_table, int accumulate_table,
                             int B, int H, int W, int nH, int shift, float scale, void* workspace,
                             long long workspace_bytes, mis_stream_t stream);
/* the same three with the window size as an argument: window 7 (bias_table [169][nH], shift in {0,3}) or window 8
 * (bias_table [225][nH], shift in {0,4}) -- MODEL.SWIN.WINDOW_SIZE = 8 with DATA.IMG_SIZE = 256 is how the
 * reference runs SwinUnet on 256 x 256 inputs (code/config.py:194-195, swin_transformer_unet_...sys.py:198-201) */
int mis_window_attention_fwd_ws(const float* qkv, long long ldq, float* out, long long ldo, const float* bias_table,
                                int B, int H, int W, int nH, int shift, float scale, int window, mis_stream_t stream);
long long mis_window_attention_workspace_bytes_ws(int B, int H, int W, int nH, int window);
int mis_window_attention_bwd_ws(const float* qkv, long long ldq, const float* dout, long long ldo, float* dqkv,
                                long long lddq, const float* bias_table, float* dbias_table, int accumulate_table,
                                int B, int H, int W, int nH, int shift, float scale, int window, void* workspace,
                                long long workspace_bytes, mis_stream_t stream);
/* mis_window_attention_bwd_ws in two halves (same results bit for bit): _parts_ws = dqkv and the per-(sample, window) partials
 * of the table gradient in `workspace` (round 6: each (sample, window, head) wave reduces its dS block to the (2 window - 1)^2
 * table entries itself), _dtable_ws = the gradient of relative_position_bias_table from them (:99-131: autograd's index_add
 * through relative_position_index).  As with mis_layernorm_bwd_{parts,final}: the table gradient feeds nothing downstream and
 * may run on another stream behind the first half -- or as a mis_colsum_batch job: mis_window_attention_table_partials gives
 * the shape of the partials (float [rows][cols] at the start of the workspace, cols in the table's own [index][head] order;
 * MIS_ERR_UNSUPPORTED when the vector-pipe kernels, whose partials are dS blocks, are selected) */
int mis_window_attention_bwd_parts_ws(const float* qkv, long long ldq, const float* dout, long long ldo, float* dqkv,
                                      long long lddq, const float* bias_table, int B, int H, int W, int nH, int shift,
                                      float scale, int window, void* workspace, long long workspace_bytes,
                                      mis_stream_t stream);
int mis_window_attention_dtable_ws(void* workspace, long long workspace_bytes, float* dbias_table, int accumulate_table,
                                   int B, int H, int W, int nH, int window, mis_stream_t stream);
int mis_window_attention_table_partials(int B, int H, int W, int nH, int window, long long* rows, int* cols);

/* ---- UNETR (reference code/networks/unetr.py, built from MONAI blocks that are NOT vendored in the reference:
 * parity of these entry points is pinned to a torch restatement of the published algorithm only) --------------------
 * full multi-head self-attention over N <= 256 tokens, head_dim 64: qkv [B*N][3*nH*64] (q | k | v, head, dim) ->
 * out [B*N][nH*64]; stats (B*nH*N*2 floats: row max, row sum) are kept for the backward */
int mis_full_attention_fwd(const float* qkv, long long ldq, float* out, long long ldo, float* stats, int B, int N,
                           int nH, float scale, mis_stream_t stream);
long long mis_full_attention_workspace_bytes(int B, int N, int nH);
int mis_full_attention_bwd(const float* qkv, long long ldq, const float* dout, long long ldo, float* dqkv,
                           long long lddq, const float* stats, int B, int N, int nH, float scale, void* workspace,
                           long long workspace_bytes, mis_stream_t stream);
/* 'b c (h p1) (w p2) (d p3) -> b (h w d) (p1 p2 p3 c)' for c = 1: x [B][1][H][W][D] -> out [B*(H/P)(W/P)(D/P)][P^3] */
int mis_patch3d_im2col(const float* x, long long x_bs, float* out, int B, int H, int W, int D, int P,
                       mis_stream_t stream);
/* out[row] = x[row] + pos[row % L] (position embeddings); dpos[l] = sum over the rows with row % L == l of dy[row] */
int mis_add_rowcycle(const float* x, long long ldx, const float* pos, float* out, long long ldo, long long M, int C,
                     int L, mis_stream_t stream);
int mis_sum_rowcycle(const float* dy, long long ld, float* dpos, long long M, int C, int L, mis_stream_t stream);

/* ---- input pipeline on the device (SURVEY s.8 row n4): the training set is resident in HBM as one float pool
 * (images) and one byte pool (labels); a batch is one gather launch that applies the reference's per-sample
 * augmentation.  Random draws stay on the host in the reference's order and arrive as B parameter records in
 * DEVICE memory.  Selected source pixels are exactly scipy's / numpy's (bit-exact outputs). */
typedef struct MisAug2D {        /* code/dataloaders/dataset.py:406-425 RandomGenerator on one slice */
    long long img_off, lab_off;  /* element offsets of the H x W slice in img_pool / lab_pool */
    int H, W;
    int mode;                    /* 0: none; 1: random_rot_flip (:79-89) np.rot90(k) then np.flip(axis);
                                    2: random_rotate (:92-96) scipy.ndimage.rotate(angle, order=0, reshape=False) */
    int k, axis, reserved;
    double m00, m01, m10, m11;   /* mode 2: scipy's rot_matrix [[cosdg, sindg], [-sindg, cosdg]] ... */
    double off0, off1;           /* ... and offset = in_center - rot_matrix @ out_center */
} MisAug2D;
/* image_out [B][1][out_h][out_w] f32, label_out [B][out_h][out_w] u8 (may be NULL): augmentation followed by
 * scipy.ndimage.zoom(.., (out_h/x, out_w/y), order=0) (:417-420) */
int mis_augment2d(const float* img_pool, const unsigned char* lab_pool, const void* params_device, int B, int out_h,
                  int out_w, float* image_out, unsigned char* label_out, mis_stream_t stream);

typedef struct MisCrop3D {       /* code/dataloaders/brats2019.py RandomRotFlip (:134-147) + RandomCrop (:84-131) */
    long long img_off, lab_off;
    int d0, d1, d2;              /* volume size (w, h, d) */
    int k, axis;                 /* np.rot90(k) in the (0,1) plane, np.flip(axis in {0,1}) */
    int o0, o1, o2;              /* crop origin in the rotated volume minus the zero padding (:99-108), may be < 0 */
} MisCrop3D;
/* image_out [B][1][p0][p1][p2] f32 (ToTensor :196-208), label_out [B][p0][p1][p2] u8 (label_bytes 1) or int64 (8) */
int mis_crop_rotflip3d(const float* img_pool, const unsigned char* lab_pool, const void* params_device, int B, int p0,
                       int p1, int p2, float* image_out, void* label_out, int label_bytes, mis_stream_t stream);

/* ---- SwinUNETR encoder (csrc/swin3d.hip): 3-D shifted-window attention and patch merging ------------------------
 * reference: net_factory_3d('swinunetr') = monai.networks.nets.SwinUNETR(img_size=(64,64,64), in_channels, out_channels,
 * feature_size=48) (code/networks/net_factory_3d.py:7,37-38).  MONAI is an un-vendored dependency: PARITY UNPINNED, the
 * published algorithm (SwinTransformerBlock.forward_part1, WindowAttention, PatchMerging "merging") is restated in
 * oracle/swinunetr.py.  Token-major activations [B][D][H][W][C], C % 4 == 0; heads of 16 channels.
 * mis_win3d_gather: windows [B*nW][n][C] <- tokens after zero padding to multiples of the window and the cyclic shift
 *   torch.roll(x, (-sd,-sh,-sw)) (inverse != 0: tokens <- windows; each is the other's gradient).
 * mis_merge3d: merged [B][D/2][H/2][W/2][8C] <- tokens in MONAI's v0.9 slot order (inverse: gradient of the tokens).
 * mis_win3d_attn_fwd / _bwd: softmax(q k^T / 4 + table[index[:n,:n]] + shift mask) v per (window, head) from qkv
 *   [BW*n][3*nH*16]; region [nW][n] int32 shift-region ids or NULL; dtable [2197][nH] deterministic. */
int mis_win3d_gather(const float* src, float* dst, int B, int D, int H, int W, int C, int wd, int wh, int ww, int sd,
                     int sh, int sw, int inverse, mis_stream_t stream);
long long mis_win3d_windows(int B, int D, int H, int W, int wd, int wh, int ww);
int mis_merge3d(const float* src, float* dst, int B, int D, int H, int W, int C, int inverse, mis_stream_t stream);
int mis_win3d_attn_fwd(const float* qkv, long long ldq, float* out, long long ldo, float* stats, const float* table,
                       const int* region, int BW, int nW, int n, int nH, mis_stream_t stream);
long long mis_win3d_attn_workspace_bytes(int BW, int n, int nH);
int mis_win3d_attn_bwd(const float* qkv, long long ldq, const float* out, const float* dout, long long ldo, float* dqkv,
                       long long lddq, const float* stats, const float* table, const int* region, float* dtable,
                       int accumulate_table, int BW, int nW, int n, int nH, void* workspace, long long workspace_bytes,
                       mis_stream_t stream);

/* ---- 1x1x1 convolution of channel-major volumes with few voxels and many channels as a batched GEMM (csrc/conv1x1_gemm.hip)
 * The contraction inside nn.Conv3d(C, 2C, 2, stride=2) / nn.ConvTranspose3d(2C, C, 2, stride=2) on V-Net's deep levels
 * (reference code/networks/vnet.py:73, :100) once the 2x2x2 taps are folded into channels (mis_space_to_depth2):
 *   y[n][co][s] (+)= bias[co] + sum_ci wt[ci][co] x[n][ci][s],   s over the S voxels of the coarse volume,
 * weights CONTRACTION-major ([Cin][Cout], row stride ldw); x / y NCDHW views (channel stride S, batch strides x_bs / y_bs).
 * Split over the input channels when (image, tile) entries would not fill the chip: partials in `workspace`
 * (mis_conv1x1_gemm_workspace_bytes; 0 = none needed), fixed-order reduction.  S, Cout and the strides multiples of 4, 16-byte
 * aligned pointers; same result as mis_conv_fwd(k = 1) up to fp32 summation order. */
long long mis_conv1x1_gemm_workspace_bytes(int N, int Cin, int Cout, long long S);
int mis_conv1x1_gemm(const float* x, long long x_bs, const float* wt, long long ldw, const float* bias, float* y, long long y_bs,
                     int N, int Cin, int Cout, long long S, int accumulate, float* workspace, long long workspace_bytes,
                     mis_stream_t stream);
/* Its weight gradient: dw[m][n] (+)= sum over the N images and S voxels of a[img][m][s] * b[img][n][s] (a = dy, b = x gives
 * [Cout][Cin]; swapped operands give the transposed layout directly -- ConvTranspose3d's parameter is input-major).  Slices over
 * (image, voxel chunk), partials in `workspace` (always needed), fixed-order reduction.  S, a_bs, b_bs multiples of 4. */
long long mis_conv1x1_wgrad_workspace_bytes(int N, int M, int Nc, long long S);
int mis_conv1x1_wgrad(const float* a, long long a_bs, const float* b, long long b_bs, float* dw, long long ldw, int N, int M, int Nc,
                      long long S, int accumulate, float* workspace, long long workspace_bytes, mis_stream_t stream);

/* Test support (never on the product path): fills the LDS of every CU with NaNs so that a kernel reading LDS it did not write
 * fails deterministically instead of depending on the previous launch.  sink: any device float (or NULL). */
int mis_debug_poison_lds(float* sink, mis_stream_t stream);
/* diagnostics: `blocks` workgroups of 4 register-light, LDS-free waves that keep one execution pipe busy for `iters` rounds
 * (kind 1 bf16 MFMA, 2 fp32 MFMA, 3 unpacked VALU, 4 packed fp32 VALU): a co-residency probe -- does a foreign wave on the
 * same SIMD change another kernel's results? (scripts/interference.py) */
int mis_debug_spin(int kind, int blocks, int iters, float* sink, mis_stream_t stream);
/* Development support (never on the product path): a device buffer of 8 x uint64 per (workgroup, wave) that a -DMIS_WR_PROF=1
 * build of the z-ring Winograd weight-gradient kernel fills with the cycles of its loop phases (scripts/wgrad_prof.py); NULL
 * switches it off.  Returns MIS_ERR_UNSUPPORTED in the product build, which never writes the buffer. */
int mis_debug_wgrad_prof(unsigned long long* buf);

#ifdef __cplusplus
}
#endif
#endif /* MIS_HIP_H */
