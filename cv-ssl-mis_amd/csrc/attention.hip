// (Shifted-)window multi-head self-attention core of SwinUnet, forward and backward.
//
// Replaces WindowAttention.forward between the qkv and proj Linear layers and the roll /
// window_partition / window_reverse plumbing of SwinTransformerBlock.forward (reference
// code/networks/swin_transformer_unet_skip_expand_decoder_sys.py:115-150, 28-60, 244-288, mask :216-238):
//
//   attn = softmax((q*scale) @ k^T + rel_pos_bias[h] (+ shift mask)) ;  out = attn @ v
//
// The qkv projection is token-wise, so it is applied to the tokens in their natural order; the
// cyclic shift and the window partition are folded into the token index each lane computes
// (token of window (wy,wx), position (iy,ix): ((wy*7+iy+shift) % H, (wx*7+ix+shift) % W)), and the
// output is written straight back to that token -- no roll / partition / reverse copies in HBM.
//
// Window = 7x7 = 49 tokens, head_dim = 32 (the only geometry the reference instantiates): one wave
// per (sample, window, head).  Two generations live here: the first (attn_fwd_kernel / attn_bwd_kernel, selected by
// MIS_ATTN_VALU=1 or for buffers beyond 2^31 bytes) runs on the fp32 vector pipe -- lane i < 49 owns query row i,
// K/V rows sit in LDS and are read as broadcasts; the second (attn_*_mfma_kernel, the default) pads the window to 64
// and runs all five matrix products on v_mfma_f32_16x16x4_f32 (measured at 48 images: forward 132 -> 79 us,
// backward 784 -> 432 us at 56^2; 335 -> 75 us at 7^2).
#include "common.h"
#include <stdlib.h>

namespace {

#define MIS_ATTN_WS 7
#define MIS_ATTN_NS ws7
#include "attention_impl.inc"
#undef MIS_ATTN_WS
#undef MIS_ATTN_NS

// window 8 (64 tokens: no padded rows): the reference's IMG_SIZE 256 / WINDOW_SIZE 8 configuration
// (code/config.py:194-195), which lets SwinUnet run the 256 x 256 inputs of BASELINE config 5
#define MIS_ATTN_WS 8
#define MIS_ATTN_NS ws8
#include "attention_impl.inc"
#undef MIS_ATTN_WS
#undef MIS_ATTN_NS

}  // namespace

// head_dim is fixed to 32; window 7 (SwinUnet tiny: heads 3/6/12/24 at C = 96..768) or 8
extern "C" int mis_window_attention_fwd_ws(const float* qkv, long long ldq, float* out, long long ldo,
                                           const float* bias_table, int B, int H, int W, int nH, int shift, float scale,
                                           int window, hipStream_t stream) {
    if (window == 7) return ws7::window_attention_fwd(qkv, ldq, out, ldo, bias_table, B, H, W, nH, shift, scale, stream);
    if (window == 8) return ws8::window_attention_fwd(qkv, ldq, out, ldo, bias_table, B, H, W, nH, shift, scale, stream);
    return MIS_ERR_UNSUPPORTED;
}

extern "C" long long mis_window_attention_workspace_bytes_ws(int B, int H, int W, int nH, int window) {
    if (window == 7) return ws7::window_attention_workspace_bytes(B, H, W, nH);
    if (window == 8) return ws8::window_attention_workspace_bytes(B, H, W, nH);
    return MIS_ERR_UNSUPPORTED;
}

extern "C" int mis_window_attention_bwd_ws(const float* qkv, long long ldq, const float* dout, long long ldo,
                                           float* dqkv, long long lddq, const float* bias_table, float* dbias_table,
                                           int accumulate_table, int B, int H, int W, int nH, int shift, float scale,
                                           int window, void* workspace, long long workspace_bytes, hipStream_t stream) {
    if (window == 7)
        return ws7::window_attention_bwd(qkv, ldq, dout, ldo, dqkv, lddq, bias_table, dbias_table, accumulate_table, B, H,
                                         W, nH, shift, scale, workspace, workspace_bytes, stream);
    if (window == 8)
        return ws8::window_attention_bwd(qkv, ldq, dout, ldo, dqkv, lddq, bias_table, dbias_table, accumulate_table, B, H,
                                         W, nH, shift, scale, workspace, workspace_bytes, stream);
    return MIS_ERR_UNSUPPORTED;
}

// mis_window_attention_bwd_ws in two halves for callers that keep the bias-table gradient off the data-gradient chain (the
// token plans run the second half beside it, with a workspace of the layer's own): _parts_ws = dqkv and the per-unit dS partials,
// _dtable_ws = the relative-position-bias-table gradient from them (reference ...sys.py:99-131: autograd's index_add through
// relative_position_index).  Same workspace size as the one-call form; same results bit for bit.
extern "C" int mis_window_attention_bwd_parts_ws(const float* qkv, long long ldq, const float* dout, long long ldo,
                                                 float* dqkv, long long lddq, const float* bias_table, int B, int H, int W,
                                                 int nH, int shift, float scale, int window, void* workspace,
                                                 long long workspace_bytes, hipStream_t stream) {
    if (window == 7)
        return ws7::window_attention_bwd(qkv, ldq, dout, ldo, dqkv, lddq, bias_table, nullptr, 0, B, H, W, nH, shift, scale,
                                         workspace, workspace_bytes, stream, 1);
    if (window == 8)
        return ws8::window_attention_bwd(qkv, ldq, dout, ldo, dqkv, lddq, bias_table, nullptr, 0, B, H, W, nH, shift, scale,
                                         workspace, workspace_bytes, stream, 1);
    return MIS_ERR_UNSUPPORTED;
}

extern "C" int mis_window_attention_dtable_ws(void* workspace, long long workspace_bytes, float* dbias_table,
                                              int accumulate_table, int B, int H, int W, int nH, int window,
                                              hipStream_t stream) {
    if (window == 7)
        return ws7::window_attention_bwd(nullptr, 0, nullptr, 0, nullptr, 0, nullptr, dbias_table, accumulate_table, B, H, W,
                                         nH, 0, 0.f, workspace, workspace_bytes, stream, 2);
    if (window == 8)
        return ws8::window_attention_bwd(nullptr, 0, nullptr, 0, nullptr, 0, nullptr, dbias_table, accumulate_table, B, H, W,
                                         nH, 0, 0.f, workspace, workspace_bytes, stream, 2);
    return MIS_ERR_UNSUPPORTED;
}

// Shape of the bias-table partials mis_window_attention_bwd_parts_ws leaves at the start of its workspace: float [rows][cols],
// cols = (2 window - 1)^2 * nH in the layout of relative_position_bias_table -- dtable = their column sum, which
// mis_window_attention_dtable_ws forms, or a mis_colsum_batch job together with the other finishing sums of a backward pass.
extern "C" int mis_window_attention_table_partials(int B, int H, int W, int nH, int window, long long* rows, int* cols) {
    if (window == 7) return ws7::window_attention_table_partials(B, H, W, nH, rows, cols);
    if (window == 8) return ws8::window_attention_table_partials(B, H, W, nH, rows, cols);
    return MIS_ERR_UNSUPPORTED;
}

// the window-7 entry points of ABI version 1
extern "C" int mis_window_attention_fwd(const float* qkv, long long ldq, float* out, long long ldo,
                                        const float* bias_table, int B, int H, int W, int nH, int shift, float scale,
                                        hipStream_t stream) {
    return mis_window_attention_fwd_ws(qkv, ldq, out, ldo, bias_table, B, H, W, nH, shift, scale, 7, stream);
}

extern "C" long long mis_window_attention_workspace_bytes(int B, int H, int W, int nH) {
    return mis_window_attention_workspace_bytes_ws(B, H, W, nH, 7);
}

extern "C" int mis_window_attention_bwd(const float* qkv, long long ldq, const float* dout, long long ldo,
                                        float* dqkv, long long lddq, const float* bias_table, float* dbias_table,
                                        int accumulate_table, int B, int H, int W, int nH, int shift, float scale,
                                        void* workspace, long long workspace_bytes, hipStream_t stream) {
    return mis_window_attention_bwd_ws(qkv, ldq, dout, ldo, dqkv, lddq, bias_table, dbias_table, accumulate_table, B, H, W,
                                       nH, shift, scale, 7, workspace, workspace_bytes, stream);
}
