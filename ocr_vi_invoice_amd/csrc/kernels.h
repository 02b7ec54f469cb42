// Non-GEMM kernels of the hot path (layout, pooling, norms, attention, ASF, DB head tail, CTC).
// Every launcher validates shapes on the host, enqueues on `stream`, never allocates or synchronises.
#pragma once
#include "common.h"

namespace ocrvi {

// float32 NCHW [N,3,H,W] -> T zero-padded NHWC4 [N,Hp,Wp,4]; image origin at (pad_t, pad_l); channel 3 = 0.
int k_nchw3_to_nhwc4_pad(int dtype, const float* x, void* y, int N, int H, int W, int pad_t, int pad_l, int Hp, int Wp, hipStream_t s);
// MaxPool2d(3, stride 2, pad 1) on NHWC T (torchvision resnet stem, backbone.py:34).
int k_maxpool3x3s2(int dtype, const void* x, void* y, int N, int H, int W, int C, hipStream_t s);
// f16x2: the detector's stem conv 7x7 / 2 (+ folded BN, ReLU) and this max-pool fused (stem_pool.hip).  xpad: the stem conv's padded NHWC4
// input [N][Hp][Wp][4]; w / bias / wscale: the stem's packed weights (pack_conv, AM_ROWS); y: [N][H / 4][W / 4][64].
bool stem_pool_eligible(int dtype, int cout, int Kp, int KH, int H, int W);
int k_stem_pool(int dtype, const void* xpad, const void* w, const float* bias, float wscale, void* y, int N, int H, int W, int Hp, int Wp,
                hipStream_t s);
// LayerNorm over the last dim (eps 1e-5, svtrv2.py:93,95,446).  x is f32 or T, out is f32 or T.
int k_layernorm(int dtype, const void* x, int x_f32, void* out, int out_f32, const float* gamma, const float* beta, int rows, int D,
                hipStream_t s);
// Fused MixingBlock MLP for the 16-bit modes (mlp_fused.hip):  x <- x + fc2(gelu(fc1(LN(x; ln_g, ln_b))))  in place on the fp32
// residual stream x [M][D]; when xn != null also writes xn [M][D] T = LN(x_new; next_g, next_b), or T(x_new) when next_g == null.
// wstream = pack_mlp_stream(fc1.weight [4D][D], fc2.weight [D][4D]) uploaded to the device; b1 [4D], b2 [D] device fp32.
bool mlp_fused_eligible(int dtype, int D);
// f16x2, D = 128 / 256 (mlp_x2.hip): same contract; the stream carries its two weight scales behind the units
bool mlp_x2_eligible(int dtype, int D);
void pack_mlp_x2_stream(const float* w1, const float* w2, int D, std::vector<char>& out);
int k_mlp_x2(float* x, void* xn, const float* ln_g, const float* ln_b, const float* next_g, const float* next_b, const void* wstream, const float* b1,
             const float* b2, int M, int D, hipStream_t s);
void pack_mlp_stream(const float* w1, const float* w2, int D, int dtype, std::vector<char>& out);
int k_mlp_fused(int dtype, float* x, void* xn, const float* ln_g, const float* ln_b, const float* next_g, const float* next_b, const void* wstream,
                const float* b1, const float* b2, int M, int D, hipStream_t s);
// f32 -> T cast (n elements).
int k_cast_from_f32(int dtype, const float* x, void* y, size_t n, hipStream_t s);
// T NHWC [N,H,W,C] (row stride ld, channel offset coff) -> float32 NCHW (test taps / API outputs).
int k_nhwc_to_nchw_f32(int dtype, const void* x, float* y, int N, int H, int W, int C, int ld, int coff, hipStream_t s);

// Multi-head self-attention, head_dim 32 (svtrv2.py:77-86, 199-213): qkv T [B*N][3*heads*32] -> out T [B*N][heads*32].
// 16-bit types: N <= 1024.  4-byte types: N <= 512 in one pass; longer sequences run as key chunks of <= 512 merged by a second kernel and
// need `scratch` = attention_scratch_bytes(..) bytes of device memory (0 for N <= 512).
int k_attention(int dtype, const void* qkv, void* out, int B, int N, int heads, hipStream_t s, void* scratch = nullptr);
size_t attention_scratch_bytes(int dtype, int B, int N, int heads);
// FRM vertical cross-attention with the precomputed query (svtrv2.py:236-243): kv T [B*H*W][2D] (token = h*W + w per image),
// vq f32 [D] -> out T [B*W][D].
int k_frm_vertical(int dtype, const void* kv, const float* vq, void* out, int B, int H, int W, int D, hipStream_t s);

// Adaptive scale fusion (neck.py:57-79) fused: bilinear(align_corners=True) taps of p3..p5 at p2 resolution, 1x1 conv over the
// virtual 1024-ch concat -> 4 scores -> softmax -> blend.  p_i: T NHWC [N,H>>i,W>>i,256]; w f32 [4][1024]; b f32 [4].
int k_asf(int dtype, const void* p2, const void* p3, const void* p4, const void* p5, const float* w, const float* b, float* scratch, void* out,
          int N, int H, int W, hipStream_t s);
size_t asf_scratch_bytes(int N, int H, int W);  // fp32 scratch for the coarse-level score maps
// DB head tail (head.py:16,28-48): y T NHWC [N,H2,W2,128] (ch 0..63 binarise branch, 64..127 threshold branch, after deconv1+BN+ReLU)
// -> ConvTranspose2d(64,1,2,2) per branch, sigmoid, step function.  w2 f32 [2][64][4], b2 f32 [2].  Outputs f32 [N,1,2*H2,2*W2]; any but
// `binary` may be null.
int k_db_tail(int dtype, const void* y, const float* w2, const float* b2, float k, float* binary, float* thresh, float* thresh_binary,
              float* bin_logits, float* thresh_logits, int N, int H2, int W2, hipStream_t s);

// DB maps from the two logit maps (head.py:28-40); thresh / thresh_binary may be null.  n = elements per map (multiple of 4).
int k_db_maps(const float* bin_logits, const float* thresh_logits, float k, float* binary, float* thresh, float* thresh_binary, size_t n,
              hipStream_t s);

// logits f32 [B*T][C] (row = b*T + t) -> log_softmax into log_probs [T][B][C] (nullable) and argmax [B][T] (nullable; first max wins).
int k_ctc_logsoftmax_argmax(const float* logits, int ld, float* log_probs, int32_t* argmax_ids, int B, int T, int C, hipStream_t s);
// log_probs f32 [T][B][C] -> argmax [B][T]
int k_ctc_argmax_tbc(const float* log_probs, int32_t* argmax_ids, int B, int T, int C, hipStream_t s);
// argmax [B][T] -> collapsed ids [B][T] (-1 padded) + lens [B] (svtrv2.py:559-566)
int k_ctc_collapse(const int32_t* argmax_ids, int32_t* ids, int32_t* lens, int B, int T, int blank, hipStream_t s);

}  // namespace ocrvi
