/* libocrvi -- C ABI of the MI355X-native (gfx950) DBNet++ / SVTRv2 / CTC inference hot path.
 *
 * The reference (ZenHKD/ocr-vi-invoice) is pure Python and has no FFI; the hot path sits behind two
 * torch.nn.Module forward APIs.  Each entry point below names the reference interface it replaces.
 * Plain pointers and sizes only -- no torch types.  All device pointers are HIP device memory on the
 * handle's device.  Every *_forward only enqueues work on `stream` (a hipStream_t passed as void*): no
 * allocation, no host synchronisation, so calls are graph-capturable.  The caller owns inputs, outputs
 * and workspace; a handle owns its (repacked) weights only.  Handles are per-device and not thread-safe.
 *
 * Errors: every function returns 0 (OCRVI_OK) or a negative code and never aborts the process;
 * ocrvi_last_error() returns a thread-local message for the last failing call.
 */
#ifndef OCRVI_H
#define OCRVI_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OCRVI_OK 0
#define OCRVI_EINVAL (-1) /* bad shape / alignment / argument (reference: ATen shape errors, svtrv2.py:425 assert) */
#define OCRVI_EHIP (-2)   /* a HIP runtime call failed; message carries hipGetErrorString */
#define OCRVI_ENOMEM (-3) /* workspace too small */
#define OCRVI_EBLOB (-4)  /* weight blob malformed or a tensor is missing / has the wrong shape */
#define OCRVI_ERANGE (-5) /* OCRVI_F16X2 only: an intermediate activation left fp16's exponent range (|x| >= 65520); reference: none -- the
                             reference is fp32 end to end (src/pipeline/pipeline2.py:312-318), so this mode says when it cannot stand in for it */

/* Arithmetic type the MFMA kernels compute in (accumulation is always fp32).
 * OCRVI_F32   fp32 operands on v_mfma_f32_16x16x4_f32: bit for bit an fp32 fmaf chain (what the reference's CPU path computes in).
 * OCRVI_F16X2 fp32-equivalent operands on the 16-bit matrix pipe: every GEMM operand element is stored as two fp16 halves
 *             x = hi + lo (>= 22 significant bits, weights scaled by a power of two per layer) and every product is formed from the
 *             three partial products hi hi + hi lo + lo hi in fp32 accumulators (the lo lo term, <= 2^-22 of a product, is dropped;
 *             conv_gemm's 64- and 32-column tiles compute all four); same API, same fp32 inputs and outputs.
 *             Supported activation range: fp16's exponent.  An intermediate activation with |x| >= 65520 cannot be represented
 *             (its hi half would be infinite): every kernel that writes f16x2 elements raises the handle's overflow flag instead of
 *             passing it on silently, and ocrvi_det_status / ocrvi_rec_status report it (below).  Below |x| = 2^-3 the absolute error
 *             of an element is 2^-25 (its lo half is a subnormal fp16 number), so tensors whose rms is under about 2^-9 lose the
 *             fp32-equivalence (relative error 2^-25 / |x|): BatchNorm / LayerNorm keep the two models' activations at O(0.1 .. 10).
 * OCRVI_BF16 / OCRVI_F16  plain 16-bit operands (throughput modes). */
typedef enum { OCRVI_F32 = 0, OCRVI_BF16 = 1, OCRVI_F16 = 2, OCRVI_F16X2 = 3 } ocrvi_dtype;

const char* ocrvi_last_error(void);
/* ABI version of this header (bumped on any signature change). */
int ocrvi_abi_version(void);

/* ------------------------------------------------------------------------------------------------
 * Detection: replaces DBNetPP(backbone='resnet50', dcn=True).eval() -- model/det/dbnet.py:6-17
 * (ResNet-50 + DCNv2 backbone model/det/backbone.py:8-60, dcn.py:41-59; FPN+ASF neck neck.py:26-79;
 * DB head head.py:32-48).
 * ------------------------------------------------------------------------------------------------ */
typedef struct ocrvi_det ocrvi_det;

typedef struct {
    int32_t dtype;   /* ocrvi_dtype */
    float k;         /* DB step-function steepness, head.py:6,28-30 (reference default 50) */
    int32_t max_batch;  /* largest N a forward call will see (sizes nothing; validated only) */
    int32_t reserved[5];
} ocrvi_det_cfg;

/* `blob` = host bytes produced by ocr_vi_invoice_amd.weights.pack_blob(fold_det(state_dict)): BN already
 * folded (eval-mode running stats), standard OIHW fp32 tensors.  Replaces load_state_dict + .to(device) +
 * .eval() of load_detection_model (src/pipeline/pipeline2.py:43-54). */
int ocrvi_det_create(int device, const void* blob, size_t blob_bytes, const ocrvi_det_cfg* cfg, ocrvi_det** out);
void ocrvi_det_destroy(ocrvi_det* h);
/* Bytes of scratch device memory one forward of shape (N,3,H,W) needs. */
int ocrvi_det_workspace_bytes(const ocrvi_det* h, int N, int H, int W, size_t* bytes);
/* Replaces DBNetPP.forward (dbnet.py:13-17).  x: float32 NCHW [N,3,H,W], H and W multiples of 32
 * (pipeline2.py:33-40 guarantees it).  Outputs are float32 [N,1,H,W]; `binary` is required, the other four
 * dict entries of head.py:42-48 may be NULL (then not written; the arithmetic that produces them is still
 * the same two-branch head). */
int ocrvi_det_forward(ocrvi_det* h, const float* x, int N, int H, int W,
                      float* binary, float* thresh, float* thresh_binary, float* bin_logits, float* thresh_logits,
                      void* workspace, size_t workspace_bytes, void* stream);
/* Test hook: copies of intermediate features as float32 NCHW (c2..c5 backbone.py:56-60, fused neck.py:79).
 * Any pointer may be NULL.  Must follow a forward on the same workspace and stream. */
/* OCRVI_F16X2 handles: OCRVI_OK, or OCRVI_ERANGE when an f16x2 kernel on this handle's device has met a value fp16's exponent cannot
 * carry since the last ocrvi_range_reset (the flag is per device and sticky; *_forward copies it to the handle asynchronously on the
 * caller's stream, so call this AFTER synchronising that stream -- never a host sync inside *_forward).  Other dtypes: always OCRVI_OK. */
int ocrvi_det_status(const ocrvi_det* h);
int ocrvi_det_debug_features(ocrvi_det* h, int N, int H, int W, float* c2, float* c3, float* c4, float* c5,
                             float* fused, void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Recognition: replaces SVTRv2(variant).eval() -- model/rec2/svtrv2.py:410-569 (inference graph; the
 * SGM branch, svtrv2.py:252-385, is training-only and not built).
 * ------------------------------------------------------------------------------------------------ */
typedef struct ocrvi_rec ocrvi_rec;

typedef struct {
    int32_t dtype;        /* ocrvi_dtype */
    int32_t dims[3];      /* VARIANTS[..]['dims']       svtrv2.py:391-407 */
    int32_t num_blocks[3];/* VARIANTS[..]['num_blocks'] */
    int32_t num_local[3]; /* VARIANTS[..]['num_local']  (first num_local blocks of a stage are LocalMixing) */
    int32_t num_classes;  /* tokenizer.num_classes = 232 (tokenizer.py:21) */
    int32_t blank_id;     /* 0 (tokenizer.py:14) */
    int32_t reserved[4];
} ocrvi_rec_cfg;

/* `blob` = pack_blob(fold_rec(state_dict, variant)).  Replaces load_recognition_model
 * (src/pipeline/pipeline2.py:72-82). */
int ocrvi_rec_create(int device, const void* blob, size_t blob_bytes, const ocrvi_rec_cfg* cfg, ocrvi_rec** out);
void ocrvi_rec_destroy(ocrvi_rec* h);
int ocrvi_rec_workspace_bytes(const ocrvi_rec* h, int B, int H, int W, size_t* bytes);
/* Replaces SVTRv2.forward(x, targets=None) (svtrv2.py:503-536) fused with decode_probs' device half
 * (svtrv2.py:555-566).  x: float32 NCHW [B,3,H,W], H % 16 == 0 and W % 4 == 0, T = W/4.
 *   log_probs [T,B,num_classes] float32 (may be NULL)
 *   argmax_ids [B,T] int32: per-step argmax over classes, first index wins ties (may be NULL)
 *   ids [B,T] int32: greedy CTC path after collapsing repeats and dropping `blank_id`, padded with -1
 *   lens [B] int32: number of valid entries of each ids row
 * ids/lens may both be NULL.  Mapping ids -> text (which also drops pad id 1, tokenizer.py:73) is host work. */
int ocrvi_rec_forward(ocrvi_rec* h, const float* x, int B, int H, int W,
                      float* log_probs, int32_t* argmax_ids, int32_t* ids, int32_t* lens,
                      void* workspace, size_t workspace_bytes, void* stream);
/* Test hook: float32 copies of backbone_norm output [B, H/16*W/4, D] (svtrv2.py:500) and FRM output
 * [B, W/4, D] (svtrv2.py:247).  Either may be NULL.  Must follow a forward on the same workspace/stream. */
int ocrvi_rec_status(const ocrvi_rec* h);   /* as ocrvi_det_status */
int ocrvi_rec_debug_features(ocrvi_rec* h, int B, int H, int W, float* backbone_norm, float* frm,
                             void* workspace, size_t workspace_bytes, void* stream);

/* The per-device f16x2 range flag behind ocrvi_{det,rec}_status: reset (asynchronous on `stream`) / read (synchronises the device). */
int ocrvi_range_reset(int device, void* stream);
int ocrvi_range_flag(int device, int* raised);

/* Standalone greedy CTC decode of caller-supplied log-probs: replaces SVTRv2.decode_probs' tensor half
 * (svtrv2.py:555-566).  log_probs [T,B,C] float32 on device. */
int ocrvi_ctc_greedy(int device, const float* log_probs, int T, int B, int C, int blank_id,
                     int32_t* argmax_ids, int32_t* ids, int32_t* lens, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Pre-processing on the device (the callers either side of the two models on the e2e path).
 * ------------------------------------------------------------------------------------------------ */
/* Replaces the detection input normalisation of src/pipeline/pipeline2.py:312-314: images uint8 HWC
 * [N,H,W,3] (device) -> float32 NCHW [N,3,H,W] = ((v/255 as float32) - mean) / std evaluated in float64. */
int ocrvi_normalize_u8(int device, const uint8_t* images, int N, int H, int W, float* out, void* stream);
/* Replaces the cv2.resize call of resize_image_for_det (src/pipeline/pipeline2.py:33-40): uint8 HWC [src_h,src_w,3] ->
 * uint8 HWC [dst_h,dst_w,3], OpenCV's 8-bit INTER_LINEAR (fixed-point) arithmetic. */
int ocrvi_resize_u8(int device, const uint8_t* src, int src_h, int src_w, uint8_t* dst, int dst_h, int dst_w, void* stream);
/* Replaces crop_image (src/det/test.py:123-130, box already reduced to its clamped bounding rect) followed by
 * preprocess_for_recognition (src/pipeline/pipeline2.py:92-128) for a batch of crops.  images uint8 HWC
 * [n_img,H,W,3]; boxes int32 [B,5] = (image index, x, y, w, h) in pixels, inside the image; out float32
 * [B,3,out_h,out_w].  w<=0 or h<=0 yields the all-zero tensor of pipeline2.py:154-156. */
int ocrvi_crop_resize_normalize(int device, const uint8_t* images, int n_img, int H, int W, const int32_t* boxes, int B,
                                int out_h, int out_w, float* out, void* stream);

/* Replaces DBPostProcessor(thresh, box_thresh, max_candidates, unclip_ratio).__call__ with .min_area (src/det/test.py:46-106) on a HOST
 * probability map prob[H*W] (the reference also runs this stage on the CPU after `.cpu().numpy()`, pipeline2.py:320-321).
 * Output: the unclipped polygons as int32 (x, y) pairs in points[2*cap_points]; box i owns points box_offsets[i] .. box_offsets[i+1]
 * (box_offsets has cap_boxes+1 entries); scores[i] is its box_score_fast value; *n_boxes the number of boxes, in the reference's order. */
int ocrvi_db_postprocess(const float* prob, int H, int W, float thresh, float box_thresh, int max_candidates, float unclip_ratio,
                         float min_area, int32_t* points, int cap_points, int32_t* box_offsets, float* scores, int cap_boxes, int* n_boxes);

/* Replaces unclip()'s Clipper call for a given offset distance (src/det/test.py:37-43: PyclipperOffset().AddPath(box, JT_ROUND,
 * ET_CLOSEDPOLYGON); Execute(distance)[0]): pts int32 (x, y) pairs of a closed polygon -> the offset polygon after Clipper's closing
 * union (raw round-join offset path, then the outline of its positive-winding region), int32 pairs in out[2*cap_pts]; *n_out points
 * (0 when the input degenerates).  Host code; ocrvi_db_postprocess calls the same routine with distance = area * unclip_ratio / length. */
int ocrvi_unclip_polygon(const int32_t* pts, int n_pts, double distance, int32_t* out, int cap_pts, int* n_out);

/* Replaces the host middle of the per-image loop for a batch of pages (src/pipeline/pipeline2.py:320-343): post_processor(prob_map)
 * (src/det/test.py:55-106) on each of the n_pages host maps prob[n_pages][H][W] -> boxes divided by (scale_w, scale_h) with the int64
 * truncation of pipeline2.py:324-328 -> crop_image's clamped bounding rectangle in the orig_h x orig_w page (src/det/test.py:123-130).
 * rects[(page * cap_per_page + i) * 5] = (page_base + page, x, y, w, h) -- the layout ocrvi_crop_resize_normalize consumes --,
 * scores (may be NULL) likewise, counts[page] = boxes of that page.  Pages are independent and run on `threads` host threads. */
int ocrvi_db_boxes_batch(const float* prob, int n_pages, int H, int W, float thresh, float box_thresh, int max_candidates,
                         float unclip_ratio, float min_area, double scale_w, double scale_h, int orig_h, int orig_w, int page_base,
                         int32_t* rects, float* scores, int cap_per_page, int32_t* counts, int threads);

/* Device half of the same stage (the reference thresholds on the host, `pred[0] > self.thresh`, src/det/test.py:57,
 * after copying the whole map, pipeline2.py:320).  prob: DEVICE float32 [n_pages,H,W] (W % 32 == 0).  Outputs, all DEVICE buffers:
 *   mask_bits  uint32 [n_pages,H,W/32]   bit (x & 31) of word (x >> 5) = prob > thresh
 *   comps      int32  [n_pages,cap,8]    per 8-connected component: x0, y0, x1, y1 (inclusive box), pixel count, root (index y*W+x of
 *                                        its first pixel in raster order), sum of round(prob * 2^20) as uint64 in ints 6..7; row order
 *                                        is arbitrary (sort by root for a canonical order)
 *   counts     int32  [n_pages]          components found (may exceed cap: rows beyond cap are not written)
 *   offsets    int64  [n_pages,cap+1]    start, in floats, of each component's box inside the page's packed region; [min(count,cap)] = total
 *   packed     float32[n_pages,pack_cap] the probability values inside the component boxes, row-major per box, back to back (left
 *                                        unwritten for a page whose total exceeds pack_cap).  offsets and packed may both be NULL.
 * workspace: ocrvi_db_components_workspace_bytes(n_pages, H, W) bytes of device memory.  Asynchronous on `stream`. */
size_t ocrvi_db_components_workspace_bytes(int n_pages, int H, int W);
int ocrvi_db_components(int device, const float* prob, int n_pages, int H, int W, float thresh, uint32_t* mask_bits, int32_t* comps,
                        int32_t* counts, int cap, long long* offsets, float* packed, long long pack_cap, void* workspace, void* stream);
/* ocrvi_db_boxes_batch fed by HOST copies of ocrvi_db_components' outputs instead of the full maps: same results (the segmentation comes
 * from mask_bits, the map is rebuilt inside the component boxes only, which is all box_score_fast reads).  skipped[page] = 1 (and
 * counts[page] = 0) for a page whose table or boxes overflowed (count > cap or total > pack_cap): process that page from its full map. */
int ocrvi_db_boxes_batch_sparse(const uint32_t* mask_bits, const int32_t* comps, const int32_t* comp_counts, int cap, const long long* offsets,
                                const float* packed, long long pack_cap, int n_pages, int H, int W, float box_thresh, int max_candidates,
                                float unclip_ratio, float min_area, double scale_w, double scale_h, int orig_h, int orig_w, int page_base,
                                int32_t* rects, float* scores, int cap_per_page, int32_t* counts, int threads, int32_t* skipped);

/* ------------------------------------------------------------------------------------------------
 * Kernel-level test/bench hooks (same kernels the models launch; used by tests/ and bench.py for
 * per-kernel parity and roofline timing).
 * ------------------------------------------------------------------------------------------------ */
/* The 27-channel offset / mask conv of DeformableConv2d alone (dcn.py:42-46: self.offset_conv, then sigmoid on channels 18..26): x float32
 * NCHW [N,C,H,W] (device), weight float32 host [27,C,3,3], bias host [27]; pad 1, stride 1 or 2; out DEVICE float32 [N,Ho,Wo,32]
 * (0..17 offsets, 18..26 mask, 27..31 zero).  Test hook: allocates and synchronises internally. */
int ocrvi_test_offset_conv(int device, int dtype, const float* x, const float* weight_host, const float* bias_host, int N, int C, int H, int W,
                           int stride, float* out, int iters, float* avg_ms);
/* Modulated deformable 3x3 conv, pad 1, dil 1 (torchvision.ops.deform_conv2d as called at dcn.py:48-57) with
 * a fused per-channel bias (+ optional ReLU) epilogue.  x float32 NCHW [N,C,H,W]; offset [N,18,Ho,Wo];
 * mask [N,9,Ho,Wo] (already sigmoided); weight float32 host [Co,C,3,3]; bias float32 host [Co] or NULL;
 * out float32 NCHW [N,Co,Ho,Wo].  Allocates and synchronises internally (test hook, not graph-capturable). */
int ocrvi_test_deform_conv(int device, int dtype, const float* x, const float* offset, const float* mask,
                           const float* weight_host, const float* bias_host, int N, int C, int H, int W, int Co,
                           int stride, int relu, float* out, int iters, float* avg_ms);
/* Plain conv2d (groups, stride (sh,sw), square kernel 1 or 3, pad = k/2) + bias + activation
 * (0 none, 1 ReLU, 2 exact GELU).  Same conventions as above. */
int ocrvi_test_conv(int device, int dtype, const float* x, const float* weight_host, const float* bias_host,
                    int N, int C, int H, int W, int Co, int ksize, int sh, int sw, int groups, int act, float* out,
                    int iters, float* avg_ms);
/* Linear / 1x1 convolution as a plain GEMM with the fused epilogue of the hot path: out[M][N] = act(a[M][K] . weight[N][K]^T + bias
 * (+ res)) (res_post = 0) or act(...) + res (res_post = 1); a, res and out are float32 DEVICE buffers (converted to/from `dtype`
 * around the call; the residual is kept in fp32 when out_f32 is set, as the recogniser's residual stream is), weight/bias HOST.  Takes
 * the persistent LDS-DMA ring GEMM whenever the shape is eligible (svtrv2.py:51-61,80-86 Linears; resnet Bottleneck 1x1 convs). */
int ocrvi_test_gemm(int device, int dtype, const float* a, const float* weight_host, const float* bias_host, const float* res, int M,
                    int K, int N, int act, int res_post, int out_f32, float* out, int iters, float* avg_ms);
/* Multi-head self-attention on a packed qkv tensor [B, N, 3*heads*32] float32 (layout of
 * qkv.reshape(B,N,3,heads,32), svtrv2.py:80) -> out [B, N, heads*32] float32 (svtrv2.py:82-85). */
/* The detector's stem: conv 7x7 stride 2 pad 3 (3 -> 64 channels) + bias + ReLU + max-pool 3x3 stride 2 pad 1 (torchvision resnet50's
 * conv1 / bn1 / relu / maxpool with BN folded, backbone.py:34).  fused != 0: the one-kernel f16x2 form (dtype must be 3), else the two
 * kernels of the other compute types.  x float32 NCHW device [N,3,H,W] (H, W multiples of 4); weight float32 host [64,3,7,7]; bias float32
 * host [64]; out float32 NCHW device [N,64,H/4,W/4].  Test hook: allocates and synchronises internally. */
int ocrvi_test_stem_pool(int device, int dtype, const float* x, const float* weight_host, const float* bias_host, int N, int H, int W, int fused,
                         float* out, int iters, float* avg_ms);
int ocrvi_test_attention(int device, int dtype, const float* qkv, int B, int N, int heads, float* out, int iters,
                         float* avg_ms);

/* The fused MixingBlock MLP of the 16-bit modes (svtrv2.py:28-39,100): x [M][D] float32 DEVICE, updated in place to
 * x + fc2(gelu(fc1(LayerNorm(x; ln_g, ln_b)))); with want_xn also xn_out [M][D] float32 DEVICE = LayerNorm(x_new; next_g, next_b) rounded to
 * `dtype` (next_g == NULL: x_new rounded to `dtype`).  Weights and vectors are HOST float32 (fc1 [4D][D], fc2 [D][4D]).  D in {128, 256, 384}. */
int ocrvi_test_mlp(int device, int dtype, float* x, const float* ln_g_host, const float* ln_b_host, const float* w1_host, const float* b1_host,
                   const float* w2_host, const float* b2_host, const float* next_g_host, const float* next_b_host, int want_xn, int M, int D,
                   float* xn_out, int iters, float* avg_ms);
/* Host-only (no GPU): the OCRVI_F16X2 weight format.  src: n fp32 values of one layer (n % 4 == 0) -> dst: n 4-byte elements, per chunk of
 * 4 consecutive elements [hi0 hi1 hi2 hi3 | lo0 lo1 lo2 lo3] (fp16), hi + lo = src * 2^s with one power of two s per layer that puts the
 * largest magnitude into [2^13, 2^14); *wscale = 2^-s.  (What ocrvi_*_create does to every GEMM weight in that mode.) */
int ocrvi_test_pack_f16x2(const float* src, size_t n, void* dst, float* wscale);

/* Per-launch HIP-event profiler (process-global, off by default).  While enabled every MFMA / bandwidth kernel launch
 * of the graphs above is bracketed by two events recorded on its launch stream.  ocrvi_prof_report synchronises those
 * events and writes a JSON object {tag: {launches, ms, flops, bytes}} (algorithmic FLOPs / bytes per tag) into buf. */
int ocrvi_prof_enable(int on);
int ocrvi_prof_reset(void);
int ocrvi_prof_report(char* buf, size_t cap);

#ifdef __cplusplus
}
#endif
#endif /* OCRVI_H */
