/* mega_hip.h -- C ABI of libmega_hip.so: the MI355X (gfx950) native kernels behind MEGA's
 * per-key-frame inference hot path.
 *
 * Boundary rules
 *   - extern "C", plain device pointers + sizes, no torch / ATen types.
 *   - every entry point returns int: 0 = MEGA_OK, 1 = bad argument, 2 = launch failure,
 *     3 = workspace too small.  Nothing is allocated inside: the caller owns outputs and workspaces.
 *   - `stream` is a hipStream_t passed as void* (NULL = the legacy default stream).  All work is
 *     enqueued on that stream; no call synchronises the device or copies to the host.
 *   - dtype codes: MEGA_F32 = 0 (exact-f32 MFMA path, parity mode), MEGA_BF16 = 1, MEGA_F16 = 2 (IEEE half: the same
 *     kernels instantiated for _Float16 operands -- the bf16 MFMA rate and bytes with 11 significant bits instead of 8,
 *     values bounded by 65 504; accumulation is f32 in every mode).  Wherever an entry point takes a dtype code, MEGA_F16
 *     is accepted where MEGA_BF16 is, unless its comment says otherwise (the split-precision plane kernels are bf16 only).
 *   - activations are NHWC; GEMM weights are OHWI ([Cout][R][S][Cin]) so both operands are K-contiguous.
 *
 * Each prototype cites the reference interface it replaces (paths relative to the reference tree
 * Scalsol/mega.pytorch).
 */
#ifndef MEGA_HIP_H
#define MEGA_HIP_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MEGA_F32 0
#define MEGA_BF16 1
#define MEGA_F16 2

#define MEGA_OK 0
#define MEGA_ERR_ARG 1
#define MEGA_ERR_LAUNCH 2
#define MEGA_ERR_WS 3

/* Conv2d (+ FrozenBatchNorm2d scale/bias) (+ residual add) (+ ReLU), implicit GEMM on MFMA.
 * Replaces torch Conv2d->cuDNN + the unfused FrozenBatchNorm2d / relu_ / += of
 * mega_core/modeling/backbone/resnet.py:324-344 (Bottleneck.forward), :155-204 (ResNetHead),
 * mega_core/layers/batch_norm.py:19-31, modeling/rpn/rpn.py:99-106 (RPNHead), and every nn.Linear
 * of the box head (H = W = R = S = 1, N = rows): roi_box_feature_extractors.py:894,:907,:826,
 * roi_box_predictors.py:50-57.
 *   in  [N][H][W][Cin]      (in_dtype)        w [Cout][R][S][Cin] (in_dtype)
 *   out [N*Ho*Wo][ldo]      (out_dtype)       y = act(acc*scale[c] + bias[c] (+ residual[m][c]))
 *   relu: 0 = identity, 1 = ReLU, 2 = LeakyReLU(0.1) (FlowNetS)
 *   scale / bias: f32 [Cout] or NULL (1 / 0); residual [M][ldr] in_dtype or NULL; ldo/ldr <= 0 -> Cout.
 *   Cin must be a multiple of 64 (bf16) / 32 (f32).  out_dtype may be MEGA_F32 with bf16 inputs. */
int mega_conv2d_nhwc(const void* in, const void* w, const float* scale, const float* bias, const void* residual,
                     void* out, int N, int H, int W, int Cin, int Cout, int R, int S, int stride, int pad, int dil,
                     int relu, int ldo, int ldr, int in_dtype, int out_dtype, void* stream);
/* Which block tile (BM * 1000 + BN, i.e. which igemm_kernel<.., BM, BN> instantiation) mega_conv2d_nhwc launches for
 * a GEMM of M = N*Ho*Wo rows, Cout columns, K = R*S*Cin: lets a profiler attribute time to the kernel symbol
 * rocprofv3 reports.  No device work. */
int mega_conv2d_nhwc_tile(int M, int Cout, int K);
/* Same, for a given operand dtype (MEGA_F32 / MEGA_BF16): kind * 1000000 + BM * 1000 + BN with kind 0 =
 * igemm_kernel<.., BM, BN> (register-staged tiles) and kind 8 = igemm8_kernel (bf16 only: LDS-DMA staging, 8 waves,
 * BM x 256 tiles, BM = 256 or 192).  Every kernel accumulates an output element over K in the same order with the
 * same MFMA instruction, so the choice never changes a result bit. */
int mega_conv2d_nhwc_plan(int M, int Cout, int K, int in_dtype);
/* The shape-complete form: which kernel mega_conv2d_nhwc[_ws] really dispatches THIS layer to (same predicates as the
 * launch path): kind 2 = igemm2_kernel (round 6: 128 x 256 tiles at two blocks per CU -- the streaming class, 1x1 with K <= 512,
 * by default), kind 4 = igemm4_kernel matrix class (round 6: the igemm8 tiles on 4 waves of 512 registers, 128 x 128 outputs per
 * wave), 3 = igemm4 streaming class (only with MEGA_IGEMM4=2), kind 6 = conv3x3_c64_kernel (layer1's 3x3 64 -> 64 conv: 3x3, stride 1, pad 1, bf16 out, no residual,
 * enough tiles), 7 = igemm8 streaming class (1x1, K <= 512), 8 = igemm8 matrix class, 0 = igemm_kernel.  -1: bad shape. */
int mega_conv2d_nhwc_plan_ex(int N, int H, int W, int Cin, int Cout, int R, int S, int stride, int pad, int dil, int ldo,
                             int has_residual, int in_dtype, int out_dtype);

/* mega_conv2d_nhwc with a caller-owned workspace of mega_conv2d_nhwc_workspace_bytes(M, Cout, K) bytes (M = N*Ho*Wo,
 * K = R*S*Cin; 0 for most layers).  With it, layers with K >= 32768 (the box head's first FC, K = 100352) run
 * split-K: three K ranges per output tile write f32 partial sums into the workspace and a second kernel adds them in a
 * fixed order and applies scale / bias / residual / activation.  The split depends on K only, never on M, so a
 * row's result does not depend on the batch it is computed in.  mega_conv2d_nhwc (no workspace) never splits. */
size_t mega_conv2d_nhwc_workspace_bytes(int M, int Cout, int K);
int mega_conv2d_nhwc_ws(const void* in, const void* w, const float* scale, const float* bias, const void* residual,
                        void* out, int N, int H, int W, int Cin, int Cout, int R, int S, int stride, int pad, int dil,
                        int relu, int ldo, int ldr, int in_dtype, int out_dtype, void* ws, size_t ws_bytes, void* stream);

/* ResNet stem: 7x7 stride-2 pad-3 conv (3->64) + FrozenBN + ReLU  (resnet.py:347-366 BaseStem.forward,
 * without the max-pool).  in: NCHW f32 [N][3][H][W]; w_tap64: [147][64] f32 with tap = (c*7+r)*7+s;
 * out: NHWC [N][Ho][Wo][64], Ho = (H-1)/2+1. */
int mega_stem_conv_bn_relu(const float* in, const float* w_tap64, const float* scale, const float* bias, void* out,
                           int N, int H, int W, int out_dtype, void* stream);
/* The same stem on the matrix cores (bf16 path): w_n176_bf16 = bf16 [64][176], row n = output channel, column
 * k = (c*7+r)*8+s for the 7 taps s of kernel row (c, r), zero for s = 7 and for the 8 pad columns; out: NHWC bf16.
 * Image pixels and weights are rounded to bf16, accumulation is f32. */
int mega_stem_conv_bn_relu_bf16(const float* in, const void* w_n176_bf16, const float* scale, const float* bias,
                                void* out, int N, int H, int W, void* stream);
/* The bf16 stem fed by the uint8 frames themselves, [N][H][W][3] RGB: the preprocessing of mega_preprocess_frames
 * (data/transforms/transforms.py:83-129: ToTensor, to_bgr255, Normalize with std 1) is applied on the patch load --
 * (float)byte - mean[c] for output channel c -- so the f32 image never exists in HBM.  Same bits as
 * mega_preprocess_frames followed by mega_stem_conv_bn_relu_bf16. */
int mega_stem_conv_bn_relu_bf16_u8(const void* frames_u8, const void* w_n176_bf16, const float* scale, const float* bias,
                                   void* out, int N, int H, int W, float mean0, float mean1, float mean2, int to_bgr,
                                   void* stream);

/* The whole stem of resnet.py:355-366 in one kernel (bf16): conv 7x7/2 + FrozenBN + ReLU + F.max_pool2d(3, 2, 1); `in` =
 * the preprocessed f32 NCHW image (u8 = 0) or the uint8 RGB frames [N][H][W][3] with the preprocessing on the patch load
 * (u8 = 1; mean / to_bgr as above); out NHWC bf16 [N][Hp][Wp][64], Hp = ((H - 1) / 2 + 1 - 1) / 2 + 1.  The stem's own
 * 64-channel map is never written.  Same bits as mega_stem_conv_bn_relu_bf16[_u8] + mega_maxpool3x3s2_nhwc. */
int mega_stem_pool_bf16(const void* in, int u8, const void* w_n176_bf16, const float* scale, const float* bias, void* out,
                        int N, int H, int W, float mean0, float mean1, float mean2, int to_bgr, void* stream);

/* mega_stem_pool_bf16 for either 16-bit type: dtype = MEGA_BF16 or MEGA_F16 is the type of w_n176 and of `out`. */
int mega_stem_pool_dt(const void* in, int u8, const void* w_n176, const float* scale, const float* bias, void* out, int N,
                      int H, int W, float mean0, float mean1, float mean2, int to_bgr, int dtype, void* stream);

/* F.max_pool2d(kernel 3, stride 2, padding 1) on NHWC (resnet.py:365). */
int mega_maxpool3x3s2_nhwc(const void* in, void* out, int N, int H, int W, int C, int dtype, void* stream);

/* nn.AvgPool2d(2, stride 2, ceil_mode=True) on NHWC (FlowNetS, mega_core/modeling/backbone/flownet.py:52,:56,:112). */
int mega_avgpool2x2_ceil_nhwc(const void* in, void* out, int N, int H, int W, int C, int dtype, void* stream);

/* ROIAlign forward.  Replaces mega_core._C.roi_align_forward (csrc/ROIAlign.h:11-25,
 * cuda/ROIAlign_cuda.cu:64-122, cpu/ROIAlign_cpu.cpp:113-219).  rois [K][5] f32 = (batch, x1, y1, x2, y2).
 *   in_nhwc : feat is [B][H][W][C] (1) or the reference's [B][C][H][W] (0)
 *   out_nhwc: out is bin-major [K][ph*pw][C] (1) or the reference's [K][C][ph][pw] (0) */
int mega_roi_align_fwd(const void* feat, const float* rois, void* out, int K, int C, int H, int W,
                       float spatial_scale, int pooled_h, int pooled_w, int sampling_ratio, int in_nhwc,
                       int out_nhwc, int dtype, int out_dtype, void* stream);

/* Greedy NMS.  Replaces mega_core._C.nms (csrc/nms.h:10-28): unsorted dets [n][4] + scores [n] ->
 * kept ORIGINAL indices, ascending, int64 (cuda/nms.cu:127-130, cpu/nms_cpu.cpp:64); *keep_cnt is a device int.
 * strict_gt = 1: suppress when IoU > thr (cuda/nms.cu:60); 0: IoU >= thr (cpu/nms_cpu.cpp:60).  n <= 8192. */
size_t mega_nms_full_workspace_bytes(int n);
int mega_nms(const float* dets, const float* scores, int n, float thr, int strict_gt, long long* keep_out,
             int* keep_cnt, void* ws, size_t ws_bytes, void* stream);

/* Batched NMS over P independent, already score-sorted problems (device-side counts, no host sync).
 *   boxes [P][nmax][4]; counts [P]; valid [P][nmax] u8 or NULL; order [P][nmax] or NULL
 *   keep_pos [P][max_keep] positions in sorted order (ascending); keep_cnt [P]
 *   flags [P][nmax] (pre-zeroed) or NULL: flags[order[pos]] = 1 for every kept box. */
size_t mega_nms_workspace_bytes(int P, int nmax);
int mega_nms_sorted(const float* boxes, const int* counts, const unsigned char* valid, const int* order, int P,
                    int nmax, float thr, int strict_gt, int max_keep, int* keep_pos, int* keep_cnt,
                    unsigned char* flags, void* ws, size_t ws_bytes, void* stream);

/* RPN proposal selection for B frames in one call.  Replaces RPNPostProcessor.forward_for_single_feature_map
 * (modeling/rpn/inference.py:76-123) incl. AnchorGenerator.grid_anchors (rpn/anchor_generator.py:73-95),
 * BoxCoder.decode (box_coder.py:52-95), clip_to_image (structures/bounding_box.py:214-219),
 * remove_small_boxes + boxlist_nms (structures/boxlist_ops.py:9-50).
 *   rpn_out [B][Hf*Wf][ldc] f32: channel a = objectness logit of anchor a, channel A + 4a + j = delta j
 *   outputs: proposals [B][post_nms_top_n][4], prop_scores [B][post_nms_top_n], prop_cnt [B] (device) */
size_t mega_rpn_select_workspace_bytes(int B, int pre_nms_top_n);
int mega_rpn_select(const float* rpn_out, const float* cell_anchors, int B, int Hf, int Wf, int A, int ldc,
                    int anchor_stride, int pre_nms_top_n, int post_nms_top_n, float nms_thresh, int strict_gt,
                    float min_size, float im_w, float im_h, float* proposals, float* prop_scores, int* prop_cnt,
                    void* ws, size_t ws_bytes, void* stream);
/* The same with prop_index [B][post_nms_top_n] (NULL allowed): the flat anchor index (y * Wf + x) * A + a of every
 * kept proposal (-1 in unused rows) = the reference's index into permute_and_flatten's (N, H*W*A) order
 * (rpn/utils.py:10-14, rpn/inference.py:93-104) after NMS: what "bit-exact proposal indices" is checked on. */
int mega_rpn_select_idx(const float* rpn_out, const float* cell_anchors, int B, int Hf, int Wf, int A, int ldc,
                        int anchor_stride, int pre_nms_top_n, int post_nms_top_n, float nms_thresh, int strict_gt,
                        float min_size, float im_w, float im_h, float* proposals, float* prop_scores, int* prop_cnt,
                        int* prop_index, void* ws, size_t ws_bytes, void* stream);

/* Box-head post-processor for one image.  Replaces PostProcessor.forward / filter_results
 * (modeling/roi_heads/box_head/inference.py:45-149): softmax, per-class decode with (wx,wy,ww,wh), clip,
 * score > score_thresh, per-class NMS, detections_per_img k-th value cut (>= kth, ties kept).
 *   logits [R][NC], deltas [R][NC*4], props [R][4], nprop device int or NULL (= R); R <= 1024
 *   outputs have capacity (NC-1)*R rows; out_cnt is a device int; probs_out [R][NC] optional. */
size_t mega_postprocess_workspace_bytes(int R, int NC);
int mega_postprocess(const float* logits, const float* deltas, const float* props, const int* nprop, int R, int NC,
                     float wx, float wy, float ww, float wh, float im_w, float im_h, float score_thresh,
                     float nms_thresh, int strict_gt, int max_det, float* out_boxes, float* out_scores,
                     long long* out_labels, int* out_cnt, float* probs_out, void* ws, size_t ws_bytes,
                     void* stream);

/* The same post-processor for B images of R rows each in one launch chain (the key frames of a step-batch: the
 * reference calls PostProcessor.forward once per key frame, tools/test_net.py -> engine/inference.py:23-46).
 *   logits [B][R][NC], deltas [B][R][NC*4], props [B][R][4], nprop device int[B] or NULL; outputs [B][(NC-1)*R]...,
 *   out_cnt device int[B].  Image b's results have the bits of mega_postprocess on image b alone (same kernels,
 *   the image is a grid dimension). */
size_t mega_postprocess_batched_workspace_bytes(int B, int R, int NC);
int mega_postprocess_batched(const float* logits, const float* deltas, const float* props, const int* nprop, int B,
                             int R, int NC, float wx, float wy, float ww, float wh, float im_w, float im_h,
                             float score_thresh, float nms_thresh, int strict_gt, int max_det, float* out_boxes,
                             float* out_scores, long long* out_labels, int* out_cnt, float* probs_out, void* ws,
                             size_t ws_bytes, void* stream);

/* Position-embedding logits of the relation module: log(relu(Wg . pe(q,k) + bg) + 1e-6).
 * Replaces extract_position_matrix + extract_position_embedding + the Wgs 1x1 conv + relu + log
 * (roi_box_feature_extractors.py:147-176,:126-144,:593-597,:630) without materialising the
 * [64][Nq][Nk] embedding.  wg_t [64][16] (Wg transposed), bg [16], dim_mat [8] = 1000^(i/8);
 * out [16][Nq][ldp] f32, ldp >= Nk (attention wants ldp % 32 == 0).  precise != 0: libm-accurate sincosf
 * (f32 parity mode); 0: two-constant range reduction + hardware sin/cos (abs error ~1e-6). */
int mega_position_logits(const float* rois_q, const float* rois_k, const float* wg_t, const float* bg,
                         const float* dim_mat, float* out, int Nq, int Nk, int ldp, int precise, void* stream);

/* Relation-attention core (roi_box_feature_extractors.py:599-646): per head h (64-wide)
 *   out[q][h*64+j] = resid[q][h*64+j] + bias_v[h*64+j]
 *                    + sum_k softmax_k( scale * q[q][h,:] . k[k][h,:] + pos[h][q][k] ) * vt[h*64+j][k]
 * q already contains the learned u vector (folded into the Wq bias), vt is V projected by Wv and stored
 * key-contiguous ([groups*64][ldv], pad columns zero).  pos / resid / bias_v may be NULL. */
int mega_relation_attention(const void* q, int ldq, const void* k, int ldk, const void* vt, int ldv,
                            const float* pos, int ldp, const void* resid, int ldr, const float* bias_v, void* out,
                            int ldo, int Nq, int Nk, int groups, float scale, int dtype, void* ws, size_t ws_bytes,
                            void* stream);
/* The key range of one call is split over mega_relation_attention_splits() block groups (flash-decoding style:
 * partial max / sum / output per split, merged by a second kernel) when ws holds at least
 * mega_relation_attention_workspace_bytes(); with ws == NULL the call runs unsplit. */
/* bf16-mode pair of the two calls above with the logits kept in bf16 and in the attention kernel's own tile order:
 * out_bf16 / pos_tiled_bf16 = [16][ceil(Nk/32)][Nq][32] bf16 (16 * ceil(Nk/32) * Nq * 64 bytes, 16-byte aligned), the
 * 32 keys of a tile stored as (h2, rq, e) with key = 8 rq + 4 h2 + e.  Half the bytes of the f32 form, written by the
 * matrix-core position kernel and read by the attention kernel in fully coalesced 2 KiB blocks one tile pair ahead. */
int mega_position_logits_tiled(const float* rois_q, const float* rois_k, const float* wg_t, const float* bg,
                               const float* dim_mat, void* out_bf16, int Nq, int Nk, void* stream);
int mega_relation_attention_tiled_pos(const void* q, int ldq, const void* k, int ldk, const void* vt, int ldv,
                                      const void* pos_tiled_bf16, const void* resid, int ldr, const float* bias_v,
                                      void* out, int ldo, int Nq, int Nk, int groups, float scale, void* ws,
                                      size_t ws_bytes, void* stream);
int mega_relation_attention_splits(int Nq, int Nk, int groups);
size_t mega_relation_attention_workspace_bytes(int Nq, int Nk, int groups);

/* Test-time frame transform on device: uint8 HWC RGB [N][H][W][3] -> f32 CHW [N][3][H][W],
 * ToTensor -> (BGR*255 if to_bgr) -> minus mean, std 1.  Replaces the CPU chain
 * mega_core/data/transforms/transforms.py:83-129 for frames already at the target size. */
int mega_preprocess_frames(const unsigned char* in, float* out, int N, int H, int W, float mean0, float mean1,
                           float mean2, int to_bgr, void* stream);

/* Test-time Resize on device: PIL / torchvision F.resize(BILINEAR) of uint8 HWC frames,
 * [N][Hi][Wi][3] -> [N][Ho][Wo][3], bit-identical to Pillow's 8-bit resampler (libImaging/Resample.c): horizontal
 * pass to uint8 (tmp [N][Hi][Wo][3], needed only when both dimensions change), then vertical.  bounds_* = int
 * [out][2] (first tap, tap count), coef_* = int [out][ksize] fixed-point (<<22) taps, both prepared on the host
 * (mega.pytorch_amd.feed.pil_bilinear_coeffs).  A pass whose size is unchanged is skipped, as in PIL.  Replaces
 * mega_core/data/transforms/transforms.py:27-66 (Resize.__call__ -> F.resize). */
int mega_resize_bilinear_u8(const unsigned char* in, unsigned char* out, unsigned char* tmp, int N, int Hi, int Wi,
                            int Ho, int Wo, const int* bounds_h, const int* coef_h, int ksize_h, const int* bounds_v,
                            const int* coef_v, int ksize_v, void* stream);

/* FGFA flow-guided aggregation (BASELINE configs[4]): bilinear warp of T frames' [features | embeddings] by their
 * flow fields (F.grid_sample bilinear / border / align_corners=False), cosine-similarity weights of the embeddings
 * against the key frame's, softmax over frames, weighted feature sum -- one fused kernel.  Replaces
 * GeneralizedRCNNFGFA.get_grid / resample / compute_weight + the softmax / sum of _forward_test
 * (mega_core/modeling/detector/generalized_rcnn_fgfa.py:45-76,:201-211).
 *   feats [T][H][W][Cf+Ce] NHWC, flow [T][2][H][W] f32, out [H][W][Cf], weights_out [T][H][W] f32 or NULL. */
int mega_fgfa_warp_aggregate(const void* feats, const float* flow, void* out, float* weights_out, int T, int H,
                             int W, int Cf, int Ce, int key, int dtype, void* stream);
/* The same with the T maps and flow fields held in a ring of T slots (the window's deques of
 * generalized_rcnn_fgfa.py:166-176 without the per-step torch.cat of 21 x 3072-channel maps): order = 1 + T device ints,
 * order[0] = slot of the key frame, order[1 + t] = slot of window position t.  Frames are visited in window order, so the
 * result has the bits of the contiguous call; one hipGraph serves every step (only the ints change). */
int mega_fgfa_warp_aggregate_ring(const void* feats, const float* flow, void* out, float* weights_out, int T, int H,
                                  int W, int Cf, int Ce, const int* order, int dtype, void* stream);

/* FlowNetS input in one kernel (round 6): the T image pairs cat([cur, frame_t]) of generalized_rcnn_fgfa.py:196-198 (the
 * "/ 255" lives in the first conv's weights), average-pooled 2 x 2 with ceil_mode (backbone/flownet.py:52,:56), written as
 * the operand of flow_conv1 with its seven horizontal taps gathered into the channel dimension:
 *   out[t][3 + h][w][s * 8 + c] = pool(pair)[t][h][w - 3 + s][c]  (c < 6; zeros for c = 6, 7, channels 56..63, taps outside
 *   the row and the three rows above / below), dtype MEGA_BF16 / MEGA_F16, [T][ceil(H/2) + 6][ceil(W/2)][64]:
 * flow_conv1 (7 x 7, stride 2, pad 3, 6 -> 64) then IS a 7 x 1 conv, stride 2, pad 0, over 64 channels (K = 448 instead of
 * the 49 x 64 of a channel-padded operand).  ring: f32 [T][3][H][W]; cur: the key frame(s), or NULL = ring slot order[0]. */
int mega_fgfa_pair_taps(const float* ring, long long ring_stride, const float* cur, long long cur_stride, const int* order,
                        void* out, int T, int H, int W, int dtype, void* stream);

/* DFF feature propagation: out[H][W][C] = bilinear warp (same grid convention as above) of the
 * key frame's NHWC feature map by flow [2][H][W], times the per-element scale map [H][W][C].  Replaces
 * GeneralizedRCNNDFF.get_grid / resample and the scale multiply (detector/generalized_rcnn_dff.py:41-60,:132-135). */
int mega_dff_warp_scale(const void* feats, const float* flow, const void* scale, void* out, int H, int W, int C,
                        int dtype, void* stream);

/* Pool / key-set assembly of the batched relation aggregation: n block copies (rows x row_bytes bytes, independent
 * source / destination row strides) in one launch -- replaces the reference's per-key-frame torch.cat of its pools
 * (roi_box_feature_extractors.py:676,:687-688,:812-814; generalized_rcnn_mega.py:213-216).  segs = array of
 *   struct { const void* src; void* dst; long long src_stride, dst_stride; int rows, row_bytes; }
 * Destinations must not overlap; addresses / strides / row lengths 2-byte aligned at least.  Bit-exact data movement. */
int mega_copy_segments(const void* segs, int n, void* stream);

/* dst[i] = bf16(src[i]) (round to nearest even), n contiguous elements, both pointers 16-byte aligned: the rounded copy
 * of the head's f32 activation stream that the bf16 Wq / Wk / Wv projections read
 * (roi_box_feature_extractors.py:584-597 run in one dtype; this is the mixed-precision seam of the bf16 mode). */
int mega_cast_f32_to_bf16(const float* src, void* dst_bf16, size_t n, void* stream);

/* The identity bottleneck of the backbone's full-resolution stage -- layer1 blocks 1, 2 of ResNet-50/101-C4
 * (backbone/resnet.py:324-344: 256 -> 64 (1x1) -> 64 (3x3, pad 1) -> 256 (1x1) channels, stride 1, FrozenBN as f32
 * scale / bias vectors (layers/batch_norm.py:19-31), residual = the block's input) -- in ONE persistent kernel:
 *   out = relu(bn3(conv3(relu(bn2(conv2(relu(bn1(conv1 x))))))) + x),   x, out: NHWC bf16 [N][H][W][256],
 * w1 [64][256], w2 [64][3][3][64], w3 [256][64] bf16 (OHWI).  The 64-channel intermediates never leave the CU.  Same MFMA,
 * K order, roundings and epilogue arithmetic as the three mega_conv2d_nhwc launches it replaces: bit-identical. */
int mega_bottleneck64_fwd(const void* x, const void* w1, const float* s1, const float* b1, const void* w2, const float* s2,
                          const float* b2, const void* w3, const float* s3, const float* b3, void* out, int N, int H, int W,
                          void* stream);

/* The stage's first block with the 1x1 downsample branch on the residual (layer1 block 0: 64 -> 64 -> 64 (3x3) -> 256,
 * backbone/resnet.py:266-276,:324-344), one persistent kernel:
 *   out = relu(bn3(conv3(relu(bn2(conv2(relu(bn1(conv1 x))))))) + bnd(convd(x))),  x NHWC bf16 [N][H][W][64], out [N][H][W][256];
 * w1 [64][64], w2 [64][3][3][64], w3 / wd [256][64].  The identity branch is computed from the x patch already in LDS and
 * rounded to bf16 before the add (as the separate launch would): bit-identical to the four mega_conv2d_nhwc launches. */
int mega_bottleneck64_ds_fwd(const void* x, const void* w1, const float* s1, const float* b1, const void* w2, const float* s2,
                             const float* b2, const void* w3, const float* s3, const float* b3, const void* wd, const float* sd,
                             const float* bd, void* out, int N, int H, int W, void* stream);

/* The two fused bottlenecks above for either 16-bit type (dtype = MEGA_BF16 / MEGA_F16: the type of x, every w and out). */
int mega_bottleneck64_fwd_dt(const void* x, const void* w1, const float* s1, const float* b1, const void* w2, const float* s2,
                             const float* b2, const void* w3, const float* s3, const float* b3, void* out, int N, int H, int W,
                             int dtype, void* stream);
int mega_bottleneck64_ds_fwd_dt(const void* x, const void* w1, const float* s1, const float* b1, const void* w2, const float* s2,
                                const float* b2, const void* w3, const float* s3, const float* b3, const void* wd, const float* sd,
                                const float* bd, void* out, int N, int H, int W, int dtype, void* stream);

/* dst [rows][3K] bf16 = [hi | lo | hi] of src [rows][K] f32 (hi = bf16(v), lo = bf16(v - hi); K % 8 == 0): the A operand
 * of a split-precision GEMM on the bf16 matrix cores against weight rows [Wh | Wh | Wl] -- v.W to ~2^-16.  Used for the
 * stage FCs of the aggregation head (roi_box_feature_extractors.py:826-827) when the activation stream is f32. */
int mega_split_f32_to_bf16x3(const float* src, void* dst_bf16, int rows, int K, void* stream);

/* ---- split-precision activation PLANES (round 5): the parity mode of the frame stage at bf16 matrix-core rates, and the
 * wide residual stream of the bf16 mode.  The reference computes these layers in fp32 (mega_core/config/defaults.py:541;
 * backbone/resnet.py:324-344 `out += identity`, rpn/rpn.py:99-106, roi_box_feature_extractors.py:894,:907).  An f32
 * activation x [M][C] lives in HBM as bf16 [M][2C] = [hi | lo], hi = bf16(x), lo = bf16(x - hi): x = hi + lo to ~2^-17. */

/* dst [rows][2K] bf16 = [hi | lo] of src [rows][K] f32 (K % 8 == 0, both 16-byte aligned). */
int mega_split_f32_to_planes(const float* src, void* dst_bf16, int rows, int K, void* stream);

/* mega_roi_align_fwd on f32 NHWC features with the pooled rows leaving as planes: out bf16 [K][2 PH PW C] = [hi | lo] of the
 * f32 row [PH PW C] (same term order as the f32 kernel, ROIAlign_cuda.cu:64-122).  C % 4 == 0. */
int mega_roi_align_fwd_planes(const float* feat, const float* rois, void* out_planes, int K, int C, int H, int W,
                              float spatial_scale, int pooled_h, int pooled_w, int sampling_ratio, void* stream);

/* conv + FrozenBN (+ split residual) + activation on plane tensors, bf16 matrix cores, f32 accumulation (igemm8 SP kernels):
 *   in       bf16 [N][H][W][ldi]; the contraction runs over Cin channels per tap, SOURCE channel k < kwrap ? k : k - kwrap
 *            (kwrap = 0: no wrap).  Split precision: ldi = 2C, Cin = 3C, kwrap = 2C reads [hi | lo | hi]; with weights
 *            packed [Wh | Wh | Wl] per tap ([Cout][R][S][3C]) the contraction is x_hi.Wh + x_lo.Wh + x_hi.Wl = x.W to ~2^-16.
 *            bf16 compute on a wide residual stream: ldi = 2C, Cin = C, kwrap = 0, plain weights (only hi is read).
 *   residual split planes [M][ldr] (ldr >= 2 Cout; 0 = 2 Cout) or NULL: hi + lo is added in f32 before the activation
 *   out_mode 0: bf16 [M][ldo];  1: split planes [M][ldo] (ldo >= 2 Cout);  2: f32 [M][ldo]     (ldo 0 = the natural width)
 *   relu     0 none, 1 ReLU, 2 LeakyReLU(0.1)
 * Cin % 64 == 0, kwrap % 64 == 0, Cout % 8 == 0, every tensor below 2 GiB (MEGA_ERR_ARG otherwise).  ws / ws_bytes: the
 * split-K workspace of mega_conv2d_nhwc_workspace_bytes(M, Cout, R S Cin) (long-K layers: f32 output without residual). */
int mega_conv2d_nhwc_sp(const void* in, int ldi, int kwrap, const void* w, const float* scale, const float* bias,
                        const void* residual, int ldr, void* out, int ldo, int out_mode, int N, int H, int W, int Cin,
                        int Cout, int R, int S, int stride, int pad, int dil, int relu, void* ws, size_t ws_bytes,
                        void* stream);

/* The plane kernels above for either 16-bit type (dtype = MEGA_BF16 / MEGA_F16: the type of the [hi | lo] pairs and of the
 * weights).  MEGA_F16 carries the TWO-PASS form of the fp16 mode (conv_mode "h2"): ldi = 2C, Cin = 2C, kwrap = 0 reads the planes
 * as [hi | lo] against weights packed [W | W] per tap, W rounded to fp16 once -- x_hi.W + x_lo.W: exact activations (to ~2^-22)
 * against single-rounded weights, f32 accumulation, twice the matrix-core work of the one-pass fp16 mode. */
int mega_split_f32_to_planes_dt(const float* src, void* dst, int rows, int K, int dtype, void* stream);
int mega_roi_align_fwd_planes_dt(const float* feat, const float* rois, void* out_planes, int K, int C, int H, int W,
                                 float spatial_scale, int pooled_h, int pooled_w, int sampling_ratio, int dtype, void* stream);
int mega_conv2d_nhwc_sp_dt(const void* in, int ldi, int kwrap, const void* w, const float* scale, const float* bias,
                           const void* residual, int ldr, void* out, int ldo, int out_mode, int N, int H, int W, int Cin,
                           int Cout, int R, int S, int stride, int pad, int dil, int relu, int dtype, void* ws, size_t ws_bytes,
                           void* stream);

/* mega_copy_segments with an f32 -> bf16 conversion on the way (source blocks f32, destination blocks bf16; row_bytes =
 * SOURCE bytes per row, a multiple of 32; 16-byte aligned on both sides): a concatenation of f32 row blocks delivered as the
 * rounded copy the bf16 projections read (roi_box_feature_extractors.py:812-814 pools with an f32 activation stream). */
int mega_copy_cast_segments(const void* segs, int n, void* stream);

/* mega_cast_f32_to_bf16 / mega_copy_cast_segments with the destination type as a code (MEGA_BF16 / MEGA_F16): the rounded
 * copies of the head's f32 activation stream that its 16-bit projections read. */
int mega_cast_f32_to_half(const float* src, void* dst, size_t n, int dtype, void* stream);
int mega_copy_cast_segments_dt(const void* segs, int n, int dtype, void* stream);

/* hipGetErrorString of the last launch failure any entry point of this library reported (MEGA_ERR_LAUNCH). */
const char* mega_last_error_string(void);

/* Several independent relation-attention problems (the key frames of one engine step-batch at the same stage) in ONE
 * launch (+ one combine launch).  Every problem runs exactly the code and the key-range split of its own
 * mega_relation_attention / mega_relation_attention_tiled_pos call: identical bits.  n <= 16; all problems share groups,
 * scale, dtype and the kind of position term (none / f32 rows `pos` with ldp / tile-ordered bf16 `pos_tiled`). */
typedef struct {
  const void* q; const void* k; const void* vt; const float* pos; const void* pos_tiled; const void* resid;
  const float* bias_v; void* out; void* ws; size_t ws_bytes;
  int ldq, ldk, ldv, ldp, ldr, ldo, Nq, Nk;
  int io_f32;      /* != 0 with dtype bf16: resid and out are f32 rows (the head's f32 activation stream, cfg.HEAD_STREAM:
                    * x + attention is never rounded to bf16 between the stages of roi_box_feature_extractors.py:806-829) */
  int nk1;         /* keys 0 .. nk1-1 come from (k, vt), keys nk1 .. Nk-1 from (k2, vt2): the [local window ; memory] key set of a
                    * MEGA stage (roi_box_feature_extractors.py:676,:687-688,:812-814 concatenate it per key frame) read in
                    * place.  0 or Nk: one segment, k2 / vt2 ignored.  Same keys in the same order: same bits. */
  const void* k2;  /* [Nk - nk1][ldk] */
  const void* vt2; /* [groups*64][ldv2]; its columns may start at any element (2-byte) address */
  int ldv2;
  int reserved;
} mega_attn_desc;
int mega_relation_attention_batched(const void* descs /* mega_attn_desc[n], host memory */, int n, int groups,
                                    float scale, int dtype, void* stream);

/* mega_position_logits_tiled for n <= 16 (query boxes, key boxes) problems in one launch. */
typedef struct { const float* rois_q; const float* rois_k; void* out_bf16; int Nq, Nk; } mega_pos_desc;
int mega_position_logits_tiled_batched(const void* descs /* mega_pos_desc[n], host memory */, int n, const float* wg_t,
                                       const float* bg, const float* dim_mat, void* stream);

/* Round 6: the head on IEEE-half operands (cfg.HEAD_DTYPE "float16": Q K^T, P V and the position term on
 * v_mfma_f32_32x32x16_f16 / 16x16x32_f16 -- the bf16 rate and bytes, 11 significant bits instead of 8).  The three
 * tile-ordered-logit entry points above with the 16-bit type as an argument: dtype = MEGA_BF16 (the calls above) or
 * MEGA_F16; the logits / Q / K / V^T are that type.  mega_relation_attention and mega_relation_attention_batched take
 * MEGA_F16 through their own dtype argument.  Replaces the same reference code as the bf16 forms:
 * roi_box_feature_extractors.py:126-176 (position embedding), :567-646 (attention_module_multi_head). */
int mega_position_logits_tiled_dt(const float* rois_q, const float* rois_k, const float* wg_t, const float* bg,
                                  const float* dim_mat, void* out16, int Nq, int Nk, int dtype, void* stream);
int mega_position_logits_tiled_batched_dt(const void* descs /* mega_pos_desc[n], host memory */, int n, const float* wg_t,
                                          const float* bg, const float* dim_mat, int dtype, void* stream);
int mega_relation_attention_tiled_pos_dt(const void* q, int ldq, const void* k, int ldk, const void* vt, int ldv,
                                         const void* pos_tiled16, const void* resid, int ldr, const float* bias_v,
                                         void* out, int ldo, int Nq, int Nk, int groups, float scale, int dtype, void* ws,
                                         size_t ws_bytes, void* stream);

/* Round 6, FGFA / DFF (BASELINE configs[4]): FlowNetS's refinement levels, mega_core/modeling/backbone/flownet.py:40-52,:94-111.
 *
 * mega_conv2d_nhwc_ws with the split count chosen by the CALLER (small-M, long-K layers whose tiles would not fill the chip:
 * FlowNetS's coarse levels on 21 pairs, the FGFA box head's fc6 on 300 rows -- roi_box_feature_extractors.py:55-118).
 * ksplit >= 1 K ranges (clamped so that every range holds a K-tile), f32 partial sums in a workspace of
 * mega_conv2d_nhwc_ks_workspace_bytes(M, Cout, K, in_dtype, ksplit) bytes, added in a fixed order by a second kernel.
 * The result depends on ksplit (summation order), never on M. */
size_t mega_conv2d_nhwc_ks_workspace_bytes(int M, int Cout, int K, int in_dtype, int ksplit);
int mega_conv2d_nhwc_ks(const void* in, const void* w, const float* scale, const float* bias, const void* residual,
                        void* out, int N, int H, int W, int Cin, int Cout, int R, int S, int stride, int pad, int dil,
                        int relu, int ldo, int ldr, int in_dtype, int out_dtype, int ksplit, void* ws, size_t ws_bytes,
                        void* stream);
/* nn.ConvTranspose2d(Cin, C, 4, stride=2) + bias + activation + crop_like + its slice of the torch.cat (flownet.py:9-13,:40-52,
 * :94-111), as ONE sub-pixel GEMM: the four output phases (a, b) of a stride-2 transposed conv are a 2 x 2 / pad 1 convolution
 * with 4 C output columns (a, b, co); GEMM row (t, m, n), column (a, b, co) is written to
 *   out[t][2 m + a - crop][2 n + b - crop][coff + co]   of an NHWC tensor [N][out_H][out_W][ldo] (dropped outside it).
 * in [N][H][W][Cin]; w4 [4 C][2][2][Cin] with w4[(a*2 + b)*C + co][r][s][ci] = Wt[ci][co][a + 2 (1 - r)][b + 2 (1 - s)]
 * (Wt = the ConvTranspose2d weight [Cin][C][4][4]); bias4 f32 [4 C] (the bias once per phase) or NULL; relu as
 * mega_conv2d_nhwc; dtype = operand AND output type (MEGA_BF16 / MEGA_F16: Cin % 64 == 0; MEGA_F32: Cin % 32 == 0);
 * C a multiple of 16 (4 C = whole 64-column tiles), ldo, coff multiples of the 16-byte vector.  ksplit / ws as mega_conv2d_nhwc_ks with M = N (H+1) (W+1), Cout = 4 C,
 * K = 4 Cin.  A quarter of the matrix work of the zero-stuffed form, no zero-stuffing / crop / cat copies. */
int mega_conv2d_nhwc_subpixel(const void* in, const void* w4, const float* bias4, void* out, int N, int H, int W, int Cin,
                              int C, int relu, int out_H, int out_W, int crop, int ldo, int coff, int dtype, int ksplit,
                              void* ws, size_t ws_bytes, void* stream);
/* The rest of a refinement level's concatenation in one pass (flownet.py:94-111): out[..., 0:Cs] = skip,
 * out[..., Cs+C : Cs+C+2] = crop(ConvTranspose2d(2, 2, 4, stride=2)(flow)) computed in f32 from the f32 weights
 * w_up [2][2][4][4] / b_up [2], out[..., Cs+C+2 : ldo] = 0.  skip [N][H2][W2][Cs], flow [N][h][w][2], out [N][H2][W2][ldo] of
 * `dtype`; crop: rows / columns of the full (2h+2) x (2w+2) map dropped at the top / left. */
int mega_flow_level_assemble(const void* skip, const void* flow, const float* w_up, const float* b_up, void* out, int N,
                             int H2, int W2, int Cs, int C, int ldo, int h, int w, int crop, int dtype, void* stream);

/* FlowNetS flow prediction (flownet.py:40-52 Convolution1..5 = nn.Conv2d(Cin, 2, 3, padding=1)), second half: by linearity the
 * 3 x 3 conv with two output channels is a 1 x 1 conv with 18 columns z[p][(r*3 + s)*2 + c] = sum_ci x[p][ci] w[c][ci][r][s]
 * (mega_conv2d_nhwc, f32 output: one pass over x, K = Cin instead of 9 Cin on 64-wide tiles) followed by
 *   out[t][y][x][c] = (sum_{r,s} z[t][y+r-1][x+s-1][(r*3+s)*2 + c]) * scale + bias[c]     (f32, taps outside the map skipped).
 * z f32 [N][H][W][ldz], ldz >= 18 and even; bias f32 [2]; out [N][H][W][2] of out_dtype (MEGA_F32 / MEGA_BF16 / MEGA_F16). */
int mega_flow_pred_finish(const float* z, int ldz, const float* bias, float scale, void* out, int N, int H, int W,
                          int out_dtype, void* stream);

/* FlowNetS flow_conv1 (flownet.py:52,:56: Conv2d(6, 64, 7, stride 2) + LeakyReLU(0.1) on cat([key, frame_t]) / 255) per FRAME
 * instead of per pair: the conv is linear before its activation, so conv(pair) = conv_key(key frame) + conv_ref(frame_t); both
 * halves [A | B] of a frame are computed once, when it enters the window (mega_conv2d_nhwc with 128 output channels over that
 * frame's tap operand, f32 out), and a key frame's 21 pairs are  out[t][p][c] = leaky(A[key][p][c] + B[t][p][c] + bias[c]).
 * ab f32 [S][P][128]; bias f32 [64]; order (device int; NULL: use `key`): order[0] = slot of the key frame; out [T][P][64] of
 * dtype (MEGA_BF16 / MEGA_F16), pair t = (key frame, the frame in slot t).  nwin > 0: order holds G rows [key slot, slot of
 * window position 0 .. nwin-1], T = G nwin, pair g nwin + t = (key frame of row g, window position t): exactly the pairs of G
 * key frames, in window order. */
int mega_flow_conv1_combine(const float* ab, const float* bias, const int* order, int key, void* out, int T, long long P,
                            int nwin, int dtype, void* stream);
/* mega_fgfa_warp_aggregate_ring with the flow fields in WINDOW order: flow [T][2][H][W] = the T pairs (key frame, window
 * position t) and nothing else; key_pos = the key frame's window position (cfg KEY_FRAME_LOCATION).  What a FlowNetS pass over
 * the exact pairs of several key frames (mega_flow_conv1_combine with nwin > 0) hands to the warp. */
int mega_fgfa_warp_aggregate_ring_pos(const void* feats, const float* flow, void* out, float* weights_out, int T, int H,
                                      int W, int Cf, int Ce, const int* order, int key_pos, int dtype, void* stream);
/* ... for G key frames in one launch: order [G][1 + T], flow [G][T][2][H][W], out [G][H][W][Cf], weights_out [G][T][H][W] or
 * NULL; the feature ring `feats` is shared.  Same bits per key frame as G separate calls. */
int mega_fgfa_warp_aggregate_ring_pos_batched(const void* feats, const float* flow, void* out, float* weights_out, int T,
                                              int H, int W, int Cf, int Ce, const int* order, int key_pos, int G, int dtype,
                                              void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MEGA_HIP_H */
