/*
 * mxdet.h -- C-ABI of libmxdet_hip.so: the MI355X (gfx950) two-stage-detector hot path.
 *
 * Drop-in boundary. The reference (jiangzhengkai/mxdetection) declares the
 * plugin slots only by name -- /root/reference/README.md:24 (`mxdetection/ops`), README.md:15-19
 * (`core/{anchor,bbox,mask,loss}`), README.md:27-32 (`models/{backbones,rpn_heads,bbox_heads,
 * mask_heads,necks,roi_extractors}`) -- and delegates all arithmetic to MXNet 1.3.0 operators
 * (README.md:37). Each entry below names the slot it sits behind and the MXNet-1.3.0 operator whose
 * role it takes; INTEGRATION.md shows the `mx.operator.CustomOp` / ctypes stub a maintainer would add.
 *
 * Conventions (every entry):
 *   - raw device pointers + explicit sizes; no torch / MXNet types; the caller owns every buffer,
 *     workspace included; the library never allocates device memory, frees, or keeps a pointer.
 *   - asynchronous on `stream` (a hipStream_t passed as void*); re-entrant; no global mutable state
 *     except the thread-local error string.
 *   - returns 0 on success or a negative MXDET_E* code; never throws, aborts or prints.
 *   - boxes are fp32 (x1,y1,x2,y2) in the legacy "+1" pixel convention; rois are (batch,x1,y1,x2,y2).
 *   - activations are bf16 channels-last [N,H,W,C]; `bf16` buffers are passed as uint16_t*.
 *   - `accumulate` on backward entries mirrors MXNet's req: 0 = kWriteTo, 1 = kAddTo.
 */
#ifndef MXDET_H_
#define MXDET_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MXDET_OK 0
#define MXDET_EINVAL (-1)
#define MXDET_ESHAPE (-2)
#define MXDET_EWORKSPACE (-3)
#define MXDET_EHIP (-4)
#define MXDET_ERCCL (-5)

#define MXDET_DTYPE_F32 0
#define MXDET_DTYPE_BF16 1

typedef void* mxdet_stream_t; /* hipStream_t */

/* thread-local description of the last failure on this thread ("" if none) */
const char* mxdet_last_error(void);
/* library version / build arch string, e.g. "mxdet-hip 0.1 gfx950" */
const char* mxdet_version(void);

/* ------------------------------------------------------------------------------------------------
 * core/bbox  (README.md:17)  -- bbox_overlaps; MXNet role: Cython bbox_overlaps / contrib.box_iou
 * out[na,nb] = IoU(a_i, b_j), legacy +1 convention. */
int mxdet_box_iou(const float* boxes_a, int64_t na, const float* boxes_b, int64_t nb, float* out,
                  mxdet_stream_t stream);

/* core/anchor (README.md:16) -- dense anchor grid of one pyramid level.
 * out[(y*W+x)*A+a] = base[a] + (x*stride, y*stride, x*stride, y*stride). */
int mxdet_generate_anchors(const float* base_anchors, int32_t A, int32_t H, int32_t W, int32_t stride,
                           float* out, mxdet_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * ops (README.md:24) -- batched NMS over B independent score-sorted lists.
 * boxes[B,n_max,4] sorted by descending score; counts[B] valid entries per list; invalid[B,n_max]
 * (may be NULL) marks boxes to drop before NMS. A box suppresses a later one when IoU > thresh.
 * keep_idx[B,n_max] receives the kept positions in ascending order, num_keep[B] their number
 * (at most max_keep each). MXNet role: contrib.box_nms / Cython cpu_nms / gpu_nms. */
size_t mxdet_nms_batched_workspace_bytes(int32_t B, int32_t n_max);
int mxdet_nms_batched(const float* boxes, const int32_t* counts, const uint8_t* invalid, int32_t B,
                      int32_t n_max, float thresh, int32_t max_keep, int32_t* keep_idx,
                      int32_t* num_keep, void* workspace, size_t workspace_bytes,
                      mxdet_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * core/evaluation (README.md:20) + ops (README.md:24) -- test-time detection post-processing
 * (MXNet-lineage role: the host-side im_detect -> per-class threshold -> nms ->
 * max_per_image loop of py-faster-rcnn / mx-rcnn).
 * cls_logits [N*R, ld_cls] (columns 0..C-1, class 0 = background) and bbox_pred [N*R, ld_reg] (class-specific
 * deltas, 4 per class, normalised: delta = pred * stds + means) in `dtype`; rois [N*R,5] = (image, x1,y1,x2,y2), the
 * first num_rois[n] rows of image n valid; im_info [N,3] = (h, w, scale).
 * score = softmax(logits) (m = max, e_c = exp(x_c - m), s = sum in class order, e_c / s); box = decode + clip.
 * Per image and foreground class: keep score > score_thresh, sort by (score desc, roi asc), greedy NMS at
 * nms_thresh; then the max_per_image best over all classes by (score desc, roi asc, class asc).
 * dets [N, max_per_image, 6] = (x1, y1, x2, y2, score, class), padding rows zero with class -1; num_dets [N]. */
size_t mxdet_detection_postprocess_workspace_bytes(int32_t N, int32_t rois_per_image, int32_t num_classes);
int mxdet_detection_postprocess(const void* cls_logits, const void* bbox_pred, int32_t dtype, int32_t ld_cls,
                                int32_t ld_reg, const float* rois, const int32_t* num_rois, const float* im_info,
                                int32_t N, int32_t rois_per_image, int32_t num_classes, const float* means,
                                const float* stds, float score_thresh, float nms_thresh, int32_t max_per_image,
                                float* dets, int32_t* num_dets, void* workspace, size_t workspace_bytes,
                                mxdet_stream_t stream);


/* ------------------------------------------------------------------------------------------------
 * rpn_heads + ops (README.md:28, :24) -- pyramid proposal generation.
 * MXNet role: contrib.Proposal / MultiProposal, or the lineage's pyramid-proposal CustomOp.
 * Per image and level: top `pre_nms_top_n` anchors by objectness logit (ties: lower anchor index
 * first) -> decode deltas at the anchors, clip to the image -> drop boxes with w or h < min_size ->
 * NMS(thresh) -> at most post_nms_top_n per level; levels merged per image by (score desc, global
 * anchor index asc) and cut to post_nms_top_n.
 *
 * Level l has H[l] x W[l] cells, A anchors per cell, feature stride stride[l]; its logits and deltas
 * live in one tensor each, addressed with element strides so NCHW and NHWC producers both fit:
 *   score(n,y,x,a)   = cls[l][n*cls_sn[l] + y*cls_sy[l] + x*cls_sx[l] + a*cls_sa[l]]
 *   delta(n,y,x,a,k) = reg[l][n*reg_sn[l] + y*reg_sy[l] + x*reg_sx[l] + (a*4+k)*reg_sc[l]]
 * Outputs: rois[N,post_nms_top_n,5] (batch index, box; zero padded), roi_scores[N,post_nms_top_n],
 * roi_anchor[N,post_nms_top_n] (global anchor index, -1 padding), num_rois[N]. */
typedef struct {
  int32_t num_levels;       /* <= 8 */
  int32_t A;                /* anchors per cell */
  int32_t H[8], W[8], stride[8];
  const void* cls[8];
  const void* reg[8];
  int64_t cls_sn[8], cls_sy[8], cls_sx[8], cls_sa[8];
  int64_t reg_sn[8], reg_sy[8], reg_sx[8], reg_sc[8];
  int32_t dtype;            /* MXDET_DTYPE_F32 | MXDET_DTYPE_BF16 (both tensors) */
  int32_t classes;          /* 0 or 1: plain. C > 1 (dense one-stage heads): A counts (anchor, class) pairs, a = anchor*C + c;
                             * deltas and base anchors are looked up with a / C (base_anchors stays [A/C,4]) */
  const float* base_anchors[8]; /* device, [A,4] per level */
} mxdet_pyramid_t;

size_t mxdet_proposal_workspace_bytes(const mxdet_pyramid_t* p, int32_t N, int32_t pre_nms_top_n);
int mxdet_proposal(const mxdet_pyramid_t* p, int32_t N, const float* im_info /* [N,3] h,w,scale */,
                   int32_t pre_nms_top_n, int32_t post_nms_top_n, float nms_thresh, float min_size,
                   float* rois, float* roi_scores, int32_t* roi_anchor, int32_t* num_rois,
                   void* workspace, size_t workspace_bytes, mxdet_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * core/anchor (README.md:16) -- anchor <-> GT assignment and RPN target sampling.
 * anchors[A_total,4]; gt_boxes[N,G_max,5] (x1,y1,x2,y2,class; rows with class < 0 are padding).
 * label = 1 if max IoU >= fg_thresh or the anchor attains some GT's maximum IoU (> 0);
 *         0 if max IoU < bg_thresh; -1 otherwise; anchors reaching outside the image by more than
 * allowed_border are -1. Then subsample to `batch_size` per image (at most fg_fraction*batch_size
 * foreground) with Philox keys (mxdet_math.h: mxdet_sample_key, streams 0/1); batch_size <= 0 keeps
 * every label (RetinaNet). Outputs: labels[N,A_total] int32, matched_gt[N,A_total] int32 (argmax GT,
 * lowest index on ties), bbox_targets[N,A_total,4] (encoded for label==1, else 0),
 * max_iou[N,A_total] (may be NULL). step_dev (may be NULL): device word that overrides `step` when
 * the launch is replayed from a hipGraph (kernel arguments are frozen at capture). */
size_t mxdet_anchor_target_workspace_bytes(int32_t N, int64_t A_total, int32_t G_max);
int mxdet_anchor_target(const float* anchors, int64_t A_total, const float* gt_boxes, int32_t N,
                        int32_t G_max, const float* im_info, float fg_thresh, float bg_thresh,
                        float allowed_border, int32_t batch_size, float fg_fraction, uint32_t seed,
                        uint32_t step, const uint32_t* step_dev, uint32_t image_offset, int32_t* labels,
                        int32_t* matched_gt,
                        float* bbox_targets, float* max_iou, void* workspace, size_t workspace_bytes,
                        mxdet_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * core/bbox (README.md:17) -- proposal-target: RoI sampling for the box head.
 * Candidates of image n = its first num_rois[n] proposals followed by its valid GT boxes.
 * fg: max IoU >= fg_thresh; bg: bg_lo <= max IoU < bg_hi. Sample min(fg_fraction*rois_per_image,
 * #fg) foreground and fill up with background, both by Philox keys (streams 2/3); selected rois are
 * written foreground first, each group in ascending candidate order; unfilled slots are padding
 * (label -1, zero box, zero weights).
 * Outputs: out_rois[N,R,5], labels[N,R] int32 (class, 0 = background), bbox_targets[N,R,4*num_reg]
 * and bbox_weights[N,R,4*num_reg] (class-specific slot, normalised (t-mean)/std), matched_gt[N,R],
 * num_fg[N]. num_reg = num_classes (class-specific) or 1 (class-agnostic). */
int mxdet_proposal_target(const float* rois, const int32_t* num_rois, int32_t rois_stride,
                          const float* gt_boxes, int32_t N, int32_t G_max, int32_t rois_per_image,
                          float fg_fraction, float fg_thresh, float bg_hi, float bg_lo,
                          int32_t num_classes, int32_t class_agnostic,
                          const float* means /* HOST [4] */, const float* stds /* HOST [4] */,
                          uint32_t seed, uint32_t step, const uint32_t* step_dev,
                          uint32_t image_offset, float* out_rois, int32_t* labels,
                          float* bbox_targets, float* bbox_weights, int32_t* matched_gt,
                          int32_t* num_fg, mxdet_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * roi_extractors (README.md:32) -- FPN level map + RoIAlign. MXNet role: contrib.ROIAlign
 * (Detectron aligned=False semantics), one call covering all pyramid levels.
 * feats[l]: bf16 [N,H[l],W[l],C] channels-last; rois[R,5]; levels[R] int32 in [lvl_min, lvl_min+L).
 * out: bf16 [R,PH,PW,C]. Bilinear taps in fp32 in the fixed order ((w1*v1+w2*v2)+w3*v3)+w4*v4,
 * samples summed iy-major then ix, times 1/count, rounded to bf16 once. */
int mxdet_fpn_level_map(const float* rois, int64_t R, int32_t lvl_min, int32_t lvl_max,
                        int32_t* levels, mxdet_stream_t stream);
typedef struct {
  int32_t num_levels, lvl_min;
  int32_t H[8], W[8];
  float spatial_scale[8];
  void* feat[8]; /* bf16 [N,H,W,C]; for bwd: fp32 gradient accumulators of the same shape */
} mxdet_feat_pyramid_t;
int mxdet_roi_align_fwd(const mxdet_feat_pyramid_t* f, int32_t N, int32_t C, const float* rois,
                        const int32_t* levels, int64_t R, int32_t PH, int32_t PW,
                        int32_t sampling_ratio, uint16_t* out, mxdet_stream_t stream);
/* bwd: scatter-adds grad_out (bf16 [R,PH,PW,C]) into fp32 [N,H,W,C] accumulators f->feat[l]
 * (caller zeroes them for accumulate = 0 semantics). fp32 atomics: order-dependent in the last bits. */
int mxdet_roi_align_bwd(const mxdet_feat_pyramid_t* f, int32_t N, int32_t C, const float* rois,
                        const int32_t* levels, int64_t R, int32_t PH, int32_t PW,
                        int32_t sampling_ratio, const uint16_t* grad_out, mxdet_stream_t stream);
/* bwd, gather form: deterministic, no atomics, no fp32 accumulators. f->feat[l] are the bf16 [N,H,W,C] GRADIENT maps
 * themselves: every pixel is written (accumulate = 0) or added to (accumulate = 1, one bf16 rounding of the sum).
 * Per pixel the contributions are summed in fp32 in a fixed order: roi index ascending; inside a roi the bilinear
 * weights are collapsed per pooling bin (they are separable), row bins before column bins -- bit-reproducible run to
 * run, last-bit different from the sample-by-sample order of mxdet_roi_align_bwd. Requires sampling_ratio > 0 and at
 * most 32 samples per axis. The per-roi records the gather reads depend only on the rois: `_prepare` writes them into
 * the workspace (it can be issued as soon as the rois exist, e.g. in the forward pass) and `_prepared` is the gather
 * without that pass; the plain call does both. */
size_t mxdet_roi_align_bwd_gather_workspace_bytes(const mxdet_feat_pyramid_t* f, int32_t N, int64_t R);
int mxdet_roi_align_bwd_gather(const mxdet_feat_pyramid_t* f, int32_t N, int32_t C, const float* rois,
                               const int32_t* levels, int64_t R, int32_t PH, int32_t PW, int32_t sampling_ratio,
                               const uint16_t* grad_out, int32_t accumulate, void* workspace, size_t workspace_bytes,
                               mxdet_stream_t stream);
int mxdet_roi_align_bwd_gather_prepare(const mxdet_feat_pyramid_t* f, int32_t N, const float* rois,
                                       const int32_t* levels, int64_t R, int32_t PH, int32_t PW, int32_t sampling_ratio,
                                       void* workspace, size_t workspace_bytes, mxdet_stream_t stream);
int mxdet_roi_align_bwd_gather_prepared(const mxdet_feat_pyramid_t* f, int32_t N, int32_t C, const float* rois,
                                        const int32_t* levels, int64_t R, int32_t PH, int32_t PW, int32_t sampling_ratio,
                                        const uint16_t* grad_out, int32_t accumulate, void* workspace,
                                        size_t workspace_bytes, mxdet_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * core/loss (README.md:19)
 * smooth-L1 (MXNet smooth_l1(scalar=sigma)): elementwise on (pred - target) * weight. */
int mxdet_smooth_l1_fwd(const float* pred, const float* target, const float* weight, int64_t n,
                        float sigma, float* out, mxdet_stream_t stream);
int mxdet_smooth_l1_bwd(const float* pred, const float* target, const float* weight,
                        const float* grad_out, int64_t n, float sigma, int32_t accumulate,
                        float* grad_pred, mxdet_stream_t stream);

/* Sigmoid focal loss (RetinaNet): logits [n,C] (dtype f32|bf16), labels[n] int32 (-1 ignore,
 * 0 background, c in 1..C foreground class c), normaliser = max(1, #foreground) computed on device.
 * loss_out[1] fp32 (sum / normaliser, fixed reduction order); grad written in the logits dtype.
 * workspace: mxdet_loss_workspace_bytes(n). */
size_t mxdet_loss_workspace_bytes(int64_t n);
int mxdet_focal_loss(const void* logits, int32_t dtype, const int32_t* labels, int64_t n, int32_t C,
                     float alpha, float gamma, float grad_scale, float* loss_out, void* grad_logits,
                     void* workspace, size_t workspace_bytes, mxdet_stream_t stream);

/* RetinaNet: per-anchor class labels (class of the matched GT for label == 1, else the -1 / 0 label) and the
 * number of foreground anchors (device word, the loss normaliser). */
int mxdet_anchor_class_labels(const int32_t* labels, const int32_t* matched_gt, const float* gt_boxes, int32_t N,
                              int64_t A_total, int32_t G_max, int32_t* cls_labels, int32_t* num_fg,
                              mxdet_stream_t stream);
/* RetinaNet losses fused fwd+bwd over one pyramid level: sigmoid focal loss on cls[cell*ld_cls + a*C + c] and
 * smooth-L1 on reg[cell*ld_reg + a*4 + k] (bf16 channels-last head outputs), both normalised by max(1, *num_fg);
 * gradients written in place of the same layout (padding channels untouched); partial sums (cls, reg) per
 * workgroup, reduced by mxdet_loss_finalize. */
int32_t mxdet_retina_loss_num_partials(int32_t N, int32_t H, int32_t W, int32_t A);
int mxdet_retina_loss_level(const uint16_t* cls, const uint16_t* reg, int32_t N, int32_t H, int32_t W, int32_t A,
                            int32_t C, int32_t ld_cls, int32_t ld_reg, const int32_t* cls_labels,
                            const float* bbox_targets, int64_t A_total, int64_t level_offset, float alpha,
                            float gamma, float sigma, const int32_t* num_fg, float loss_scale, uint16_t* grad_cls,
                            uint16_t* grad_reg, float* partial, mxdet_stream_t stream);

/* RPN losses fused fwd+bwd over one pyramid level's head output.
 * head: bf16 [N,H,W,Cpad] channels-last, channel a = objectness logit of anchor a, channel
 * A + 4a + k = delta k of anchor a. labels / bbox_targets are the mxdet_anchor_target outputs
 * (global anchor order), level_offset = index of this level's first anchor.
 * cls: sigmoid BCE over label >= 0; reg: smooth-L1(sigma) over label == 1; both * norm
 * (norm = 1 / sampled batch, chosen by the caller). Partial sums are appended to
 * partial[2*num_blocks] and reduced in fixed order by mxdet_loss_finalize.
 * grad_head: bf16 [N,H,W,Cpad] = d(loss)/d(head) * loss_scale (channels >= 5A are written 0).
 * Per-workgroup partial sums (cls, reg) go to partial[2*i + {0,1}], i < mxdet_rpn_loss_num_partials;
 * the caller lays the levels' partial ranges back to back and reduces them in index order with
 * mxdet_loss_finalize, so the scalar losses are reproducible bit for bit. */
int32_t mxdet_rpn_loss_num_partials(int32_t N, int32_t H, int32_t W);
int mxdet_rpn_loss_level(const uint16_t* head, int32_t N, int32_t H, int32_t W, int32_t A,
                         int32_t Cpad, const int32_t* labels, const float* bbox_targets,
                         int64_t A_total, int64_t level_offset, float sigma, float norm,
                         float loss_scale, uint16_t* grad_head, float* partial,
                         mxdet_stream_t stream);
/* out[j] = sum over i < count of partial[i*ncomp + j], fixed order, one workgroup */
int mxdet_loss_finalize(const float* partial, int32_t count, int32_t ncomp, float* out,
                        mxdet_stream_t stream);

/* Box-head losses fused fwd+bwd. cls_logits: [R,num_classes] (dtype), softmax CE with ignore label
 * -1, normalised by norm; bbox_pred [R,4*num_reg] vs targets/weights, smooth-L1(sigma) * norm.
 * loss_out[2] = (cls, reg); grads written in the logits dtype scaled by loss_scale. */
int mxdet_rcnn_loss(const void* cls_logits, const void* bbox_pred, int32_t dtype, int32_t ld_cls,
                    int32_t ld_reg, const int32_t* labels, const float* bbox_targets,
                    const float* bbox_weights, int64_t R, int32_t num_classes, int32_t reg_dim,
                    float sigma, float norm, float loss_scale, float* loss_out, void* grad_cls,
                    void* grad_reg, void* workspace, size_t workspace_bytes, mxdet_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * core/mask + mask_heads (README.md:18, :30) -- Mask R-CNN.
 * mask targets: for roi r (batch, box) with matched GT g = matched_gt[r] and class labels[r] > 0, resample the
 * instance bitmask gt_masks[batch, g] ([N,G_max,H,W] u8, 0/1) inside the box to S x S (bilinear at bin centres,
 * RoIAlign aligned=False tap rules) and threshold at 0.5. cls_out[r] = class (1..) or -1 for non-foreground rois. */
int mxdet_mask_target(const float* rois, const int32_t* matched_gt, const int32_t* labels,
                      const uint8_t* gt_masks, int64_t R, int32_t G_max, int32_t H, int32_t W, int32_t S,
                      uint8_t* targets, int32_t* cls_out, mxdet_stream_t stream);
/* x [R,H,W,4*C] (channel = (dy*2+dx)*C + c) -> y [R,2H,2W,C] (inverse = 0) or back (inverse = 1): with a 1x1
 * conv C_in -> 4*C this is Deconvolution(kernel=2, stride=2). */
int mxdet_pixel_shuffle2(const uint16_t* x, int64_t R, int32_t H, int32_t W, int32_t C, int32_t inverse,
                         uint16_t* y, mxdet_stream_t stream);
/* backward of conv1x1 -> ReLU -> pixel shuffle in one pass: dx[r,h,w,(dy*2+dx)*C+c] = dy_up[r,2h+dy,2w+dx,c] where the
 * packed activation act (same layout as dx, [R,H,W,4*C]) is > 0, else 0 */
int mxdet_pixel_shuffle2_inv_relu(const uint16_t* dy_up, const uint16_t* act, int64_t R, int32_t H, int32_t W, int32_t C,
                                  uint16_t* dx, mxdet_stream_t stream);
/* per-pixel sigmoid BCE on channel cls[r]-1 of logits [R,S,S,Cpad] (bf16), normalised by (#fg rois * S*S);
 * grad (bf16, same shape) is fully written; loss_out[1] fp32, fixed-order reduction. */
size_t mxdet_mask_loss_workspace_bytes(int64_t R, int32_t S);
int mxdet_mask_loss(const uint16_t* logits, const int32_t* cls, const uint8_t* targets, int64_t R, int32_t S,
                    int32_t Cpad, float loss_scale, float* loss_out, uint16_t* grad, void* workspace,
                    size_t workspace_bytes, mxdet_stream_t stream);
/* Inference paste-back (MXNet-lineage role: sigmoid -> cv2.resize(mask, (w, h)) -> > 0.5 -> place at the box, on the
 * host). dets [R,6] = (x1,y1,x2,y2,score,class) in the frame of the OUTPUT masks; logits [R,S,S,Cpad] bf16.
 * prob = sigmoid(logit of channel class-1); box corners rounded half-even to integers, w = x2-x1+1, h = y2-y1+1;
 * pixel (x,y) inside the box samples prob bilinearly at ((x-x1+0.5)*S/w-0.5, (y-y1+0.5)*S/h-0.5) (clamped, fp32,
 * unfused: top/bottom rows first, then vertical) and is 1 where the value > thresh. masks [R,H,W] u8, W % 4 == 0;
 * rows with class <= 0 are all zero. */
size_t mxdet_mask_paste_workspace_bytes(int64_t R, int32_t S);
int mxdet_mask_paste(const uint16_t* logits, const float* dets, int64_t R, int32_t S, int32_t Cpad, int32_t H, int32_t W,
                     float thresh, uint8_t* masks, void* workspace, size_t workspace_bytes, mxdet_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * backbones / necks / rpn_heads / bbox_heads / mask_heads (README.md:27-31) -- dense contractions.
 * MXNet roles: Convolution (+ BatchNorm(use_global_stats) + Activation + elemwise_add),
 * FullyConnected, Pooling, UpSampling. All bf16 in / fp32 accumulate on MFMA / bf16 out.
 *
 * conv2d: x bf16 [N,H,W,Cin] channels-last, w bf16 [Cout,KH,KW,Cin], y bf16 [N,Ho,Wo,Cout].
 * Epilogue: y = act( conv + bias[c] + residual ), residual optional, of shape y or -- with
 * res_upsample = 1 -- [N,ceil(Ho/2),ceil(Wo/2),Cout] read nearest-neighbour (FPN top-down add).
 * A fully-connected layer is the 1x1 case with H = W = 1 and N = rows. */
typedef struct {
  int32_t N, H, W, Cin;      /* input  */
  int32_t Cout, KH, KW;      /* filter */
  int32_t stride, pad;       /* same in both dims */
  int32_t Ho, Wo;            /* output; must equal floor((H + 2*pad - KH)/stride) + 1 etc. */
  int32_t relu;              /* fwd: apply ReLU;   dgrad: multiply by (mask > 0) */
  int32_t res_upsample;      /* fwd only */
  int32_t accumulate;        /* dgrad / wgrad: add into the output instead of overwriting */
  /* Optional hint (fwd / dgrad, single launches): device memory the caller will read NEXT -- the filter of the layer
   * that follows -- or null. Every workgroup reads a slice of it (up to 16 KiB) while its epilogue runs, so the bytes
   * sit in the Infinity Cache when the next launch starts: a layer's filter was last touched a whole step earlier and
   * its first touch otherwise costs an HBM round trip in front of every workgroup's K loop (measured: a C5 3x3 layer
   * 32.7 us with a cold filter, 24.0 us with a warm one). The contents are never used. */
  const void* prefetch;
  int64_t prefetch_bytes;
  /* Optional 1-bit ReLU mask, byte [pixel][channel / 8], bit k = (value of channel 8 j + k) > 0 (the channel count must
   * be a multiple of 8). fwd: if non-null the kernel ALSO writes the mask of the values it stores (the activation's
   * ReLU mask for the backward pass, 1/16 of its bytes). dgrad with relu = 1: if non-null it is read INSTEAD of the
   * 16-bit relu_mask operand (which may then be null) -- same result, 1/16 of the operand traffic: the expand-layer
   * data gradients are HBM-bound and their mask is a forward activation that is cold in every cache. */
  void* relu_bits;
} mxdet_conv_desc_t;

int mxdet_conv2d_fwd(const mxdet_conv_desc_t* d, const uint16_t* x, const uint16_t* w,
                     const float* bias, const uint16_t* residual, uint16_t* y,
                     mxdet_stream_t stream);
/* Chained forward: y2 = act2( act( conv3x3(x, w) + bias ) (*) w2 + bias2 + residual2 ), act = ReLU if d->relu, act2 = ReLU if
 * relu2 -- the tail of a bottleneck block (conv2 -> conv3 + shortcut) whose intermediate map nobody needs afterwards, i.e.
 * a FROZEN block (models/backbones, /root/reference/README.md:27: ResNet stage C2 with frozen_stages = 1). d describes the
 * 3x3 (stride 1, pad 1, Cin % 64 == 0, Cout == 64); w2 is [cout2][1][1][64] with cout2 == 256; residual2 / y2 are
 * [N, Ho, Wo, cout2]. One launch, the 64-channel intermediate never reaches memory; bit-identical to mxdet_conv2d_fwd
 * twice. d->prefetch is honoured; d->relu_bits and d->res_upsample must be unset.
 * w3 != NULL adds a third convolution in the same launch: y3 = act3( y2 (*) w3 + bias3 ), w3 [cout3][1][1][cout2] with
 * cout3 == 64, y3 [N, Ho, Wo, cout3] -- the NEXT block's conv1, computed from the block output's stored bf16 values as
 * they leave the workgroup (bit-identical to a further mxdet_conv2d_fwd on y2). */
int mxdet_conv2d_fwd_chain(const mxdet_conv_desc_t* d, const uint16_t* x, const uint16_t* w, const float* bias,
                           const uint16_t* w2, const float* bias2, int32_t cout2, int32_t relu2,
                           const uint16_t* residual2, uint16_t* y2, const uint16_t* w3, const float* bias3,
                           int32_t cout3, int32_t relu3, uint16_t* y3, mxdet_stream_t stream);
/* Forward with the reduction split over `ksplit` ranges of 64-channel slices, for 1x1 / stride-1 layers with a long
 * reduction on few rows (fully connected layers on pooled rois): raw fp32 tiles in the caller's workspace, folded in
 * split order (deterministic) with bias / residual / ReLU by a second kernel. Last-bit different from mxdet_conv2d_fwd
 * (another summation order); ksplit = 1 is that call. */
size_t mxdet_conv2d_fwd_splitk_workspace_bytes(const mxdet_conv_desc_t* d, int32_t ksplit);
int mxdet_conv2d_fwd_splitk(const mxdet_conv_desc_t* d, const uint16_t* x, const uint16_t* w, const float* bias,
                            const uint16_t* residual, uint16_t* y, int32_t ksplit, void* workspace,
                            size_t workspace_bytes, mxdet_stream_t stream);
/* dgrad: dx[N,H,W,Cin] = (sum over taps dy[N,Ho,Wo,Cout] * w  +  residual) * (relu_mask > 0), with
 * wt = the same filter stored [Cin,KH,KW,Cout] (mxdet_filter_transpose). residual (bf16, shape of dx;
 * may be NULL) is the gradient arriving over a parallel branch (identity shortcut / sibling conv);
 * d->accumulate with residual == NULL means residual = dx. The mask factor applies only if d->relu:
 * relu_mask is then the forward activation x (bf16, post-ReLU). */
int mxdet_conv2d_dgrad(const mxdet_conv_desc_t* d, const uint16_t* dy, const uint16_t* wt,
                       const uint16_t* residual, const uint16_t* relu_mask, uint16_t* dx,
                       mxdet_stream_t stream);
/* wgrad: dw fp32 [Cout,KH,KW,Cin] (+)= sum over pixels dy * x; deterministic split-K through
 * workspace slabs reduced in fixed order; db fp32 [Cout] (may be NULL) (+)= sum over pixels dy. */
size_t mxdet_conv2d_wgrad_workspace_bytes(const mxdet_conv_desc_t* d);
int mxdet_conv2d_wgrad(const mxdet_conv_desc_t* d, const uint16_t* x, const uint16_t* dy, float* dw,
                       float* db, void* workspace, size_t workspace_bytes, mxdet_stream_t stream);
/* Grouped convolutions: independent forward convolutions (kind 0) or stride-1 data gradients (kind 1) that can share
 * one tile configuration -- the 3x3 of every pyramid level of an RPN / RetinaNet head, the FPN output convs -- as ONE
 * grid: the small levels ride in the shadow of the large ones instead of running latency-bound on their own.
 * Same protocol as the grouped weight gradients: _plan writes a host table (+ tile configuration and grid size), the
 * caller uploads it once, mxdet_conv2d_grouped launches it. Per item the semantics are exactly those of
 * mxdet_conv2d_fwd (src = x, filt = w, dst = y; bias / residual / desc.relu / desc.res_upsample) or
 * mxdet_conv2d_dgrad (src = dy, filt = wt, dst = dx; residual / relu_mask / desc.accumulate). */
typedef struct {
  mxdet_conv_desc_t desc;
  const void* src;
  const void* filt;
  const float* bias;
  const void* residual;
  const void* relu_mask;
  void* dst;
} mxdet_conv_item_t;
size_t mxdet_conv2d_grouped_table_bytes(int32_t n);
int mxdet_conv2d_grouped_plan(const mxdet_conv_item_t* items, int32_t n, int32_t kind, void* table_host,
                              size_t table_bytes, int32_t* cfg, int32_t* grid);
int mxdet_conv2d_grouped(const void* table_dev, int32_t n, int32_t kind, int32_t cfg, int32_t grid,
                         mxdet_stream_t stream);
/* Grouped weight gradients: the wgrads of many layers (a ResNet stage, the FPN, a head) in ONE launch pair (MFMA
 * workgroups of all layers in one grid + one fold launch). At batch 2 a layer's own grid is one or two workgroups per
 * CU; a group runs at the occupancy of the largest layers, needs less split-K and two launches instead of 2 per layer.
 * Usage: fill items[] (descriptor + device pointers, accumulate honoured per item), call _plan once per group -- it
 * writes a table of mxdet_conv2d_wgrad_grouped_table_bytes(n) bytes to HOST memory and reports the workspace size and
 * the two grid sizes --, copy the table to device memory (it stays valid while the pointers do, e.g. across hipGraph
 * replays), then launch with mxdet_conv2d_wgrad_grouped. Results equal mxdet_conv2d_wgrad's up to the split-K
 * partition (fp32 sums in a different, still fixed, order). */
typedef struct {
  mxdet_conv_desc_t desc;
  const void* x;    /* bf16 [N,H,W,Cin] */
  const void* dy;   /* bf16 [N,Ho,Wo,Cout] */
  float* dw;        /* f32 [Cout,KH,KW,Cin] */
  float* db;        /* f32 [Cout] or NULL */
} mxdet_wgrad_item_t;
size_t mxdet_conv2d_wgrad_grouped_table_bytes(int32_t n);
int mxdet_conv2d_wgrad_grouped_plan(const mxdet_wgrad_item_t* items, int32_t n, void* table_host, size_t table_bytes,
                                    size_t* workspace_bytes, int32_t* grid_wgrad, int32_t* grid_big,
                                    int32_t* grid_reduce);
int mxdet_conv2d_wgrad_grouped(const void* table_dev, int32_t n, int32_t grid_wgrad, int32_t grid_big,
                               int32_t grid_reduce, void* workspace, size_t workspace_bytes, size_t workspace_needed,
                               mxdet_stream_t stream);
/* The same launch in parts (bit 0: the three-tap kernel of the 3x3 / stride 1 items, grid_big workgroups; bit 1: the
 * one-tap kernel + bias workgroups, grid_wgrad; bit 2: the fold). The two tile kernels are independent of each other
 * (the 3x3 tiles are MFMA-bound, the 1x1 tiles HBM-bound: a caller may issue them on two streams); the fold needs both. */
int mxdet_conv2d_wgrad_grouped_parts(const void* table_dev, int32_t n, int32_t grid_wgrad, int32_t grid_big,
                                     int32_t grid_reduce, int32_t parts, void* workspace, size_t workspace_bytes,
                                     size_t workspace_needed, mxdet_stream_t stream);
/* w [Cout,KH,KW,Cin] -> wt [Cin,KH,KW,Cout] (bf16) */
int mxdet_filter_transpose(const uint16_t* w, int32_t Cout, int32_t KH, int32_t KW, int32_t Cin,
                           uint16_t* wt, mxdet_stream_t stream);

/* All filters of a model in one launch. descs_dev: device array of ndesc records
 * { const uint16_t* w; uint16_t* wt; int32 Cout, taps, Cin, tile0; } (32 bytes each) where tile0 is the running
 * sum of ceil(Cin/64)*ceil(Cout/64)*taps over the preceding records, total_tiles the overall sum. */
int mxdet_filter_transpose_batched(const void* descs_dev, int32_t ndesc, int32_t total_tiles,
                                   mxdet_stream_t stream);

/* stem: 7x7 stride-2 pad-3 convolution reading the NCHW fp32/bf16 image [N,3,H,W] directly
 * (coalesced plane reads), + bias + ReLU, writing bf16 [N,Ho,Wo,64]; w bf16 [64,7,7,3]. */
int mxdet_stem_conv7x7(const void* image, int32_t dtype, int32_t N, int32_t H, int32_t W,
                       const uint16_t* w, const float* bias, uint16_t* y, mxdet_stream_t stream);
/* the same followed by the 3x3 stride-2 pad-1 max pooling, in one pass (the stem map is never written): y bf16
 * [N,Hp,Wp,64] with Ho = (H-1)/2+1, Hp = (Ho-1)/2+1 (same for W); bits equal to maxpool3x3s2(stem_conv7x7) up to the
 * order of the fp32 accumulation inside the convolution */
int mxdet_stem_conv7x7_pool(const void* image, int32_t dtype, int32_t N, int32_t H, int32_t W,
                            const uint16_t* w, const float* bias, uint16_t* y, mxdet_stream_t stream);
/* 3x3 stride-2 pad-1 max pooling, bf16 channels-last */
int mxdet_maxpool3x3s2(const uint16_t* x, int32_t N, int32_t H, int32_t W, int32_t C, uint16_t* y,
                       mxdet_stream_t stream);
/* y[N,ceil(H/2),ceil(W/2),C] = x[N, 2i, 2j, C]  (FPN P6) and its adjoint (scatter into zeros / add) */
int mxdet_subsample2(const uint16_t* x, int32_t N, int32_t H, int32_t W, int32_t C, uint16_t* y,
                     mxdet_stream_t stream);
int mxdet_subsample2_bwd(const uint16_t* dy, int32_t N, int32_t H, int32_t W, int32_t C,
                         int32_t accumulate, uint16_t* dx, mxdet_stream_t stream);
/* adjoint of the nearest-neighbour 2x upsample: dcoarse[N,Hc,Wc,C] (+)= sum of the 2x2 fine cells */
int mxdet_upsample2_bwd(const uint16_t* dfine, int32_t N, int32_t Hf, int32_t Wf, int32_t C,
                        int32_t accumulate, uint16_t* dcoarse, mxdet_stream_t stream);
/* One-stage (RetinaNet) test-time detection, MXNet-lineage role: per level top-k of sigmoid scores over (anchor, class),
 * decode, concatenate levels, per-class NMS, max_per_image -- a host loop in the lineage.
 * p describes the head outputs with p->classes = C foreground classes and p->A = anchors_per_cell * C:
 *   score(n,y,x,a*C+c) is the class logit, delta(n,y,x,a,k) the box delta (plain encoding, no normalisation).
 * Per image and level the pre_nms_top_n largest logits (ties: lower (cell, anchor, class) index first) are decoded and
 * clipped; the candidates of all levels are merged by (logit desc, index asc) and cut to 4096; score = sigmoid(logit);
 * then exactly as mxdet_detection_postprocess: per class keep score > score_thresh, greedy NMS at nms_thresh, and the
 * max_per_image best by (score desc, candidate rank asc, class asc). dets [N,max_per_image,6], class in 1..C. */
size_t mxdet_retina_detect_workspace_bytes(const mxdet_pyramid_t* p, int32_t N, int32_t pre_nms_top_n);
int mxdet_retina_detect(const mxdet_pyramid_t* p, int32_t N, const float* im_info, int32_t pre_nms_top_n,
                        float score_thresh, float nms_thresh, int32_t max_per_image, float* dets, int32_t* num_dets,
                        void* workspace, size_t workspace_bytes, mxdet_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * process_data (README.md:23) -- the step in front of the path.
 * MXNet-lineage role: the host-side cv2 pipeline  flip -> cv2.resize(fx=fy=scale, INTER_LINEAR) -> BGR->RGB,
 * minus pixel mean (over std) -> NCHW -> zero-pad the batch to a common size. Here one launch turns the decoded
 * 8-bit frames of a batch into the bf16 NCHW tensor the stem kernel reads.
 *
 * Resize arithmetic = OpenCV's 8-bit bilinear (the reference names cv2 as its resizer, README.md:53-56, no version
 * pinned; restated from its published imgproc/resize.cpp, parity with a cv2 build is UNPINNED -- cv2 is not in the
 * image): with inv = 1/scale in double,
 *   fx = (float)((dx + 0.5) * inv - 0.5); sx = floor(fx); fx -= sx; sx < 0 -> (0, 0); sx >= sw-1 -> (sw-1, 0)
 *   a1 = rint(fx * 2048), a0 = rint((1 - fx) * 2048)           (11-bit coefficients, round half even)
 *   row(y) = S[y][sx] * a0 + S[y][min(sx+1, sw-1)] * a1;         same for dy -> (sy, b0, b1)
 *   v = (((b0 * (row(sy) >> 4)) >> 16) + ((b1 * (row(sy+1) >> 4)) >> 16) + 2) >> 2
 * then out = (v - mean[c]) / std[c] in fp32 (IEEE, unfused), rounded to bf16 (nearest even).
 * flip mirrors the SOURCE columns (flip-then-resize, the order the lineage uses); swap_rb reads source channel 2-c.
 * Pixels outside (dst_h, dst_w) of their image are written as zero: out is [N, 3, Hp, Wp], Wp % 8 == 0. */
typedef struct {
  const uint8_t* src;     /* device pointer, [src_h, src_w, 3] 8-bit interleaved, row pitch src_w*3 */
  int32_t src_h, src_w;
  int32_t dst_h, dst_w;   /* rint(src * scale), <= Hp / Wp */
  int32_t flip;           /* mirror columns */
  int32_t pad_;
  double inv_scale;       /* 1 / scale */
} mxdet_image_desc_t;
#define MXDET_PREPROCESS_MAX_BATCH 64
int mxdet_image_preprocess(const mxdet_image_desc_t* images /* host array */, int32_t N, int32_t Hp, int32_t Wp,
                           const float* mean3 /* host */, const float* std3 /* host */, int32_t swap_rb,
                           uint16_t* out, mxdet_stream_t stream);

/* Instance masks from polygons (datasets: COCO "segmentation" lists), written straight at network resolution.
 * verts [V,2] f32 (x,y) already scaled/flipped into the resized image; poly_start [P+1] i32 offsets into verts
 * (polygons of one instance adjacent, instances in n*G+g order); inst_first [N*G+1] i32 offsets into the polygon list. masks [N,G,H,W] u8: 1 where the pixel centre (x+0.5, y+0.5) is inside an odd number of edges
 * of ANY polygon of the instance (union of even-odd fills), else 0; instances without polygons are all zero.
 * Edge rule: (y0 <= py) != (y1 <= py) and px < x0 + (py - y0) * (x1 - x0) / (y1 - y0), fp32 unfused.
 * pycocotools' frPoly walks a 5x upsampled boundary instead; the two differ only on boundary pixels (unpinned). */
int mxdet_polygon_masks(const float* verts, const int32_t* poly_start, const int32_t* inst_first, int32_t N,
                        int32_t G, int32_t H, int32_t W, uint8_t* masks, mxdet_stream_t stream);

/* elementwise helpers on channels-last tensors */
int mxdet_add_bf16(const uint16_t* a, const uint16_t* b, int64_t n, uint16_t* out,
                   mxdet_stream_t stream);
int mxdet_relu_bwd_bf16(const uint16_t* dy, const uint16_t* y, int64_t n, uint16_t* dx,
                        mxdet_stream_t stream);
int mxdet_f32_to_bf16(const float* x, int64_t n, uint16_t* y, mxdet_stream_t stream);
/* y[i] (+)= bf16(x[i]) as fp32->bf16 with optional mask y*(mask>0): used to fold the RoIAlign
 * fp32 gradient accumulators into the bf16 pyramid gradients */
int mxdet_f32_accum_to_bf16(const float* x, int64_t n, int32_t accumulate, uint16_t* y,
                            mxdet_stream_t stream);
/* layout converters at the boundary (MXNet tensors are NCHW) */
int mxdet_nchw_to_nhwc_bf16(const void* x, int32_t dtype, int32_t N, int32_t C, int32_t H, int32_t W,
                            uint16_t* y, mxdet_stream_t stream);
int mxdet_nhwc_to_nchw_f32(const uint16_t* x, int32_t N, int32_t C, int32_t H, int32_t W, float* y,
                           mxdet_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * optimizer step (MXNet role: sgd_mom_update after kvstore reduce).
 * Flat fp32 parameter / gradient / momentum arenas of n elements:
 *   g = grad * rescale (+ wd * w);  m = momentum * m + g;  w -= lr_mult[i] * lr * m
 * and the bf16 working copy of w is refreshed in the same pass. per-element scale arrays may be
 * NULL. */
int mxdet_sgd_momentum_update(float* w, const float* grad, float* mom, uint16_t* w_bf16, int64_t n,
                              float lr, float momentum, float wd, float rescale,
                              mxdet_stream_t stream);
/* Same update with the learning rate read from device memory (lr_dev[0]) when the kernel runs: a step captured into
 * a hipGraph follows the warm-up / step schedule (MXNet role: lr_scheduler feeding the optimizer) without re-capture. */
int mxdet_sgd_momentum_update_sched(float* w, const float* grad, float* mom, uint16_t* w_bf16, int64_t n,
                                    const float* lr_dev, float momentum, float wd, float rescale,
                                    mxdet_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * data-parallel gradient exchange (MXNet role: kvstore push/pull, /root/reference/README.md:37; SURVEY.md
 * sections 8a10, 8b, 8e). One process per GPU; the only exchange of the hot path is the fp32 sum of contiguous slices
 * ("buckets") of the flat gradient arena, over RCCL (xGMI). `mxdet_comm_t` is the one explicit handle of the library:
 * an RCCL communicator + a side stream of its own + a ring of events, created and destroyed by the caller. The library
 * resolves RCCL at run time (the copy already mapped into the process, else librccl.so.1); without it every entry
 * below returns MXDET_ERCCL.
 *
 *   rank 0: mxdet_comm_unique_id(id) -> the caller hands the MXDET_COMM_ID_BYTES bytes to every rank by any channel
 *   every rank, with its device current: mxdet_comm_create(id, world, rank, &comm)           (collective)
 *   per bucket, as soon as backward has finalised it: mxdet_allreduce_bucket(comm, g + lo, hi - lo, s, &ticket)
 *       the sum is ordered behind everything enqueued on stream `s` so far and runs on the communicator's side
 *       stream: `s` itself does not wait (the rest of backward overlaps the exchange);
 *   before the bucket is consumed: mxdet_comm_wait(comm, ticket, t) makes stream `t` wait for that bucket's sum
 *       (ticket -1: for every bucket issued so far).
 * Sums are fp32, in place, in RCCL's fixed order for the communicator (run-to-run identical); the 1/world average is
 * the optimizer's `rescale`. At most MXDET_COMM_MAX_INFLIGHT buckets may be un-waited at a time. */
typedef struct mxdet_comm mxdet_comm_t;
#define MXDET_COMM_ID_BYTES 128
#define MXDET_COMM_MAX_INFLIGHT 64
int mxdet_comm_unique_id(uint8_t* id /* host, MXDET_COMM_ID_BYTES */);
int mxdet_comm_create(const uint8_t* id, int32_t world, int32_t rank, mxdet_comm_t** comm_out);
int mxdet_comm_destroy(mxdet_comm_t* comm);
int mxdet_allreduce_bucket(mxdet_comm_t* comm, float* grad, int64_t count, mxdet_stream_t stream,
                           int32_t* ticket_out);
int mxdet_comm_wait(mxdet_comm_t* comm, int32_t ticket, mxdet_stream_t stream);
/* replicate `bytes` bytes of `buf` from rank `root` (initial weights); ordered on `stream` itself */
int mxdet_comm_broadcast(mxdet_comm_t* comm, void* buf, size_t bytes, int32_t root, mxdet_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* MXDET_H_ */
