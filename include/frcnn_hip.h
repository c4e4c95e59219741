/*
 * frcnn_hip.h -- C ABI of libfrcnn_hip.so: the Faster R-CNN region-proposal and
 * RoI hot path as hand-written HIP kernels for gfx950 (MI355X).
 *
 * The reference (ttjh1234/ObjectDetection) has no FFI: its boundary is the set
 * of Python functions in Faster_RCNN/module/.  Each entry point below replaces
 * the TensorFlow ops one of those functions runs; the citation names the
 * reference lines it stands in for.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless the comment says "host";
 *   - boxes are float32 [ymin, xmin, ymax, xmax], normalized;
 *   - anchor linear index idx = (y*W + x)*A + a          (module/data.py:117);
 *   - feature maps are addressed through explicit element strides
 *     (sb, sh, sw, sc); channels-last (sc == 1) is the fast layout;
 *   - the caller owns inputs, outputs and workspace; the library allocates
 *     nothing and keeps no caller pointer.  Its only process-wide state: the
 *     tuning values below (atomics; the environment is read once, when the
 *     library is loaded -- no entry point calls getenv) and the scratch array
 *     of frcnn_nms_static.  Every other entry point is re-entrant;
 *   - work is ordered on `stream` (a hipStream_t passed as void*); no call
 *     synchronises the device;
 *   - return value: FRCNN_OK or a negative FRCNN_E_* code; nothing throws.
 *     Data-dependent failures (short sample batches) are reported on the
 *     device in the per-image `status`/`counts` outputs, never by a sync.
 */
#ifndef FRCNN_HIP_H
#define FRCNN_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FRCNN_OK            0
#define FRCNN_E_BADARG     -1   /* null pointer, non-positive size, bad flag   */
#define FRCNN_E_WORKSPACE  -2   /* workspace missing or too small              */
#define FRCNN_E_LAUNCH     -3   /* hipGetLastError() != hipSuccess after launch */
#define FRCNN_E_UNSUPPORTED -4  /* size beyond what the kernels are built for  */
#define FRCNN_E_STATIC_LIMIT -5 /* needs more than the built-in scratch array holds */

/* flag bits written by frcnn_iou_assign */
#define FRCNN_F_VALID     1u    /* all four anchor coords strictly in (0,1)    */
#define FRCNN_F_POSITIVE  2u    /* valid and best IoU  > pos_thr               */
#define FRCNN_F_NEGATIVE  4u    /* valid and best IoU  < neg_thr               */

/* per-image sampler status (device side) */
#define FRCNN_S_OK          0
#define FRCNN_S_SHORT_BATCH 1   /* fewer negatives than batch - n_pos_taken    */

typedef void* frcnn_stream_t;

int         frcnn_abi_version(void);
const char* frcnn_status_string(int status);

/* Tuning values: thresholds that select between kernels computing the same result (no
 * counterpart in the reference).  key is the short name or its environment variable:
 *   "nms_gk"            FRCNN_NMS_GK            mask workgroups per column block (<= 0: by batch)
 *   "roi_band_min_wg"   FRCNN_ROI_BAND_MIN_WG   band-resident RoI forward from n workgroups on and
 *                                               for any RoI count (n < 0: never)
 *   "bwd_cx"            FRCNN_BWD_CX            gather backward: feature cells per wave (1|2);
 *                                               patch-owner backward: patch shape, 10 * rows +
 *                                               columns (44|34|46|26; default by launch size)
 *   "bwd_chunk"         FRCNN_BWD_CHUNK         RoIs per launch of the gather backward (<= 0: by batch)
 *   "bwd_band"          FRCNN_BWD_BAND          patch-owner RoI backward (pool 2, crops up to
 *                                               16 x 16): 0 never (the gather kernel runs instead)
 *   "nms_fused_max_b"   FRCNN_NMS_FUSED_MAX_B   single-launch NMS for batches up to n images
 *   "roi_band_split"    FRCNN_ROI_BAND_SPLIT    band-resident RoI forward: ranges an image's RoIs are
 *                                               split into (<= 0: by launch size)
 *   "topk_merge_max_runs" FRCNN_TOPK_MERGE_MAX_RUNS frcnn_topk_sort: calls of up to n blocks of 1,024
 *                                               scores (all images together) sort each block in one
 *                                               wave and rank across blocks, two launches instead
 *                                               of the radix passes' eight (0: never)
 * A variable present in the environment when the library is loaded gives the initial value;
 * frcnn_set_tuning replaces it (FRCNN_TUNING_UNSET restores the built-in choice).  Returns
 * FRCNN_E_BADARG for an unknown key. */
#define FRCNN_TUNING_UNSET ((long long)(-9223372036854775807LL - 1))
int frcnn_set_tuning(const char* key, long long value);
int frcnn_get_tuning(const char* key, long long* value);

/* ---- a1: anchor grid -------------------------------------------------------
 * Replaces make_anchor(), Faster_RCNN/module/anchor.py:10-44 (31x31x9) and
 * shipdetection/train.py:128-162 (32x32x12).  float64 math, one cast to f32.
 * scales/ratios are HOST arrays (n_scales*n_ratios == A <= 64).
 * base_w/base_h: the reference divides both axes by 500 (base_w == base_h).
 * anchors: [H*W*A*4] float32. */
int frcnn_anchors(float* anchors, int H, int W,
                  const double* scales, int n_scales,
                  const double* ratios, int n_ratios,
                  double base_w, double base_h, frcnn_stream_t stream);

/* ---- a2 + first half of a3: anchors x GT IoU, argmax, flags ----------------
 * Replaces the broadcast IoU and argmax/threshold logic of patch_batch,
 * Faster_RCNN/module/data.py:79-110.
 * anchors [N,4]; gt [B,G,4] (zero rows pad); outputs per image:
 * max_iou [B,N] f32, best_gt [B,N] i32 (first maximum), flags [B,N] u8
 * (FRCNN_F_*).  iou_full (optional, may be NULL) receives the whole [B,N,G]
 * matrix for parity checks. */
int frcnn_iou_assign(const float* anchors, int N, const float* gt, int B, int G,
                     float pos_thr, float neg_thr,
                     float* max_iou, int32_t* best_gt, uint8_t* flags,
                     float* iou_full, frcnn_stream_t stream);

/* ---- second half of a3 + a4: RPN mini-batch sampling, gather and encode ----
 * Replaces Faster_RCNN/module/data.py:111-149: ordered compaction of the
 * positive list and of the negative list (negatives are positions within the
 * valid-anchor list, decoded as anchor ids -- kept on purpose, data.py:110,118),
 * three shuffles, gathers, corner-based target encode.
 * Shuffle = order by (key, list position).  keys_pos/keys_neg [B,N] and
 * keys_all [B,batch] are uint32 keys per list position; pass NULL for all
 * three to use Philox-4x32-10 counter keys from (seed, image_base + b).
 * Outputs per image: batch_label [B,batch] f32, batch_anchor/batch_gt/
 * batch_reg_gt [B,batch,4] f32, gt_list [B,batch] i64, batch_pos [B,batch,3]
 * i64 (y,x,a), counts [B,4] i32 = {P, Nneg, n_pos_taken, status}.
 * N <= 2^24, batch <= 1024. */
int frcnn_rpn_sample(const float* anchors, int H, int W, int A,
                     const float* gt, int B, int G,
                     const int32_t* best_gt, const uint8_t* flags,
                     const uint32_t* keys_pos, const uint32_t* keys_neg,
                     const uint32_t* keys_all, uint64_t seed, int image_base,
                     int batch, int max_pos,
                     float* batch_label, float* batch_anchor, float* batch_gt,
                     int64_t* gt_list, int64_t* batch_pos, float* batch_reg_gt,
                     int32_t* counts, frcnn_stream_t stream);

/* ---- a4 / a10: corner-based target encode -------------------------------------
 * Replaces Faster_RCNN/module/data.py:143-149 (clip_wh = 0) and :252-259
 * (clip_wh = 1: anchor width/height clipped to [1e-2,1]).  [n,4] each. */
int frcnn_encode(const float* anchor, const float* gt, int64_t n, int clip_wh,
                 float* target, frcnn_stream_t stream);

/* ---- a5 / a12: decode deltas and clip ------------------------------------------
 * Replaces inverse_trans(), Faster_RCNN/module/data.py:154-176 (clip_wh = 0,
 * anchors [n_anchor,4] broadcast over B images, deltas [B,n_anchor,4]) and
 * generate_coord(), :179-198 (clip_wh = 1, call with B = 1 and
 * n_anchor = total boxes).  boxes [B,n_anchor,4]. */
int frcnn_decode_clip(const float* anchors, const float* deltas, int B, int64_t n_anchor,
                      int clip_wh, float* boxes, frcnn_stream_t stream);

/* ---- a6 (first part): per-image score top-k, sorted, with fused decode ----------
 * Replaces tf.math.top_k + index split + inverse_trans + gather_nd,
 * Faster_RCNN/module/model.py:45-48 (= module/data.py:203-206).
 * scores [B,N]; descending, equal scores -> lower index first (-0 == +0).
 * top_scores [B,K] f32, top_idx [B,K] i32.  If anchors and deltas are both
 * non-NULL, cand_boxes [B,K,4] receives the decoded+clipped box of each
 * selected anchor (decode of the K winners only; same values as decoding all
 * N and gathering).  1 <= K <= min(N, 16384).  No workspace. */
int frcnn_topk(const float* scores, int B, int N, int K,
               const float* anchors, const float* deltas,
               float* top_scores, int32_t* top_idx, float* cand_boxes,
               frcnn_stream_t stream);

/* The same result by a stable radix sort through global memory, from small workgroups: any
 * 1 <= K <= N, and it shares the chip with a concurrent bandwidth-bound kernel (frcnn_topk's
 * workgroups need a CU to themselves).  A call of a few images (tuning value
 * "topk_merge_max_runs") takes two launches instead: one wave sorts each block of 1,024
 * scores, then every score's place is its place in its block plus the smaller scores of all
 * other blocks of its image.  workspace: frcnn_topk_sort_workspace_bytes(B,N) bytes (about
 * 16 bytes per score; up to 30 for a call that takes the two-launch form under the tuning
 * value in force when the size is asked for), 16-byte aligned.  A workspace that only covers
 * the radix passes is always accepted: the call then takes them. */
size_t frcnn_topk_sort_workspace_bytes(int B, int N);
/* 1 if a call of this size with this much workspace takes the two-launch sorted-runs form, 0 if
 * the radix passes (no counterpart in the reference; for tests and bench lines). */
int frcnn_topk_sort_plan(int B, int N, size_t workspace_bytes);
int frcnn_topk_sort(const float* scores, int B, int N, int K,
                    const float* anchors, const float* deltas,
                    float* top_scores, int32_t* top_idx, float* cand_boxes,
                    void* workspace, size_t workspace_bytes, frcnn_stream_t stream);

/* ---- a6 (second part): greedy NMS ------------------------------------------------
 * Replaces tf.image.combined_non_max_suppression(boxes, scores, post_n, post_n,
 * iou_threshold), Faster_RCNN/module/model.py:51 (= module/data.py:209), one
 * class.  boxes [B,K,4] already in descending-score order (ties: earlier
 * first).  Bitmask kernel (wavefront ballot) + scan, in chunks of 1024
 * candidates with per-image early exit.
 * proposed [B,post_n,4] = kept boxes clipped to [0,1], zero padded;
 * keep_pos [B,post_n] i32 = kept positions in 0..K-1, -1 padded (may be NULL);
 * keep_count [B] i32 (may be NULL).
 * workspace: frcnn_nms_workspace_bytes(B,K,post_n) bytes, 256-byte aligned
 * (about 256 KB + 6 x post_n bytes per image; independent of K); its
 * contents on entry do not matter (nothing has to be cleared between calls).  Only kernel
 * launches are issued, so the call can be recorded into a HIP graph. */
size_t frcnn_nms_workspace_bytes(int B, int K, int post_n);
int frcnn_nms(const float* boxes, int B, int K, float iou_threshold, int post_n,
              float* proposed, int32_t* keep_pos, int32_t* keep_count,
              void* workspace, size_t workspace_bytes, frcnn_stream_t stream);

/* The same NMS in ONE kernel launch (plus two tiny launches that clear its counters), for a few
 * images at a time (B <= 8): the reference's own inference call runs one image
 * (Faster_RCNN/model_train.py:741-743), where the chunked form above is a chain of 24 dependent
 * launches.  A persistent grid: one 1,024-thread workgroup per image resolves a chunk of 1,024
 * candidates by relaxation over all its columns at once, the other workgroups compute the mask
 * tiles of the following chunk beside it; hand-off through counters in the workspace.  Same
 * arguments and bit-identical results; workspace frcnn_nms_fused_workspace_bytes(B,K,post_n)
 * bytes (about 0.5 MB per image), 256-byte aligned, contents on entry do not matter.  The grid's
 * waits are bounded: should it ever fail to become resident as a whole (it is at most 256
 * workgroups), it drains after about a second and keep_count[b] = -1 tells. */
size_t frcnn_nms_fused_workspace_bytes(int B, int K, int post_n);
int frcnn_nms_fused(const float* boxes, int B, int K, float iou_threshold, int post_n,
                    float* proposed, int32_t* keep_pos, int32_t* keep_count,
                    void* workspace, size_t workspace_bytes, frcnn_stream_t stream);

/* The same NMS for callers that cannot allocate device memory (code under HIP graph capture,
 * tight serving loops, harnesses that forbid hipMalloc).  No counterpart in the reference.
 * Same arguments, results and kernels as frcnn_nms, minus the workspace: the masks, the kept
 * index list and the counters live in a device array built into the library
 * (FRCNN_NMS_STATIC_BYTES per GPU, held until the process ends).  The call allocates and frees
 * nothing.  Returns FRCNN_E_STATIC_LIMIT when frcnn_nms_workspace_bytes(B,K,post_n) exceeds
 * FRCNN_NMS_STATIC_BYTES (e.g. post_n = 2000: up to 61 images per call; the need does not
 * depend on K).  Nothing has to be cleared between calls (the first chunk of a call never reads
 * the array).  All calls on one GPU share the array: calls that overlap on one GPU (two
 * streams, two threads) race -- order them, or use frcnn_nms with a workspace per stream.
 * The array's address is looked up on the first call on each device; make that call outside
 * graph capture. */
#define FRCNN_NMS_STATIC_BYTES (16u << 20)
int frcnn_nms_static(const float* boxes, int B, int K, float iou_threshold, int post_n,
                     float* proposed, int32_t* keep_pos, int32_t* keep_count,
                     frcnn_stream_t stream);

/* ---- a7 + a8: fused bilinear RoI crop + max-pool, forward --------------------------
 * Replaces tf.image.crop_and_resize(fmap, boxes, box_idx, (crop_h,crop_w)),
 * Faster_RCNN/module/model.py:53-55, followed by MaxPool2D((pool,pool)),
 * Faster_RCNN/module/model.py:30, without materialising the crops.
 * fmap [B,H,W,C] through strides (elements); rois [B,R,4];
 * out [B,R,crop_h/pool,crop_w/pool,C] contiguous.  pool in {1,2}; pool = 1
 * gives the plain crop.  winner (may be NULL) [same shape] u8: which sample of
 * the pool window won, row-major, first maximum. */
int frcnn_roi_crop_pool_fwd(const float* fmap, int B, int H, int W, int C,
                            int64_t sb, int64_t sh, int64_t sw, int64_t sc,
                            const float* rois, int R, int crop_h, int crop_w, int pool,
                            float* out, uint8_t* winner, frcnn_stream_t stream);

/* The same forward with a caller-provided workspace of frcnn_roi_fwd_workspace_bytes(B,R) bytes
 * (272 per RoI + a 16-byte header, 16-byte aligned, contents on entry do not matter), for the band-resident
 * kernel: a workgroup keeps a band of one image's feature rows for 32 channels in LDS and
 * serves every RoI of the image from there, so the feature map enters each compute unit once
 * instead of once per RoI (module/model.py:53-55 and :30 as above; same results bit for bit).
 * Taken when the call has at least 768 workgroups (B * C/32 * bands: three per compute unit)
 * and at least 1024 RoIs per image, the layout is channels-last with C % 32 == 0 and the crop at
 * most 16 x 16; every other call is forwarded to frcnn_roi_crop_pool_fwd, so the entry point
 * accepts whatever that one accepts.  On the 600x1000 workload (C = 512, R = 2000) the band
 * kernel takes 3.16 ms per 64 images against 4.0 ms for the per-RoI kernel, and loses below 16
 * images per call.  The tuning value "roi_band_min_wg" = n replaces the workgroup
 * threshold and lifts the RoI threshold (n < 0: never).  frcnn_roi_fwd_plan tells which kernel
 * a call with a channels-last, 16-byte aligned map and a full workspace would run (1: band,
 * 0: per RoI).
 * Planar feature maps ([B,C,H,W] contiguous, i.e. sw == 1, sh == W, sc == H*W, sb == C*H*W): when
 * the workspace holds B*H*W*C*4 bytes beyond frcnn_roi_fwd_workspace_bytes(B,R), the map is
 * transposed into it once (frcnn_nchw_to_nhwc) and the channels-last kernels run on the copy;
 * with a smaller workspace the any-stride kernel runs on the original. */
size_t frcnn_roi_fwd_workspace_bytes(int B, int R);
int frcnn_roi_fwd_plan(int B, int H, int W, int C, int R, int crop_h, int crop_w, int pool,
                       int want_winner);
int frcnn_roi_crop_pool_fwd_ws(const float* fmap, int B, int H, int W, int C,
                               int64_t sb, int64_t sh, int64_t sw, int64_t sc,
                               const float* rois, int R, int crop_h, int crop_w, int pool,
                               float* out, uint8_t* winner, void* workspace,
                               size_t workspace_bytes, frcnn_stream_t stream);

/* The two halves of frcnn_roi_crop_pool_fwd_ws for callers that know the RoIs before the
 * feature map is due (module/model.py:51 produces them, :53-55 consumes them): frcnn_roi_fwd_prepare
 * writes the per-RoI sample geometry into the workspace if a channels-last call of these
 * dimensions would run the band kernel (returns 1; 0: nothing to prepare; < 0: error), e.g. on the
 * stream that computed the proposals; frcnn_roi_crop_pool_fwd_prepared is frcnn_roi_crop_pool_fwd_ws
 * without that step (same arguments, same rois, ordered after the prepare call by the caller; call it
 * only after prepare returned 1 for the same dimensions and winner choice).  The geometry carries a
 * header with the dimensions it was made for: a band kernel that finds another header (a workspace
 * prepare never wrote, or wrote for another call) returns without writing `out` rather than follow
 * foreign descriptors.  A planar map is transposed by the second call as before. */
int frcnn_roi_fwd_prepare(const float* rois, int B, int H, int W, int C, int R, int crop_h,
                          int crop_w, int pool, int want_winner, void* workspace,
                          size_t workspace_bytes, frcnn_stream_t stream);
int frcnn_roi_crop_pool_fwd_prepared(const float* fmap, int B, int H, int W, int C,
                                     int64_t sb, int64_t sh, int64_t sw, int64_t sc,
                                     const float* rois, int R, int crop_h, int crop_w, int pool,
                                     float* out, uint8_t* winner, void* workspace,
                                     size_t workspace_bytes, frcnn_stream_t stream);

/* [B,C,H,W] contiguous -> [B,H,W,C] contiguous (dst != src), tiled through LDS: the layout
 * change in front of the channels-last RoI kernels for callers whose backbone emits planar
 * maps (the reference's maps are channels-last already, module/model.py:12-17). */
int frcnn_nchw_to_nhwc(const float* src, int B, int C, int H, int W, float* dst,
                       frcnn_stream_t stream);

/* ---- a13: adjoint of a7 + a8 ---------------------------------------------------------
 * The reference never differentiates the crop (stop_gradient,
 * Faster_RCNN/module/model.py:71); this is the exact adjoint, atomic-free: each
 * feature cell is owned by one thread group which gathers its contributions in
 * the fixed order RoI ascending, sample (sy,sx) row-major, corner TL,TR,BL,BR.
 * dY/winner as written by the forward; dfmap [B,H,W,C] contiguous channels-last, fully
 * overwritten.  C % 4 == 0; crops above 32 x 32 need C <= 2048. */
int frcnn_roi_crop_pool_bwd(const float* dY, const uint8_t* winner, const float* rois,
                            int B, int R, int H, int W, int C,
                            int crop_h, int crop_w, int pool,
                            float* dfmap, frcnn_stream_t stream);

/* ---- a9 + a10: proposal <-> GT labelling and target encode ------------------------------
 * Replaces Faster_RCNN/module/data.py:215-259.  proposed [B,R,4], gt [B,G,4],
 * gt_label_in [B,G] i64.  Outputs: p_label [B,R] i64 (bg_label = background),
 * gt_mask [B,R] i32, gt_onehot [B,R,num_classes] f32 (may be NULL),
 * gt_coord [B,R,4] f32, best_gt [B,R] i32 (may be NULL), max_iou [B,R] f32
 * (may be NULL; the clipped, floored IoU the label is decided on). */
int frcnn_label_proposals(const float* proposed, const float* gt, const int64_t* gt_label_in,
                          int B, int R, int G, float fg_thr, float iou_floor,
                          int bg_label, int num_classes,
                          int64_t* p_label, int32_t* gt_mask, float* gt_onehot,
                          float* gt_coord, int32_t* best_gt, float* max_iou,
                          frcnn_stream_t stream);

/* ---- a11: proposal sampling ---------------------------------------------------------------
 * Replaces Faster_RCNN/module/data.py:266-316 for any B.  Per image:
 * n_pos = min(#(p_label != bg), max_pos) shuffled positives, batch - n_pos
 * shuffled negatives, concatenation shuffled.  Keys as in frcnn_rpn_sample
 * (keys_pos/keys_neg [B,R], keys_all [B,batch]; all NULL -> Philox).
 * tindex [B,batch] i64; counts [B,4] i32 = {P, Nneg, n_pos_taken, status}. */
int frcnn_sample_proposals(const int64_t* p_label, int B, int R, int bg_label,
                           const uint32_t* keys_pos, const uint32_t* keys_neg,
                           const uint32_t* keys_all, uint64_t seed, int image_base,
                           int batch, int max_pos,
                           int64_t* tindex, int32_t* counts, frcnn_stream_t stream);

/* ---- next row f1: detection post-process ------------------------------------------------------
 * Replaces Faster_RCNN/model_train.py:745-759 (per image): class argmax, gather of that
 * class's offsets, generate_coord (module/data.py:179-198), drop background,
 * tf.image.non_max_suppression_with_scores(coord, score, max_out, iou_threshold) (hard NMS;
 * equal scores: lower proposal index first), keep score >= score_threshold.
 * proposed [B,R,4], cls_prob [B,R,num_classes], reg [B,R,num_classes,4].
 * decoded [B,R,4] and cls_arg [B,R] receive every proposal's box / class; boxes
 * [B,max_out,4], scores [B,max_out], classes / src [B,max_out] i32 (-1 padded; src =
 * proposal index), count [B].  R <= 16384, max_out <= 1024. */
int frcnn_detect_postprocess(const float* proposed, const float* cls_prob, const float* reg,
                             int B, int R, int num_classes, int bg_label, int max_out,
                             float iou_threshold, float score_threshold,
                             float* decoded, int32_t* cls_arg, float* boxes, float* scores,
                             int32_t* classes, int32_t* src, int32_t* count,
                             frcnn_stream_t stream);

/* ---- next row f2: loss assembly ---------------------------------------------------------------
 * RPN side.  Replaces Faster_RCNN/model_train.py:626-635 with the losses of :598-600
 * (tf.keras.losses.BinaryCrossentropy on probabilities; Loss_bbr, module/loss.py:4-14):
 * gather pred_obj [B,H*W*A] and pred_reg [B,H*W*A,4] at pos [B,S,3] i64 = (y,x,a) (the
 * batch_pos of frcnn_rpn_sample); label [B,S] f32 (1 / 0); reg_gt [B,S,4].
 *   loss[0] = mean over B*S of the binary cross-entropy (clip to [1e-7, 1-1e-7], +1e-7 in
 *             the logs, as Keras does);
 *   loss[1] = mean over the 4*P coordinates of the P samples with label 1 of smooth-L1
 *             (threshold 1); 0 when P = 0;
 *   loss[2] = 2*loss[0] + loss[1].
 * partial [B,4] f32 receives per image {BCE sum, smooth-L1 sum, P, #out-of-range positions}.
 * d_obj [B,H*W*A] / d_reg [B,H*W*A,4] (either may be NULL): d loss[2] / d prediction, zero at
 * anchors outside the mini-batch (written in full by the call). */
int frcnn_rpn_loss(const float* pred_obj, const float* pred_reg, const int64_t* pos,
                   const float* label, const float* reg_gt, int B, int H, int W, int A, int S,
                   float* loss, float* partial, float* d_obj, float* d_reg,
                   frcnn_stream_t stream);

/* Head side.  Replaces Faster_RCNN/model_train.py:639-657 (CategoricalCrossentropy on
 * probabilities: row normalised by its sum, clipped to [1e-7, 1-1e-7]):
 * cls_prob [B,R,K], reg [B,R,K,4] gathered at tindex [B,S] i64 and, for the regression, at
 * class argmax(label) (first maximum); label [B,S,K] f32; gt_mask [B,S] i32 (non-zero =
 * foreground); gt_coord [B,S,4] (clipped to [-100,100] here, model_train.py:643).
 * loss / partial as above (means over B*S rows and over 4*P coordinates).
 * d_cls [B,R,K] / d_reg [B,R,K,4] (either may be NULL): d loss[2] / d prediction, written in
 * full.  K <= 64. */
int frcnn_head_loss(const float* cls_prob, const float* reg, const int64_t* tindex,
                    const float* label, const int32_t* gt_mask, const float* gt_coord,
                    int B, int R, int S, int K, float* loss, float* partial,
                    float* d_cls, float* d_reg, frcnn_stream_t stream);

/* ---- next row f3: input side -------------------------------------------------------------------
 * Replaces Faster_RCNN/module/data.py:9-52 (data_preprocess / augmentation_data /
 * random_augmentation) and the padding of shipdetection/train.py:36-37 for a batch:
 * boxes [total,4] f32 and labels [total] i64 (may be NULL) hold every image's objects back to
 * back, image b owning rows offsets[b] .. offsets[b+1]-1 (offsets [B+1] i32).  Output
 * gt [B,G,4] / gt_label [B,G] (may be NULL): the first G objects, zero padded; count [B] i32
 * (may be NULL) = objects the image has (more than G: the rest is dropped; the reference
 * raises on a negative pad there).  With use_flip != 0 the boxes of a flipped image become
 * [ymin, 1-xmax, ymax, 1-xmin] BEFORE padding (padding rows stay 0).  Flip decision:
 * flip [B] u8, or NULL -> Philox counter key (stream 6, image image_base+b) > 2^31, the
 * "uniform > 0.5" of module/data.py:44-45. */
int frcnn_pad_gt(const float* boxes, const int64_t* labels, const int32_t* offsets,
                 const uint8_t* flip, int use_flip, uint64_t seed, int image_base,
                 int B, int G, float* gt, int64_t* gt_label, int32_t* count,
                 frcnn_stream_t stream);

/* tf.image.flip_left_right on the images the same decision flips (module/data.py:25):
 * images / out [B,H,W,C] f32 contiguous, out != images; images not flipped are copied. */
int frcnn_flip_images(const float* images, const uint8_t* flip, uint64_t seed, int image_base,
                      int B, int H, int W, int C, float* out, frcnn_stream_t stream);

/* ---- next row f4: the head's first dense layer on the pooled tensor ---------------------------
 * Replaces TimeDistributed(Flatten) + TimeDistributed(Dense(4096, activation='relu')),
 * Faster_RCNN/module/model.py:31-32, on the tensor frcnn_roi_crop_pool_fwd writes: a RoI's
 * [ph,pw,C] block IS its flattened (h,w,c) feature row, so x = pooled viewed as [M = B*R,
 * K = ph*pw*C] and no transpose or copy sits between the pool and the GEMM.
 * w [K,N] row-major (the Keras kernel layout), bias [N] (may be NULL), out [M,N];
 * relu != 0 applies max(., 0).  float32 in and out on the float32 matrix cores
 * (v_mfma_f32_32x32x2_f32): out[m][n] = act(fmaf-chain over k = 0..K-1 of x[m][k]*w[k][n],
 * from 0, one rounding per product, + bias[n]) -- one accumulator per output, k ascending, no
 * split over k, so the result does not depend on tile sizes.  K % 4 == 0, N % 4 == 0; x and w
 * 16-byte aligned.  No workspace. */
int frcnn_fc1_fwd(const float* x, const float* w, const float* bias, int64_t M, int K, int N,
                  int relu, float* out, frcnn_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* FRCNN_HIP_H */
