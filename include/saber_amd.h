/* saber_amd.h - C-ABI of the MI355X-native slice-wise SAM2 engine.
 *
 * Drop-in boundary for the hot path of chanzuckerberg/saber.  Each entry point
 * names the reference interface it replaces (paths relative to the reference tree):
 *
 *   saber_engine_create / set_weight / finalize
 *       <- build_sam2(cfg, ckpt, device, apply_postprocessing=True)   saber/adapters/sam2/automask.py:61-63
 *          get_sam2_checkpoint(cfg)                                   saber/pretrained_weights.py:174-202
 *   saber_prepare
 *       <- prep.prepare(image, to_rgb=True)                           saber/utils/preprocessing.py:67-80
 *          (contrast :4-18, normalize :20-37)
 *   saber_encode
 *       <- SAM2ImagePredictor.set_image(crop)  (third-party sam2; call site automask.py:66 / predictor.py:70)
 *   saber_decode_points
 *       <- SAM2ImagePredictor._predict(points, labels, mask_input, multimask_output, return_logits=True)
 *   saber_amg_generate
 *       <- FilteredSAM2MaskGenerator.generate -> SAM2AutomaticMaskGenerator.generate
 *                                                                     saber/adapters/sam2/amg.py:161-183
 *                                                                     saber/adapters/sam2/predictor.py:70
 *   saber_label_plane
 *       <- the paint loop of propagationSegmenter.slice_by_slice      saber/segmenters/propagation.py:185-186
 *   saber_separate_masks
 *       <- separate_masks(vol_masks, min_mask_area)                   saber/segmenters/utils.py:88-131
 *   saber_classifier_*
 *       <- Predictor(model_config, model_weights).predict / SAM2Classifier.forward
 *                                                                     saber/classifier/models/predictor.py:117-175, SAM2.py:118-197
 *   saber_smooth_labels / saber_gaussian_smoothing_3d
 *       <- fast_3d_gaussian_smoothing(volume, scale, deviceID)        saber/filters/masks.py:230-287
 *          gaussian_smoothing_3d(volume, sigma, device)               saber/filters/gaussian.py:76-138
 *
 * Conventions: plain pointers and sizes only; every *_dev pointer is device memory owned by the
 * caller (e.g. a PyTorch-ROCm tensor's data_ptr()); `stream` is a hipStream_t (NULL = default
 * stream).  Every function returns 0 on success or a negative saber_status; the message is kept
 * per handle (saber_last_error).  Nothing aborts or exits.  A handle is bound to one device and
 * must not be used concurrently from two threads; different handles are independent
 * (reference threading contract: saber/utils/parallelization.py:100-135).
 */
#ifndef SABER_AMD_H
#define SABER_AMD_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct saber_engine saber_engine;

enum saber_status {
    SABER_OK = 0,
    SABER_ERR_INVALID = -1,   /* bad argument / unsupported configuration (ValueError in the reference) */
    SABER_ERR_STATE = -2,     /* call order violated (RuntimeError in the reference)                    */
    SABER_ERR_HIP = -3,       /* HIP runtime failure                                                     */
    SABER_ERR_CAPACITY = -4,  /* caller-provided output capacity exceeded                                */
    SABER_ERR_RANGE = -5      /* overflow sentinel: NaN / inf in the 16-bit arithmetic (saber_engine_check_finite) */
};

enum saber_dtype { SABER_U16 = 0, SABER_F32 = 1 };

/* cfgAMG (saber/adapters/sam2/amg.py:7-17) + the un-passed upstream defaults of
 * SAM2AutomaticMaskGenerator (mask_threshold 0.0, crop_nms_thresh 0.7, crop_overlap_ratio 512/1500). */
typedef struct saber_amg_params {
    int points_per_side;
    int points_per_batch;
    float pred_iou_thresh;
    float stability_score_thresh;
    float stability_score_offset;
    float mask_threshold;
    float box_nms_thresh;
    int crop_n_layers;
    float crop_nms_thresh;
    float crop_overlap_ratio;
    int crop_n_points_downscale_factor;
    int use_m2m;
    int multimask_output;
} saber_amg_params;

/* One AMG record: the non-array fields of the SAM-AMG dict consumed downstream
 * (saber/segmenters/base.py:128-133, saber/segmenters/utils.py:36-39,70). */
typedef struct saber_mask_meta {
    int32_t area;
    float bbox_xywh[4];
    float predicted_iou;
    float stability_score;
    float point_xy[2];
    float crop_box_xywh[4];
} saber_mask_meta;

/* trunk: "tiny" | "small" | "base" | "large" (the four sam2.1 Hiera trunks SABER can name, saber/adapters/base.py:28-33;
 * anything else is SABER_ERR_INVALID).  max_images: crops encoded per batched pass / feature slots kept resident.
 * max_prompts: prompts decoded per batched pass. */
int saber_engine_create(int device_id, const char* trunk, int max_images, int max_prompts, saber_engine** out);
void saber_engine_destroy(saber_engine* e);
/* e == NULL returns the calling thread's last create-time error. */
const char* saber_last_error(const saber_engine* e);

/* Upload one fp32 tensor under its upstream checkpoint key (e.g. "image_encoder.trunk.blocks.0.attn.qkv.weight"). */
int saber_engine_set_weight(saber_engine* e, const char* name, const float* host_data, const int64_t* shape, int ndim);
/* Check completeness, build bf16 / fused tables.  Must precede any compute call. */
int saber_engine_finalize(saber_engine* e);

/* K0.  img_dev: (H,W) uint16 or float32.  out_dev: (H,W) float32 in [0,1]. */
int saber_prepare(saber_engine* e, const void* img_dev, int dtype, int H, int W, float* out_dev, void* stream);
/* K0 for an (H,W,3) float32 array, as the reference's segment_image_2d treats RGB input (saber/adapters/sam2/predictor.py:58-59 ->
 * saber/utils/preprocessing.py:67-80): uniform_filter(size=500, reflect) over all three axes of the array (channel axis included),
 * clip +-3 sigma, ONE global min/max.  img_dev, out_dev: (H,W,3) interleaved float32.  Like saber_prepare it needs a created
 * handle only (no weights). */
int saber_prepare_rgb(saber_engine* e, const float* img_dev, int H, int W, float* out_dev, void* stream);

/* Encode n crops of one image.  img_dev: (H,W) [channels==1] or (H,W,3) float32 in [0,1];
 * crop_boxes_host: n x [x0,y0,x1,y1].  Features stay resident in slots slot0..slot0+n-1.
 * channels == -1: one grey plane that already carries the model's input normalisation; it is replicated to three channels WITHOUT the
 * ImageNet statistics (the video path's frames, saber/adapters/preprocessing.py:55-56; a stack of Z frames is a (Z*1024, 1024) image
 * whose crop boxes are the frames, so a window of frames is one batched pass). */
int saber_encode(saber_engine* e, const float* img_dev, int H, int W, int channels, const int* crop_boxes_host, int n,
                 int slot0, void* stream);
/* Copy a slot's features out as NCHW fp32: image_embed (256,64,64), feat_s0 (32,256,256), feat_s1 (64,128,128).
 * Any pointer may be NULL. */
int saber_get_features(saber_engine* e, int slot, float* image_embed_dev, float* feat_s0_dev, float* feat_s1_dev, void* stream);

/* Video (memory) path, SURVEY.md 8f-1 (reference: saber/adapters/sam2/predictor.py:232-348 driving upstream SAM2Base.track_step): a tracked
 * frame's image embedding is replaced by its memory-conditioned version before the mask decoder runs.  Tokens cross this boundary as
 * (4096, 256) fp32 in row-major (y * 64 + x) order.  get: the slot's image_embed (with no_mem_embed, as saber_encode leaves it).
 * set: overwrite it (the slot keeps its high-resolution features).  saber_get_decoder_tokens: the (n, 8, 256) output tokens of the last
 * saber_decode_points call ([obj, iou, mask0..3, point, pad]); upstream projects one mask token to the object pointer. */
int saber_get_embed_tokens(saber_engine* e, int slot, float* out_tokens_dev, void* stream);
int saber_set_embed_tokens(saber_engine* e, int slot, const float* tokens_dev, void* stream);
int saber_get_decoder_tokens(saber_engine* e, int n, float* out_dev, void* stream);

/* Slot state in and out of a handle (SURVEY.md 8e "Propagation path": the per-frame image encodes of one tomogram are independent and
 * shard over ranks; the tracking chain that consumes them is sequential.  Upstream keeps these per-frame features in
 * inference_state["cached_features"], sam2_video_predictor.py).  n consecutive slots from slot0, each as three fp32 device arrays in the
 * engine's token order: image_embed 4096 x 256, feat_s1 16384 x 64, feat_s0 65536 x 32.  import marks the slots as encoded; only
 * meaningful between handles built from the same model configuration and weights. */
int saber_export_slots(saber_engine* e, int slot0, int n, float* emb_dev, float* fs1_dev, float* fs0_dev, void* stream);
int saber_import_slots(saber_engine* e, int slot0, int n, const float* emb_dev, const float* fs1_dev, const float* fs0_dev, void* stream);

/* Decode n point prompts against a slot.  pts_dev: (n,2) in model pixels (0..1024); labels_dev: (n) or NULL (=1).
 * mask_in_dev: (n,256,256) low-res logits or NULL.  multimask: 3 masks per prompt, else 1 (dynamic selection).
 * Outputs: lowres (n,M,256,256), iou (n,M), obj (n) - any may be NULL. */
int saber_decode_points(saber_engine* e, int slot, const float* pts_dev, const int* labels_dev, int n, int multimask,
                        const float* mask_in_dev, float* out_lowres_dev, float* out_iou_dev, float* out_obj_dev, void* stream);

/* The same with SEVERAL points per prompt (reference: upstream SAM2's prompt encoder as SAM2VideoPredictor.add_new_points_or_box feeds it -
 * clicks with labels 1 / 0, a box as its two corners with labels 2 / 3 in front of the clicks, label -1 = not a point; one padding point is
 * appended by the engine as upstream does when no box tensor is passed).  pts_dev: (n, points_per_prompt, 2), labels_dev: (n, points_per_prompt).
 * points_per_prompt == 1 is saber_decode_points.  More than one point makes 8 + (points_per_prompt - 1) decoder tokens per prompt: the bf16
 * kernels are built for 8, so such prompts are decoded in the EXACT precision mode only (SABER_ERR_STATE otherwise). */
int saber_decode_prompts(saber_engine* e, int slot, const float* pts_dev, const int* labels_dev, int n, int points_per_prompt, int multimask,
                         const float* mask_in_dev, float* out_lowres_dev, float* out_iou_dev, float* out_obj_dev, void* stream);

/* Automatic mask generation for one image.  out_bits_dev: (max_masks, H, ceil(W/32)) uint32, bit b of word w of row y
 * = pixel (y, 32w+b).  out_meta_host: max_masks records.  Synchronises the stream before returning (a count is returned).
 * More than max_masks results: SABER_ERR_CAPACITY with *out_count = the capacity needed (nothing is written).
 * params->points_per_batch is accepted for cfgAMG parity and does NOT change any result: prompts are independent of each other in
 * SAM2's decoder, so the engine decodes them in batches of its own max_prompts (upstream batches only to bound memory). */
int saber_amg_generate(saber_engine* e, const float* img_dev, int H, int W, int channels, const saber_amg_params* params,
                       uint32_t* out_bits_dev, int max_masks, saber_mask_meta* out_meta_host, int* out_count, void* stream);

/* Weight format of the Hiera stage-2 / stage-3 block GEMMs (qkv, proj, fc1, fc2: 94 % of the encoder's weights), to be chosen before
 * saber_engine_finalize.  SABER_WEIGHTS_FP8_E4M3 (BASELINE configs[4], "fp8 weights"): every output row is quantised to OCP e4m3fn with
 * one power-of-two scale per row (round to nearest even, saturating at 448).  The MFMA operands stay bf16 (activations are bf16, and the
 * quantised value times its scale is exact in bf16), accumulation fp32: this is the numerics of an fp8-weight checkpoint, not an
 * fp8-operand kernel.
 * SABER_WEIGHTS_MXFP8 (BASELINE configs[4], "fp8 weights on CDNA4 fp8 MFMA"; Hiera-L): qkv of the blocks that keep their width and both MLP
 * layers of every stage-2 / stage-3 block are stored in the OCP MX format (e4m3fn elements, one e8m0 power-of-two scale per 32
 * K-elements: the smallest with amax <= 448 * scale) and RUN on v_mfma_scale_f32_16x16x128_f8f6f4 (csrc/gemm_fp8.hip, twice the bf16
 * MFMA rate): their activation operands are quantised to the same format where they are produced (LayerNorm -> MX, GELU epilogue ->
 * MX), accumulation and the residual stream stay fp32, attention and attn.proj stay bf16.
 * Both are narrower than the reference's precision: never the default; their price against fp32 is reported by tests/test_gpu_fp8.py. */
#define SABER_WEIGHTS_BF16 0
#define SABER_WEIGHTS_FP8_E4M3 1
#define SABER_WEIGHTS_MXFP8 2
int saber_engine_set_weight_format(saber_engine* e, int format);

/* Arithmetic precision of the model (encoder + prompt / mask decoder; everything around them is fp32 or integer in either mode).
 * SABER_PRECISION_BF16 (default, the production path): bf16 MFMA operands, fp32 accumulation, fp32 residual stream and statistics -
 *   3-8e-3 rel-RMS from the reference's fp32 arithmetic after 48 Hiera blocks.
 * SABER_PRECISION_EXACT: every operand, stored activation and statistic in fp32 (GEMMs on the fp32-input MFMA, exact-erf GELU, the mask
 *   decoder as the unfolded composition upstream executes): the reference's own precision (saber/utils/io.py:127-132 runs fp32, autocast
 *   commented out), ~1e-6 from the fp32 CPU oracle; eight to nine times slower (1.2 s per cfgAMG-default slice) - a verification mode, never the default.
 * SABER_PRECISION_FP16 (round 4): the production kernels compiled for IEEE half operands (v_mfma_f32_16x16x32_f16: the bf16 forms' rate)
 *   - 10 mantissa bits on every GEMM / attention operand with fp32 accumulation, i.e. the operand width of the TF32 arithmetic the
 *   reference enables on its GPUs (saber/utils/io.py:127-130).  Same kernels, schedules, token order, workspaces and C-ABI as bf16; weights
 *   and stored activations are fp16.  What fp16 gives up is RANGE (largest finite value 65 504): saber_engine_finalize fails loudly when a
 *   weight, or a LayerNorm output bound |gamma| sqrt(C) + |beta|, cannot be represented, and at RUN time the overflow sentinel below
 *   (saber_engine_check_finite; built into saber_amg_generate) turns an activation that left the range into SABER_ERR_RANGE instead of
 *   NaN masks.  Conversions do NOT saturate: a clamped activation would be a silently wrong result.  The choice is made ONCE, before
 *   saber_engine_finalize (the weights are converted there); a handle finalized in one 16-bit type cannot be switched to the other.
 *   Not available together with the fp8 weight formats.
 * Call it with EXACT once BEFORE saber_engine_finalize (the fp32 weight copies are kept only then: +0.9 GB for Hiera-L); afterwards the
 * mode can be switched back and forth between EXACT and the handle's 16-bit type between calls on the same handle (EXACT then FP16 before
 * finalize makes a handle with both).  hipGraph replay is bypassed in exact mode. */
#define SABER_PRECISION_BF16 0
#define SABER_PRECISION_EXACT 1
#define SABER_PRECISION_FP16 2
int saber_engine_set_precision(saber_engine* e, int precision);

/* Overflow sentinel of the 16-bit modes (round 5).  A stored fp16 activation beyond 65 504 is an inf from there on: fp32
 * accumulators, the fp32 residual stream and every LayerNorm / softmax statistic carry it (as inf or NaN) into (a) the three feature maps of
 * the encoder pass, (b) the decoder batch's predicted IoUs / hypernetwork outputs and (c) the low-res logits.  The engine counts non-finite
 * values there on the device, on the caller's stream: one HBM-bound scan of the feature maps per encoder pass (16 MB per crop), two tiny
 * scans per decoder batch, one fma per stored pixel inside dec_upscale.  saber_amg_generate reads the counters with the records at its one
 * synchronisation and returns SABER_ERR_RANGE (nothing of that call's output may be used).  After saber_encode / saber_decode_points /
 * saber_decode_prompts (asynchronous, no synchronisation inside) call saber_engine_check_finite where the host synchronises anyway: it
 * waits for `stream`, returns SABER_ERR_RANGE with a message naming the stage when any counter is non-zero, and clears the counters.
 * bf16 handles run the same checks (their range is fp32's: a hit means NaN / inf in the input or the weights); the exact mode does not. */
int saber_engine_check_finite(saber_engine* e, void* stream);

/* hipGraph replay of saber_amg_generate's launch sequences (BASELINE configs[4]: "hipGraph-captured per-slice encode+decode"): the batched
 * encoder pass and each decoder batch are run eagerly the first time their shapes are seen on a handle, captured the second time and
 * replayed from then on (needs a non-default stream; on by default, SABER_AMD_GRAPHS=0 or saber_engine_set_graphs(e, 0) turns it off).
 * saber_engine_graph_stats: sequences captured / replayed so far on this handle. */
int saber_engine_set_graphs(saber_engine* e, int enable);
/* Co-residency experiment (round 4): the batched encoder passes of saber_amg_generate run on `stream` instead of the
 * call's own stream (fenced by events on both sides: results are identical); NULL restores the default.  With CU-masked streams
 * (saber_k_stream_create_cu_range) one handle's MFMA-bound encoder and another handle's HBM-bound decoder can be given disjoint sets of CUs. */
int saber_engine_set_encoder_stream(saber_engine* e, void* stream);
int saber_engine_graph_stats(const saber_engine* e, int* captures, int* replays);

/* Host synchronisations (hipStreamSynchronize) the last saber_amg_generate call on this handle needed: 2 per group of crops decoded
 * together + 1 at the end (7 for cfgAMG's default 1 + 4 + 16 crop pyramid), more only when a scratch buffer had to grow. */
int saber_amg_last_syncs(const saber_engine* e);
/* Where the generator's post-processing runs: 1 (default) = on the device - IoU / stability / crop-edge filters, the
 * per-crop box NMS, the cross-crop NMS and the compaction of the survivors are kernels on the caller's stream and a slice costs ONE host
 * synchronisation; 0 = the host restatement of round 1-2 (3 synchronisations).  Identical masks, order and records either way
 * (tests/test_gpu_graphs.py); generators with more than 12 288 candidates per image use the host path. */
int saber_engine_set_device_amg(saber_engine* e, int enable);
/* IoU pruning of the m2m pass (on by default; results are identical either way): a refined candidate reports its mask 0's IoU prediction or
 * the best of the other three (dynamic multimask selection, upstream sam2 MaskDecoder._dynamic_multimask_via_stability); when all four
 * predictions are <= params->pred_iou_thresh it cannot pass the `predicted_iou > pred_iou_thresh` filter of the mask generator
 * (upstream automatic_mask_generator._process_batch), so its masks are neither upscaled nor read.  saber_amg_last_pruning: how many of the
 * last saber_amg_generate call's m2m candidates were skipped. */
int saber_engine_set_iou_pruning(saber_engine* e, int enable);
int saber_amg_last_pruning(const saber_engine* e, int64_t* pruned, int64_t* m2m_candidates);

/* plane[y][x] = (position in order_host)+1 of the LAST mask covering the pixel, 0 if none. */
int saber_label_plane(saber_engine* e, const uint32_t* bits_dev, const int* order_host, int n, int H, int W,
                      uint16_t* plane_dev, void* stream);

/* out_inter_dev[i*n+j] = |mask_i AND mask_j| in pixels for the n bit-packed masks: the integer counts behind
 * remove_duplicate_masks' IoU (saber/segmenters/utils.py:21-29); IoU = inter / (area_i + area_j - inter). */
int saber_mask_pair_intersections(saber_engine* e, const uint32_t* bits_dev, int n, int H, int W, int32_t* out_inter_dev, void* stream);

/* 3-D connected components of the stitched label volume (replaces saber.segmenters.utils.separate_masks,
 * saber/segmenters/utils.py:88-131, the last step of propagationSegmenter.slice_by_slice, propagation.py:189): foreground =
 * plane value != 0, 26-connectivity, components below min_mask_area * 10 voxels dropped (utils.py:113-119), survivors numbered
 * 1..K in scipy.ndimage.label's order.  planes_dev: (Z,H,W) uint16, out_dev: (Z,H,W) uint32, both on the engine's device;
 * Z*H*W < 2^31.  Synchronises the stream (the label count is returned). */
int saber_separate_masks(saber_engine* e, const uint16_t* planes_dev, int Z, int H, int W, int min_mask_area, uint32_t* out_dev,
                         int* out_n_labels, void* stream);

/* Per-label adaptive 3-D Gaussian smoothing of a label volume (replaces saber.filters.masks.fast_3d_gaussian_smoothing,
 * saber/filters/masks.py:230-287; applied to the segmenter's output by segment_tomogram_core, saber/entry_points/inference_core.py:68-74
 * with scale = 0.05).  For each label value v != 0, ascending: sigma = scale * 2 (3 |vol == v| / 4 pi)^(1/3) (masks.py:289-309), separable
 * zero-padded Gaussian of int(6 sigma + 1) (made odd) taps along x, y, z in fp32 (gaussian.py:97-131), out[field > 0.5] = (uint8) v,
 * later labels overwriting earlier ones.  labels_dev: (Z,H,W) of elem_bytes 1, 2 or 4 (unsigned), values <= 2^22; out_dev: (Z,H,W)
 * uint8 (label values wrap modulo 256 exactly as the reference's uint8 result array does).  *out_n_labels = labels found.
 * Only a created handle is needed (no weights / finalize).  Synchronises the stream (per-label statistics come back to the host). */
int saber_smooth_labels(saber_engine* e, const void* labels_dev, int elem_bytes, int Z, int H, int W, double scale, uint8_t* out_dev,
                        int* out_n_labels, void* stream);

/* The separable filter itself on one 0/1 mask with a given sigma (saber/filters/gaussian.py:76-138): mask_dev (Z,H,W) uint8,
 * out_dev (Z,H,W) float32. */
int saber_gaussian_smoothing_3d(saber_engine* e, const uint8_t* mask_dev, int Z, int H, int W, double sigma, float* out_dev, void* stream);

/* ---- domain-expert classifier filter on the engine's image embeddings ----
 * Replaces saber.classifier.models.predictor.Predictor (saber/classifier/models/predictor.py:9-60 construction, :117-175 predict) with the
 * SAM2Classifier model (saber/classifier/models/SAM2.py:21-197) behind it; hook: saber2D._apply_classifier -> filters.apply_classifier
 * (saber/segmenters/base.py:159-176, saber/filters/masks.py:8-21).
 * A classifier is bound to a finalized engine handle (its Hiera encoder is the classifier's frozen backbone) and is used from the same
 * thread as that handle.  Weights: the state_dict entries of SAM2Classifier under their own names ("projection.0.weight", ...,
 * "projection.1.running_var", "projection.2.weight", ..., "classifier.4.bias"; num_batches_tracked is not needed), fp32 host arrays.
 * Eval-mode BatchNorm is folded into the convolutions at finalize. */
typedef struct saber_classifier saber_classifier;
int saber_classifier_create(saber_engine* e, int num_classes, saber_classifier** out);
void saber_classifier_destroy(saber_classifier* c);
int saber_classifier_set_weight(saber_classifier* c, const char* name, const float* host_data, const int64_t* shape, int ndim);
int saber_classifier_finalize(saber_classifier* c);
/* Predictor.predict: image_dev (H,W) float32 (the grey image handed to apply_classifier), masks_dev (n,H,W) uint8 candidate masks.
 * Per mask: whole-image z-score (monai NormalizeIntensity), crop_and_resize_adaptive (margin 1.5, 320x320, bilinear image / nearest
 * mask), masks whose resized crop holds fewer than min_area pixels are skipped (their row of probs stays 0, predictor.py:139-141,170-173),
 * SAM2 image embedding of the crop, ROI/RONI features, head, softmax.  probs_host: (n, num_classes) float32 HOST memory.
 * Synchronises the stream (bounding boxes, areas and the probabilities come back to the host). */
int saber_classifier_predict(saber_classifier* c, const float* image_dev, int H, int W, const uint8_t* masks_dev, int n, int min_area,
                             float* probs_host, void* stream);
/* SAM2Classifier.forward after the backbone, on the embeddings engine slots 0..k-1 hold now (saber_encode / saber_set_embed_tokens):
 * mask_crops_dev (k,320,320) uint8 -> probs_host (k, num_classes). */
int saber_classifier_head(saber_classifier* c, const uint8_t* mask_crops_dev, int k, float* probs_host, void* stream);
/* The 320x320 image crops / binarised mask crops the last saber_classifier_predict call made for its first n masks (device copies). */
int saber_classifier_get_crops(saber_classifier* c, int n, float* crops_out_dev, uint8_t* masks_out_dev, void* stream);

/* Per-launch HIP-event profiling of the engine's own kernels, by kernel class (events are recorded on the
 * stream the kernels are launched on).  Class order: 0 gemm_bf16, 1 hiera_attention, 2 layernorm,
 * 3 decoder_attention, 4 elementwise, 5 image_ops, 6 mask_post, 7 decoder_t2i, 8 decoder_i2t, 9 decoder_upscale, 10 gemm_mxfp8 (the GEMMs on the
 * fp8 MFMA, weight format SABER_WEIGHTS_MXFP8 only).  flops / bytes are ALGORITHMIC. */
typedef struct saber_profile_class { int64_t launches; double ms; double flops; double bytes; } saber_profile_class;
#define SABER_PROFILE_CLASSES 11
int saber_profile_begin(saber_engine* e);
int saber_profile_end(saber_engine* e, saber_profile_class* out, int n_classes);

/* Algorithmic work counters (FLOPs per call) for roofline reporting. */
double saber_encoder_flops(const saber_engine* e);
double saber_decoder_flops_per_prompt(void);

#ifdef __cplusplus
}
#endif
#endif
