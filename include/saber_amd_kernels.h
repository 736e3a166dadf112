/* saber_amd_kernels.h - kernel-level C-ABI (plain device pointers, no engine handle).
 *
 * These expose the individual gfx950 kernels so that parity tests can check each one against the
 * CPU oracle in isolation, and so that an integrator can reuse e.g. the GEMM or the K0/K8 kernels.
 * Each returns 0 or -1; on -1 saber_k_last_error() holds the message (thread-local).
 * The torch ops they stand in for (executed by the third-party sam2 package under the reference
 * call site saber/adapters/sam2/predictor.py:70) are named per function; see SURVEY.md 8a.
 */
#ifndef SABER_AMD_KERNELS_H
#define SABER_AMD_KERNELS_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

const char* saber_k_last_error(void);
/* must be called once per device before the first kernel call (sets large-LDS attributes) */
int saber_k_init(int device_id);

/* nn.Linear / 1x1 conv:  C = epi(A[M,K] . W[N,K]^T + bias); A, W bf16 (uint16 storage), bias/res fp32.
 * act: 0 none, 1 GELU(erf), 2 ReLU, 3 sigmoid.  out_f32 / out_bf16 / bias / res may be NULL.
 * pool4: rows 4q..4q+3 max-pooled into row q. res_shift / res_mod: residual row = (row >> res_shift) % res_mod. */
int saber_k_gemm(const uint16_t* A, const uint16_t* W, const float* bias, const float* res, float* out_f32, uint16_t* out_bf16,
                 int M, int N, int K, int act, int act_last, int pool4, int res_shift, int res_mod, void* stream);

/* Same GEMM with explicit leading dimensions; w_kpad = 1 declares W rows zero-padded to a multiple of 64 in K (ldw >= padded K),
 * the layout the engine uploads weights in (enables the direct-to-LDS kernel for K = 144, 288). */
int saber_k_gemm_ld(const uint16_t* A, int lda, const uint16_t* W, int ldw, int w_kpad, const float* bias, const float* res, float* out_f32,
                    uint16_t* out_bf16, int M, int N, int K, int act, void* stream);

/* Residual step + the LayerNorm that follows it in one kernel (Hiera MultiScaleBlock: attn.proj + shortcut -> norm2, mlp.layers.1 +
 * residual -> norm1 of the next block):  y = A.W^T + bias + res -> out_f32 (and out_bf16 = bf16(y) if not NULL);
 * ln_out = bf16(LayerNorm(y) * ln_gamma + ln_beta).  N must be 144, 288 or 576 (the workgroup owns whole rows); W rows zero-padded
 * to a multiple of 64 in K (ldw >= padded K); res may alias out_f32 (in-place residual stream). */
int saber_k_gemm_rowln(const uint16_t* A, int lda, const uint16_t* W, int ldw, const float* bias, const float* res, float* out_f32,
                       uint16_t* out_bf16, const float* ln_gamma, const float* ln_beta, float ln_eps, uint16_t* ln_out, int M, int N, int K,
                       void* stream);

/* nn.LayerNorm over the last dim; fp32 in, fp32 and/or bf16 out. */
int saber_k_layernorm(const float* x, const float* gamma, const float* beta, float eps, float* out_f32, uint16_t* out_bf16,
                      int rows, int C, int act, void* stream);

/* Hiera MultiScaleAttention core on contiguous windows (head_dim 72): qkv bf16 [tokens][3*heads*72]
 * -> out bf16 [tokens_q][heads*72]; nk keys per window, q_pool: queries max-pooled over 4 consecutive rows. */
int saber_k_hiera_attention(const uint16_t* qkv, uint16_t* out, int n_windows, int nk, int heads, int q_pool, void* stream);
/* Same for every trunk: head_dim 72 (large) | 96 (tiny, small) | 56 (base+); any nk (the 14x14 / 7x7 windows of the smaller
 * trunks are 196 / 49 keys); key_mask (optional, device): one byte per key of a window, zero-padded to a multiple of 128 bytes,
 * 0 = the key takes no part (window-padding rows in the global-attention blocks of those trunks). */
int saber_k_hiera_attention_ex(const uint16_t* qkv, uint16_t* out, int n_windows, int nk, int heads, int head_dim, int q_pool,
                               const uint8_t* key_mask, void* stream);

/* two-way-transformer attention (fp32 in, bf16 out), contiguous [B][n][heads*hd]. */
int saber_k_dec_attention(const float* q, const float* k, const float* v, uint16_t* out, int B, int nq, int nk, int heads, int hd,
                          int k_shared, void* stream);

/* K0: prep.prepare on a (H,W) slice; ws_dev: 4*H*W floats of scratch, minmax_dev: 2 uint32 */
int saber_k_prepare(const void* img, int dtype, int H, int W, float* out, float* ws_dev, uint32_t* minmax_dev, void* stream);

/* K8: upsample n low-res (256x256) logit maps to a crop, counts / bbox / bit-pack.  stats: n x 8 int32
 * (area, inter, union, x0, y0, x1, y1, pad). bits: n x H x ceil(W/32). */
int saber_k_mask_post(const float* lowres, int n, int crop_x0, int crop_y0, int crop_w, int crop_h, int H, int W, float thr,
                      float offset, uint32_t* bits, int32_t* stats, void* stream);

/* ---- SAM2 video (memory) path: the per-frame arithmetic the reference's segment_volume (saber/adapters/sam2/predictor.py:232-348) makes the
 * third-party video predictor run besides encoder and mask decoder (upstream sam2/modeling/{memory_attention,memory_encoder,sam2_base}.py).
 * Channels-last fp32 tensors [pixels][C], pixels in row-major (y, x) order. */
/* axial rotary encoding of q / k (RoPEAttention): rows < n_rot are rotated (token = row % side^2 on a side x side grid, theta 10000), the
 * rest (object-pointer tokens) copied; fp32 and/or bf16 out */
int saber_k_rope(const float* x, int64_t rows, int n_rot, int C, int side, float theta, float* out_f32, uint16_t* out_bf16, void* stream);
/* P[row][0..n) = softmax(scale * S[row][0..n)) as bf16; columns n..ld_p zeroed */
int saber_k_softmax_rows(const float* S, int64_t ld_s, int64_t rows, int n, float scale, uint16_t* P, int64_t ld_p, void* stream);
/* Conv2d(k3, s2, p1) of the memory encoder's mask down-sampler; w (Cout,Cin,3,3) */
int saber_k_conv3x3s2(const float* in, int H, int W, int Cin, const float* w, const float* b, int Cout, float* out, void* stream);
/* the same with the weights pre-arranged as (3,3,Cin,Cout) (Cout a multiple of 4): the layout the tracking loop keeps per model */
int saber_k_conv3x3s2_t(const float* in, int H, int W, int Cin, const float* wt, const float* b, int Cout, float* out, void* stream);
/* plane[y][x] = label wherever logits (Hv,Wv) > thr at the nearest source pixel of the output pixel centre; any_flag (optional, device
 * int) is OR-ed with 1 when a pixel was painted.  SAM2Adapter.segment_volume's _apply, saber/adapters/sam2/predictor.py:288-298. */
int saber_k_paint_nearest(const float* logits, int Hv, int Wv, float thr, int label, uint16_t* plane, int H, int W, int* any_flag, void* stream);
/* out (n,H,W) bytes, 1 where bit (x & 31) of bits[n][y][x >> 5] is set: the bool `segmentation` arrays of SAM2AutomaticMaskGenerator's
 * dict list (saber/adapters/sam2/automask.py:50-56 hands them to the segmenters), unpacked before the copy to the host */
int saber_k_unpack_masks(const uint32_t* bits, int n, int H, int W, uint8_t* out, void* stream);
/* depth-wise Conv2d(k7, p3) of the memory fuser's ConvNeXt blocks; w (C,1,7,7) */
int saber_k_dwconv7(const float* in, int H, int W, int C, const float* w, const float* b, float* out, void* stream);
/* the video predictor's mask_downsample: Conv2d(1, 1, k4, s4) */
/* saber_k_dwconv7 with the weights given as (7,7,C) [tap][channel] */
int saber_k_dwconv7_t(const float* in, int H, int W, int C, const float* wt, const float* b, float* out, void* stream);
int saber_k_conv4x4s4(const float* in, int H, int W, const float* w, const float* b, float* out, void* stream);
/* F.interpolate(mode="bilinear", align_corners=False, antialias) of n planes, fused post transform: 0 none, 1 a*sigmoid(v)+c, 2 a*(v>0)+c, 3 a*v+c, 4 (v>=a) */
int saber_k_resize_plane(const float* in, int n_planes, int H, int W, float* out, int Ho, int Wo, int antialias, int post, float a, float c, void* stream);
/* MXFP8 GEMM on the block-scaled MFMA v_mfma_scale_f32_16x16x128_f8f6f4 (row g-1): out = act(A . W^T + bias) (+ res).  Operands in the OCP MX
 * format: e4m3fn elements ([M][lda] / [N][ldw] bytes, K contiguous, zero-padded to Kp, a multiple of 128) + one e8m0 scale byte per 32
 * K-elements, stored K-step-major: S[Kp / 128][rows][4] (rows >= the operand's rows rounded up to whole tiles: a multiple of 768 for A, of 192 for W).
 * Exactly one output: out_f32 (+ res, both with leading dimension ldc; out_bf16 may be given too and receives a copy) | out_bf16 | out_mx
 * (+ out_mx_scales [N / 128][out_mx_rows][4]: the result as the next GEMM's MX operand, N % 128 == 0).  act: 0 none, 1 GELU. */
int saber_k_gemm_mx(const uint8_t* A, int64_t lda, const uint8_t* SA, int64_t sa_rows, const uint8_t* W, int64_t ldw, const uint8_t* SW, int64_t sw_rows, const float* bias,
                    const float* res, float* out_f32, uint16_t* out_bf16, uint8_t* out_mx, uint8_t* out_mx_scales, int64_t out_mx_rows, int64_t ldc, int64_t M, int N, int Kp,
                    int act, void* stream);
/* bf16 [M][C] -> MX (C % 32 == 0): per 32-element block the smallest power-of-two scale with amax <= 448 * scale, elements round to nearest even */
int saber_k_quant_mx(const uint16_t* x, int64_t ldx, int C, uint8_t* out, int64_t ldo, int Kp, uint8_t* scales, int64_t scale_rows, int64_t M, void* stream);
/* LayerNorm over C of fp32 rows, written straight as an MX operand */
int saber_k_ln_mx(const float* x, int64_t ldx, const float* gamma, const float* beta, float eps, int C, uint8_t* out, int64_t ldo, int Kp, uint8_t* scales, int64_t scale_rows,
                  int64_t M, void* stream);
/* torchvision.ops.nms as the mask generator's device path runs it (csrc/amg_device.hip: stable descending score order, suppress box IoU >
 * iou_thresh, fp32): boxes (n,4) xyxy, scores (n), n <= 12288; scratch: n * 64 bytes; keep_out (n) receives the kept indices in score order,
 * count_out their number.  One workgroup. */
int saber_k_box_nms(const float* boxes_xyxy, const float* scores, int n, float iou_thresh, void* scratch, int* keep_out, int* count_out, void* stream);
/* one-head attention of 256 channels, flash style (the memory attention of the video path, upstream MemoryAttentionLayer self / cross attention):
 * out = bf16(softmax(scale Q K^T) V + bias_v); Q [n_q][256], K, V [n_keys][256] bf16 row-major, n_q a multiple of 64; ws: scratch of at least
 * (n_q / 64) * 8 * 64 * 258 floats for the split over the keys, or NULL */
int saber_k_flash256(const uint16_t* Q, const uint16_t* K, const uint16_t* V, int n_q, int n_keys, float scale, const float* bias_v, uint16_t* out, float* ws,
                    int64_t ws_floats, void* stream);
/* one axis (0: rows, 1: columns) of scipy.ndimage.gaussian_filter(sigma, mode="mirror", truncate=4) on n planes of H x W: the anti-aliasing filter
 * skimage.transform.resize applies before it down-samples a tomogram slice to the model's 1024 px (saber/adapters/preprocessing.py:21) */
int saber_k_gauss_mirror(const float* in, float* out, int n_planes, int H, int W, int axis, double sigma, void* stream);
/* out = x + alpha * g[c] * y  (x, g may be NULL) */
int saber_k_axpy(const float* x, const float* y, const float* g, float alpha, int64_t rows, int C, float* out, void* stream);
/* out = x + y[row % y_rows] as bf16 and/or fp32 (y may be NULL) */
int saber_k_add_to_bf16(const float* x, const float* y, int y_rows, uint16_t* out_bf16, float* out_f32, int64_t rows, int C, void* stream);
/* widen stored bf16 (the memory bank keeps spatial memories in bf16, as upstream does) to fp32 */
int saber_k_bf16_to_f32(const uint16_t* x, int64_t n, float* out, void* stream);
/* batched bf16 GEMM C[b] = A[b] . W[b]^T + bias with explicit leading dimensions and batch strides */
int saber_k_gemm_batched(const uint16_t* A, int lda, int64_t strideA, const uint16_t* W, int ldw, int64_t strideW, const float* bias, float* out_f32,
                         int ldcf, int64_t strideCf, uint16_t* out_bf16, int ldcb, int64_t strideCb, int M, int N, int K, int batch, void* stream);

/* engine token order helpers (DESIGN.md "token order") */
/* Folded image->token attention of the two-way transformer (reference: sam2 TwoWayAttentionBlock.cross_attn_image_to_token +
 * norm4, called from sam2/modeling/sam/transformer.py via saber/adapters/sam2/automask.py's predictor):
 * Xout[p][n] = LN(x_n + softmax_heads(x_n Kt_p^T + kscale * blockdiag_h(tk_p[h]) peq_n[h] + cb_p) VtT_p^T + bo).
 * X [P or 1][4096][256] bf16 (x_batch_stride 0 = shared); Kt [P][64][256] bf16 = the token keys folded through W_q (8 heads x 8
 * tokens, log2e/sqrt(d) folded in); the positional term (pe_n Kt^T in the plain formula) enters through peq = pe W_q^T
 * [4096][128] bf16 (model constant) and the un-folded token keys tk [P*8][128] f32; cb [P][64]; VtT [P][256][64] bf16. */
int saber_k_dec_i2t(const uint16_t* X, int64_t x_batch_stride, const uint16_t* peq, const uint16_t* Kt, const float* tk, float kscale, const float* cb,
                    const uint16_t* VtT, const float* bo, const float* gamma, const float* beta, float eps, uint16_t* Xout, int P, void* stream);

/* Folded token->image attention (cross_attn_token_to_image / final_attn_token_to_image):
 * out[p][t] = Wv (sum_n softmax_n(Qt_p[h,t].x_n + qscale * tq_p[h,t].pek_n[h]) x_n) + bv.
 * Qt [P][64][256] bf16 (queries folded through W_k), pek = pe W_k^T [4096][128] bf16 (model constant), tq [P*8][128] f32 (un-folded
 * projected queries), part_ws [P*split*64*256] f32, ml_ws [P*split*64*2] f32, out [P][8][128] bf16. */
int saber_k_dec_t2i(const uint16_t* X, int64_t x_batch_stride, const uint16_t* pek, const uint16_t* Qt, const float* tq, float qscale, float* part_ws,
                    float* ml_ws, int P, int split, const uint16_t* Wv, const float* bv, uint16_t* out, void* stream);

/* Development (co-residency experiments, tools/cu_mask_bench.py): a HIP stream restricted to CUs first_cu .. first_cu + n_cus - 1
 * (hipExtStreamCreateWithCUMask), and its release. */
int saber_k_stream_create_cu_range(int first_cu, int n_cus, void** out_stream);
int saber_k_stream_destroy(void* stream);
/* 16-bit operand type of every kernel-level entry point called from THIS thread (thread-local): 0 = bf16 (default), 1 = IEEE fp16.  The
 * uint16_t operands and outputs of saber_k_gemm*, saber_k_layernorm, saber_k_hiera_attention*, saber_k_dec_*, saber_k_flash256, ... are
 * then fp16 bit patterns; same kernels, compiled for v_mfma_f32_16x16x32_f16.  Returns the previous setting. */
int saber_k_set_operand_type(int f16);
/* the host-side fp32 -> IEEE half (round to nearest even) conversion saber_engine_finalize applies to the weights in SABER_PRECISION_FP16
 * (host pointers; no device needed) */
void saber_k_host_f32_to_f16(const float* in, uint16_t* out, int64_t n);
/* development hook: bit flags read by experimental kernel variants (0 in production) */
void saber_k_set_debug(int flags);
/* development: device buffer (uint64 per block, wave and phase) that instrumented kernels fill with s_memtime sums; NULL = off */
void saber_k_set_stamp_buffer(void* dev);

int saber_k_perm_index(int y, int x, int stage);

#ifdef __cplusplus
}
#endif
#endif
