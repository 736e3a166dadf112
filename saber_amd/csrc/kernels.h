// Internal launcher interface between the engine (engine.hip) and the gfx950 kernels.
// Every launcher returns nullptr on success or a static error string (no exceptions,
// no abort): errors surface through the C-ABI status + saber_last_error().
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef uint16_t bf16_t;

enum { ACT_NONE = 0, ACT_GELU = 1, ACT_RELU = 2, ACT_SIGMOID = 3 };

// ------------------------------------------------------------------ gemm.hip
struct GemmParams {
    const bf16_t* A = nullptr; int64_t lda = 0; int64_t strideA = 0;
    const bf16_t* W = nullptr; int64_t ldw = 0; int64_t strideW = 0;  // [N][K] row-major
    const float* bias = nullptr; int64_t strideBias = 0;
    const float* res = nullptr; int64_t ldres = 0; int64_t strideRes = 0;  // fp32 residual
    float* Cf = nullptr; int64_t ldcf = 0; int64_t strideCf = 0;
    bf16_t* Cb = nullptr; int64_t ldcb = 0; int64_t strideCb = 0;
    int M = 0, N = 0, K = 0;
    int act = ACT_NONE;
    int act_last = 0;    // apply the activation after the residual add (act(dc2(x) + feat_s0))
    int res_shift = 0;   // residual row = (row >> res_shift) ...
    int res_mod = 0;     // ... then % res_mod when res_mod > 0 (broadcast over prompts)
    int pool4 = 0;       // rows 4q..4q+3 max-pooled into output row q (Hiera q-pool shortcut)
    int batch = 1;
    int dbg = 0;         // development flags (saber_k_set_debug), 0 in production
    unsigned long long* stamps = nullptr;   // development: in-kernel cycle stamps (tools/gemm_stamps.py)
    int rev = 0;         // walk the M tiles from the last to the first (see Finalizer / eng_gemm: consecutive GEMMs of the encoder alternate)
    int w_kpad = 0;      // W rows are zero-padded to a multiple of 64 in K (ldw >= padded K): enables the direct-to-LDS kernel
    // row-owner GEMM + LayerNorm (gemm_rowln.hip): ln_out = bf16(LayerNorm(Cf row) * ln_gamma + ln_beta)
    const float* ln_gamma = nullptr; const float* ln_beta = nullptr; float ln_eps = 1e-6f;
    bf16_t* ln_out = nullptr; int64_t ldln = 0;
    const bf16_t* Wpk = nullptr;     // W packed per K-step for the row-owner kernel (launch_pack_w_kstep)
};
const char* launch_gemm(const GemmParams& p, hipStream_t stream);
const char* gemm_init_device();
// y = A.W^T + bias + res (fp32 -> Cf, optional bf16 copy -> Cb) and ln_out = bf16(LayerNorm(y)) in one kernel; N in {144, 288, 576}
bool gemm_rowln_supported(const GemmParams& p);
const char* launch_gemm_rowln(const GemmParams& p, hipStream_t stream);
const char* gemm_rowln_init_device();
// one wave per SIMD, 128x128 wave tiles (gemm_w1.hip): bf16-output GEMMs with K % 32 == 0
bool gemm_w1_supported(const GemmParams& p);
const char* launch_gemm_w1(const GemmParams& p, hipStream_t stream);
const char* gemm_w1_init_device();
// W [N][ldw] (rows zero-padded) -> Wpk[ceil(K/32)][N][32] with the LDS chunk permutation of the row-owner kernel applied
size_t gemm_rowln_packed_elems(int N, int K);
const char* launch_pack_w_kstep(const bf16_t* W, int ldw, int N, int K, bf16_t* out, hipStream_t s);

// ------------------------------------------------------------------ gemm_fp8.hip
// C = act((A8 . W8^T) * sa[m] * sw[n] + bias) (+ res): e4m3 operands on the block-scaled fp8 MFMA, see the file header
struct GemmMxParams {
    const uint8_t* A = nullptr; int64_t lda = 0;      // [M][lda] e4m3, rows zero-padded to Kp
    const uint8_t* SA = nullptr; int64_t sa_rows = 0; // e8m0 block scales [Kp / 128][sa_rows][4], sa_rows >= ceil(M / 768) * 768
    const uint8_t* W = nullptr; int64_t ldw = 0;      // [N][ldw] e4m3, rows zero-padded to Kp
    const uint8_t* SW = nullptr; int64_t sw_rows = 0; // [Kp / 128][sw_rows][4], sw_rows >= ceil(N / 192) * 192
    const float* bias = nullptr;
    // exactly one output form: bf16 (Cb alone) | MX (C8 + SC: e4m3 [M][ldc8] + [N / 128][sc_rows][4], the next GEMM's A operand) | fp32 + residual (Cf, + optional bf16 copy Cb)
    bf16_t* Cb = nullptr; int64_t ldcb = 0;
    uint8_t* C8 = nullptr; int64_t ldc8 = 0; uint8_t* SC = nullptr; int64_t sc_rows = 0;
    float* Cf = nullptr; int64_t ldcf = 0; const float* res = nullptr; int64_t ldres = 0;
    int64_t M = 0; int N = 0, Kp = 0, act = ACT_NONE;
};
const char* launch_gemm_mx(const GemmMxParams& p, hipStream_t s);
// bf16 [M][C] -> MX: e4m3 [M][Kp] (zero-padded) + e8m0 block scales SC[Kp / 128][sc_rows][4]
const char* launch_quant_mx_bf16(const bf16_t* x, int64_t ldx, int C, uint8_t* out, int64_t ldo, int Kp, uint8_t* SC, int64_t sc_rows, int64_t M, hipStream_t s);
// LayerNorm of fp32 rows straight into MX
const char* launch_ln_mx(const float* x, int64_t ldx, const float* gamma, const float* beta, float eps, int C, uint8_t* out, int64_t ldo, int Kp, uint8_t* SC, int64_t sc_rows,
                         int64_t M, hipStream_t s);

// ------------------------------------------------------------------ layernorm.hip
struct LayerNormParams {
    const float* x = nullptr; int64_t ldx = 0;
    const float* gamma = nullptr; const float* beta = nullptr; float eps = 1e-6f;
    float* out_f = nullptr; bf16_t* out_bf = nullptr; bf16_t* out_bf_add = nullptr; int64_t ldo = 0;
    const float* addvec = nullptr; int add_mod = 0;  // out_bf_add = bf16(y + addvec[row % add_mod])
    const uint8_t* row_valid = nullptr; int valid_mod = 0;  // rows with row_valid[row % valid_mod] == 0 are written as zeros (window padding)
    int rows = 0, C = 0, act = ACT_NONE;
};
const char* launch_layernorm(const LayerNormParams& p, hipStream_t s);
// out[img][r][:] = idx[r] >= 0 ? in[img][idx[r]][:] : 0   (re-ordering between the engine's token orders; fp32 rows of C floats)
const char* launch_gather_rows(const float* in, int64_t in_rows, float* out, int64_t out_rows, const int* idx, int C, int n_images, hipStream_t s);
const char* launch_add_to_bf16(const float* x, const float* y, int ymod, bf16_t* out_bf, float* out_f, int64_t rows, int C,
                               hipStream_t s);

// ------------------------------------------------------------------ attention_hiera.hip
// qkv: bf16 [tokens][3*heads*hd]; out: bf16 [tokens_q][heads*hd]; hd = 72 | 96 | 56; windows are contiguous runs of nk rows.
// kmask (optional): one byte per key of a window (the same for every window), zero-padded to a multiple of 128 bytes; keys whose
// byte is 0 take no part in the softmax (window-padding rows of the tiny/small/base+ trunks in their global-attention blocks).
const char* launch_hiera_attention(const bf16_t* qkv, bf16_t* out, int n_windows, int nk, int heads, int hd, int q_pool,
                                   const uint8_t* kmask, hipStream_t s);
const char* hiera_attention_init_device();

// ------------------------------------------------------------------ image_ops.hip
const char* launch_prepare_u16(const uint16_t* img, int H, int W, float* out, float* ws, unsigned int* minmax, hipStream_t s);
const char* launch_prepare_f32(const float* img, int H, int W, float* out, float* ws, unsigned int* minmax, hipStream_t s);
const char* launch_prepare_rgb_f32(const float* img, int H, int W, float* out, float* ws, unsigned int* minmax, hipStream_t s);   // (H,W,3) interleaved
const char* launch_resize_normalize(const float* img, int H, int W, int channels, const int* crops_dev, int n, float* out, int res,
                                    hipStream_t s);
const char* launch_patch_embed(const float* pix, const float* wt, const float* bias, const float* pos, float* out, int n_images,
                               int C, int res, hipStream_t s);
const char* image_ops_init_device();

// ------------------------------------------------------------------ decoder_ops.hip
struct PromptWeights {
    const float* gauss;         // [2][128]
    const float* point_embed;   // [4][256]
    const float* not_a_point;   // [256]
    const float* out_tokens;    // [6][256]: obj, iou, mask0..3
};
const char* launch_prompt_tokens(const float* pts, const int* labels, int P, PromptWeights w, float* tokens, hipStream_t s);
const char* launch_prompt_tokens_multi(const float* pts, const int* labels, int P, int K, PromptWeights w, float* tokens, hipStream_t s);   // K points per prompt: 7 + K tokens

struct MaskEmbedWeights {
    const float *w1, *b1, *g1, *be1;   // conv 1->4 k2s2 [4][4], LN2d(4)
    const float *w2, *b2, *g2, *be2;   // conv 4->16 k2s2 [16][16], LN2d(16)
    const float *w3, *b3;              // conv 16->256 1x1 [256][16]
};
// src[p][tok] = image_embed[tok] + mask_downscaling(mask_in[p]) (token order = engine order on the 64x64 grid)
// Which image a prompt reads: prompt p of a launch uses the tensor at base + ((p + off) / div) * stride.
// One tensor per prompt: {stride, 1, 0}; one tensor shared by every prompt: {0, 1, 0}; several crops batched into one
// launch, `div` consecutive prompts per crop (slot): {slot stride, div, index of the launch's first prompt within the batch}.
struct XMap { int64_t stride; int div; int off; };
const char* launch_mask_embed_src(const float* mask_in, int P, const float* image_embed, XMap emb_map, const float* pos, MaskEmbedWeights w,
                                  float* src_f, bf16_t* src_bf, bf16_t* srcpos_bf, float clamp_abs, hipStream_t s, int raw4_q0 = -1);   // clamp_abs > 0: mask_in is clamped to +-clamp_abs on load

// h2[p][tok][16] (bf16, engine token order) = the 16-channel hidden vector of the mask-prompt embedding: the first two stages of
// mask_downscaling (conv k2s2 1->4, LN2d, GELU, conv k2s2 4->16, LN2d, GELU) on the 4 x 4 logit patch of each token.  The 1 x 1 conv to 256
// channels + image_embed ("src") is assembled tile by tile inside the layer-0 kernels of the two-way transformer (XBuild below) and never
// reaches HBM: 32 B per token instead of 512 B written once and read twice.
const char* launch_mask_hidden(const float* mask_in, int P, MaskEmbedWeights w, bf16_t* h2, float clamp_abs, hipStream_t s, int raw4_q0 = -1);
// How a layer-0 kernel assembles X0 = bf16(image_embed + b3 + h2 . W3^T) for prompt p: embb = image_embed + b3 (fp32 [slots][4096 x 256] in the
// order of launch_embb_tiles) of slot (p + map.off) / map.div; h2 as above; w3 = mask_downscaling.6.weight fp32 [256][16].  Same arithmetic as
// mask_embed_src_kernel<true> (one K = 16 MFMA per 16 x 16 block with the fp32 image_embed + b3 as its C operand): bit-identical tiles.
// embb in the tile builders' order: out[(row / 16) * 16 + ch / 16][lane = (ch % 16 / 4) * 16 + row % 16][ch % 4] = emb[row][ch] + b3[ch]
// (one KB per (16-row tile, 16-channel tile): the C operand of one MFMA of the builders, read by one load instruction)
const char* launch_embb_tiles(const float* emb, const float* b3, float* out, hipStream_t s);
struct XBuild { const float* embb = nullptr; XMap map{0, 1, 0}; const bf16_t* h2 = nullptr; const float* w3 = nullptr; };

// fp32 multi-head attention for the two-way transformer.  q [B][nq][heads*hd], k/v [B or 1][nk][heads*hd].
// Output is bf16 (it always feeds the out_proj GEMM).
const char* launch_dec_attention(const float* q, const float* k, const float* v, bf16_t* out, int B, int nq, int nk, int heads,
                                 int hd, int64_t q_bs, int64_t k_bs, int64_t v_bs, int64_t o_bs, hipStream_t s);

// masks[p][k][y][x] = sum_c hyper[p][k][c] * up[p][perm(y,x)][c]  (up: bf16 [P][65536][32], engine token order)
const char* launch_mask_dot(const bf16_t* up, const float* hyper, int P, float* masks4, hipStream_t s);
// multimask: out[p][0..2] = masks4[p][1..3], iou_out = iou4[:,1:]; else dynamic single-mask selection (delta 0.05 / thr 0.98)
const char* launch_mask_pick(const float* masks4, const float* iou4, int P, int multimask, float* out_iou, int* out_sel, hipStream_t s, const uint8_t* live = nullptr);
// live[p] = iou4[p][0] > thr || max(iou4[p][1..3]) > thr: whether a single-mask (dynamic multimask) candidate can pass a `predicted IoU > thr` filter at all
const char* launch_iou_live_flags(const float* iou4, int P, float thr, uint8_t* live, unsigned long long* counters, hipStream_t s);
const char* launch_mask_select(const float* masks4, const float* iou4, int P, int multimask, float* out_masks, float* out_iou,
                               int* counts_ws, hipStream_t s);

// ------------------------------------------------------------------ decoder_fused.hip
const char* launch_dec_fold(const float* a, const bf16_t* W, const float* bias, int mode, float scale, bf16_t* out, float* cb, int P, hipStream_t s);
// pek / peq: the dense positional encoding projected by the attention's image-side weight, bf16 [4096][128] in engine token order
// (pe W_k^T for tokens->image, pe W_q^T for image->tokens); tq / tk: the token-side projections fp32 [P*8][128]; scale: head_dim^-0.5 log2(e)
const char* launch_dec_t2i(const bf16_t* X, XMap xm, const bf16_t* pek, const bf16_t* Qt, const float* tq, float qscale, float* Opart, float* ML,
                           int P, int split, const bf16_t* Wv, const float* bv, bf16_t* out, hipStream_t s, const XBuild* build = nullptr);
const char* launch_dec_i2t(const bf16_t* X, XMap xm, const bf16_t* peq, const bf16_t* Kt, const float* tk, float kscale, const float* cb, const bf16_t* Vt,
                           const float* bo, const float* gamma, const float* beta, float eps, bf16_t* Xout, int P, hipStream_t s, const XBuild* build = nullptr);
const char* launch_dec_upscale(const bf16_t* X, const bf16_t* W1, const float* b1, const float* ln_g, const float* ln_b, const bf16_t* W2p,
                               const float* b2, const float* fs1, const float* fs0, XMap slot_map, const float* hyper, float* masks4, int P,
                               hipStream_t s, const uint8_t* live = nullptr, const float* iou4 = nullptr, int multimask = 0);
// live: optional per-prompt flags, prompts with 0 are skipped; iou4 ([P][4], optional): only the planes a multimask / single-mask selection can return are computed
const char* decoder_fused_init_device();

// ------------------------------------------------------------------ decoder_tokens.hip
// One segment of the token side of the two-way transformer (everything between two image-side kernels) as one launch; see the file header.
struct TokLin { const bf16_t* w = nullptr; const float* b = nullptr; int ldw = 0; int n = 0; };
struct TokLn { const float* g = nullptr; const float* b = nullptr; };
struct TokSeg {
    int P = 0;
    float* queries = nullptr; const float* tok_pe = nullptr;
    float kscale = 0.f;
    // (1) queries = LN(queries + o_proj(t_att))                         [after a tokens -> image attention]
    const bf16_t* t_att = nullptr; TokLin att_o; TokLn att_ln; float att_eps = 1e-5f;
    // (2) queries = LN3(queries + mlp(queries)); image -> tokens operands of this layer
    int do_mlp = 0;
    TokLin mlp1, mlp2; TokLn ln3;
    TokLin i2t_k, i2t_v; const bf16_t* i2t_qT = nullptr; const float* i2t_qb = nullptr; const bf16_t* i2t_o = nullptr;
    float* tk_out = nullptr; bf16_t* fold_k = nullptr; float* fold_cb = nullptr; bf16_t* fold_v = nullptr;
    // (3) self attention of the tokens + LN1
    int do_self = 0, self_first = 0;
    TokLin sa_q, sa_k, sa_v, sa_o; TokLn ln1;
    // (4) operands of the next tokens -> image attention
    int do_t2i = 0;
    TokLin t2i_q; const bf16_t* t2i_kT = nullptr; float* tq_out = nullptr; bf16_t* fold_q = nullptr;
    // (5) heads: IoU, object score, hypernetwork MLPs
    int do_heads = 0;
    TokLin iou[3], obj[3], hyper[3];
    float* iou4 = nullptr; float* obj_out = nullptr; float* hyper_out = nullptr;
};

const char* launch_dec_tokens(const TokSeg& s, hipStream_t st);
const char* decoder_tokens_init_device();

// K8: bilinear upsample of 256x256 logits to the crop, threshold / stability counts / bbox / bit-packing.
struct MaskStats { int area; int inter; int uni; int x0; int y0; int x1; int y1; int pad; };
const char* launch_mask_post(const float* lowres, const int* idx, int n, int crop_x0, int crop_y0, int crop_w, int crop_h, int H,
                             int W, float thr, float offset, uint32_t* bits, MaskStats* stats, hipStream_t s, const uint8_t* pass = nullptr);   // pass: optional per-mask flags, masks with 0 are skipped
// ------------------------------------------------------------------ amg_device.hip: filters, box NMS and compaction of the mask generator on the device
struct DevCand { float box[4]; float iou, stab; float pt[2]; int crop[4]; int area; int slot; float score; int pad; };   // slot: index of the bit mask in the K8 scratch
struct DevCrop { int box[4]; int kbase; int nm; int pt0; int M; };   // one crop of a decoded batch: first candidate / candidates / first grid point (batch-relative), masks per point
struct saber_mask_meta;
const char* amg_device_init();
const char* launch_box_nms_probe(const float* boxes, const float* scores, int n, float thr, void* tmp /* n x 64 bytes */, int* keep, int* count, hipStream_t s);
const char* launch_amg_plane(const float* iou, const int* sel, int n, int plane_mode, float thr, int* plane, uint8_t* pass, hipStream_t s);
const char* launch_amg_crops(const DevCrop* crops, int n_crops, int max_nm, const MaskStats* stats, const uint8_t* pass, const float* iou, const float* crop_pts, float stab_thr,
                             float nms_thr, int H, int W, int slot_base, DevCand* tmp, DevCand* keep, int* counts, DevCand* surv, int* nsurv, int cap, hipStream_t s);
const char* launch_amg_final(DevCand* surv, const int* nsurv, int cap, int multi_crop, float nms_thr, int max_masks, int* final_slots, saber_mask_meta* meta, int* out_count,
                             const uint32_t* scratch, uint32_t* out_bits, int64_t words, hipStream_t s);
// paint label planes: plane[y][x] = max over i (in order) ... later masks overwrite earlier ones (propagation.py:185-186)
const char* launch_gather_masks(const uint32_t* src, const int* idx, uint32_t* dst, int n, int64_t words, hipStream_t s);
const char* launch_label_plane(const uint32_t* bits, const int* order, int n, int H, int W, uint16_t* plane, hipStream_t s);
// inter[i][j] = popcount(mask_i & mask_j) on bit-packed masks (n x words uint32)
const char* launch_pair_intersections(const uint32_t* bits, int n, int64_t words, int* inter, hipStream_t s);
// gather token-major [tokens][C] (engine order, stage s grid) -> NCHW fp32 for inspection / parity tests
const char* launch_unpermute_nchw(const float* tok, int C, int stage, float* out, hipStream_t s);

// ------------------------------------------------------------------ video_ops.hip (SAM2 memory path, channels-last fp32, row-major pixels)
const char* launch_rope(const float* x, int64_t rows, int n_rot, int C, int side, float theta, float* out_f, bf16_t* out_bf, hipStream_t s);
const char* launch_softmax_rows(const float* S, int64_t lds_, int64_t rows, int n, float scale, bf16_t* P, int64_t ldp, hipStream_t s);
const char* launch_conv3x3s2(const float* in, int H, int W, int Cin, const float* w, const float* b, int Cout, float* out, hipStream_t s);
const char* launch_unpack_masks(const uint32_t* bits, int n, int H, int W, uint8_t* out, hipStream_t s);
const char* launch_paint_nearest(const float* logits, int Hv, int Wv, float thr, int label, uint16_t* plane, int H, int W, int* any_flag, hipStream_t s);
const char* launch_conv3x3s2_t(const float* in, int H, int W, int Cin, const float* wt, const float* b, int Cout, float* out, hipStream_t s);
const char* launch_dwconv7_t(const float* in, int H, int W, int C, const float* wt, const float* b, float* out, hipStream_t s);   // weights [49][C]
const char* launch_dwconv7(const float* in, int H, int W, int C, const float* w, const float* b, float* out, hipStream_t s);
const char* launch_conv4x4s4(const float* in, int H, int W, const float* w, const float* b, float* out, hipStream_t s);
const char* launch_resize_plane(const float* in, int n_planes, int H, int W, float* out, int Ho, int Wo, int antialias, int post, float a, float c, hipStream_t s);
// softmax(scale Q K^T) V + bias_v for ONE head of 256 channels (memory attention of the video path): Q [n_q][256], K / V [n_keys][256] bf16
// row-major, out bf16 [n_q][256]; ws: >= (n_q / 64) * 8 * 64 * 258 floats of scratch for the split over the keys (may be NULL: no split)
const char* launch_flash256(const bf16_t* Q, const bf16_t* K, const bf16_t* V, int n_q, int n_keys, float scale, const float* bias_v, bf16_t* out, float* ws,
                            size_t ws_floats, hipStream_t s);
const char* launch_gauss_mirror(const float* in, float* out, int n_planes, int H, int W, int axis, double sigma, hipStream_t s);
const char* launch_axpy(const float* x, const float* y, const float* g, float alpha, int64_t rows, int C, float* out, hipStream_t s);
const char* launch_bf16_to_f32(const bf16_t* x, int64_t n, float* out, hipStream_t s);
