// Launchers of the kernel files that are compiled once per 16-bit operand type (included by kernels.h inside namespace op_bf16 and
// inside namespace op_f16: no include guard on purpose).  "bf16" in a name or comment = "the build's 16-bit operand type".
// ------------------------------------------------------------------ gemm.hip, gemm_rowln.hip, layernorm.hip
const char* launch_gemm(const GemmParams& p, hipStream_t stream);
const char* gemm_init_device();
// y = A.W^T + bias + res (fp32 -> Cf, optional bf16 copy -> Cb) and ln_out = bf16(LayerNorm(y)) in one kernel; N in {144, 288, 576}
bool gemm_rowln_supported(const GemmParams& p);
const char* launch_gemm_rowln(const GemmParams& p, hipStream_t stream);
const char* gemm_rowln_init_device();
// W [N][ldw] (rows zero-padded) -> Wpk[ceil(K/32)][N][32] with the LDS chunk permutation of the row-owner kernel applied
size_t gemm_rowln_packed_elems(int N, int K);
const char* launch_pack_w_kstep(const bf16_t* W, int ldw, int N, int K, bf16_t* out, hipStream_t s);
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
const char* launch_prompt_tokens(const float* pts, const int* labels, int P, PromptWeights w, float* tokens, hipStream_t s);
const char* launch_prompt_tokens_multi(const float* pts, const int* labels, int P, int K, PromptWeights w, float* tokens, hipStream_t s);   // K points per prompt: 7 + K tokens
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
// dec_i2t of layer l fused with the tokens -> image attention that follows it (round 5): X' is written as before and consumed as the next
// attention's keys / values from LDS; `out` = that attention's output after v_proj (what launch_dec_t2i writes).  P workgroups of four waves.
const char* launch_dec_i2t_t2i(const bf16_t* X, XMap xm, const bf16_t* peq, const bf16_t* Kt, const float* tk, float kscale, const float* cb, const bf16_t* VtT,
                               const float* bo, const float* gamma, const float* beta, float eps, bf16_t* Xout, int P,
                               const bf16_t* pek, const bf16_t* Qt, const float* tq, float qscale, const bf16_t* Wv, const float* bv, bf16_t* out, hipStream_t s);
const char* launch_dec_upscale(const bf16_t* X, const bf16_t* W1, const float* b1, const float* ln_g, const float* ln_b, const bf16_t* W2p,
                               const float* b2, const float* fs1, const float* fs0, XMap slot_map, const float* hyper, float* masks4, int P,
                               hipStream_t s, const uint8_t* live = nullptr, const float* iou4 = nullptr, int multimask = 0, unsigned int* sentinel = nullptr);
// sentinel (optional): += number of lanes that stored a NaN / inf logit
// live: optional per-prompt flags, prompts with 0 are skipped; iou4 ([P][4], optional): only the planes a multimask / single-mask selection can return are computed
const char* decoder_fused_init_device();
const char* launch_dec_tokens(const TokSeg& s, hipStream_t st);
const char* decoder_tokens_init_device();
const char* launch_mask_post(const float* lowres, const int* idx, int n, int crop_x0, int crop_y0, int crop_w, int crop_h, int H,
                             int W, float thr, float offset, uint32_t* bits, MaskStats* stats, hipStream_t s, const uint8_t* pass = nullptr);   // pass: optional per-mask flags, masks with 0 are skipped
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
