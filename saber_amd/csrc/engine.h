// Engine state shared by engine.hip (encode / decode) and amg.hip (automatic mask generation).
#pragma once
#include <hip/hip_runtime.h>
#include <functional>
#include <map>
#include <set>
#include <string>
#include <vector>

#include "../../include/saber_amd.h"
#include "kernels.h"

struct BlockSpec { int din, dout, heads, window, q_stride; };

struct HostTensor { std::vector<int64_t> shape; std::vector<float> data; };

struct LinW { const bf16_t* w = nullptr; const float* b = nullptr; int out = 0, in = 0, ldw = 0; const bf16_t* wpk = nullptr; int wpk_n = 0; const float* wf = nullptr;
              const uint8_t* w8 = nullptr; const uint8_t* sw8 = nullptr; int kp8 = 0, sw_rows = 0; };   // w8 / sw8: MXFP8 copy (e4m3 [out][kp8] + e8m0 [kp8 / 128][sw_rows][4], gemm_fp8.hip), weight format SABER_WEIGHTS_MXFP8 only   // wf: dense fp32 [out][in] copy, kept only when the exact-precision mode was requested before finalize (exact.hip)   // wpk: K-step-packed copy (gemm_rowln.hip)  // rows zero-padded to ldw = ceil(in/64)*64
struct LnW { const float* g = nullptr; const float* b = nullptr; };

struct BlockW { LnW n1, n2; LinW qkv, proj, fc1, fc2, sc; };

struct AttnW { LinW q, k, v, o; const bf16_t* pe_proj = nullptr; const bf16_t* img_wT = nullptr; };  // img_wT: the image-side projection weight transposed, bf16 [256][128] (k_proj of tokens->image, q_proj of image->tokens): operand of the folds in decoder_tokens.hip  // pe_proj: dense PE projected by the image-side weight (k: tokens->image, q: image->tokens), bf16 [4096][128]
struct DecLayerW { AttnW self_attn, t2i, i2t; LnW n1, n2, n3, n4; LinW mlp1, mlp2; };

enum { PC_GEMM = 0, PC_HIERA_ATTN, PC_LAYERNORM, PC_DEC_ATTN, PC_ELEMENTWISE, PC_IMAGE, PC_MASK_POST, PC_DEC_T2I, PC_DEC_I2T, PC_DEC_UPSCALE, PC_GEMM_MX, PC_N };
struct ProfRec { int cls; double flops; double bytes; hipEvent_t a, b; };

struct saber_engine {
    int device = 0;
    std::string trunk;
    int max_images = 1, max_prompts = 64;
    std::string err;
    bool finalized = false;
    int weight_format = 0;          // SABER_WEIGHTS_*
    int precision = 0;              // SABER_PRECISION_*: the mode compute calls run in
    bool op_f16 = false;            // 16-bit operand type of this handle's weights / workspaces (fixed at finalize): false = bf16, true = IEEE fp16
    bool keep_f32 = false;          // fp32 weight copies were requested before finalize (exact mode available)
    void* exact_ws = nullptr;       // exact.hip's workspaces (allocated on first use)

    // model description (tiny / small / base+ / large)
    int embed_dim = 0;
    int head_dim = 0;                 // embed_dim / heads of stage 0: 96 (tiny, small), 56 (base+), 72 (large)
    int pe_bkg = 7;                   // side of the background pos_embed that is bicubically resized to the 256^2 grid
    // Token layout per stage.  large: the bit-interleaved order of common.h in every stage.  tiny/small/base+ (14x14 and 7x7
    // windows on the 64^2 / 32^2 grids): stages 2 and 3 are stored window-major WITH the reference's window padding rows
    // (70^2 = 25 windows x 196 rows, 35^2 = 25 x 49), rows of a 14x14 window ordered by 2x2 pooling group, so windows stay
    // contiguous runs of rows, a pooling group stays 4 consecutive rows and row r of stage 3 is rows 4r..4r+3 of stage 2.
    bool padded = false;
    int tok_rows[4] = {65536, 16384, 4096, 1024};
    const uint8_t* valid[4] = {nullptr, nullptr, nullptr, nullptr};   // per stage: 1 = real token, 0 = window padding row
    const uint8_t* kmask2 = nullptr;  // valid[2] zero-padded to a multiple of 128 (key mask of the global-attention blocks)
    const int* pack_idx = nullptr;    // [tok_rows[2]]: engine-order row of the 64^2 grid, or -1 for a padding row
    const int* unpack_idx = nullptr;  // [4096]: padded-layout row of each engine-order row
    std::vector<BlockSpec> blocks;
    std::vector<int> stage_ends;
    std::vector<int> stage_dims;

    std::map<std::string, HostTensor> host_w;
    std::vector<void*> allocs;

    // encoder weights
    const float *pe_wt = nullptr, *pe_bias = nullptr, *pos_table = nullptr;
    std::vector<BlockW> bw;
    LinW neck3, neck2, s1, s0;  // lateral 32^2, (64^2 + no_mem), composed conv_s1.neck, composed conv_s0.neck

    // decoder weights
    PromptWeights pw{};
    MaskEmbedWeights mw{};
    const float* no_mask_embed = nullptr;
    const float* dense_pe = nullptr;  // [4096][256], engine token order
    const bf16_t* dense_pe_bf = nullptr;
    DecLayerW dl[2];
    AttnW final_attn; LnW final_ln;
    LinW dc1, dc2; LnW up_ln; const bf16_t* dc2p = nullptr;
    LinW hyper[3];  // stacked over the 4 mask tokens (batched GEMM)
    LinW iou_head[3], obj_head[3];

    // encoder workspace
    float *pix = nullptr, *xa = nullptr, *xb = nullptr, *lat3 = nullptr;
    bf16_t *xn = nullptr, *qkv = nullptr, *att = nullptr, *hid = nullptr;
    // device-side post-processing of the mask generator (amg_device.hip): filters, per-crop and cross-crop box NMS, compaction
    bool amg_device = true;
    uint8_t* amg_pass = nullptr; DevCand *amg_tmp = nullptr, *amg_keep = nullptr, *amg_surv = nullptr; DevCrop* amg_crops_dev = nullptr;
    int *amg_counts = nullptr, *amg_nsurv = nullptr, *amg_final_slots = nullptr, *amg_count_dev = nullptr; float* amg_crop_pts = nullptr;
    saber_mask_meta* amg_meta_dev = nullptr; size_t amg_dev_cap = 0, amg_dev_masks_cap = 0;
    bool iou_prune = true;          // AMG m2m pass: skip the mask upscaling of candidates whose predicted IoUs cannot pass pred_iou_thresh (identical results)
    uint8_t* live = nullptr;        // per-prompt flags of the decode chunk in progress
    unsigned long long* prune_counters = nullptr;   // device: [0] pruned, [1] seen (accumulated by iou_live_flags_kernel)
    unsigned int* nonfinite = nullptr;              // device: overflow-sentinel counters (engine.hip "sentinel"); the 16 bytes behind prune_counters[0..1]
    int64_t amg_last_pruned = 0, amg_last_m2m = 0;    // statistics of the last saber_amg_generate call (bench.py)
    int decode_n_pts = 1;           // points per prompt of the decode call in progress (saber_decode_prompts; exact precision only when > 1)
    uint8_t *xn8_s = nullptr, *hid8_s = nullptr; int64_t mx_rows = 0;   // MXFP8 weight format: scale panels of the MX activations (their e4m3 bytes reuse xn / hid); mx_rows = panel rows
    bf16_t* sb[4] = {nullptr, nullptr, nullptr, nullptr};
    int* crops_dev = nullptr;
    // resident features per slot
    float *emb = nullptr, *fs1 = nullptr, *fs0 = nullptr;
    // per-slot first-pass shared tensors (src0 = image_embed + no_mask_embed)
    bf16_t* src0_bf = nullptr;
    float* embb = nullptr;            // per slot: image_embed + mask_downscaling.6.bias (the C operand of the in-kernel X0 tiles, XBuild)
    bf16_t* h2_bf = nullptr;          // [max_prompts][4096][16]: hidden vectors of the mask-prompt embedding (launch_mask_hidden)
    std::vector<char> slot_embb_valid;
    std::vector<char> slot_valid, slot_shared_valid;

    // decoder workspace (per chunk of max_prompts prompts)
    float *tok_pe = nullptr, *queries = nullptr, *tq = nullptr, *tk = nullptr, *tv = nullptr;
    bf16_t *t_bf0 = nullptr, *t_bf1 = nullptr, *t_att = nullptr, *t_hid = nullptr;
    bf16_t* keys_bf = nullptr;                                   // image tokens of each prompt [P][4096][256]
    bf16_t *fold_q = nullptr, *fold_k = nullptr, *fold_v = nullptr;  // folded operands [P][64][256]
    float *fold_cb = nullptr, *t2i_part = nullptr, *t2i_ml = nullptr;
    float *masks4 = nullptr, *hyper_out = nullptr, *iou4 = nullptr, *head_tmp = nullptr;
    bf16_t *head_bf0 = nullptr, *head_bf1 = nullptr;
    int* counts_ws = nullptr;
    float* dec_out_masks = nullptr;  // [max_prompts][3][65536] staging when caller passes NULL
    float* dec_out_iou = nullptr;

    // prepare workspace
    float* prep_ws = nullptr; size_t prep_ws_elems = 0; unsigned int* prep_minmax = nullptr;

    // AMG workspace (grown on demand)
    float *amg_prep = nullptr; size_t amg_prep_elems = 0;
    int* amg_sel = nullptr; bool amg_m2m_sized = false;
    float *amg_pts = nullptr, *amg_low1 = nullptr, *amg_low2 = nullptr, *amg_iou1 = nullptr, *amg_iou2 = nullptr, *amg_pts2 = nullptr;
    size_t amg_prompts_cap = 0, amg_pts_cap = 0;
    uint32_t* amg_bits = nullptr; size_t amg_bits_words = 0;        // masks kept across crops (persistent, grown on demand)
    uint32_t* amg_crop_bits = nullptr; size_t amg_crop_words = 0;   // one crop's pred_iou survivors
    MaskStats* amg_stats = nullptr; int* amg_idx = nullptr; size_t amg_stats_cap = 0;
    int* order_dev = nullptr; size_t order_cap = 0;

    // hipGraph replay of the AMG driver's launch sequences (one batched encoder pass, 12 decoder batches per slice with the default
    // pyramid): a sequence is run eagerly the first time its arguments are seen, captured the second time, replayed from then on
    bool graphs_on = true;
    std::map<std::string, hipGraphExec_t> graphs;
    std::set<std::string> graph_seen, graph_bad;
    int graph_replays = 0, graph_captures = 0;
    float* amg_img = nullptr; size_t amg_img_elems = 0;     // engine-owned copy of the caller's image: a stable address for the captured launches
    int* crops_pin = nullptr;                                // pinned host copies of the crop boxes of encoder passes: 8 slots of 256 ints.  Slot 0 belongs
                                                             // to the AMG driver (its H2D copy is captured into a hipGraph and reads the slot at replay time);
                                                             // saber_encode cycles through slots 1..7, each guarded by an event recorded behind its H2D copy, so
                                                             // back-to-back calls without a stream synchronisation never overwrite boxes a queued copy has yet to read
    hipEvent_t crops_ev[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    int crops_next = 1;
    int *rm_to_eng = nullptr, *eng_to_rm = nullptr;   // 64x64 grid: row-major (y * 64 + x) <-> engine token order (video path, on demand)
    int amg_last_syncs = 0;           // host synchronisations of the last saber_amg_generate call (saber_amg_last_syncs)
    // co-residency experiment (saber_engine_set_encoder_stream): the mask generator's encoder passes run on this stream, fenced against the
    // caller's stream by two events; nullptr = everything on the caller's stream
    hipStream_t enc_stream = nullptr;
    hipEvent_t enc_ev[2] = {nullptr, nullptr};

    // optional per-launch HIP-event profiling (saber_profile_begin / saber_profile_end)
    bool prof_on = false;
    std::vector<ProfRec> prof;
    std::vector<hipEvent_t> ev_pool; size_t ev_used = 0;
};

void prof_begin(saber_engine* e, int cls, double flops, double bytes, hipStream_t s);
void prof_end(saber_engine* e, hipStream_t s);
// profiled launch: records a HIP event pair around `call` on stream `s` when profiling is on
#define ENG_KP(e, cls, flops, bytes, call)                               \
    do {                                                                 \
        prof_begin((e), (cls), (flops), (bytes), s);                     \
        const char* _m = (call);                                         \
        prof_end((e), s);                                                \
        if (_m) return eng_fail((e), SABER_ERR_INVALID, _m);             \
    } while (0)

int eng_fail(saber_engine* e, int code, const std::string& msg);
// h[0..2]: the sentinel counters as read from the device; SABER_OK or SABER_ERR_RANGE with a message that names the stage
int eng_check_finite_counts(saber_engine* e, const unsigned int* h);
// Binds the calling thread to the engine's device for the duration of one C-ABI call and restores the caller's current device on
// return (a caller whose torch current device is M must not find it switched to the engine's device N afterwards).
// It also selects the kernels' 16-bit operand type for the call (kernels.h: the launchers dispatch on the calling thread's setting).
struct DeviceGuard {
    int prev = -1;
    int prev_op = 0;
    hipError_t st = hipSuccess;
    explicit DeviceGuard(int dev, bool op_f16) {
        prev_op = g_saber_op_f16; g_saber_op_f16 = op_f16 ? 1 : 0;
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != dev) st = hipSetDevice(dev); else prev = -1;
    }
    ~DeviceGuard() { g_saber_op_f16 = prev_op; if (prev >= 0) (void)hipSetDevice(prev); }
};
#define ENG_DEVICE(e)                                                                                  \
    DeviceGuard _dev_guard((e)->device, (e)->op_f16);                                                               \
    if (_dev_guard.st != hipSuccess) return eng_fail((e), SABER_ERR_HIP, std::string("hipSetDevice: ") + hipGetErrorString(_dev_guard.st))
#define ENG_HIP(e, call)                                                                              \
    do {                                                                                              \
        hipError_t _st = (call);                                                                      \
        if (_st != hipSuccess) return eng_fail((e), SABER_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(_st)); \
    } while (0)
#define ENG_K(e, call)                                                   \
    do {                                                                 \
        const char* _m = (call);                                         \
        if (_m) return eng_fail((e), SABER_ERR_INVALID, _m);             \
    } while (0)

// hipGraph replay of a fixed launch sequence (engine.hip)
int eng_graphed(saber_engine* e, const std::string& key, hipStream_t s, const std::function<int()>& body);
// drops every captured sequence (their launches hold addresses of workspaces that are about to be released)
void eng_graphs_flush(saber_engine* e);
// internal entry points used by amg.hip
int eng_encode(saber_engine* e, const float* img_dev, int H, int W, int channels, const int* crops_host, int n, int slot0, hipStream_t s);
// per_slot > 0: the prompts span consecutive slots, per_slot prompts each (crops of one AMG layer decoded in one batch);
// per_slot = 0: all n prompts read slot `slot`.
// mask_clamp > 0: the mask prompt is clamped to +-mask_clamp as it is read (SAM2ImagePredictor clamps the logits it returns to +-32)
int eng_decode(saber_engine* e, int slot, int per_slot, const float* pts_dev, const int* labels_dev, int n, int multimask,
               const float* mask_in_dev, float mask_clamp, float* out_lowres, float* out_iou, float* out_obj, hipStream_t s);
// out_raw4: out_lowres receives ALL 4 low-res planes of every prompt ([n][4][256*256]) and no selection copy is made: out_iou holds the
// IoUs of planes 1-3 (multimask) or of the chosen plane, out_sel the chosen plane (single-mask mode).  mask_in_raw4: mask_in_dev is such a
// buffer from a multimask decode and prompt q refines plane 1 + q % 3 of its prompt q / 3.
int eng_decode_ex(saber_engine* e, int slot, int per_slot, const float* pts_dev, const int* labels_dev, int n, int multimask,
                  const float* mask_in_dev, int mask_in_raw4, float mask_clamp, float* out_lowres, int out_raw4, float* out_iou, float* out_obj,
                  int* out_sel, hipStream_t s, float prune_iou_thr = 0.f);   // prune_iou_thr > 0 (single-mask raw-plane decodes): candidates whose four predicted IoUs are all <= it skip the mask upscaling
// exact-precision mode (exact.hip): the Hiera blocks + neck of n images already patch-embedded in e->xa; one chunk of the decoder
int exact_encode_blocks(saber_engine* e, int n, int slot0, hipStream_t s);
int exact_decode_core(saber_engine* e, int slot0, int per_slot, int p_base, const float* pts, const int* labels, int P, const float* mask_in,
                      float mask_clamp, int mask_in_q0, float* out_obj, float* masks4, hipStream_t s, int n_pts = 1);
int exact_chunk_prompts(const saber_engine* e);
void exact_release(saber_engine* e);
template <typename T> int eng_alloc(saber_engine* e, T** p, size_t count);
int eng_alloc_bytes(saber_engine* e, void** p, size_t bytes);
void eng_free(saber_engine* e, void* p);
// grow-on-demand workspaces: waits for the stream, releases the previous allocation (if any) and allocates `count` elements
template <typename T> int eng_regrow(saber_engine* e, T** p, size_t count, hipStream_t s) {
    if (*p) {
        hipError_t st = hipStreamSynchronize(s);
        if (st != hipSuccess) return eng_fail(e, SABER_ERR_HIP, std::string("hipStreamSynchronize: ") + hipGetErrorString(st));
        eng_free(e, *p);
        *p = nullptr;
    }
    return eng_alloc(e, p, count);
}
