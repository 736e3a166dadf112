// Internal launcher interface between the engine (engine.hip) and the gfx950 kernels.
// Every launcher returns nullptr on success or a static error string (no exceptions,
// no abort): errors surface through the C-ABI status + saber_last_error().
//
// The sources of the kernels that take 16-bit MFMA operands are compiled twice (common.h "OPERAND TYPE", op_wrap.hip): their launchers
// exist as op_bf16::launch_x and op_f16::launch_x (declared in kernels_op.h).  Host code (engine.hip, amg.hip, classifier.hip,
// capi_kernels.hip, ...) calls plain launch_x(...): the forwarders at the end of this header pick the namespace from the calling thread's
// operand type, which every C-ABI entry point sets from the engine handle's precision mode (DeviceGuard, engine.h) and the kernel-level
// C-ABI from saber_k_set_operand_type.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

// development switches shared by all kernel files (capi_kernels.hip: saber_k_set_debug, saber_k_set_stamp_buffer)
extern int g_saber_debug_flags;
extern unsigned long long* g_saber_stamp_buf;
// compute units of the calling thread's CURRENT device (capi_kernels.hip): queried once per device, thread-safe (the launchers kept
// function-local `static int n_cu`s - a data race between worker threads, and the first caller's device for everybody)
int saber_cu_count();
// operand type of the calling thread: 0 = bf16, 1 = fp16 (capi_kernels.hip)
extern thread_local int g_saber_op_f16;
inline bool saber_op_is_f16() { return g_saber_op_f16 != 0; }

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


// ------------------------------------------------------------------ image_ops.hip
const char* launch_prepare_u16(const uint16_t* img, int H, int W, float* out, float* ws, unsigned int* minmax, hipStream_t s);
const char* launch_prepare_f32(const float* img, int H, int W, float* out, float* ws, unsigned int* minmax, hipStream_t s);
const char* launch_prepare_rgb_f32(const float* img, int H, int W, float* out, float* ws, unsigned int* minmax, hipStream_t s);   // (H,W,3) interleaved
const char* launch_resize_normalize(const float* img, int H, int W, int channels, const int* crops_dev, int n, float* out, int res,
                                    hipStream_t s);
const char* launch_patch_embed(const float* pix, const float* wt, const float* bias, const float* pos, float* out, int n_images,
                               int C, int res, hipStream_t s);
const char* image_ops_init_device();
// overflow sentinel: adds the number of 4-value groups of p[0..n) that hold a NaN / inf to *counter
const char* launch_nonfinite_scan(const float* p, int64_t n, unsigned int* counter, hipStream_t s);

// ------------------------------------------------------------------ decoder_ops.hip
struct PromptWeights {
    const float* gauss;         // [2][128]
    const float* point_embed;   // [4][256]
    const float* not_a_point;   // [256]
    const float* out_tokens;    // [6][256]: obj, iou, mask0..3
};

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

struct XBuild { const float* embb = nullptr; XMap map{0, 1, 0}; const bf16_t* h2 = nullptr; const float* w3 = nullptr; };




// ------------------------------------------------------------------ decoder_tokens.hip
// One segment of the token side of the two-way transformer (everything between two image-side kernels) as one launch; see the file header.
struct TokLin { const bf16_t* w = nullptr; const float* b = nullptr; int ldw = 0; int n = 0;
                const bf16_t* wpk = nullptr; int npk = 0; };      // wpk: optional launch_pack_w_kstep copy of the npk rows of w (a 16 x 32 fragment = one contiguous KB)
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
    const bf16_t* mlp1_pk = nullptr; const bf16_t* mlp2_pk = nullptr;     // launch_pack_w_kstep copies of mlp1.w [2048][256] / mlp2.w [256][2048] (required with do_mlp)
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


// K8: bilinear upsample of 256x256 logits to the crop, threshold / stability counts / bbox / bit-packing.
struct MaskStats { int area; int inter; int uni; int x0; int y0; int x1; int y1; int pad; };
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



// ------------------------------------------------------------------ launchers of the twice-compiled kernel files
namespace op_bf16 {
#include "kernels_op.h"
}
namespace op_f16 {
#include "kernels_op.h"
}
#ifndef SABER_OP_NS      // host files only: inside a kernel file (namespace op_*) unqualified names are that namespace's own launchers
#define SABER_OP_FWD(fn)                                                                                                  \
    template <class... A> inline auto fn(A&&... a) {                                                                      \
        return saber_op_is_f16() ? op_f16::fn(static_cast<A&&>(a)...) : op_bf16::fn(static_cast<A&&>(a)...);              \
    }
// one-time kernel attribute setup: both builds
#define SABER_OP_INIT(fn)                                                                                                 \
    inline const char* fn() { const char* m = op_bf16::fn(); return m ? m : op_f16::fn(); }
SABER_OP_INIT(gemm_init_device) SABER_OP_INIT(gemm_rowln_init_device) SABER_OP_INIT(hiera_attention_init_device)
SABER_OP_INIT(decoder_fused_init_device) SABER_OP_INIT(decoder_tokens_init_device)
SABER_OP_FWD(launch_gemm) SABER_OP_FWD(gemm_rowln_supported) SABER_OP_FWD(launch_gemm_rowln) SABER_OP_FWD(gemm_rowln_packed_elems) SABER_OP_FWD(launch_pack_w_kstep)
SABER_OP_FWD(launch_layernorm) SABER_OP_FWD(launch_gather_rows) SABER_OP_FWD(launch_add_to_bf16) SABER_OP_FWD(launch_hiera_attention)
SABER_OP_FWD(launch_prompt_tokens) SABER_OP_FWD(launch_prompt_tokens_multi) SABER_OP_FWD(launch_mask_embed_src) SABER_OP_FWD(launch_mask_hidden) SABER_OP_FWD(launch_embb_tiles)
SABER_OP_FWD(launch_dec_attention) SABER_OP_FWD(launch_mask_dot) SABER_OP_FWD(launch_mask_pick) SABER_OP_FWD(launch_iou_live_flags) SABER_OP_FWD(launch_mask_select)
SABER_OP_FWD(launch_dec_fold) SABER_OP_FWD(launch_dec_t2i) SABER_OP_FWD(launch_dec_i2t) SABER_OP_FWD(launch_dec_i2t_t2i) SABER_OP_FWD(launch_dec_upscale) SABER_OP_FWD(launch_dec_tokens)
SABER_OP_FWD(launch_mask_post) SABER_OP_FWD(launch_gather_masks) SABER_OP_FWD(launch_label_plane) SABER_OP_FWD(launch_pair_intersections) SABER_OP_FWD(launch_unpermute_nchw)
SABER_OP_FWD(launch_rope) SABER_OP_FWD(launch_softmax_rows) SABER_OP_FWD(launch_conv3x3s2) SABER_OP_FWD(launch_unpack_masks) SABER_OP_FWD(launch_paint_nearest)
SABER_OP_FWD(launch_conv3x3s2_t) SABER_OP_FWD(launch_dwconv7_t) SABER_OP_FWD(launch_dwconv7) SABER_OP_FWD(launch_conv4x4s4) SABER_OP_FWD(launch_resize_plane)
SABER_OP_FWD(launch_flash256) SABER_OP_FWD(launch_gauss_mirror) SABER_OP_FWD(launch_axpy) SABER_OP_FWD(launch_bf16_to_f32)
#undef SABER_OP_FWD
#undef SABER_OP_INIT
#endif
