// Kernel-level C-ABI (include/saber_amd_kernels.h): thin wrappers over the launchers.
#include <string>

#include "../../include/saber_amd_kernels.h"
#include "common.h"
#include "kernels.h"

#include <atomic>

static thread_local std::string g_kerr;
static int kfail(const char* m) { g_kerr = m ? m : "unknown"; return -1; }
static int kcheck(const char* m) {
    if (m) return kfail(m);
    hipError_t st = hipGetLastError();
    if (st != hipSuccess) return kfail(hipGetErrorString(st));
    return 0;
}

extern "C" const char* saber_k_last_error(void) { return g_kerr.c_str(); }

extern "C" int saber_k_init(int device_id) {
    if (hipSetDevice(device_id) != hipSuccess) return kfail("hipSetDevice failed");
    const char* m = gemm_init_device();
    if (!m) m = gemm_rowln_init_device();
    if (!m) m = hiera_attention_init_device();
    if (!m) m = image_ops_init_device();
    if (!m) m = decoder_fused_init_device();
    if (!m) m = amg_device_init();
    (void)hipGetLastError();
    if (m) return kfail(m);
    return 0;
}

extern "C" int saber_k_gemm(const uint16_t* A, const uint16_t* W, const float* bias, const float* res, float* out_f32,
                            uint16_t* out_bf16, int M, int N, int K, int act, int act_last, int pool4, int res_shift, int res_mod,
                            void* stream) {
    GemmParams p;
    p.A = A; p.lda = K; p.W = W; p.ldw = K; p.bias = bias; p.res = res; p.ldres = N; p.Cf = out_f32; p.ldcf = N; p.Cb = out_bf16; p.ldcb = N;
    p.M = M; p.N = N; p.K = K; p.act = act; p.act_last = act_last; p.pool4 = pool4; p.res_shift = res_shift; p.res_mod = res_mod;
    return kcheck(launch_gemm(p, (hipStream_t)stream));
}

extern "C" int saber_k_gemm_ld(const uint16_t* A, int lda, const uint16_t* W, int ldw, int w_kpad, const float* bias, const float* res,
                               float* out_f32, uint16_t* out_bf16, int M, int N, int K, int act, void* stream) {
    GemmParams p;
    p.A = A; p.lda = lda; p.W = W; p.ldw = ldw; p.bias = bias; p.res = res; p.ldres = N; p.Cf = out_f32; p.ldcf = N; p.Cb = out_bf16; p.ldcb = N;
    p.M = M; p.N = N; p.K = K; p.act = act; p.w_kpad = w_kpad;
    // the widest 16-bit-output GEMMs read W packed per K-step (the engine packs qkv / mlp.layers.0 once at finalize: engine.hip); here: into a
    // scratch kept per thread, so that the kernel-level tests and tools/gemm_bench.py reach the kernels the engine runs
    if (w_kpad && out_bf16 && !out_f32 && !res && (K % 64) == 0 && (N & 7) == 0 && ((int64_t)M * N >= (int64_t)1024 * 65536 || (g_saber_debug_flags & 128))) {
        static thread_local bf16_t* scratch = nullptr;
        static thread_local size_t scratch_elems = 0;
        const size_t need = gemm_rowln_packed_elems(N, K);
        if (need > scratch_elems) {
            if (scratch) { (void)hipDeviceSynchronize(); (void)hipFree(scratch); scratch = nullptr; scratch_elems = 0; }
            if (hipMalloc(reinterpret_cast<void**>(&scratch), need * sizeof(bf16_t)) != hipSuccess) return kfail("gemm: scratch allocation failed");
            scratch_elems = need;
        }
        if (const char* m = launch_pack_w_kstep(W, ldw, N, K, scratch, (hipStream_t)stream)) return kfail(m);
        p.Wpk = scratch;
    }
    return kcheck(launch_gemm(p, (hipStream_t)stream));
}

extern "C" int saber_k_gemm_rowln(const uint16_t* A, int lda, const uint16_t* W, int ldw, const float* bias, const float* res, float* out_f32,
                                  uint16_t* out_bf16, const float* ln_gamma, const float* ln_beta, float ln_eps, uint16_t* ln_out, int M, int N, int K,
                                  void* stream) {
    // the kernel reads W packed per K-step (the engine packs its weights once at finalize); here: into a scratch kept per thread
    static thread_local bf16_t* scratch = nullptr;
    static thread_local size_t scratch_elems = 0;
    const size_t need = gemm_rowln_packed_elems(N, K);
    if (need > scratch_elems) {
        if (scratch) { (void)hipDeviceSynchronize(); (void)hipFree(scratch); scratch = nullptr; scratch_elems = 0; }
        if (hipMalloc(reinterpret_cast<void**>(&scratch), need * sizeof(bf16_t)) != hipSuccess) return kfail("gemm_rowln: scratch allocation failed");
        scratch_elems = need;
    }
    if (const char* m = launch_pack_w_kstep(W, ldw, N, K, scratch, (hipStream_t)stream)) return kfail(m);
    GemmParams p;
    p.A = A; p.lda = lda; p.W = W; p.ldw = ldw; p.w_kpad = 1; p.bias = bias; p.res = res; p.ldres = N; p.Cf = out_f32; p.ldcf = N; p.Cb = out_bf16; p.ldcb = N;
    p.M = M; p.N = N; p.K = K; p.ln_gamma = ln_gamma; p.ln_beta = ln_beta; p.ln_eps = ln_eps; p.ln_out = ln_out; p.ldln = N; p.Wpk = scratch;
    return kcheck(launch_gemm_rowln(p, (hipStream_t)stream));
}

extern "C" int saber_k_layernorm(const float* x, const float* gamma, const float* beta, float eps, float* out_f32, uint16_t* out_bf16,
                                 int rows, int C, int act, void* stream) {
    LayerNormParams p;
    p.x = x; p.ldx = C; p.gamma = gamma; p.beta = beta; p.eps = eps; p.out_f = out_f32; p.out_bf = out_bf16; p.ldo = C; p.rows = rows; p.C = C; p.act = act;
    return kcheck(launch_layernorm(p, (hipStream_t)stream));
}

extern "C" int saber_k_hiera_attention(const uint16_t* qkv, uint16_t* out, int n_windows, int nk, int heads, int q_pool, void* stream) {
    return kcheck(launch_hiera_attention(qkv, out, n_windows, nk, heads, 72, q_pool, nullptr, (hipStream_t)stream));
}

extern "C" int saber_k_hiera_attention_ex(const uint16_t* qkv, uint16_t* out, int n_windows, int nk, int heads, int head_dim, int q_pool,
                                          const uint8_t* key_mask, void* stream) {
    return kcheck(launch_hiera_attention(qkv, out, n_windows, nk, heads, head_dim, q_pool, key_mask, (hipStream_t)stream));
}

extern "C" int saber_k_dec_attention(const float* q, const float* k, const float* v, uint16_t* out, int B, int nq, int nk, int heads,
                                     int hd, int k_shared, void* stream) {
    const int64_t C = (int64_t)heads * hd;
    return kcheck(launch_dec_attention(q, k, v, out, B, nq, nk, heads, hd, nq * C, k_shared ? 0 : nk * C, k_shared ? 0 : nk * C, nq * C,
                                       (hipStream_t)stream));
}

extern "C" int saber_k_prepare(const void* img, int dtype, int H, int W, float* out, float* ws_dev, uint32_t* minmax_dev, void* stream) {
    if (dtype == 0) return kcheck(launch_prepare_u16((const uint16_t*)img, H, W, out, ws_dev, minmax_dev, (hipStream_t)stream));
    if (dtype == 1) return kcheck(launch_prepare_f32((const float*)img, H, W, out, ws_dev, minmax_dev, (hipStream_t)stream));
    return kfail("prepare: dtype must be 0 (u16) or 1 (f32)");
}

extern "C" int saber_k_mask_post(const float* lowres, int n, int crop_x0, int crop_y0, int crop_w, int crop_h, int H, int W, float thr,
                                 float offset, uint32_t* bits, int32_t* stats, void* stream) {
    return kcheck(launch_mask_post(lowres, nullptr, n, crop_x0, crop_y0, crop_w, crop_h, H, W, thr, offset, bits,
                                   reinterpret_cast<MaskStats*>(stats), (hipStream_t)stream));
}

extern "C" int saber_k_perm_index(int y, int x, int stage) { return perm_index(y, x, stage); }

extern "C" int saber_k_dec_i2t(const uint16_t* X, int64_t x_batch_stride, const uint16_t* peq, const uint16_t* Kt, const float* tk, float kscale, const float* cb,
                               const uint16_t* VtT, const float* bo, const float* gamma, const float* beta, float eps, uint16_t* Xout, int P,
                               void* stream) {
    return kcheck(launch_dec_i2t(X, XMap{x_batch_stride, 1, 0}, peq, Kt, tk, kscale, cb, VtT, bo, gamma, beta, eps, Xout, P, (hipStream_t)stream));
}

extern "C" int saber_k_dec_t2i(const uint16_t* X, int64_t x_batch_stride, const uint16_t* pek, const uint16_t* Qt, const float* tq, float qscale, float* part_ws,
                               float* ml_ws, int P, int split, const uint16_t* Wv, const float* bv, uint16_t* out, void* stream) {
    return kcheck(launch_dec_t2i(X, XMap{x_batch_stride, 1, 0}, pek, Qt, tq, qscale, part_ws, ml_ws, P, split, Wv, bv, out, (hipStream_t)stream));
}

int g_saber_debug_flags = 0;
unsigned long long* g_saber_stamp_buf = nullptr;   // development: device buffer for in-kernel cycle stamps (nullptr in production)
extern "C" void saber_k_set_stamp_buffer(void* dev) { g_saber_stamp_buf = (unsigned long long*)dev; }
extern "C" void saber_k_set_debug(int flags) { g_saber_debug_flags = flags; }
// 16-bit operand type of the kernel-level entry points called from THIS thread: 0 = bf16 (default), 1 = fp16 (common.h "OPERAND TYPE").
// The engine-level C-ABI sets it per call from the handle's precision mode and restores it on return (DeviceGuard, engine.h).
thread_local int g_saber_op_f16 = 0;
bf16_t saber_host_f2h(float f);          // engine.hip
// host-side fp32 -> IEEE half conversion the engine converts its weights with (exposed for the CPU test against numpy.float16)
extern "C" void saber_k_host_f32_to_f16(const float* in, uint16_t* out, int64_t n) { for (int64_t i = 0; i < n; ++i) out[i] = saber_host_f2h(in[i]); }
// A HIP stream whose kernels run on the first n_cus compute units of `first_cu`.. (hipExtStreamCreateWithCUMask; bit i of the mask = CU i in
// the runtime's numbering, which interleaves the XCDs: consecutive indices are spread over all eight) - the co-residency experiment of
// VERDICT r03 item 4 (tools/cu_mask_bench.py).  The handle can be wrapped in torch.cuda.ExternalStream.
extern "C" int saber_k_stream_create_cu_range(int first_cu, int n_cus, void** out_stream) {
    if (!out_stream || first_cu < 0 || n_cus < 1 || first_cu + n_cus > 1024) return kfail("stream_create_cu_range: bad argument");
    uint32_t mask[32] = {0};
    for (int c = first_cu; c < first_cu + n_cus; ++c) mask[c >> 5] |= 1u << (c & 31);
    hipStream_t s = nullptr;
    const hipError_t st = hipExtStreamCreateWithCUMask(&s, (uint32_t)((first_cu + n_cus + 31) / 32), mask);
    if (st != hipSuccess) return kfail(hipGetErrorString(st));
    *out_stream = (void*)s;
    return 0;
}
extern "C" int saber_k_stream_destroy(void* stream) { return hipStreamDestroy((hipStream_t)stream) == hipSuccess ? 0 : kfail("hipStreamDestroy failed"); }
int saber_cu_count() {
    static std::atomic<int> table[64];                      // zero-initialised; 0 = not queried yet
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    int n = table[dev].load(std::memory_order_relaxed);
    if (n > 0) return n;
    hipDeviceProp_t pr;
    n = (hipGetDeviceProperties(&pr, dev) == hipSuccess && pr.multiProcessorCount > 0) ? pr.multiProcessorCount : 256;
    table[dev].store(n, std::memory_order_relaxed);         // (every thread that races here computes the same value)
    return n;
}

extern "C" int saber_k_set_operand_type(int f16) { const int prev = g_saber_op_f16; g_saber_op_f16 = f16 ? 1 : 0; return prev; }

// ------------------------------------------------------------------------------------------------ video (memory) path kernels
extern "C" int saber_k_rope(const float* x, int64_t rows, int n_rot, int C, int side, float theta, float* out_f32, uint16_t* out_bf16, void* stream) {
    return kcheck(launch_rope(x, rows, n_rot, C, side, theta, out_f32, out_bf16, (hipStream_t)stream));
}
extern "C" int saber_k_softmax_rows(const float* S, int64_t ld_s, int64_t rows, int n, float scale, uint16_t* P, int64_t ld_p, void* stream) {
    return kcheck(launch_softmax_rows(S, ld_s, rows, n, scale, P, ld_p, (hipStream_t)stream));
}
extern "C" int saber_k_conv3x3s2(const float* in, int H, int W, int Cin, const float* w, const float* b, int Cout, float* out, void* stream) {
    return kcheck(launch_conv3x3s2(in, H, W, Cin, w, b, Cout, out, (hipStream_t)stream));
}
extern "C" int saber_k_conv3x3s2_t(const float* in, int H, int W, int Cin, const float* wt, const float* b, int Cout, float* out, void* stream) {
    return kcheck(launch_conv3x3s2_t(in, H, W, Cin, wt, b, Cout, out, (hipStream_t)stream));
}
extern "C" int saber_k_paint_nearest(const float* logits, int Hv, int Wv, float thr, int label, uint16_t* plane, int H, int W, int* any_flag, void* stream) {
    return kcheck(launch_paint_nearest(logits, Hv, Wv, thr, label, plane, H, W, any_flag, (hipStream_t)stream));
}
extern "C" int saber_k_unpack_masks(const uint32_t* bits, int n, int H, int W, uint8_t* out, void* stream) {
    if (n > 0 && (!bits || !out)) return kcheck("unpack_masks: null pointer");
    return kcheck(launch_unpack_masks(bits, n, H, W, out, (hipStream_t)stream));
}
extern "C" int saber_k_dwconv7(const float* in, int H, int W, int C, const float* w, const float* b, float* out, void* stream) {
    return kcheck(launch_dwconv7(in, H, W, C, w, b, out, (hipStream_t)stream));
}
extern "C" int saber_k_dwconv7_t(const float* in, int H, int W, int C, const float* wt, const float* b, float* out, void* stream) {
    return kcheck(launch_dwconv7_t(in, H, W, C, wt, b, out, (hipStream_t)stream));
}
extern "C" int saber_k_conv4x4s4(const float* in, int H, int W, const float* w, const float* b, float* out, void* stream) {
    return kcheck(launch_conv4x4s4(in, H, W, w, b, out, (hipStream_t)stream));
}
extern "C" int saber_k_resize_plane(const float* in, int n_planes, int H, int W, float* out, int Ho, int Wo, int antialias, int post, float a, float c, void* stream) {
    return kcheck(launch_resize_plane(in, n_planes, H, W, out, Ho, Wo, antialias, post, a, c, (hipStream_t)stream));
}
extern "C" int saber_k_gemm_mx(const uint8_t* A, int64_t lda, const uint8_t* SA, int64_t sa_rows, const uint8_t* W, int64_t ldw, const uint8_t* SW, int64_t sw_rows, const float* bias,
                               const float* res, float* out_f32, uint16_t* out_bf16, uint8_t* out_mx, uint8_t* out_mx_scales, int64_t out_mx_rows, int64_t ldc, int64_t M, int N, int Kp,
                               int act, void* stream) {
    GemmMxParams p;
    p.A = A; p.lda = lda; p.SA = SA; p.sa_rows = sa_rows; p.W = W; p.ldw = ldw; p.SW = SW; p.sw_rows = sw_rows; p.bias = bias;
    p.res = res; p.ldres = ldc; p.Cf = out_f32; p.ldcf = ldc; p.Cb = out_bf16; p.ldcb = ldc; p.C8 = out_mx; p.ldc8 = ldc; p.SC = out_mx_scales; p.sc_rows = out_mx_rows;
    p.M = M; p.N = N; p.Kp = Kp; p.act = act;
    return kcheck(launch_gemm_mx(p, (hipStream_t)stream));
}
extern "C" int saber_k_quant_mx(const uint16_t* x, int64_t ldx, int C, uint8_t* out, int64_t ldo, int Kp, uint8_t* scales, int64_t scale_rows, int64_t M, void* stream) {
    return kcheck(launch_quant_mx_bf16(x, ldx, C, out, ldo, Kp, scales, scale_rows, M, (hipStream_t)stream));
}
extern "C" int saber_k_ln_mx(const float* x, int64_t ldx, const float* gamma, const float* beta, float eps, int C, uint8_t* out, int64_t ldo, int Kp, uint8_t* scales,
                             int64_t scale_rows, int64_t M, void* stream) {
    return kcheck(launch_ln_mx(x, ldx, gamma, beta, eps, C, out, ldo, Kp, scales, scale_rows, M, (hipStream_t)stream));
}
extern "C" int saber_k_box_nms(const float* boxes_xyxy, const float* scores, int n, float iou_thresh, void* scratch, int* keep_out, int* count_out, void* stream) {
    return kcheck(launch_box_nms_probe(boxes_xyxy, scores, n, iou_thresh, scratch, keep_out, count_out, (hipStream_t)stream));
}
extern "C" int saber_k_flash256(const uint16_t* Q, const uint16_t* K, const uint16_t* V, int n_q, int n_keys, float scale, const float* bias_v, uint16_t* out, float* ws,
                                int64_t ws_floats, void* stream) {
    return kcheck(launch_flash256(Q, K, V, n_q, n_keys, scale, bias_v, out, ws, (size_t)ws_floats, (hipStream_t)stream));
}
extern "C" int saber_k_gauss_mirror(const float* in, float* out, int n_planes, int H, int W, int axis, double sigma, void* stream) {
    return kcheck(launch_gauss_mirror(in, out, n_planes, H, W, axis, sigma, (hipStream_t)stream));
}
extern "C" int saber_k_axpy(const float* x, const float* y, const float* g, float alpha, int64_t rows, int C, float* out, void* stream) {
    return kcheck(launch_axpy(x, y, g, alpha, rows, C, out, (hipStream_t)stream));
}
extern "C" int saber_k_add_to_bf16(const float* x, const float* y, int y_rows, uint16_t* out_bf16, float* out_f32, int64_t rows, int C, void* stream) {
    return kcheck(launch_add_to_bf16(x, y, y_rows, out_bf16, out_f32, rows, C, (hipStream_t)stream));
}
// batched GEMM with explicit strides (scores / values of the memory attention, one batch entry per tracked object)
extern "C" int saber_k_gemm_batched(const uint16_t* A, int lda, int64_t strideA, const uint16_t* W, int ldw, int64_t strideW, const float* bias, float* out_f32,
                                    int ldcf, int64_t strideCf, uint16_t* out_bf16, int ldcb, int64_t strideCb, int M, int N, int K, int batch, void* stream) {
    GemmParams p;
    p.A = A; p.lda = lda; p.strideA = strideA; p.W = W; p.ldw = ldw; p.strideW = strideW; p.bias = bias; p.Cf = out_f32; p.ldcf = ldcf; p.strideCf = strideCf;
    p.Cb = out_bf16; p.ldcb = ldcb; p.strideCb = strideCb; p.M = M; p.N = N; p.K = K; p.batch = batch;
    return kcheck(launch_gemm(p, (hipStream_t)stream));
}
extern "C" int saber_k_bf16_to_f32(const uint16_t* x, int64_t n, float* out, void* stream) { return kcheck(launch_bf16_to_f32(x, n, out, (hipStream_t)stream)); }
