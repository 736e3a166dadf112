// Engine: weights, workspaces, the batched Hiera-L encoder pass and the batched prompt decoder.
// C-ABI entry points are declared in include/saber_amd.h (each cites the reference interface it replaces).
#include "engine.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <functional>
#include <utility>

#include "common.h"

static thread_local std::string g_create_err;

int eng_fail(saber_engine* e, int code, const std::string& msg) {
    if (e) e->err = msg; else g_create_err = msg;
    return code;
}

int eng_alloc_bytes(saber_engine* e, void** p, size_t bytes) {
    *p = nullptr;
    if (bytes == 0) bytes = 16;
    hipError_t st = hipMalloc(p, bytes);
    if (st != hipSuccess) return eng_fail(e, SABER_ERR_HIP, std::string("hipMalloc(") + std::to_string(bytes) + "): " + hipGetErrorString(st));
    e->allocs.push_back(*p);
    return SABER_OK;
}
void eng_free(saber_engine* e, void* p) {
    if (!p) return;
    for (size_t i = 0; i < e->allocs.size(); ++i)
        if (e->allocs[i] == p) { e->allocs.erase(e->allocs.begin() + i); break; }
    (void)hipFree(p);
}
template <typename T> int eng_alloc(saber_engine* e, T** p, size_t count) { return eng_alloc_bytes(e, reinterpret_cast<void**>(p), count * sizeof(T)); }
template int eng_alloc<float>(saber_engine*, float**, size_t);
template int eng_alloc<bf16_t>(saber_engine*, bf16_t**, size_t);
template int eng_alloc<int>(saber_engine*, int**, size_t);
template int eng_alloc<uint32_t>(saber_engine*, uint32_t**, size_t);
template int eng_alloc<MaskStats>(saber_engine*, MaskStats**, size_t);
template int eng_alloc<double>(saber_engine*, double**, size_t);
template int eng_alloc<uint8_t>(saber_engine*, uint8_t**, size_t);
template int eng_alloc<DevCand>(saber_engine*, DevCand**, size_t);
template int eng_alloc<DevCrop>(saber_engine*, DevCrop**, size_t);
template int eng_alloc<saber_mask_meta>(saber_engine*, saber_mask_meta**, size_t);

#define TRY(x) do { int _r = (x); if (_r != SABER_OK) return _r; } while (0)

// ------------------------------------------------------------------------------------------------ model description
struct TrunkSpec { const char* name; int embed_dim, heads0; int stages[4]; int globals[3]; int window_spec[4]; int pe_bkg; };
// configs/sam2.1/sam2.1_hiera_{t,s,b+,l}.yaml of the third-party sam2 package (selected by saber/pretrained_weights.py:183-188)
static const TrunkSpec kTrunks[4] = {
    {"tiny", 96, 1, {1, 2, 7, 2}, {5, 7, 9}, {8, 4, 14, 7}, 7},
    {"small", 96, 1, {1, 2, 11, 2}, {7, 10, 13}, {8, 4, 14, 7}, 7},
    {"base", 112, 2, {2, 3, 16, 3}, {12, 16, 20}, {8, 4, 14, 7}, 14},
    {"large", 144, 2, {2, 6, 36, 4}, {23, 33, 43}, {8, 4, 16, 8}, 7},
};

static const char* hiera_spec(saber_engine* e, const TrunkSpec& t) {
    e->embed_dim = t.embed_dim;
    e->head_dim = t.embed_dim / t.heads0;
    e->pe_bkg = t.pe_bkg;
    e->padded = t.window_spec[2] == 14;
    if (e->padded) {
        if (t.window_spec[3] != 7 || t.stages[2] < 2) return "trunk spec: padded layout needs 14/7 windows and >= 2 blocks in stage 2";
        e->tok_rows[2] = 25 * 196;
        e->tok_rows[3] = 25 * 49;
    } else if (t.window_spec[2] != 16 || t.window_spec[3] != 8) return "trunk spec: unsupported window sizes";
    if (t.window_spec[0] != 8 || t.window_spec[1] != 4) return "trunk spec: unsupported window sizes";
    e->blocks.clear(); e->stage_ends.clear(); e->stage_dims.clear();
    int idx = 0;
    for (int s = 0; s < 4; ++s) {
        const int dim = t.embed_dim << s, heads = t.heads0 << s;
        e->stage_dims.push_back(dim);
        for (int b = 0; b < t.stages[s]; ++b, ++idx) {
            BlockSpec bs;
            const bool first = s > 0 && b == 0;
            bs.dout = dim;
            bs.din = first ? dim / 2 : dim;
            bs.heads = heads;
            bs.window = first ? t.window_spec[s - 1] : t.window_spec[s];
            for (int g : t.globals)
                if (g == idx) {
                    if (s != 2 || first) return "trunk spec: global attention blocks must lie inside stage 2";
                    bs.window = 0;
                }
            bs.q_stride = first ? 2 : 1;
            e->blocks.push_back(bs);
        }
        e->stage_ends.push_back(idx - 1);
    }
    return nullptr;
}

// ------------------------------------------------------------------------------------------------ C-ABI: lifecycle
// ------------------------------------------------------------------------------------------------ hipGraph replay
// body() issues a fixed launch sequence on stream s (no host synchronisation, no allocation).  `key` names the sequence AND every pointer /
// size baked into its launches.  First sight: eager (lazy one-time setup inside launchers happens here); second: captured + launched;
// afterwards: one hipGraphLaunch.  Anything that goes wrong while capturing marks the key bad and the sequence stays eager.
int eng_graphed(saber_engine* e, const std::string& key, hipStream_t s, const std::function<int()>& body) {
    if (!e->graphs_on || e->prof_on || s == nullptr || e->precision == SABER_PRECISION_EXACT || e->graph_bad.count(key)) return body();
    auto it = e->graphs.find(key);
    if (it != e->graphs.end()) {
        ENG_HIP(e, hipGraphLaunch(it->second, s));
        ++e->graph_replays;
        return SABER_OK;
    }
    if (!e->graph_seen.count(key)) { e->graph_seen.insert(key); return body(); }
    if (e->graphs.size() >= 256 || e->graph_seen.size() >= 4096) return body();      // a caller that never repeats a shape: stay eager, bounded memory
    if (hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal) != hipSuccess) { (void)hipGetLastError(); e->graph_bad.insert(key); return body(); }
    const int r = body();
    hipGraph_t g = nullptr;
    const hipError_t st = hipStreamEndCapture(s, &g);
    if (r != SABER_OK || st != hipSuccess || !g) {
        if (g) (void)hipGraphDestroy(g);
        (void)hipGetLastError();
        e->graph_bad.insert(key);
        return body();                              // nothing ran while capturing: run it now, eagerly
    }
    hipGraphExec_t exec = nullptr;
    const hipError_t si = hipGraphInstantiate(&exec, g, nullptr, nullptr, 0);
    (void)hipGraphDestroy(g);
    if (si != hipSuccess || !exec) { (void)hipGetLastError(); e->graph_bad.insert(key); return body(); }
    e->graphs[key] = exec;
    ++e->graph_captures;
    ENG_HIP(e, hipGraphLaunch(exec, s));
    return SABER_OK;
}
void eng_graphs_flush(saber_engine* e) {
    if (e->graphs.empty() && e->graph_seen.empty()) return;
    (void)hipDeviceSynchronize();                 // a replay may still be running
    for (auto& kv : e->graphs) (void)hipGraphExecDestroy(kv.second);
    e->graphs.clear(); e->graph_seen.clear(); e->graph_bad.clear();
}
extern "C" int saber_engine_set_weight_format(saber_engine* e, int format) {
    if (!e) return SABER_ERR_INVALID;
    if (e->finalized) return eng_fail(e, SABER_ERR_STATE, "set_weight_format after finalize");
    if (format != SABER_WEIGHTS_BF16 && format != SABER_WEIGHTS_FP8_E4M3 && format != SABER_WEIGHTS_MXFP8) return eng_fail(e, SABER_ERR_INVALID, "set_weight_format: unknown format");
    e->weight_format = format;
    return SABER_OK;
}
extern "C" int saber_engine_set_precision(saber_engine* e, int precision) {
    if (!e) return SABER_ERR_INVALID;
    if (precision != SABER_PRECISION_BF16 && precision != SABER_PRECISION_EXACT && precision != SABER_PRECISION_FP16) return eng_fail(e, SABER_ERR_INVALID, "set_precision: unknown precision");
    if (!e->finalized) {
        if (precision == SABER_PRECISION_EXACT) e->keep_f32 = true;
        else e->op_f16 = precision == SABER_PRECISION_FP16;      // the 16-bit type the weights will be converted to
    } else {
        if (precision == SABER_PRECISION_EXACT && !e->keep_f32)
            return eng_fail(e, SABER_ERR_STATE, "set_precision: the exact mode needs the fp32 weight copies; request it once before saber_engine_finalize");
        if (precision == SABER_PRECISION_FP16 && !e->op_f16)
            return eng_fail(e, SABER_ERR_STATE, "set_precision: this handle was finalized with bf16 weights; choose SABER_PRECISION_FP16 before saber_engine_finalize");
        if (precision == SABER_PRECISION_BF16 && e->op_f16)
            return eng_fail(e, SABER_ERR_STATE, "set_precision: this handle was finalized with fp16 weights; it runs in SABER_PRECISION_FP16 (or EXACT)");
    }
    e->precision = precision;
    return SABER_OK;
}
extern "C" int saber_engine_set_iou_pruning(saber_engine* e, int enable) {
    if (!e) return SABER_ERR_INVALID;
    if ((bool)enable != e->iou_prune) { e->iou_prune = enable != 0; eng_graphs_flush(e); }      // (captured decode sequences contain the choice)
    return SABER_OK;
}
extern "C" int saber_engine_set_encoder_stream(saber_engine* e, void* stream) {
    if (!e) return SABER_ERR_INVALID;
    if ((hipStream_t)stream != e->enc_stream) { eng_graphs_flush(e); e->enc_stream = (hipStream_t)stream; }     // (captured encoder passes belong to their stream)
    return SABER_OK;
}
extern "C" int saber_engine_set_graphs(saber_engine* e, int enable) {
    if (!e) return SABER_ERR_INVALID;
    e->graphs_on = enable != 0;
    return SABER_OK;
}
extern "C" int saber_engine_graph_stats(const saber_engine* e, int* captures, int* replays) {
    if (!e) return SABER_ERR_INVALID;
    if (captures) *captures = e->graph_captures;
    if (replays) *replays = e->graph_replays;
    return SABER_OK;
}

extern "C" int saber_engine_create(int device_id, const char* trunk, int max_images, int max_prompts, saber_engine** out) {
    if (!out) return eng_fail(nullptr, SABER_ERR_INVALID, "saber_engine_create: out is NULL");
    *out = nullptr;
    if (!trunk) return eng_fail(nullptr, SABER_ERR_INVALID, "saber_engine_create: trunk is NULL");
    const std::string t(trunk);
    if (t != "tiny" && t != "small" && t != "base" && t != "large")
        return eng_fail(nullptr, SABER_ERR_INVALID, "cfg must be one of tiny/small/base/large, got '" + t + "'");
    const TrunkSpec* spec = nullptr;
    for (const TrunkSpec& k : kTrunks) if (t == k.name) spec = &k;
    if (max_images < 1 || max_images > 64 || max_prompts < 1 || max_prompts > 4096)
        return eng_fail(nullptr, SABER_ERR_INVALID, "saber_engine_create: max_images must be 1..64 and max_prompts 1..4096");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return eng_fail(nullptr, SABER_ERR_HIP, "no HIP device visible: the MI355X engine cannot run");
    if (device_id < 0 || device_id >= ndev) return eng_fail(nullptr, SABER_ERR_INVALID, "device_id out of range");
    if (hipSetDevice(device_id) != hipSuccess) return eng_fail(nullptr, SABER_ERR_HIP, "hipSetDevice failed");
    saber_engine* e = new saber_engine();
    e->device = device_id;
    e->trunk = t;
    e->max_images = max_images;
    e->max_prompts = max_prompts;
    if (const char* g = getenv("SABER_AMD_GRAPHS")) e->graphs_on = atoi(g) != 0;
    if (const char* m = hiera_spec(e, *spec)) { delete e; return eng_fail(nullptr, SABER_ERR_INVALID, m); }
    {
        const char* m = gemm_init_device();
            if (!m) m = gemm_rowln_init_device();
        if (!m) m = amg_device_init();
        if (!m) m = hiera_attention_init_device();
        if (!m) m = image_ops_init_device();
        if (!m) m = decoder_fused_init_device();
        if (!m) m = decoder_tokens_init_device();
        (void)hipGetLastError();
        if (m) { delete e; return eng_fail(nullptr, SABER_ERR_HIP, std::string("kernel attribute setup: ") + m); }
    }
    *out = e;
    return SABER_OK;
}

extern "C" void saber_engine_destroy(saber_engine* e) {
    if (!e) return;
    (void)hipSetDevice(e->device);
    (void)hipDeviceSynchronize();
    for (auto& kv : e->graphs) (void)hipGraphExecDestroy(kv.second);
    exact_release(e);
    if (e->crops_pin) (void)hipHostFree(e->crops_pin);
    for (hipEvent_t ev : e->crops_ev) if (ev) (void)hipEventDestroy(ev);
    for (hipEvent_t ev : e->enc_ev) if (ev) (void)hipEventDestroy(ev);
    for (void* p : e->allocs) (void)hipFree(p);
    delete e;
}

extern "C" const char* saber_last_error(const saber_engine* e) { return e ? e->err.c_str() : g_create_err.c_str(); }

extern "C" int saber_engine_set_weight(saber_engine* e, const char* name, const float* host, const int64_t* shape, int ndim) {
    if (!e) return SABER_ERR_INVALID;
    if (!name || !host || !shape || ndim < 1 || ndim > 4) return eng_fail(e, SABER_ERR_INVALID, "set_weight: bad argument");
    if (e->finalized) return eng_fail(e, SABER_ERR_STATE, "set_weight after finalize");
    HostTensor t;
    int64_t n = 1;
    for (int i = 0; i < ndim; ++i) { if (shape[i] <= 0) return eng_fail(e, SABER_ERR_INVALID, "set_weight: non-positive dim"); t.shape.push_back(shape[i]); n *= shape[i]; }
    t.data.assign(host, host + n);
    e->host_w[name] = std::move(t);
    return SABER_OK;
}

// ------------------------------------------------------------------------------------------------ finalize helpers
static inline bf16_t host_f2bf(float f) {
    uint32_t u; memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (bf16_t)((u >> 16) | 0x40);  // NaN stays NaN
    u += 0x7fffu + ((u >> 16) & 1u);
    return (bf16_t)(u >> 16);
}
// fp32 -> IEEE half bits, round to nearest even (subnormals included; overflow -> infinity, which the finalize range check rejects first)
bf16_t saber_host_f2h(float f) {
    uint32_t u; memcpy(&u, &f, 4);
    const uint32_t sign = (u >> 16) & 0x8000u, a = u & 0x7fffffffu;
    if (a > 0x7f800000u) return (bf16_t)(sign | 0x7e00u);                     // NaN
    if (a >= 0x477ff000u) return (bf16_t)(sign | 0x7c00u);                    // >= 65520 rounds to infinity
    if (a < 0x33000001u) return (bf16_t)sign;                                 // <= 2^-25 rounds to zero
    const int ex = (int)(a >> 23) - 127;
    uint32_t man = (a & 0x7fffffu) | 0x800000u;                               // 24-bit significand
    int shift = 13;                                                           // normal: keep 10 fraction bits
    uint32_t hexp = (uint32_t)(ex + 15);
    if (ex < -14) { shift = 13 + (-14 - ex); hexp = 0; }                      // subnormal half
    const uint32_t half_ulp = 1u << (shift - 1), mask = (1u << shift) - 1u;
    uint32_t q = man >> shift;
    const uint32_t rem = man & mask;
    if (rem > half_ulp || (rem == half_ulp && (q & 1u))) ++q;
    // q carries the implicit bit for normals (bit 10): adding it to (hexp - 1) << 10 handles the mantissa overflow into the exponent
    const uint32_t h = hexp ? ((hexp - 1) << 10) + q : q;
    return (bf16_t)(sign | h);
}
static inline bf16_t host_f2op(float f, bool f16) { return f16 ? saber_host_f2h(f) : host_f2bf(f); }

struct Finalizer {
    saber_engine* e;
    int status = SABER_OK;
    const HostTensor* get(const std::string& name, std::vector<int64_t> shape) {
        if (status != SABER_OK) return nullptr;
        auto it = e->host_w.find(name);
        if (it == e->host_w.end()) { status = eng_fail(e, SABER_ERR_INVALID, "missing weight tensor '" + name + "'"); return nullptr; }
        if (it->second.shape != shape) {
            std::string got = "(", want = "(";
            for (auto d : it->second.shape) got += std::to_string(d) + ",";
            for (auto d : shape) want += std::to_string(d) + ",";
            status = eng_fail(e, SABER_ERR_INVALID, "weight '" + name + "' has shape " + got + ") expected " + want + ")");
            return nullptr;
        }
        return &it->second;
    }
    const float* up_f32(const std::vector<float>& v) {
        if (status != SABER_OK) return nullptr;
        float* d = nullptr;
        status = eng_alloc(e, &d, v.size());
        if (status != SABER_OK) return nullptr;
        if (hipMemcpy(d, v.data(), v.size() * 4, hipMemcpyHostToDevice) != hipSuccess) { status = eng_fail(e, SABER_ERR_HIP, "weight upload failed"); return nullptr; }
        return d;
    }
    const bf16_t* up_bf16(const std::vector<float>& v) {
        if (status != SABER_OK) return nullptr;
        std::vector<bf16_t> h(v.size());
        if (e->op_f16) {      // fp16 operands: the type's range is the one thing it gives up against bf16 - fail loudly, never produce infinities
            for (size_t i = 0; i < v.size(); ++i)
                if (!(std::fabs(v[i]) <= 65504.0f)) { status = eng_fail(e, SABER_ERR_INVALID, "SABER_PRECISION_FP16: a weight of magnitude " + std::to_string(std::fabs(v[i])) + " exceeds the fp16 range (65504); use SABER_PRECISION_BF16 for this checkpoint"); return nullptr; }
        }
        for (size_t i = 0; i < v.size(); ++i) h[i] = host_f2op(v[i], e->op_f16);
        bf16_t* d = nullptr;
        status = eng_alloc(e, &d, h.size());
        if (status != SABER_OK) return nullptr;
        if (hipMemcpy(d, h.data(), h.size() * 2, hipMemcpyHostToDevice) != hipSuccess) { status = eng_fail(e, SABER_ERR_HIP, "weight upload failed"); return nullptr; }
        return d;
    }
    // fp8 weight format (saber_engine_set_weight_format): OCP e4m3fn values with one power-of-two scale per output row, so that
    // value * scale is exact in bf16 (what an fp8-operand kernel that applies the scale to its fp32 accumulators computes)
    bool q8 = false;
    static float e4m3_round(float x) {
        const float a = std::fabs(x);
        if (!(a > 0.f)) return 0.f;
        int ex; (void)std::frexp(a, &ex);                      // a = f * 2^ex, f in [0.5, 1)
        const int e2 = std::max(ex - 1, -6);                   // binade (subnormals share 2^-6)
        const float quantum = std::ldexp(1.0f, e2 - 3);        // 3 mantissa bits
        const float q = std::min(std::nearbyint(a / quantum) * quantum, 448.0f);   // round to nearest even, saturate
        return x < 0.f ? -q : q;
    }
    void quantise_rows_e4m3(std::vector<float>& w, int rows, int cols) {
        for (int r = 0; r < rows; ++r) {
            float mx = 0.f;
            for (int c = 0; c < cols; ++c) mx = std::max(mx, std::fabs(w[(size_t)r * cols + c]));
            if (!(mx > 0.f)) continue;
            const float scale = std::ldexp(1.0f, (int)std::ceil(std::log2(mx / 448.0f)));
            for (int c = 0; c < cols; ++c) w[(size_t)r * cols + c] = e4m3_round(w[(size_t)r * cols + c] / scale) * scale;
        }
    }
    // MXFP8 weight format (SABER_WEIGHTS_MXFP8): e4m3 elements + one e8m0 scale per 32 K-elements, the operand format of
    // v_mfma_scale_f32_16x16x128_f8f6f4 (gemm_fp8.hip; the rule of oracle/fp8_ref.py: mx_quantise).  w is replaced by the de-quantised
    // values (exact in bf16), so every other path of the engine (bf16 kernels, exact mode) sees the same weights.
    bool mx8 = false;
    static uint8_t e4m3_byte(float q) {          // q already on the e4m3 grid, |q| <= 448
        const float a = std::fabs(q);
        if (!(a > 0.f)) return 0;
        int ex; const float f = std::frexp(a, &ex);            // a = f 2^ex, f in [0.5, 1)
        const int E = ex - 1;
        uint8_t body;
        if (E < -6) body = (uint8_t)std::lrint(std::ldexp(a, 9));                                   // subnormal: multiples of 2^-9
        else body = (uint8_t)(((E + 7) << 3) | (int)std::lrint((f * 2.0f - 1.0f) * 8.0f));
        return (q < 0.f ? 0x80 : 0) | body;
    }
    void quantise_mx(LinW* l, std::vector<float>& w, int rows, int cols) {
        const int kp = (cols + 127) / 128 * 128, rp = (rows + 191) / 192 * 192, nb = cols / 32;
        if (cols % 32) { status = eng_fail(e, SABER_ERR_INVALID, "mxfp8: weight K must be a multiple of 32"); return; }
        std::vector<uint8_t> q((size_t)rows * kp, 0), sc((size_t)(kp / 128) * rp * 4, 127);
        for (int r = 0; r < rows; ++r)
            for (int b = 0; b < nb; ++b) {
                float* v = &w[(size_t)r * cols + 32 * b];
                float mx = 0.f;
                for (int k = 0; k < 32; ++k) mx = std::max(mx, std::fabs(v[k]));
                int ex = -127;
                if (mx > 0.f) {
                    int fe; const float f = std::frexp(mx, &fe);                                      // mx = f 2^fe
                    ex = std::max(fe - 1 - 8 + (f * 2.0f > 1.75f ? 1 : 0), -127);
                }
                for (int k = 0; k < 32; ++k) {
                    const float qv = e4m3_round(std::ldexp(v[k], -ex));
                    q[(size_t)r * kp + 32 * b + k] = e4m3_byte(qv);
                    v[k] = std::ldexp(qv, ex);
                }
                sc[((size_t)(b >> 2) * rp + r) * 4 + (b & 3)] = (uint8_t)(ex + 127);
            }
        void* d = nullptr;
        status = eng_alloc_bytes(e, &d, q.size());
        if (status == SABER_OK && hipMemcpy(d, q.data(), q.size(), hipMemcpyHostToDevice) != hipSuccess) status = eng_fail(e, SABER_ERR_HIP, "weight upload failed");
        l->w8 = (const uint8_t*)d;
        if (status != SABER_OK) return;
        status = eng_alloc_bytes(e, &d, sc.size());
        if (status == SABER_OK && hipMemcpy(d, sc.data(), sc.size(), hipMemcpyHostToDevice) != hipSuccess) status = eng_fail(e, SABER_ERR_HIP, "weight upload failed");
        l->sw8 = (const uint8_t*)d; l->kp8 = kp; l->sw_rows = rp;
    }
    // [rows][cols] fp32 -> bf16 with every row zero-padded to a multiple of 64 (direct-to-LDS GEMM contract)
    void up_lin(LinW* l, const std::vector<float>& w_in, int rows, int cols) {
        std::vector<float> wq;
        if (q8) { wq = w_in; quantise_rows_e4m3(wq, rows, cols); }
        if (mx8) { wq = w_in; quantise_mx(l, wq, rows, cols); if (status != SABER_OK) return; }
        const std::vector<float>& w = (q8 || mx8) ? wq : w_in;
        const int ld = (cols + 63) / 64 * 64;
        std::vector<float> padded((size_t)rows * ld, 0.0f);
        for (int r = 0; r < rows; ++r) std::copy(w.begin() + (size_t)r * cols, w.begin() + (size_t)(r + 1) * cols, padded.begin() + (size_t)r * ld);
        l->w = up_bf16(padded);
        if (e->keep_f32) l->wf = up_f32(w);       // (of the quantised values when the e4m3 weight format is on)
        l->out = rows; l->in = cols; l->ldw = ld;
    }
    LinW lin(const std::string& prefix, int out, int in) {
        LinW l; l.out = out; l.in = in;
        const HostTensor* w = get(prefix + ".weight", {out, in});
        const HostTensor* b = get(prefix + ".bias", {out});
        if (!w || !b) return l;
        up_lin(&l, w->data, out, in); l.b = up_f32(b->data);
        return l;
    }
    LinW lin_conv1x1(const std::string& prefix, int out, int in) {
        LinW l; l.out = out; l.in = in;
        const HostTensor* w = get(prefix + ".weight", {out, in, 1, 1});
        const HostTensor* b = get(prefix + ".bias", {out});
        if (!w || !b) return l;
        up_lin(&l, w->data, out, in); l.b = up_f32(b->data);
        return l;
    }
    LnW ln(const std::string& prefix, int c) {
        LnW l;
        const HostTensor* g = get(prefix + ".weight", {c});
        const HostTensor* b = get(prefix + ".bias", {c});
        if (!g || !b) return l;
        if (e->op_f16) {      // a normalised row has |x^| <= sqrt(C - 1): bound of the LayerNorm output that becomes an fp16 operand
            float mg = 0.f, mb = 0.f;
            for (float x : g->data) mg = std::max(mg, std::fabs(x));
            for (float x : b->data) mb = std::max(mb, std::fabs(x));
            if (!(mg * std::sqrt((float)c) + mb <= 65504.0f)) { status = eng_fail(e, SABER_ERR_INVALID, "SABER_PRECISION_FP16: LayerNorm '" + prefix + "' can produce values beyond the fp16 range (max |gamma| sqrt(C) + max |beta| > 65504); use SABER_PRECISION_BF16 for this checkpoint"); return l; }
        }
        l.g = up_f32(g->data); l.b = up_f32(b->data);
        return l;
    }
    // image_side: "k" (tokens -> image attentions) or "q" (image -> tokens): that projection's weight is also kept transposed
    AttnW attn(const std::string& prefix, int internal, const char* image_side = nullptr) {
        AttnW a;
        a.q = lin(prefix + ".q_proj", internal, 256);
        a.k = lin(prefix + ".k_proj", internal, 256);
        a.v = lin(prefix + ".v_proj", internal, 256);
        a.o = lin(prefix + ".out_proj", 256, internal);
        if (image_side && internal == 128) {
            const HostTensor* w = get(prefix + "." + image_side + "_proj.weight", {internal, 256});
            if (w) {
                std::vector<float> t((size_t)256 * 128);
                for (int r = 0; r < 128; ++r)
                    for (int c = 0; c < 256; ++c) t[(size_t)c * 128 + r] = w->data[(size_t)r * 256 + c];
                a.img_wT = up_bf16(t);
            }
        }
        return a;
    }
};

// bicubic (A = -0.75, align_corners=False) sample of a (h,w) plane at output (oy, ox) of an (H,W) grid
static double cubic_w(double t, int k) {
    const double A = -0.75;
    switch (k) {
        case 0: { const double x = t + 1.0; return ((A * x - 5.0 * A) * x + 8.0 * A) * x - 4.0 * A; }
        case 1: return ((A + 2.0) * t - (A + 3.0)) * t * t + 1.0;
        case 2: { const double x = 1.0 - t; return ((A + 2.0) * x - (A + 3.0)) * x * x + 1.0; }
        default: { const double x = 2.0 - t; return ((A * x - 5.0 * A) * x + 8.0 * A) * x - 4.0 * A; }
    }
}
static double bicubic_sample(const float* plane, int h, int w, int oy, int ox, int H, int W) {
    const double sy = (double)h / H * (oy + 0.5) - 0.5, sx = (double)w / W * (ox + 0.5) - 0.5;
    const int iy = (int)std::floor(sy), ix = (int)std::floor(sx);
    const double ty = sy - iy, tx = sx - ix;
    double acc = 0.0;
    for (int j = 0; j < 4; ++j) {
        const int yy = std::min(std::max(iy - 1 + j, 0), h - 1);
        double row = 0.0;
        for (int i = 0; i < 4; ++i) {
            const int xx = std::min(std::max(ix - 1 + i, 0), w - 1);
            row += cubic_w(tx, i) * plane[yy * w + xx];
        }
        acc += cubic_w(ty, j) * row;
    }
    return acc;
}

extern "C" int saber_engine_finalize(saber_engine* e) {
    if (!e) return SABER_ERR_INVALID;
    if (e->finalized) return eng_fail(e, SABER_ERR_STATE, "finalize called twice");
    if (e->op_f16 && e->weight_format != SABER_WEIGHTS_BF16) return eng_fail(e, SABER_ERR_INVALID, "SABER_PRECISION_FP16 runs with the default weight format only (the fp8 formats are defined on bf16 operands)");
    ENG_DEVICE(e);
    Finalizer F{e};
    const int C0 = e->embed_dim;
    const std::string t = "image_encoder.trunk.";
    // ---- patch embed: wt[tap][c]
    {
        const HostTensor* w = F.get(t + "patch_embed.proj.weight", {C0, 3, 7, 7});
        const HostTensor* b = F.get(t + "patch_embed.proj.bias", {C0});
        const int bkg = e->pe_bkg;
        const HostTensor* pe = F.get(t + "pos_embed", {1, C0, bkg, bkg});
        const HostTensor* pw = F.get(t + "pos_embed_window", {1, C0, 8, 8});
        if (F.status != SABER_OK) return F.status;
        std::vector<float> wt((size_t)147 * C0);
        for (int c = 0; c < C0; ++c)
            for (int tap = 0; tap < 147; ++tap) wt[(size_t)tap * C0 + c] = w->data[(size_t)c * 147 + tap];
        e->pe_wt = F.up_f32(wt);
        e->pe_bias = F.up_f32(b->data);
        std::vector<float> pos((size_t)65536 * C0);
        for (int y = 0; y < 256; ++y)
            for (int x = 0; x < 256; ++x) {
                const size_t row = (size_t)perm_index256(y, x);
                for (int c = 0; c < C0; ++c)
                    pos[row * C0 + c] = (float)bicubic_sample(pe->data.data() + (size_t)c * bkg * bkg, bkg, bkg, y, x, 256, 256) +
                                        pw->data[(size_t)c * 64 + (y & 7) * 8 + (x & 7)];
            }
        e->pos_table = F.up_f32(pos);
    }
    // ---- blocks
    e->bw.resize(e->blocks.size());
    for (size_t i = 0; i < e->blocks.size(); ++i) {
        const BlockSpec& bs = e->blocks[i];
        const std::string b = t + "blocks." + std::to_string(i) + ".";
        BlockW& w = e->bw[i];
        w.n1 = F.ln(b + "norm1", bs.din);
        F.q8 = e->weight_format == SABER_WEIGHTS_FP8_E4M3 && bs.dout >= 4 * e->embed_dim;      // stages 2 and 3: 94 % of the encoder's weights
        // MXFP8: the GEMMs that run on the fp8 MFMA (stages 2 and 3: qkv of the blocks that keep their width, both MLP layers of all)
        const bool mxs = e->weight_format == SABER_WEIGHTS_MXFP8 && bs.dout >= 4 * e->embed_dim;
        F.mx8 = mxs && bs.din == bs.dout;
        w.qkv = F.lin(b + "attn.qkv", 3 * bs.dout, bs.din);
        F.mx8 = false;
        w.proj = F.lin(b + "attn.proj", bs.dout, bs.dout);
        w.n2 = F.ln(b + "norm2", bs.dout);
        F.mx8 = mxs;
        w.fc1 = F.lin(b + "mlp.layers.0", 4 * bs.dout, bs.dout);
        w.fc2 = F.lin(b + "mlp.layers.1", bs.dout, 4 * bs.dout);
        F.q8 = false; F.mx8 = false;                    // (the stage-transition shortcut projection stays bf16)
        if (bs.din != bs.dout) w.sc = F.lin(b + "proj", bs.dout, bs.din);
        if (F.status != SABER_OK) return F.status;
        // qkv / fc1 run on the persistent 256x256 kernel: the same K-step-packed copy makes each of its W pieces one contiguous KB
        if (!e->padded) {
            for (LinW* l : {&w.qkv, &w.fc1}) {
                bf16_t* d = nullptr;
                TRY(eng_alloc(e, &d, gemm_rowln_packed_elems(l->out, l->in)));
                if (const char* m = launch_pack_w_kstep(l->w, l->ldw, l->out, l->in, d, nullptr)) return eng_fail(e, SABER_ERR_INVALID, m);
                l->wpk = d;
            }
        }
        // residual widths the row-owner GEMM + LayerNorm kernel covers: a K-step-packed copy of the two weights it replaces
        if (!e->padded && (bs.dout == 144 || bs.dout == 288 || bs.dout == 576)) {
            for (LinW* l : {&w.proj, &w.fc2}) {
                bf16_t* d = nullptr;
                TRY(eng_alloc(e, &d, gemm_rowln_packed_elems(l->out, l->in)));
                if (const char* m = launch_pack_w_kstep(l->w, l->ldw, l->out, l->in, d, nullptr)) return eng_fail(e, SABER_ERR_INVALID, m);
                l->wpk = d;
            }
        }
    }
    // ---- neck (+ conv_s0 / conv_s1 composed with their lateral convs; no_mem_embed folded into the 64^2 bias)
    {
        const std::string nk = "image_encoder.neck.convs.";
        const int dims[4] = {C0 * 8, C0 * 4, C0 * 2, C0};
        const HostTensor *nw[4], *nb[4];
        for (int i = 0; i < 4; ++i) {
            nw[i] = F.get(nk + std::to_string(i) + ".conv.weight", {256, dims[i], 1, 1});
            nb[i] = F.get(nk + std::to_string(i) + ".conv.bias", {256});
        }
        const HostTensor* nm = F.get("no_mem_embed", {1, 1, 256});
        const HostTensor* s0w = F.get("sam_mask_decoder.conv_s0.weight", {32, 256, 1, 1});
        const HostTensor* s0b = F.get("sam_mask_decoder.conv_s0.bias", {32});
        const HostTensor* s1w = F.get("sam_mask_decoder.conv_s1.weight", {64, 256, 1, 1});
        const HostTensor* s1b = F.get("sam_mask_decoder.conv_s1.bias", {64});
        if (F.status != SABER_OK) return F.status;
        F.up_lin(&e->neck3, nw[0]->data, 256, dims[0]); e->neck3.b = F.up_f32(nb[0]->data);
        std::vector<float> b2(nb[1]->data);
        for (int c = 0; c < 256; ++c) b2[c] += nm->data[c];
        F.up_lin(&e->neck2, nw[1]->data, 256, dims[1]); e->neck2.b = F.up_f32(b2);
        auto compose = [&](const HostTensor* sw, const HostTensor* sbias, int so, const HostTensor* lw, const HostTensor* lb, int li, LinW* out) {
            std::vector<float> w((size_t)so * li), b(so);
            for (int o = 0; o < so; ++o) {
                double bb = sbias->data[o];
                for (int m = 0; m < 256; ++m) bb += (double)sw->data[(size_t)o * 256 + m] * lb->data[m];
                b[o] = (float)bb;
                for (int k = 0; k < li; ++k) {
                    double a = 0.0;
                    for (int m = 0; m < 256; ++m) a += (double)sw->data[(size_t)o * 256 + m] * lw->data[(size_t)m * li + k];
                    w[(size_t)o * li + k] = (float)a;
                }
            }
            F.up_lin(out, w, so, li); out->b = F.up_f32(b);
        };
        compose(s1w, s1b, 64, nw[2], nb[2], dims[2], &e->s1);
        compose(s0w, s0b, 32, nw[3], nb[3], dims[3], &e->s0);
    }
    // ---- prompt encoder
    {
        const std::string p = "sam_prompt_encoder.";
        const HostTensor* g = F.get(p + "pe_layer.positional_encoding_gaussian_matrix", {2, 128});
        const HostTensor* nap = F.get(p + "not_a_point_embed.weight", {1, 256});
        const HostTensor* nme = F.get(p + "no_mask_embed.weight", {1, 256});
        std::vector<float> pemb;
        for (int k = 0; k < 4; ++k) {
            const HostTensor* pk = F.get(p + "point_embeddings." + std::to_string(k) + ".weight", {1, 256});
            if (pk) pemb.insert(pemb.end(), pk->data.begin(), pk->data.end());
        }
        const std::string d = "sam_mask_decoder.";
        const HostTensor* obj = F.get(d + "obj_score_token.weight", {1, 256});
        const HostTensor* iou = F.get(d + "iou_token.weight", {1, 256});
        const HostTensor* mt = F.get(d + "mask_tokens.weight", {4, 256});
        if (F.status != SABER_OK) return F.status;
        std::vector<float> ot;
        ot.insert(ot.end(), obj->data.begin(), obj->data.end());
        ot.insert(ot.end(), iou->data.begin(), iou->data.end());
        ot.insert(ot.end(), mt->data.begin(), mt->data.end());
        e->pw.gauss = F.up_f32(g->data);
        e->pw.point_embed = F.up_f32(pemb);
        e->pw.not_a_point = F.up_f32(nap->data);
        e->pw.out_tokens = F.up_f32(ot);
        e->no_mask_embed = F.up_f32(nme->data);
        // dense PE on the 64x64 grid in engine order
        std::vector<float> dpe((size_t)4096 * 256);
        for (int ty = 0; ty < 64; ++ty)
            for (int tx = 0; tx < 64; ++tx) {
                const size_t row = (size_t)perm_index(ty, tx, 2);
                const float x = 2.0f * ((tx + 0.5f) / 64.0f) - 1.0f, y = 2.0f * ((ty + 0.5f) / 64.0f) - 1.0f;
                for (int f = 0; f < 128; ++f) {
                    const float a = 6.283185307179586f * (x * g->data[f] + y * g->data[128 + f]);
                    dpe[row * 256 + f] = sinf(a);
                    dpe[row * 256 + 128 + f] = cosf(a);
                }
            }
        e->dense_pe = F.up_f32(dpe);
        e->dense_pe_bf = F.up_bf16(dpe);
        const std::string m = p + "mask_downscaling.";
        const HostTensor *w1 = F.get(m + "0.weight", {4, 1, 2, 2}), *b1 = F.get(m + "0.bias", {4});
        const HostTensor *g1 = F.get(m + "1.weight", {4}), *be1 = F.get(m + "1.bias", {4});
        const HostTensor *w2 = F.get(m + "3.weight", {16, 4, 2, 2}), *b2 = F.get(m + "3.bias", {16});
        const HostTensor *g2 = F.get(m + "4.weight", {16}), *be2 = F.get(m + "4.bias", {16});
        const HostTensor *w3 = F.get(m + "6.weight", {256, 16, 1, 1}), *b3 = F.get(m + "6.bias", {256});
        if (F.status != SABER_OK) return F.status;
        e->mw.w1 = F.up_f32(w1->data); e->mw.b1 = F.up_f32(b1->data); e->mw.g1 = F.up_f32(g1->data); e->mw.be1 = F.up_f32(be1->data);
        e->mw.w2 = F.up_f32(w2->data); e->mw.b2 = F.up_f32(b2->data); e->mw.g2 = F.up_f32(g2->data); e->mw.be2 = F.up_f32(be2->data);
        e->mw.w3 = F.up_f32(w3->data); e->mw.b3 = F.up_f32(b3->data);
    }
    // ---- mask decoder
    {
        const std::string d = "sam_mask_decoder.";
        // dec_tokens_kernel streams its weights through registers, one 16 x 32 fragment per wave-load: in the K-step-packed copy a fragment is one
        // contiguous KB (eight full lines) instead of sixteen half lines of sixteen rows (round 5: segments 154 / 130 -> 129 / 102 us for the MLP alone)
        auto pack_for_tokens = [&](LinW* lw, int rows) -> int {
            bf16_t* dpk = nullptr;
            TRY(eng_alloc(e, &dpk, gemm_rowln_packed_elems(rows, lw->in)));
            if (const char* m = launch_pack_w_kstep(lw->w, lw->ldw, rows, lw->in, dpk, nullptr)) return eng_fail(e, SABER_ERR_INVALID, m);
            lw->wpk = dpk; lw->wpk_n = rows;
            return SABER_OK;
        };
        for (int l = 0; l < 2; ++l) {
            const std::string L = d + "transformer.layers." + std::to_string(l) + ".";
            DecLayerW& w = e->dl[l];
            w.self_attn = F.attn(L + "self_attn", 256);
            w.n1 = F.ln(L + "norm1", 256);
            w.t2i = F.attn(L + "cross_attn_token_to_image", 128, "k");
            w.n2 = F.ln(L + "norm2", 256);
            w.mlp1 = F.lin(L + "mlp.layers.0", 2048, 256);
            w.mlp2 = F.lin(L + "mlp.layers.1", 256, 2048);
            if (F.status != SABER_OK) return F.status;
            // dec_tokens_kernel streams these two through registers, one 16 x 32 fragment per wave-load: in the K-step-packed copy a fragment is
            // one contiguous KB (eight full lines) instead of sixteen half lines of sixteen rows
            for (LinW* lw : {&w.mlp1, &w.mlp2}) TRY(pack_for_tokens(lw, lw->out));
            w.n3 = F.ln(L + "norm3", 256);
            w.n4 = F.ln(L + "norm4", 256);
            w.i2t = F.attn(L + "cross_attn_image_to_token", 128, "q");
            if (F.status != SABER_OK) return F.status;
            for (LinW* lw : {&w.self_attn.q, &w.self_attn.k, &w.self_attn.v, &w.self_attn.o, &w.t2i.q, &w.t2i.o, &w.i2t.k, &w.i2t.v}) TRY(pack_for_tokens(lw, lw->out));
        }
        e->final_attn = F.attn(d + "transformer.final_attn_token_to_image", 128, "k");
        if (F.status != SABER_OK) return F.status;
        for (LinW* lw : {&e->final_attn.q, &e->final_attn.o}) TRY(pack_for_tokens(lw, lw->out));
        e->final_ln = F.ln(d + "transformer.norm_final_attn", 256);
        // ConvTranspose2d(k2,s2) as a GEMM: N index = (ky*2+kx)*Cout + co
        auto convT = [&](const std::string& prefix, int cin, int cout, LinW* out) {
            const HostTensor* w = F.get(prefix + ".weight", {cin, cout, 2, 2});
            const HostTensor* b = F.get(prefix + ".bias", {cout});
            if (!w || !b) return;
            std::vector<float> wt((size_t)4 * cout * cin), bt((size_t)4 * cout);
            for (int pos = 0; pos < 4; ++pos)
                for (int co = 0; co < cout; ++co) {
                    bt[(size_t)pos * cout + co] = b->data[co];
                    for (int ci = 0; ci < cin; ++ci)
                        wt[((size_t)pos * cout + co) * cin + ci] = w->data[(((size_t)ci * cout + co) * 2 + (pos >> 1)) * 2 + (pos & 1)];
                }
            F.up_lin(out, wt, 4 * cout, cin); out->b = F.up_f32(bt);
        };
        convT(d + "output_upscaling.0", 256, 64, &e->dc1);
        e->up_ln = F.ln(d + "output_upscaling.1", 64);
        convT(d + "output_upscaling.3", 64, 32, &e->dc2);
        {   // dc2 weight with the contraction index permuted to the k-slot order phase A leaves in registers:
            // slot 8g+j of k-step ks <- channel 32ks + 4g + j (j<4) | 32ks + 16 + 4g + (j-4) (j>=4)
            const HostTensor* w = F.get(d + "output_upscaling.3.weight", {64, 32, 2, 2});
            if (w) {
                std::vector<float> wp((size_t)128 * 64);
                for (int pos = 0; pos < 4; ++pos)
                    for (int co = 0; co < 32; ++co)
                        for (int ks = 0; ks < 2; ++ks)
                            for (int g = 0; g < 4; ++g)
                                for (int jj = 0; jj < 8; ++jj) {
                                    const int ci = jj < 4 ? 32 * ks + 4 * g + jj : 32 * ks + 16 + 4 * g + (jj - 4);
                                    wp[((size_t)pos * 32 + co) * 64 + ks * 32 + g * 8 + jj] = w->data[(((size_t)ci * 32 + co) * 2 + (pos >> 1)) * 2 + (pos & 1)];
                                }
                e->dc2p = F.up_bf16(wp);
            }
        }
        const int hdims[3][2] = {{256, 256}, {256, 256}, {32, 256}};
        for (int l = 0; l < 3; ++l) {
            std::vector<float> w, b;
            for (int k = 0; k < 4; ++k) {
                const std::string pfx = d + "output_hypernetworks_mlps." + std::to_string(k) + ".layers." + std::to_string(l);
                const HostTensor* wk = F.get(pfx + ".weight", {hdims[l][0], hdims[l][1]});
                const HostTensor* bk = F.get(pfx + ".bias", {hdims[l][0]});
                if (!wk || !bk) return F.status;
                w.insert(w.end(), wk->data.begin(), wk->data.end());
                b.insert(b.end(), bk->data.begin(), bk->data.end());
            }
            F.up_lin(&e->hyper[l], w, 4 * hdims[l][0], hdims[l][1]); e->hyper[l].out = hdims[l][0]; e->hyper[l].b = F.up_f32(b);
        }
        const int iou_out[3] = {256, 256, 4}, obj_out[3] = {256, 256, 1};
        for (int l = 0; l < 3; ++l) {
            e->iou_head[l] = F.lin(d + "iou_prediction_head.layers." + std::to_string(l), iou_out[l], 256);
            e->obj_head[l] = F.lin(d + "pred_obj_score_head.layers." + std::to_string(l), obj_out[l], 256);
        }
        if (F.status != SABER_OK) return F.status;
        for (int l = 0; l < 3; ++l) {
            TRY(pack_for_tokens(&e->hyper[l], 4 * e->hyper[l].out));          // the four hypernetwork MLPs stacked along the rows
            TRY(pack_for_tokens(&e->iou_head[l], e->iou_head[l].out));
            TRY(pack_for_tokens(&e->obj_head[l], e->obj_head[l].out));
        }
    }
    if (F.status != SABER_OK) return F.status;

    // ---- token-layout tables of the padded-window trunks (engine.h)
    if (e->padded) {
        const int R2 = e->tok_rows[2], R3 = e->tok_rows[3];
        std::vector<int> pack(R2), unpack(4096, -1);
        std::vector<uint8_t> v2(((size_t)R2 + 127) / 128 * 128, 0), v3(R3, 0);
        for (int r = 0; r < R2; ++r) {
            const int w = r / 196, i = r % 196, g = i >> 2, j = i & 3;
            const int y = (w / 5) * 14 + (g / 7) * 2 + (j >> 1), x = (w % 5) * 14 + (g % 7) * 2 + (j & 1);
            pack[r] = (y < 64 && x < 64) ? perm_index(y, x, 2) : -1;
            if (pack[r] >= 0) { v2[r] = 1; unpack[pack[r]] = r; }
        }
        for (int r = 0; r < R3; ++r) {
            const int w = r / 49, g = r % 49;
            v3[r] = ((w / 5) * 7 + g / 7 < 32 && (w % 5) * 7 + g % 7 < 32) ? 1 : 0;
        }
        for (int i = 0; i < 4096; ++i) if (unpack[i] < 0) return eng_fail(e, SABER_ERR_INVALID, "internal: token layout table is not a bijection");
        auto up_bytes = [&](const void* src, size_t bytes, const void** dst) -> int {
            void* d = nullptr;
            TRY(eng_alloc_bytes(e, &d, bytes));
            ENG_HIP(e, hipMemcpy(d, src, bytes, hipMemcpyHostToDevice));
            *dst = d;
            return SABER_OK;
        };
        const void* d = nullptr;
        TRY(up_bytes(pack.data(), pack.size() * 4, &d)); e->pack_idx = (const int*)d;
        TRY(up_bytes(unpack.data(), unpack.size() * 4, &d)); e->unpack_idx = (const int*)d;
        TRY(up_bytes(v2.data(), v2.size(), &d)); e->kmask2 = (const uint8_t*)d; e->valid[2] = e->kmask2;
        TRY(up_bytes(v3.data(), v3.size(), &d)); e->valid[3] = (const uint8_t*)d;
    }

    // ---- workspaces
    const size_t B = e->max_images, P = e->max_prompts;
    const size_t C0s = (size_t)C0;
    TRY(eng_alloc(e, &e->pix, B * 3 * 1024 * 1024));
    TRY(eng_alloc(e, &e->xa, B * 65536 * C0s));
    TRY(eng_alloc(e, &e->xb, B * 65536 * C0s));
    TRY(eng_alloc(e, &e->xn, B * 65536 * C0s));
    TRY(eng_alloc(e, &e->qkv, B * 65536 * 6 * C0s));
    TRY(eng_alloc(e, &e->att, B * 65536 * C0s));
    TRY(eng_alloc(e, &e->hid, B * 65536 * 4 * C0s));
    if (e->weight_format == SABER_WEIGHTS_MXFP8) {
        if (e->padded) return eng_fail(e, SABER_ERR_INVALID, "the MXFP8 weight format is built for the unpadded trunk layout (Hiera-L, BASELINE configs[4])");
        // scale panels of the MX activations (K-step-major, [K / 128][mx_rows][4]); the e4m3 bytes themselves reuse xn / hid
        e->mx_rows = (int64_t)((B * 4096 + 767) / 768 * 768);      // whole tiles of either MX kernel (256 / 192 rows)
        TRY(eng_alloc(e, &e->xn8_s, (size_t)(8 * C0s / 128 + 1) * e->mx_rows * 4));
        TRY(eng_alloc(e, &e->hid8_s, (size_t)(32 * C0s / 128 + 1) * e->mx_rows * 4));
    }
    for (int s = 0; s < 4; ++s) TRY(eng_alloc(e, &e->sb[s], B * e->tok_rows[s] * (C0s << s)));
    TRY(eng_alloc(e, &e->lat3, B * e->tok_rows[3] * 256));
    TRY(eng_alloc(e, &e->crops_dev, B * 4));
    TRY(eng_alloc(e, &e->emb, B * 4096 * 256));
    TRY(eng_alloc(e, &e->fs1, B * 16384 * 64));
    TRY(eng_alloc(e, &e->fs0, B * 65536 * 32));
    TRY(eng_alloc(e, &e->src0_bf, B * 4096 * 256));
    TRY(eng_alloc(e, &e->embb, B * 4096 * 256));
    e->slot_valid.assign(B, 0);
    e->slot_shared_valid.assign(B, 0);
    e->slot_embb_valid.assign(B, 0);

    TRY(eng_alloc(e, &e->live, P));
    TRY(eng_alloc(e, &e->prune_counters, 4));                   // [0] pruned, [1] seen, [2..3] = the four 32-bit sentinel counters (engine.h: nonfinite)
    ENG_HIP(e, hipMemset(e->prune_counters, 0, 32));
    e->nonfinite = reinterpret_cast<unsigned int*>(e->prune_counters + 2);
    TRY(eng_alloc(e, &e->tok_pe, P * 8 * 256));
    TRY(eng_alloc(e, &e->queries, P * 8 * 256));
    TRY(eng_alloc(e, &e->tq, P * 8 * 256));
    TRY(eng_alloc(e, &e->tk, P * 8 * 256));
    TRY(eng_alloc(e, &e->tv, P * 8 * 256));
    TRY(eng_alloc(e, &e->t_bf0, P * 8 * 256));
    TRY(eng_alloc(e, &e->t_bf1, P * 8 * 256));
    TRY(eng_alloc(e, &e->t_att, P * 8 * 256));
    TRY(eng_alloc(e, &e->t_hid, P * 8 * 2048));
    TRY(eng_alloc(e, &e->keys_bf, P * 4096 * 256));
    TRY(eng_alloc(e, &e->h2_bf, P * 4096 * 16));
    TRY(eng_alloc(e, &e->fold_q, P * 64 * 256));
    TRY(eng_alloc(e, &e->fold_k, P * 64 * 256));
    TRY(eng_alloc(e, &e->fold_v, P * 64 * 256));
    TRY(eng_alloc(e, &e->fold_cb, P * 64));
    TRY(eng_alloc(e, &e->t2i_part, P * 8 * 64 * 256));
    TRY(eng_alloc(e, &e->t2i_ml, P * 8 * 64 * 2));
    TRY(eng_alloc(e, &e->masks4, P * 4 * 65536));
    TRY(eng_alloc(e, &e->hyper_out, P * 128));
    TRY(eng_alloc(e, &e->iou4, P * 4));
    TRY(eng_alloc(e, &e->head_tmp, P * 4));
    TRY(eng_alloc(e, &e->head_bf0, 4 * P * 256));
    TRY(eng_alloc(e, &e->head_bf1, 4 * P * 256));
    TRY(eng_alloc(e, &e->counts_ws, 2 * P));
    TRY(eng_alloc(e, &e->dec_out_masks, P * 3 * 65536));
    TRY(eng_alloc(e, &e->dec_out_iou, P * 3));
    if (!e->prep_minmax) TRY(eng_alloc(e, &e->prep_minmax, 4));
    // dense positional encoding projected by the image-side weight of every cross attention (model constants for dec_t2i / dec_i2t)
    {
        auto project = [&](AttnW& a, const LinW& w) -> int {
            bf16_t* d = nullptr;
            TRY(eng_alloc(e, &d, (size_t)4096 * 128));
            GemmParams g;
            g.A = e->dense_pe_bf; g.lda = 256; g.W = w.w; g.ldw = w.ldw; g.w_kpad = 1; g.M = 4096; g.N = 128; g.K = 256; g.Cb = d; g.ldcb = 128;
            if (const char* m = launch_gemm(g, nullptr)) return eng_fail(e, SABER_ERR_INVALID, m);
            a.pe_proj = d;
            return SABER_OK;
        };
        for (int l = 0; l < 2; ++l) {
            TRY(project(e->dl[l].t2i, e->dl[l].t2i.k));
            TRY(project(e->dl[l].i2t, e->dl[l].i2t.q));
        }
        TRY(project(e->final_attn, e->final_attn.k));
    }
    ENG_HIP(e, hipDeviceSynchronize());
    e->host_w.clear();
    e->finalized = true;
    return SABER_OK;
}

// ------------------------------------------------------------------------------------------------ profiling
void prof_begin(saber_engine* e, int cls, double flops, double bytes, hipStream_t s) {
    if (!e->prof_on) return;
    while (e->ev_pool.size() < e->ev_used + 2) { hipEvent_t ev; if (hipEventCreate(&ev) != hipSuccess) { e->prof_on = false; return; } e->ev_pool.push_back(ev); }
    ProfRec r; r.cls = cls; r.flops = flops; r.bytes = bytes; r.a = e->ev_pool[e->ev_used++]; r.b = e->ev_pool[e->ev_used++];
    (void)hipEventRecord(r.a, s);
    e->prof.push_back(r);
}
void prof_end(saber_engine* e, hipStream_t s) {
    if (!e->prof_on || e->prof.empty()) return;
    (void)hipEventRecord(e->prof.back().b, s);
}
extern "C" int saber_profile_begin(saber_engine* e) {
    if (!e) return SABER_ERR_INVALID;
    e->prof.clear(); e->ev_used = 0; e->prof_on = true;
    return SABER_OK;
}
extern "C" int saber_profile_end(saber_engine* e, saber_profile_class* out, int n_classes) {
    if (!e || !out || n_classes < PC_N) return e ? eng_fail(e, SABER_ERR_INVALID, "profile_end: need room for all classes") : SABER_ERR_INVALID;
    ENG_DEVICE(e);
    ENG_HIP(e, hipDeviceSynchronize());
    e->prof_on = false;
    for (int i = 0; i < n_classes; ++i) { out[i].launches = 0; out[i].ms = 0.0; out[i].flops = 0.0; out[i].bytes = 0.0; }
    for (const ProfRec& r : e->prof) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, r.a, r.b) != hipSuccess) continue;
        out[r.cls].launches += 1; out[r.cls].ms += ms; out[r.cls].flops += r.flops; out[r.cls].bytes += r.bytes;
    }
    e->prof.clear(); e->ev_used = 0;
    return SABER_OK;
}

// ------------------------------------------------------------------------------------------------ helpers
static GemmParams mk_gemm(const bf16_t* A, int64_t lda, int M, const LinW& w) {
    GemmParams p;
    p.A = A; p.lda = lda; p.W = w.w; p.ldw = w.ldw; p.w_kpad = 1; p.bias = w.b; p.M = M; p.N = w.out; p.K = w.in; p.Wpk = w.wpk;
    return p;
}
// algorithmic HBM bytes of one GEMM launch: every operand once (A, W, bias-free), every output once, the residual read once
static double gemm_bytes(const GemmParams& g, bool rowln = false) {
    const double M = g.M, N = g.N, K = g.K, Mo = g.pool4 ? M / 4 : M, b = g.batch > 0 ? g.batch : 1;
    double by = M * K * 2 + N * K * 2;
    if (g.Cf) by += Mo * N * 4;
    if (g.Cb) by += Mo * N * 2;
    if (g.res) by += (g.res_shift || g.res_mod ? 0.0 : Mo * N * 4);
    if (rowln) by += Mo * N * 2;
    return by * b;
}
static const char* ln_run(const float* x, const LnW& w, float eps, int rows, int C, float* out_f, bf16_t* out_bf, int act, hipStream_t s,
                          bf16_t* out_bf_add = nullptr, const float* addvec = nullptr, int add_mod = 0, const uint8_t* row_valid = nullptr,
                          int valid_mod = 0) {
    LayerNormParams p;
    p.row_valid = row_valid; p.valid_mod = valid_mod;
    p.x = x; p.ldx = C; p.gamma = w.g; p.beta = w.b; p.eps = eps; p.out_f = out_f; p.out_bf = out_bf; p.out_bf_add = out_bf_add;
    p.ldo = C; p.addvec = addvec; p.add_mod = add_mod; p.rows = rows; p.C = C; p.act = act;
    return launch_layernorm(p, s);
}

// ------------------------------------------------------------------------------------------------ K0
static int prepare_common(saber_engine* e, const void* img_dev, int dtype, int H, int W, int channels, float* out_dev, hipStream_t s) {
    // K0 needs the device binding and scratch only: it also runs on a handle without weights (saber_amd.utils.preprocessing.prepare)
    if (!img_dev || !out_dev || H <= 0 || W <= 0) return eng_fail(e, SABER_ERR_INVALID, "prepare: bad argument");
    if (!e->prep_minmax) TRY(eng_alloc(e, &e->prep_minmax, 4));
    const size_t need = (size_t)4 * H * W * channels;
    if (e->prep_ws_elems < need) {
        TRY(eng_regrow(e, &e->prep_ws, need, s));
        e->prep_ws_elems = need;
    }
    if (channels == 3) {
        if (dtype != SABER_F32) return eng_fail(e, SABER_ERR_INVALID, "prepare: (H,W,3) input must be float32");
        ENG_KP(e, PC_IMAGE, 0.0, 0.0, launch_prepare_rgb_f32((const float*)img_dev, H, W, out_dev, e->prep_ws, e->prep_minmax, s));
    } else if (dtype == SABER_U16) ENG_KP(e, PC_IMAGE, 0.0, 0.0, launch_prepare_u16((const uint16_t*)img_dev, H, W, out_dev, e->prep_ws, e->prep_minmax, s));
    else if (dtype == SABER_F32) ENG_KP(e, PC_IMAGE, 0.0, 0.0, launch_prepare_f32((const float*)img_dev, H, W, out_dev, e->prep_ws, e->prep_minmax, s));
    else return eng_fail(e, SABER_ERR_INVALID, "prepare: dtype must be SABER_U16 or SABER_F32");
    ENG_HIP(e, hipGetLastError());
    return SABER_OK;
}
extern "C" int saber_prepare(saber_engine* e, const void* img_dev, int dtype, int H, int W, float* out_dev, void* stream) {
    if (!e) return SABER_ERR_INVALID;
    ENG_DEVICE(e);
    return prepare_common(e, img_dev, dtype, H, W, 1, out_dev, (hipStream_t)stream);
}
extern "C" int saber_prepare_rgb(saber_engine* e, const float* img_dev, int H, int W, float* out_dev, void* stream) {
    if (!e) return SABER_ERR_INVALID;
    ENG_DEVICE(e);
    return prepare_common(e, img_dev, SABER_F32, H, W, 3, out_dev, (hipStream_t)stream);
}

// ------------------------------------------------------------------------------------------------ encoder
static inline int mx_kp(int k) { return (k + 127) / 128 * 128; }
static GemmMxParams mk_gemm_mx(saber_engine* e, const uint8_t* A, const uint8_t* SA, int64_t M, const LinW& l) {
    GemmMxParams g;
    g.A = A; g.lda = l.kp8; g.SA = SA; g.sa_rows = e->mx_rows; g.W = l.w8; g.ldw = l.kp8; g.SW = l.sw8; g.sw_rows = l.sw_rows; g.bias = l.b;
    g.M = M; g.N = l.out; g.Kp = l.kp8;
    return g;
}

int eng_encode(saber_engine* e, const float* img_dev, int H, int W, int channels, const int* crops_host, int n, int slot0, hipStream_t s) {
    if (!e->finalized) return eng_fail(e, SABER_ERR_STATE, "engine not finalized");
    if (!img_dev || !crops_host || n < 1 || n > e->max_images || slot0 < 0 || slot0 + n > e->max_images)
        return eng_fail(e, SABER_ERR_INVALID, "encode: bad argument (n / slot range exceeds max_images?)");
    for (int i = 0; i < n; ++i) {
        const int* c = crops_host + 4 * i;
        if (c[0] < 0 || c[1] < 0 || c[2] > W || c[3] > H || c[2] <= c[0] || c[3] <= c[1]) return eng_fail(e, SABER_ERR_INVALID, "encode: crop box outside the image");
    }
    // (through a pinned engine-owned buffer: when this pass is captured into a hipGraph the copy node reads its source at replay time)
    if (!e->crops_pin) ENG_HIP(e, hipHostMalloc(reinterpret_cast<void**>(&e->crops_pin), sizeof(int) * 256 * 8));
    const int* src = crops_host;
    int slot = 0;
    if (crops_host != e->crops_pin) {      // not the AMG driver's own slot: take the next ring slot once its previous copy has run
        slot = e->crops_next;
        e->crops_next = slot == 7 ? 1 : slot + 1;
        if (!e->crops_ev[slot]) ENG_HIP(e, hipEventCreateWithFlags(&e->crops_ev[slot], hipEventDisableTiming));
        else ENG_HIP(e, hipEventSynchronize(e->crops_ev[slot]));
        memcpy(e->crops_pin + 256 * slot, crops_host, sizeof(int) * 4 * n);
        src = e->crops_pin + 256 * slot;
    }
    ENG_HIP(e, hipMemcpyAsync(e->crops_dev, src, sizeof(int) * 4 * n, hipMemcpyHostToDevice, s));
    if (slot) ENG_HIP(e, hipEventRecord(e->crops_ev[slot], s));
    ENG_KP(e, PC_IMAGE, 0.0, 0.0, launch_resize_normalize(img_dev, H, W, channels, e->crops_dev, n, e->pix, 1024, s));
    ENG_KP(e, PC_IMAGE, 0.0, 0.0, launch_patch_embed(e->pix, e->pe_wt, e->pe_bias, e->pos_table, e->xa, n, e->embed_dim, 1024, s));
    if (e->precision == SABER_PRECISION_EXACT) {
        TRY(exact_encode_blocks(e, n, slot0, s));
        for (int i = 0; i < n; ++i) { e->slot_valid[slot0 + i] = 1; e->slot_shared_valid[slot0 + i] = 0; e->slot_embb_valid[slot0 + i] = 0; }
        return SABER_OK;
    }
    float* x = e->xa;
    float* xalt = e->xb;
    int tokens = 65536;  // per image, current stage
    int stage = 0;
    // The LayerNorm that follows each residual step (norm2 after attn.proj, norm1 of the NEXT block after mlp.layers.1) is computed in
    // the epilogue of the GEMM that produces the row (gemm_rowln.hip) wherever the residual width allows (144 / 288 / 576: stages 0-2 of
    // Hiera-L); the padded-window trunks keep the separate pass (their norm1 also zeroes the window-padding rows).
    static const bool no_rowln = getenv("SABER_AMD_NO_ROWLN") != nullptr;      // development A/B switch
    const bool fuse = !e->padded && !no_rowln;
    const size_t nblocks = e->blocks.size();
    // window-padding rows (padded layout): the reference pads the normalised tokens with zeros before qkv
    ENG_KP(e, PC_LAYERNORM, 0.0, 0.0, ln_run(x, e->bw[0].n1, 1e-6f, n * tokens, e->blocks[0].din, nullptr, e->xn, ACT_NONE, s, nullptr, nullptr, 0, e->valid[0], tokens));
    // Consumers walk the M tiles AGAINST their producers: a kernel that starts on the rows its predecessor wrote last finds them in the
    // 256-MB Infinity Cache (walking the same way, every row has been evicted by the time it is read: LRU streaming).  Measured where
    // the tensor in between is larger than the cache: qkv (297 MB in stage 3) -> window attention 10.73 -> 10.23 ms per slice; neutral
    // for the GEMM consumers.  Development flag 512 restores the forward walks.
    const int snake = (g_saber_debug_flags & 512) ? 0 : 1;
    uint8_t* const xn8 = reinterpret_cast<uint8_t*>(e->xn);     // MX activations (weight format MXFP8) live in the bf16 buffers they replace
    uint8_t* const hid8 = reinterpret_cast<uint8_t*>(e->hid);
    for (size_t i = 0; i < nblocks; ++i) {
        const BlockSpec& bs = e->blocks[i];
        const BlockW& w = e->bw[i];
        const int N = n * tokens;
        // e->xn holds norm1(x) of this block
        float* xres = x;
        int Nq = N;
        if (bs.din != bs.dout) {
            GemmParams g = mk_gemm(e->xn, bs.din, N, w.sc);
            g.Cf = xalt; g.ldcf = bs.dout; g.pool4 = 1;
            ENG_KP(e, PC_GEMM, 2.0 * g.M * (double)g.N * g.K * g.batch, gemm_bytes(g), launch_gemm(g, s));
            xres = xalt;
            Nq = N / 4;
        }
        if (w.qkv.w8) {      // MXFP8: xn arrived as an MX operand (ln_mx below / at the end of the previous block)
            GemmMxParams g = mk_gemm_mx(e, xn8, e->xn8_s, N, w.qkv);
            g.Cb = e->qkv; g.ldcb = 3 * bs.dout;
            ENG_KP(e, PC_GEMM_MX, 2.0 * g.M * (double)g.N * w.qkv.in, (double)g.M * (g.Kp + 2.0 * g.N), launch_gemm_mx(g, s));
        } else {
            GemmParams g = mk_gemm(e->xn, bs.din, N, w.qkv);
            g.Cb = e->qkv; g.ldcb = 3 * bs.dout;
            g.rev = snake;               // attention walks forward: it starts on the rows qkv wrote last
            ENG_KP(e, PC_GEMM, 2.0 * g.M * (double)g.N * g.K * g.batch, gemm_bytes(g), launch_gemm(g, s));
        }
        const int nk = bs.window > 0 ? bs.window * bs.window : tokens;
        // global blocks of the padded layout: one "window" per image whose padding rows are masked out as keys
        const uint8_t* kmask = (bs.window == 0 && e->valid[stage]) ? e->kmask2 : nullptr;
        ENG_KP(e, PC_HIERA_ATTN, 4.0 * e->head_dim * (double)N * (bs.q_stride > 1 ? nk / 4 : nk) * bs.heads, 0.0,
               launch_hiera_attention(e->qkv, e->att, N / nk, nk, bs.heads, e->head_dim, bs.q_stride > 1, kmask, s));
        {
            GemmParams g = mk_gemm(e->att, bs.dout, Nq, w.proj);
            g.Cf = xres; g.ldcf = bs.dout; g.res = xres; g.ldres = bs.dout;
            g.ln_gamma = w.n2.g; g.ln_beta = w.n2.b; g.ln_eps = 1e-6f; g.ln_out = e->xn; g.ldln = bs.dout;
            g.rev = snake;               // against the attention kernel that wrote its operand
            if (w.fc1.w8) {  // MXFP8 MLP: norm2 is written straight as the MX operand of mlp.layers.0
                g.ln_gamma = nullptr; g.ln_beta = nullptr; g.ln_out = nullptr;
                ENG_KP(e, PC_GEMM, 2.0 * g.M * (double)g.N * g.K * g.batch, gemm_bytes(g), launch_gemm(g, s));
                ENG_KP(e, PC_LAYERNORM, 0.0, 0.0, launch_ln_mx(xres, bs.dout, w.n2.g, w.n2.b, 1e-6f, bs.dout, xn8, mx_kp(bs.dout), mx_kp(bs.dout), e->xn8_s, e->mx_rows, Nq, s));
            } else if (fuse && gemm_rowln_supported(g)) {
                ENG_KP(e, PC_GEMM, 2.0 * g.M * (double)g.N * g.K * g.batch, gemm_bytes(g, true), launch_gemm_rowln(g, s));
            } else {
                ENG_KP(e, PC_GEMM, 2.0 * g.M * (double)g.N * g.K * g.batch, gemm_bytes(g), launch_gemm(g, s));
                ENG_KP(e, PC_LAYERNORM, 0.0, 0.0, ln_run(xres, w.n2, 1e-6f, Nq, bs.dout, nullptr, e->xn, ACT_NONE, s));
            }
        }
        if (bs.din != bs.dout) { std::swap(x, xalt); tokens /= 4; ++stage; }
        if (w.fc1.w8) {      // hidden activations leave the epilogue as the MX operand of mlp.layers.1
            GemmMxParams g = mk_gemm_mx(e, xn8, e->xn8_s, Nq, w.fc1);
            g.C8 = hid8; g.ldc8 = 4 * bs.dout; g.SC = e->hid8_s; g.sc_rows = e->mx_rows; g.act = ACT_GELU;
            ENG_KP(e, PC_GEMM_MX, 2.0 * g.M * (double)g.N * w.fc1.in, (double)g.M * (g.Kp + 1.0 * g.N), launch_gemm_mx(g, s));
        } else {
            GemmParams g = mk_gemm(e->xn, bs.dout, Nq, w.fc1);
            g.Cb = e->hid; g.ldcb = 4 * bs.dout; g.act = ACT_GELU;
            ENG_KP(e, PC_GEMM, 2.0 * g.M * (double)g.N * g.K * g.batch, gemm_bytes(g), launch_gemm(g, s));
        }
        const bool to_padded = e->padded && bs.din != bs.dout && stage == 2;
        const bool has_next = i + 1 < nblocks;
        if (w.fc2.w8) {
            GemmMxParams g = mk_gemm_mx(e, hid8, e->hid8_s, Nq, w.fc2);
            g.Cf = x; g.ldcf = bs.dout; g.res = x; g.ldres = bs.dout;
            if ((int)i == e->stage_ends[stage]) { g.Cb = e->sb[stage]; g.ldcb = bs.dout; }
            ENG_KP(e, PC_GEMM_MX, 2.0 * g.M * (double)g.N * w.fc2.in, (double)g.M * (g.Kp + 8.0 * g.N), launch_gemm_mx(g, s));
            if (has_next) {   // norm1 of the next block: as an MX operand when its qkv runs on the fp8 MFMA, bf16 for the stage-transition block
                if (e->bw[i + 1].qkv.w8)
                    ENG_KP(e, PC_LAYERNORM, 0.0, 0.0, launch_ln_mx(x, bs.dout, e->bw[i + 1].n1.g, e->bw[i + 1].n1.b, 1e-6f, bs.dout, xn8, mx_kp(bs.dout), mx_kp(bs.dout), e->xn8_s, e->mx_rows, Nq, s));
                else
                    ENG_KP(e, PC_LAYERNORM, 0.0, 0.0, ln_run(x, e->bw[i + 1].n1, 1e-6f, n * tokens, bs.dout, nullptr, e->xn, ACT_NONE, s));
            }
        } else {
            GemmParams g = mk_gemm(e->hid, 4 * bs.dout, Nq, w.fc2);
            g.Cf = x; g.ldcf = bs.dout; g.res = x; g.ldres = bs.dout;
            g.rev = snake;               // fc1 walked forward: the last 256 MB of the hidden tensor are still in the Infinity Cache
            if ((int)i == e->stage_ends[stage]) { g.Cb = e->sb[stage]; g.ldcb = bs.dout; }
            bool fused = false;
            if (fuse && has_next) {
                g.ln_gamma = e->bw[i + 1].n1.g; g.ln_beta = e->bw[i + 1].n1.b; g.ln_eps = 1e-6f; g.ln_out = e->xn; g.ldln = bs.dout;
                fused = gemm_rowln_supported(g);
            }
            if (fused) ENG_KP(e, PC_GEMM, 2.0 * g.M * (double)g.N * g.K * g.batch, gemm_bytes(g, true), launch_gemm_rowln(g, s));
            else ENG_KP(e, PC_GEMM, 2.0 * g.M * (double)g.N * g.K * g.batch, gemm_bytes(g), launch_gemm(g, s));
            if (to_padded) {
                // the 64^2 grid leaves the stage-transition block in the bit-interleaved order; the 14x14 windows of the blocks that
                // follow need the window-major padded layout (padding rows start as zeros and are re-zeroed by every norm1)
                ENG_KP(e, PC_ELEMENTWISE, 0.0, 0.0, launch_gather_rows(x, 4096, xalt, e->tok_rows[2], e->pack_idx, bs.dout, n, s));
                std::swap(x, xalt);
                tokens = e->tok_rows[2];
            }
            if (has_next && !fused)   // norm1 of the next block (its width is this block's dout)
                ENG_KP(e, PC_LAYERNORM, 0.0, 0.0, ln_run(x, e->bw[i + 1].n1, 1e-6f, n * tokens, bs.dout, nullptr, e->xn, ACT_NONE, s, nullptr, nullptr, 0, e->valid[stage], tokens));
        }
    }
    // neck
    {
        GemmParams g = mk_gemm(e->sb[3], e->stage_dims[3], n * e->tok_rows[3], e->neck3);
        g.Cf = e->lat3; g.ldcf = 256;
        ENG_KP(e, PC_GEMM, 2.0 * g.M * (double)g.N * g.K * g.batch, gemm_bytes(g), launch_gemm(g, s));
    }
    {
        GemmParams g = mk_gemm(e->sb[2], e->stage_dims[2], n * e->tok_rows[2], e->neck2);
        float* emb_slot = e->emb + (size_t)slot0 * 4096 * 256;
        g.Cf = e->padded ? e->xa : emb_slot; g.ldcf = 256; g.res = e->lat3; g.ldres = 256; g.res_shift = 2;
        ENG_KP(e, PC_GEMM, 2.0 * g.M * (double)g.N * g.K * g.batch, gemm_bytes(g), launch_gemm(g, s));
        // padded layout -> the decoder's bit-interleaved order of the 64^2 grid (the residual stream buffers are free by now)
        if (e->padded) ENG_KP(e, PC_ELEMENTWISE, 0.0, 0.0, launch_gather_rows(e->xa, e->tok_rows[2], emb_slot, 4096, e->unpack_idx, 256, n, s));
    }
    {
        GemmParams g = mk_gemm(e->sb[1], e->stage_dims[1], n * 16384, e->s1);
        g.Cf = e->fs1 + (size_t)slot0 * 16384 * 64; g.ldcf = 64;
        ENG_KP(e, PC_GEMM, 2.0 * g.M * (double)g.N * g.K * g.batch, gemm_bytes(g), launch_gemm(g, s));
    }
    {
        GemmParams g = mk_gemm(e->sb[0], e->stage_dims[0], n * 65536, e->s0);
        g.Cf = e->fs0 + (size_t)slot0 * 65536 * 32; g.ldcf = 32;
        ENG_KP(e, PC_GEMM, 2.0 * g.M * (double)g.N * g.K * g.batch, gemm_bytes(g), launch_gemm(g, s));
    }
    for (int i = 0; i < n; ++i) { e->slot_valid[slot0 + i] = 1; e->slot_shared_valid[slot0 + i] = 0; e->slot_embb_valid[slot0 + i] = 0; }
    // sentinel: a 16-bit activation that left its type's range (fp16: 65 504) is an inf from there on - through the fp32 accumulators, the
    // fp32 residual stream and every LayerNorm / softmax statistic - so it ends in the pass's three feature maps: one HBM-bound scan (16 MB per crop)
    ENG_KP(e, PC_ELEMENTWISE, 0.0, (double)n * 4096 * 256 * 4, launch_nonfinite_scan(e->emb + (size_t)slot0 * 4096 * 256, (int64_t)n * 4096 * 256, e->nonfinite + 0, s));
    ENG_KP(e, PC_ELEMENTWISE, 0.0, (double)n * 16384 * 64 * 4, launch_nonfinite_scan(e->fs1 + (size_t)slot0 * 16384 * 64, (int64_t)n * 16384 * 64, e->nonfinite + 0, s));
    ENG_KP(e, PC_ELEMENTWISE, 0.0, (double)n * 65536 * 32 * 4, launch_nonfinite_scan(e->fs0 + (size_t)slot0 * 65536 * 32, (int64_t)n * 65536 * 32, e->nonfinite + 0, s));
    ENG_HIP(e, hipGetLastError());
    return SABER_OK;
}

// Overflow sentinel of the 16-bit arithmetic modes (include/saber_amd.h: saber_engine_check_finite).  Three device counters, written by
// launches on the caller's stream: [0] non-finite values in the feature maps of an encoder pass, [1] in a decoder batch's predicted IoUs /
// hypernetwork outputs (every token-side quantity feeds them), [2] low-res logits stored by dec_upscale.  Sticky until read.
int eng_check_finite_counts(saber_engine* e, const unsigned int* h) {
    if (!(h[0] | h[1] | h[2])) return SABER_OK;
    return eng_fail(e, SABER_ERR_RANGE, std::string("non-finite values in the ") + (e->op_f16 ? "fp16" : "bf16") + " arithmetic: " + std::to_string(h[0]) +
                    " in encoder features, " + std::to_string(h[1]) + " in decoder IoU / hypernetwork heads, " + std::to_string(h[2]) + " in low-res mask logits" +
                    (e->op_f16 ? " - an activation left the range of IEEE half (65 504); results of this call are invalid: use precision bf16 (or exact) for this model / input"
                               : " - the input image or the weights hold NaN / inf; results of this call are invalid"));
}
extern "C" int saber_engine_check_finite(saber_engine* e, void* stream) {
    if (!e) return SABER_ERR_INVALID;
    if (!e->finalized || !e->nonfinite) return SABER_OK;
    ENG_DEVICE(e);
    unsigned int h[4] = {0, 0, 0, 0};
    ENG_HIP(e, hipMemcpyAsync(h, e->nonfinite, 16, hipMemcpyDeviceToHost, (hipStream_t)stream));
    ENG_HIP(e, hipStreamSynchronize((hipStream_t)stream));
    if (h[0] | h[1] | h[2]) ENG_HIP(e, hipMemsetAsync(e->nonfinite, 0, 16, (hipStream_t)stream));
    return eng_check_finite_counts(e, h);
}

extern "C" int saber_encode(saber_engine* e, const float* img_dev, int H, int W, int channels, const int* crop_boxes_host, int n,
                            int slot0, void* stream) {
    if (!e) return SABER_ERR_INVALID;
    ENG_DEVICE(e);
    return eng_encode(e, img_dev, H, W, channels, crop_boxes_host, n, slot0, (hipStream_t)stream);
}

extern "C" int saber_get_features(saber_engine* e, int slot, float* image_embed, float* feat_s0, float* feat_s1, void* stream) {
    if (!e) return SABER_ERR_INVALID;
    if (!e->finalized) return eng_fail(e, SABER_ERR_STATE, "engine not finalized");
    if (slot < 0 || slot >= e->max_images || !e->slot_valid[slot]) return eng_fail(e, SABER_ERR_STATE, "get_features: slot holds no encoded image; call saber_encode first");
    ENG_DEVICE(e);
    hipStream_t s = (hipStream_t)stream;
    if (image_embed) ENG_KP(e, PC_ELEMENTWISE, 0.0, 0.0, launch_unpermute_nchw(e->emb + (size_t)slot * 4096 * 256, 256, 2, image_embed, s));
    if (feat_s1) ENG_KP(e, PC_ELEMENTWISE, 0.0, 0.0, launch_unpermute_nchw(e->fs1 + (size_t)slot * 16384 * 64, 64, 1, feat_s1, s));
    if (feat_s0) ENG_KP(e, PC_ELEMENTWISE, 0.0, 0.0, launch_unpermute_nchw(e->fs0 + (size_t)slot * 65536 * 32, 32, 0, feat_s0, s));
    ENG_HIP(e, hipGetLastError());
    return SABER_OK;
}

// ------------------------------------------------------------------------------------------------ video path access to a slot
// The memory path replaces a frame's image embedding by its memory-conditioned version before the mask decoder runs
// (upstream SAM2Base.track_step: pix_feat_with_mem -> _forward_sam_heads).  Tokens cross this boundary in row-major (y, x) order.
static int ensure_rm_tables(saber_engine* e) {
    if (e->rm_to_eng) return SABER_OK;
    std::vector<int> a(4096), b(4096);
    for (int y = 0; y < 64; ++y)
        for (int x = 0; x < 64; ++x) { const int r = y * 64 + x, g = perm_index(y, x, 2); a[r] = g; b[g] = r; }
    TRY(eng_alloc(e, &e->rm_to_eng, 4096));
    TRY(eng_alloc(e, &e->eng_to_rm, 4096));
    ENG_HIP(e, hipMemcpy(e->rm_to_eng, a.data(), 4096 * sizeof(int), hipMemcpyHostToDevice));
    ENG_HIP(e, hipMemcpy(e->eng_to_rm, b.data(), 4096 * sizeof(int), hipMemcpyHostToDevice));
    return SABER_OK;
}
int eng_rm_tables(saber_engine* e) { return ensure_rm_tables(e); }
extern "C" int saber_get_embed_tokens(saber_engine* e, int slot, float* out_tokens_dev, void* stream) {
    if (!e) return SABER_ERR_INVALID;
    if (!e->finalized) return eng_fail(e, SABER_ERR_STATE, "engine not finalized");
    if (slot < 0 || slot >= e->max_images || !e->slot_valid[slot] || !out_tokens_dev) return eng_fail(e, SABER_ERR_STATE, "get_embed_tokens: slot holds no encoded image; call saber_encode first");
    ENG_DEVICE(e);
    TRY(ensure_rm_tables(e));
    hipStream_t s = (hipStream_t)stream;
    ENG_KP(e, PC_ELEMENTWISE, 0.0, 0.0, launch_gather_rows(e->emb + (size_t)slot * 4096 * 256, 4096, out_tokens_dev, 4096, e->rm_to_eng, 256, 1, s));
    return SABER_OK;
}
extern "C" int saber_set_embed_tokens(saber_engine* e, int slot, const float* tokens_dev, void* stream) {
    if (!e) return SABER_ERR_INVALID;
    if (!e->finalized) return eng_fail(e, SABER_ERR_STATE, "engine not finalized");
    if (slot < 0 || slot >= e->max_images || !e->slot_valid[slot] || !tokens_dev) return eng_fail(e, SABER_ERR_STATE, "set_embed_tokens: slot holds no encoded image; call saber_encode first");
    ENG_DEVICE(e);
    TRY(ensure_rm_tables(e));
    hipStream_t s = (hipStream_t)stream;
    ENG_KP(e, PC_ELEMENTWISE, 0.0, 0.0, launch_gather_rows(tokens_dev, 4096, e->emb + (size_t)slot * 4096 * 256, 4096, e->eng_to_rm, 256, 1, s));
    e->slot_shared_valid[slot] = 0;        // image_embed + no_mask_embed of this slot must be rebuilt
    e->slot_embb_valid[slot] = 0;
    return SABER_OK;
}
// Slot state across engine handles / ranks (propagation path: the per-frame encodes of a tomogram shard over ranks and are
// all-gathered).  A slot is three fp32 arrays in the engine's own token order: image_embed 4096 x 256, feat_s1 16384 x 64, feat_s0
// 65536 x 32 (16 MiB); export / import are plain device copies, so a slot imported from another handle of the same model decodes
// bit-identically to one encoded here.
extern "C" int saber_export_slots(saber_engine* e, int slot0, int n, float* emb_dev, float* fs1_dev, float* fs0_dev, void* stream) {
    if (!e) return SABER_ERR_INVALID;
    if (!e->finalized) return eng_fail(e, SABER_ERR_STATE, "engine not finalized");
    if (n < 0 || slot0 < 0 || slot0 + n > e->max_images || (n > 0 && (!emb_dev || !fs1_dev || !fs0_dev))) return eng_fail(e, SABER_ERR_INVALID, "export_slots: bad argument");
    for (int i = 0; i < n; ++i)
        if (!e->slot_valid[slot0 + i]) return eng_fail(e, SABER_ERR_STATE, "export_slots: slot holds no encoded image; call saber_encode first");
    ENG_DEVICE(e);
    hipStream_t s = (hipStream_t)stream;
    if (n == 0) return SABER_OK;
    ENG_HIP(e, hipMemcpyAsync(emb_dev, e->emb + (size_t)slot0 * 4096 * 256, (size_t)n * 4096 * 256 * sizeof(float), hipMemcpyDeviceToDevice, s));
    ENG_HIP(e, hipMemcpyAsync(fs1_dev, e->fs1 + (size_t)slot0 * 16384 * 64, (size_t)n * 16384 * 64 * sizeof(float), hipMemcpyDeviceToDevice, s));
    ENG_HIP(e, hipMemcpyAsync(fs0_dev, e->fs0 + (size_t)slot0 * 65536 * 32, (size_t)n * 65536 * 32 * sizeof(float), hipMemcpyDeviceToDevice, s));
    return SABER_OK;
}
extern "C" int saber_import_slots(saber_engine* e, int slot0, int n, const float* emb_dev, const float* fs1_dev, const float* fs0_dev, void* stream) {
    if (!e) return SABER_ERR_INVALID;
    if (!e->finalized) return eng_fail(e, SABER_ERR_STATE, "engine not finalized");
    if (n < 0 || slot0 < 0 || slot0 + n > e->max_images || (n > 0 && (!emb_dev || !fs1_dev || !fs0_dev))) return eng_fail(e, SABER_ERR_INVALID, "import_slots: bad argument");
    ENG_DEVICE(e);
    hipStream_t s = (hipStream_t)stream;
    if (n == 0) return SABER_OK;
    ENG_HIP(e, hipMemcpyAsync(e->emb + (size_t)slot0 * 4096 * 256, emb_dev, (size_t)n * 4096 * 256 * sizeof(float), hipMemcpyDeviceToDevice, s));
    ENG_HIP(e, hipMemcpyAsync(e->fs1 + (size_t)slot0 * 16384 * 64, fs1_dev, (size_t)n * 16384 * 64 * sizeof(float), hipMemcpyDeviceToDevice, s));
    ENG_HIP(e, hipMemcpyAsync(e->fs0 + (size_t)slot0 * 65536 * 32, fs0_dev, (size_t)n * 65536 * 32 * sizeof(float), hipMemcpyDeviceToDevice, s));
    for (int i = 0; i < n; ++i) { e->slot_valid[slot0 + i] = 1; e->slot_shared_valid[slot0 + i] = 0; e->slot_embb_valid[slot0 + i] = 0; }
    return SABER_OK;
}
// the 8 tokens of the first n prompts of the LAST decode call after the two-way transformer ([obj, iou, mask0..3, point, pad] x 256 fp32):
// the video path projects one mask token to the object pointer (upstream obj_ptr_proj(sam_output_token))
extern "C" int saber_get_decoder_tokens(saber_engine* e, int n, float* out_dev, void* stream) {
    if (!e) return SABER_ERR_INVALID;
    if (!e->finalized) return eng_fail(e, SABER_ERR_STATE, "engine not finalized");
    if (n < 1 || n > e->max_prompts || !out_dev) return eng_fail(e, SABER_ERR_INVALID, "get_decoder_tokens: n must be 1..max_prompts");
    ENG_DEVICE(e);
    ENG_HIP(e, hipMemcpyAsync(out_dev, e->queries, sizeof(float) * (size_t)n * 8 * 256, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return SABER_OK;
}

// ------------------------------------------------------------------------------------------------ decoder
static int ensure_shared(saber_engine* e, int slot, hipStream_t s) {
    if (e->slot_shared_valid[slot]) return SABER_OK;
    const size_t o256 = (size_t)slot * 4096 * 256;
    // src0 = image_embed + no_mask_embed (identical for every first-pass prompt of this crop)
    ENG_KP(e, PC_ELEMENTWISE, 0.0, 0.0, launch_add_to_bf16(e->emb + o256, e->no_mask_embed, 1, e->src0_bf + o256, nullptr, 4096, 256, s));
    e->slot_shared_valid[slot] = 1;
    return SABER_OK;
}

static int ensure_embb(saber_engine* e, int slot, hipStream_t s) {
    if (e->slot_embb_valid[slot]) return SABER_OK;
    const size_t o256 = (size_t)slot * 4096 * 256;
    // image_embed + b3 (fp32): what mask_embed_src_kernel adds before its MFMA; the layer-0 kernels load it as their tiles' C operand
    ENG_KP(e, PC_ELEMENTWISE, 0.0, 0.0, launch_embb_tiles(e->emb + o256, e->mw.b3, e->embb + o256, s));
    e->slot_embb_valid[slot] = 1;
    return SABER_OK;
}

// One chunk of P prompts.  The prompts may span several consecutive slots (crops of one AMG layer batched together): prompt p of
// the chunk reads the features of slot slot0 + (p_base + p) / per_slot.
// raw4_out: the chunk's 4 low-res planes per prompt are written there and stay (no selection copy: out_iou / out_sel say which to read);
// mask_in_q0 >= 0: mask_in is the raw 4-plane output of a multimask decode, this chunk's first prompt refines global candidate mask_in_q0
static int decode_chunk(saber_engine* e, int slot0, int per_slot, int p_base, const float* pts, const int* labels, int P, int multimask,
                        const float* mask_in, float mask_clamp, float* out_lowres, float* out_iou, float* out_obj, hipStream_t s,
                        float* raw4_out = nullptr, int* out_sel = nullptr, int mask_in_q0 = -1, float prune_iou_thr = 0.f) {
    if (e->precision == SABER_PRECISION_EXACT) {
        float* m4 = raw4_out ? raw4_out : e->masks4;
        TRY(exact_decode_core(e, slot0, per_slot, p_base, pts, labels, P, mask_in, mask_clamp, mask_in_q0, out_obj, m4, s, e->decode_n_pts));
        float* oi = out_iou ? out_iou : e->dec_out_iou;
        if (raw4_out) { ENG_KP(e, PC_ELEMENTWISE, 0.0, 0.0, launch_mask_pick(raw4_out, e->iou4, P, multimask, oi, out_sel, s)); return SABER_OK; }
        float* om = out_lowres ? out_lowres : e->dec_out_masks;
        ENG_KP(e, PC_ELEMENTWISE, 0.0, 0.0, launch_mask_select(e->masks4, e->iou4, P, multimask, om, oi, e->counts_ws, s));
        return SABER_OK;
    }
    if (e->decode_n_pts != 1) return eng_fail(e, SABER_ERR_STATE, "prompts of several points (clicks, boxes) are decoded in the exact precision mode only: the bf16 kernels are built for 8 decoder tokens per prompt");
    const int T = 8;
    const int PT = P * T;
    const size_t o256 = (size_t)slot0 * 4096 * 256;
    const bool shared = (mask_in == nullptr);
    const int slot_last = slot0 + (p_base + P - 1) / per_slot;
    const XMap slots{(int64_t)4096 * 256, per_slot, p_base};          // per-slot tensors of 4096 x 256 elements
    const XMap per_prompt{(int64_t)4096 * 256, 1, 0};
    const float kScale = 0.25f * 1.4426950408889634f;  // head_dim 16 ^ -0.5 * log2(e): scores live in the exp2 domain
    int split = 1;
    while (split < 8 && P * split < 512) split *= 2;
    ENG_KP(e, PC_ELEMENTWISE, 0.0, 0.0, launch_prompt_tokens(pts, labels, P, e->pw, e->tok_pe, s));
    ENG_HIP(e, hipMemcpyAsync(e->queries, e->tok_pe, sizeof(float) * PT * 256, hipMemcpyDeviceToDevice, s));
    const bf16_t* X;      // image tokens of each prompt, bf16 [4096][256], engine order
    XMap xm;
    XBuild xb;
    const XBuild* build = nullptr;          // layer 0 of a mask-prompted decode: X0 tiles assembled in the kernels
    // Opt-in experiment (round 3, SABER_AMD_XBUILD=1, read per call): assemble the m2m prompts' src inside layer 0's dec_t2i / dec_i2t instead
    // of materialising it.  Bit-identical and 55 GB per slice less HBM traffic, but SLOWER on this memory system: the fp32 image_embed
    // tiles (4 MB per prompt and pass, from the Infinity Cache) cost more than the 2 MB of bf16 src from HBM they replace, and with two
    // slices in flight they evict the encoder's working set from that cache (same-box A/B: 156.7 vs 144.3 ms per slice).  DESIGN.md section 4.
    const bool no_build = getenv("SABER_AMD_XBUILD") == nullptr;
    if (shared) {
        for (int sl = slot0 + p_base / per_slot; sl <= slot_last; ++sl) TRY(ensure_shared(e, sl, s));
        X = e->src0_bf + o256; xm = slots;
    } else if (no_build) {
        ENG_KP(e, PC_ELEMENTWISE, 0.0, (double)P * (65536.0 * 4 + 4096.0 * 256 * 2), launch_mask_embed_src(mask_in, P, e->emb + o256, slots, e->dense_pe, e->mw, nullptr, e->keys_bf, nullptr, mask_clamp, s, mask_in_q0));
        X = e->keys_bf; xm = per_prompt;
    } else {
        // src = image_embed + mask-prompt embedding is never materialised: only the 16-channel hidden vectors go to HBM (32 B per token
        // against 512 B), the layer-0 kernels assemble the tiles (XBuild)
        for (int sl = slot0 + p_base / per_slot; sl <= slot_last; ++sl) TRY(ensure_embb(e, sl, s));
        ENG_KP(e, PC_ELEMENTWISE, 0.0, (double)P * (65536.0 * 4 + 4096.0 * 16 * 2), launch_mask_hidden(mask_in, P, e->mw, e->h2_bf, mask_clamp, s, mask_in_q0));
        xb.embb = e->embb + o256; xb.map = slots; xb.h2 = e->h2_bf; xb.w3 = e->mw.w3;
        build = &xb;
        X = nullptr; xm = per_prompt;
    }

    static const bool no_tokfuse = getenv("SABER_AMD_NO_TOKFUSE") != nullptr;      // development A/B switch: the ~58 separate token-side launches
    if (!no_tokfuse) {
        // token side as four fused segments (decoder_tokens.hip) around the five image-side kernels
        auto lin = [](const LinW& l) { TokLin t; t.w = l.w; t.b = l.b; t.ldw = l.ldw; t.n = l.out; t.wpk = l.wpk; t.npk = l.wpk_n; return t; };
        auto lnw = [](const LnW& l) { TokLn t; t.g = l.g; t.b = l.b; return t; };
        auto base = [&]() { TokSeg g; g.P = P; g.queries = e->queries; g.tok_pe = e->tok_pe; g.kscale = kScale; return g; };
        auto with_t2i = [&](TokSeg& g, const AttnW& a) { g.do_t2i = 1; g.t2i_q = lin(a.q); g.t2i_kT = a.img_wT; g.tq_out = e->tq; g.fold_q = e->fold_q; };
        auto with_self = [&](TokSeg& g, const DecLayerW& w, int first) {
            g.do_self = 1; g.self_first = first; g.sa_q = lin(w.self_attn.q); g.sa_k = lin(w.self_attn.k); g.sa_v = lin(w.self_attn.v); g.sa_o = lin(w.self_attn.o); g.ln1 = lnw(w.n1);
        };
        auto with_att_out = [&](TokSeg& g, const AttnW& a, const LnW& ln) { g.t_att = e->t_att; g.att_o = lin(a.o); g.att_ln = lnw(ln); g.att_eps = 1e-5f; };
        auto with_mlp_i2t = [&](TokSeg& g, const DecLayerW& w) {
            g.do_mlp = 1; g.mlp1 = lin(w.mlp1); g.mlp2 = lin(w.mlp2); g.mlp1_pk = w.mlp1.wpk; g.mlp2_pk = w.mlp2.wpk; g.ln3 = lnw(w.n3);
            g.i2t_k = lin(w.i2t.k); g.i2t_v = lin(w.i2t.v); g.i2t_qT = w.i2t.img_wT; g.i2t_qb = w.i2t.q.b; g.i2t_o = w.i2t.o.w;
            g.tk_out = e->tk; g.fold_k = e->fold_k; g.fold_cb = e->fold_cb; g.fold_v = e->fold_v;
        };
        const double tflops = 2.0 * PT * 256.0;     // per 256 x 1 column of weights
        auto run_t2i = [&](const AttnW& a) -> int {
            ENG_KP(e, PC_DEC_T2I, 4.0 * 64 * 4096.0 * 256 * P, (double)P * 4096 * 256 * 2, launch_dec_t2i(X, xm, a.pe_proj, e->fold_q, e->tq, kScale, e->t2i_part, e->t2i_ml, P, split, a.v.w, a.v.b, e->t_att, s, build));
            return SABER_OK;
        };
        auto run_i2t = [&](const DecLayerW& w) -> int {
            ENG_KP(e, PC_DEC_I2T, 4.0 * 64 * 4096.0 * 256 * P, (double)P * 4096 * 256 * 4, launch_dec_i2t(X, xm, w.i2t.pe_proj, e->fold_k, e->tk, kScale, e->fold_cb, e->fold_v, w.i2t.o.b, w.n4.g, w.n4.b, 1e-5f, e->keys_bf, P, s, build));
            X = e->keys_bf; xm = per_prompt; build = nullptr;
            return SABER_OK;
        };
        // Opt-in (round 5, SABER_AMD_FUSE_I2T_T2I=1, read per call): image -> tokens of a layer and the tokens -> image attention that follows
        // it in ONE kernel (dec_i2t_t2i_kernel: X' is the next attention's key / value block while it is still in LDS).  Same results as the
        // two launches up to the order of the online softmax; measured slower (one wave per SIMD): DESIGN.md section 8.2.
        const bool fuse = getenv("SABER_AMD_FUSE_I2T_T2I") != nullptr && P >= 128;
        auto run_i2t_t2i = [&](const DecLayerW& w, const AttnW& a) -> int {
            if (!fuse || build != nullptr) { TRY(run_i2t(w)); return run_t2i(a); }
            ENG_KP(e, PC_DEC_I2T, 8.0 * 64 * 4096.0 * 256 * P, (double)P * 4096 * 256 * 4,
                   launch_dec_i2t_t2i(X, xm, w.i2t.pe_proj, e->fold_k, e->tk, kScale, e->fold_cb, e->fold_v, w.i2t.o.b, w.n4.g, w.n4.b, 1e-5f, e->keys_bf, P,
                                      a.pe_proj, e->fold_q, e->tq, kScale, a.v.w, a.v.b, e->t_att, s));
            X = e->keys_bf; xm = per_prompt; build = nullptr;
            return SABER_OK;
        };
        {   // S0: self attention of layer 0, operands of its tokens -> image attention
            TokSeg g = base(); with_self(g, e->dl[0], 1); with_t2i(g, e->dl[0].t2i);
            ENG_KP(e, PC_DEC_ATTN, tflops * (4 * 256 + 128 + 128), 0.0, launch_dec_tokens(g, s));
        }
        TRY(run_t2i(e->dl[0].t2i));
        {   // S1: rest of layer 0 on the token side, self attention of layer 1, operands of its tokens -> image attention
            TokSeg g = base(); with_att_out(g, e->dl[0].t2i, e->dl[0].n2); with_mlp_i2t(g, e->dl[0]); with_self(g, e->dl[1], 0); with_t2i(g, e->dl[1].t2i);
            ENG_KP(e, PC_DEC_ATTN, tflops * (128 + 4096 + 4 * 128 + 4 * 256 + 256), 0.0, launch_dec_tokens(g, s));
        }
        TRY(run_i2t_t2i(e->dl[0], e->dl[1].t2i));
        {   // S2: rest of layer 1, operands of the final tokens -> image attention
            TokSeg g = base(); with_att_out(g, e->dl[1].t2i, e->dl[1].n2); with_mlp_i2t(g, e->dl[1]); with_t2i(g, e->final_attn);
            ENG_KP(e, PC_DEC_ATTN, tflops * (128 + 4096 + 4 * 128 + 256), 0.0, launch_dec_tokens(g, s));
        }
        TRY(run_i2t_t2i(e->dl[1], e->final_attn));
        {   // S3: final output projection + LayerNorm, IoU / object-score / hypernetwork heads
            TokSeg g = base(); with_att_out(g, e->final_attn, e->final_ln);
            g.do_heads = 1;
            for (int l = 0; l < 3; ++l) { g.iou[l] = lin(e->iou_head[l]); g.obj[l] = lin(e->obj_head[l]); g.hyper[l] = lin(e->hyper[l]); }
            g.iou4 = e->iou4; g.obj_out = out_obj; g.hyper_out = e->hyper_out;
            ENG_KP(e, PC_DEC_ATTN, tflops * (128 + 6 * 512.0 / 8), 0.0, launch_dec_tokens(g, s));
        }
    } else {
    // tokens -> image: fold q into 64 rows of dimension 256, stream X once (dec_t2i), un-fold with v_proj
    auto t2i = [&](const AttnW& a, const LnW& ln) -> int {
        ENG_KP(e, PC_ELEMENTWISE, 0.0, 0.0, launch_add_to_bf16(e->queries, e->tok_pe, PT, e->t_bf0, nullptr, PT, 256, s));
        GemmParams g = mk_gemm(e->t_bf0, 256, PT, a.q);
        g.Cf = e->tq; g.ldcf = 128;
        ENG_KP(e, PC_GEMM, 2.0 * g.M * (double)g.N * g.K, gemm_bytes(g), launch_gemm(g, s));
        ENG_KP(e, PC_ELEMENTWISE, 0.0, 0.0, launch_dec_fold(e->tq, a.k.w, nullptr, 0, kScale, e->fold_q, nullptr, P, s));
        ENG_KP(e, PC_DEC_T2I, 4.0 * 64 * 4096.0 * 256 * P, (double)P * 4096 * 256 * 2, launch_dec_t2i(X, xm, a.pe_proj, e->fold_q, e->tq, kScale, e->t2i_part, e->t2i_ml, P, split, a.v.w, a.v.b, e->t_att, s, build));
        g = mk_gemm(e->t_att, 128, PT, a.o);
        g.Cf = e->queries; g.ldcf = 256; g.res = e->queries; g.ldres = 256;
        ENG_KP(e, PC_GEMM, 2.0 * g.M * (double)g.N * g.K, gemm_bytes(g), launch_gemm(g, s));
        ENG_KP(e, PC_LAYERNORM, 0.0, 0.0, ln_run(e->queries, ln, 1e-5f, PT, 256, e->queries, nullptr, ACT_NONE, s));
        return SABER_OK;
    };

    for (int l = 0; l < 2; ++l) {
        const DecLayerW& w = e->dl[l];
        // (1) self attention of the tokens
        ENG_KP(e, PC_ELEMENTWISE, 0.0, 0.0, launch_add_to_bf16(e->queries, l == 0 ? nullptr : e->tok_pe, PT, e->t_bf0, nullptr, PT, 256, s));
        const bf16_t* vin = e->t_bf0;
        if (l > 0) { ENG_KP(e, PC_ELEMENTWISE, 0.0, 0.0, launch_add_to_bf16(e->queries, nullptr, 1, e->t_bf1, nullptr, PT, 256, s)); vin = e->t_bf1; }
        GemmParams g = mk_gemm(e->t_bf0, 256, PT, w.self_attn.q); g.Cf = e->tq; g.ldcf = 256; ENG_KP(e, PC_GEMM, 2.0 * g.M * (double)g.N * g.K, gemm_bytes(g), launch_gemm(g, s));
        g = mk_gemm(e->t_bf0, 256, PT, w.self_attn.k); g.Cf = e->tk; g.ldcf = 256; ENG_KP(e, PC_GEMM, 2.0 * g.M * (double)g.N * g.K, gemm_bytes(g), launch_gemm(g, s));
        g = mk_gemm(vin, 256, PT, w.self_attn.v); g.Cf = e->tv; g.ldcf = 256; ENG_KP(e, PC_GEMM, 2.0 * g.M * (double)g.N * g.K, gemm_bytes(g), launch_gemm(g, s));
        ENG_KP(e, PC_DEC_ATTN, 0.0, 0.0, launch_dec_attention(e->tq, e->tk, e->tv, e->t_att, P, T, T, 8, 32, T * 256, T * 256, T * 256, T * 256, s));
        g = mk_gemm(e->t_att, 256, PT, w.self_attn.o);
        g.Cf = e->queries; g.ldcf = 256;
        if (l > 0) { g.res = e->queries; g.ldres = 256; }
        ENG_KP(e, PC_GEMM, 2.0 * g.M * (double)g.N * g.K, gemm_bytes(g), launch_gemm(g, s));
        ENG_KP(e, PC_LAYERNORM, 0.0, 0.0, ln_run(e->queries, w.n1, 1e-5f, PT, 256, e->queries, nullptr, ACT_NONE, s));
        // (2) tokens -> image
        TRY(t2i(w.t2i, w.n2));
        // (3) MLP
        ENG_KP(e, PC_ELEMENTWISE, 0.0, 0.0, launch_add_to_bf16(e->queries, nullptr, 1, e->t_bf0, nullptr, PT, 256, s));
        g = mk_gemm(e->t_bf0, 256, PT, w.mlp1); g.Cb = e->t_hid; g.ldcb = 2048; g.act = ACT_RELU; ENG_KP(e, PC_GEMM, 2.0 * g.M * (double)g.N * g.K, gemm_bytes(g), launch_gemm(g, s));
        g = mk_gemm(e->t_hid, 2048, PT, w.mlp2); g.Cf = e->queries; g.ldcf = 256; g.res = e->queries; g.ldres = 256; ENG_KP(e, PC_GEMM, 2.0 * g.M * (double)g.N * g.K, gemm_bytes(g), launch_gemm(g, s));
        ENG_KP(e, PC_LAYERNORM, 0.0, 0.0, ln_run(e->queries, w.n3, 1e-5f, PT, 256, e->queries, nullptr, ACT_NONE, s));
        // (4) image -> tokens, fused with the residual and norm4: X <- LN(X + attn)
        ENG_KP(e, PC_ELEMENTWISE, 0.0, 0.0, launch_add_to_bf16(e->queries, e->tok_pe, PT, e->t_bf0, nullptr, PT, 256, s));
        ENG_KP(e, PC_ELEMENTWISE, 0.0, 0.0, launch_add_to_bf16(e->queries, nullptr, 1, e->t_bf1, nullptr, PT, 256, s));
        g = mk_gemm(e->t_bf0, 256, PT, w.i2t.k); g.Cf = e->tk; g.ldcf = 128; ENG_KP(e, PC_GEMM, 2.0 * g.M * (double)g.N * g.K, gemm_bytes(g), launch_gemm(g, s));
        g = mk_gemm(e->t_bf1, 256, PT, w.i2t.v); g.Cf = e->tv; g.ldcf = 128; ENG_KP(e, PC_GEMM, 2.0 * g.M * (double)g.N * g.K, gemm_bytes(g), launch_gemm(g, s));
        ENG_KP(e, PC_ELEMENTWISE, 0.0, 0.0, launch_dec_fold(e->tk, w.i2t.q.w, w.i2t.q.b, 0, kScale, e->fold_k, e->fold_cb, P, s));
        ENG_KP(e, PC_ELEMENTWISE, 0.0, 0.0, launch_dec_fold(e->tv, w.i2t.o.w, nullptr, 1, 1.0f, e->fold_v, nullptr, P, s));
        ENG_KP(e, PC_DEC_I2T, 4.0 * 64 * 4096.0 * 256 * P, (double)P * 4096 * 256 * 4, launch_dec_i2t(X, xm, w.i2t.pe_proj, e->fold_k, e->tk, kScale, e->fold_cb, e->fold_v, w.i2t.o.b, w.n4.g, w.n4.b, 1e-5f, e->keys_bf, P, s, build));
        X = e->keys_bf; xm = per_prompt; build = nullptr;
    }
    TRY(t2i(e->final_attn, e->final_ln));

    // heads
    ENG_KP(e, PC_ELEMENTWISE, 0.0, 0.0, launch_add_to_bf16(e->queries, nullptr, 1, e->t_bf0, nullptr, PT, 256, s));
    auto mlp3 = [&](const LinW* L, const bf16_t* A, int last_act, float* outf, int ldo) -> int {
        GemmParams g = mk_gemm(A, T * 256, P, L[0]); g.Cb = e->head_bf0; g.ldcb = 256; g.act = ACT_RELU; ENG_KP(e, PC_GEMM, 2.0 * g.M * (double)g.N * g.K, gemm_bytes(g), launch_gemm(g, s));
        g = mk_gemm(e->head_bf0, 256, P, L[1]); g.Cb = e->head_bf1; g.ldcb = 256; g.act = ACT_RELU; ENG_KP(e, PC_GEMM, 2.0 * g.M * (double)g.N * g.K, gemm_bytes(g), launch_gemm(g, s));
        g = mk_gemm(e->head_bf1, 256, P, L[2]); g.Cf = outf; g.ldcf = ldo; g.act = last_act; ENG_KP(e, PC_GEMM, 2.0 * g.M * (double)g.N * g.K, gemm_bytes(g), launch_gemm(g, s));
        return SABER_OK;
    };
    TRY(mlp3(e->iou_head, e->t_bf0 + 1 * 256, ACT_SIGMOID, e->iou4, 4));
    if (out_obj) TRY(mlp3(e->obj_head, e->t_bf0, ACT_NONE, out_obj, 1));
    {   // 4 hypernetwork MLPs as batched GEMMs (batch = mask token)
        GemmParams g = mk_gemm(e->t_bf0 + 2 * 256, T * 256, P, e->hyper[0]);
        g.batch = 4; g.strideA = 256; g.strideW = 256 * 256; g.strideBias = 256;
        g.Cb = e->head_bf0; g.ldcb = 256; g.strideCb = (int64_t)P * 256; g.act = ACT_RELU;
        ENG_KP(e, PC_GEMM, 2.0 * g.M * (double)g.N * g.K * 4, gemm_bytes(g), launch_gemm(g, s));
        g = mk_gemm(e->head_bf0, 256, P, e->hyper[1]);
        g.batch = 4; g.strideA = (int64_t)P * 256; g.strideW = 256 * 256; g.strideBias = 256;
        g.Cb = e->head_bf1; g.ldcb = 256; g.strideCb = (int64_t)P * 256; g.act = ACT_RELU;
        ENG_KP(e, PC_GEMM, 2.0 * g.M * (double)g.N * g.K * 4, gemm_bytes(g), launch_gemm(g, s));
        g = mk_gemm(e->head_bf1, 256, P, e->hyper[2]);
        g.batch = 4; g.strideA = (int64_t)P * 256; g.strideW = 32 * 256; g.strideBias = 32;
        g.Cf = e->hyper_out; g.ldcf = 128; g.strideCf = 32;
        ENG_KP(e, PC_GEMM, 2.0 * g.M * (double)g.N * g.K * 4, gemm_bytes(g), launch_gemm(g, s));
    }
    }   // (separate token-side launches)
    // sentinel: every token-side quantity of the batch ends in the predicted IoUs and the hypernetwork outputs (a NaN / inf row of the image-token
    // state reaches them through the tokens -> image attentions: its score is NaN, so is the softmax sum); the logits are checked where they are stored
    ENG_KP(e, PC_ELEMENTWISE, 0.0, 0.0, launch_nonfinite_scan(e->iou4, (int64_t)P * 4, e->nonfinite + 1, s));
    ENG_KP(e, PC_ELEMENTWISE, 0.0, 0.0, launch_nonfinite_scan(e->hyper_out, (int64_t)P * 128, e->nonfinite + 1, s));
    // IoU pruning (the AMG m2m pass): a single-mask candidate reports iou[0] or max(iou[1..3]) (dynamic multimask selection); when all four are
    // <= the caller's pred_iou_thresh it fails that filter whatever its masks look like, so its 1 MB of planes is neither computed nor read
    const uint8_t* live = nullptr;
    if (prune_iou_thr > 0.f && raw4_out && !multimask && e->iou_prune) {
        ENG_KP(e, PC_ELEMENTWISE, 0.0, 0.0, launch_iou_live_flags(e->iou4, P, prune_iou_thr, e->live, e->prune_counters, s));
        live = e->live;
    }
    // upscaling head fused with the hypernetwork product (dec_upscale_kernel)
    ENG_KP(e, PC_DEC_UPSCALE, (double)P * 2.0 * (4096.0 * 256 * 256 + 16384.0 * 64 * 128 + 65536.0 * 32 * 4), (double)P * (4096.0 * 256 * 2 + 4 * 65536.0 * 4),
           launch_dec_upscale(X, e->dc1.w, e->dc1.b, e->up_ln.g, e->up_ln.b, e->dc2p, e->dc2.b, e->fs1 + (size_t)slot0 * 16384 * 64,
                              e->fs0 + (size_t)slot0 * 65536 * 32, XMap{0, per_slot, p_base}, e->hyper_out, raw4_out ? raw4_out : e->masks4, P, s, live,
                              getenv("SABER_AMD_ALL_PLANES") ? nullptr : e->iou4, multimask, e->nonfinite + 2));      // (development A/B: all four planes)
    float* oi = out_iou ? out_iou : e->dec_out_iou;
    if (raw4_out) {
        ENG_KP(e, PC_ELEMENTWISE, 0.0, 0.0, launch_mask_pick(raw4_out, e->iou4, P, multimask, oi, out_sel, s, live));
        return SABER_OK;
    }
    float* om = out_lowres ? out_lowres : e->dec_out_masks;
    ENG_KP(e, PC_ELEMENTWISE, 0.0, 0.0, launch_mask_select(e->masks4, e->iou4, P, multimask, om, oi, e->counts_ws, s));
    return SABER_OK;
}

int eng_decode(saber_engine* e, int slot, int per_slot, const float* pts_dev, const int* labels_dev, int n, int multimask,
               const float* mask_in_dev, float mask_clamp, float* out_lowres, float* out_iou, float* out_obj, hipStream_t s) {
    return eng_decode_ex(e, slot, per_slot, pts_dev, labels_dev, n, multimask, mask_in_dev, 0, mask_clamp, out_lowres, 0, out_iou, out_obj, nullptr, s);
}
int eng_decode_ex(saber_engine* e, int slot, int per_slot, const float* pts_dev, const int* labels_dev, int n, int multimask,
                  const float* mask_in_dev, int mask_in_raw4, float mask_clamp, float* out_lowres, int out_raw4, float* out_iou, float* out_obj,
                  int* out_sel, hipStream_t s, float prune_iou_thr) {
    if (!e->finalized) return eng_fail(e, SABER_ERR_STATE, "engine not finalized");
    if (!pts_dev || n < 0 || per_slot < 0) return eng_fail(e, SABER_ERR_INVALID, "decode: bad argument");
    if (per_slot == 0 || per_slot > n) per_slot = n > 0 ? n : 1;      // every prompt reads slot `slot`
    const int nslots = n > 0 ? (n + per_slot - 1) / per_slot : 1;
    for (int sl = slot; sl < slot + nslots; ++sl)
        if (sl < 0 || sl >= e->max_images || !e->slot_valid[sl]) return eng_fail(e, SABER_ERR_STATE, "decode: slot holds no encoded image; call saber_encode first");
    const int M = multimask ? 3 : 1;
    const int K = e->decode_n_pts;                  // points per prompt (1 except under saber_decode_prompts)
    const int chunk = e->precision == SABER_PRECISION_EXACT ? std::max(1, exact_chunk_prompts(e) * 8 / (7 + K)) : e->max_prompts;
    for (int p0 = 0; p0 < n; p0 += chunk) {
        const int P = std::min(chunk, n - p0);
        const float* min_ = !mask_in_dev ? nullptr : mask_in_raw4 ? mask_in_dev : mask_in_dev + (size_t)p0 * 65536;
        TRY(decode_chunk(e, slot, per_slot, p0, pts_dev + 2 * (size_t)p0 * K, labels_dev ? labels_dev + (size_t)p0 * K : nullptr, P, multimask,
                         min_, mask_clamp,
                         out_lowres && !out_raw4 ? out_lowres + (size_t)p0 * M * 65536 : nullptr, out_iou ? out_iou + (size_t)p0 * M : nullptr,
                         out_obj ? out_obj + p0 : nullptr, s, out_raw4 ? out_lowres + (size_t)p0 * 4 * 65536 : nullptr,
                         out_sel ? out_sel + p0 : nullptr, mask_in_dev && mask_in_raw4 ? p0 : -1, prune_iou_thr));
    }
    ENG_HIP(e, hipGetLastError());
    return SABER_OK;
}

extern "C" int saber_decode_points(saber_engine* e, int slot, const float* pts_dev, const int* labels_dev, int n, int multimask,
                                   const float* mask_in_dev, float* out_lowres_dev, float* out_iou_dev, float* out_obj_dev, void* stream) {
    if (!e) return SABER_ERR_INVALID;
    ENG_DEVICE(e);
    return eng_decode(e, slot, 0, pts_dev, labels_dev, n, multimask, mask_in_dev, 0.f, out_lowres_dev, out_iou_dev, out_obj_dev, (hipStream_t)stream);
}

extern "C" int saber_decode_prompts(saber_engine* e, int slot, const float* pts_dev, const int* labels_dev, int n, int points_per_prompt, int multimask,
                                    const float* mask_in_dev, float* out_lowres_dev, float* out_iou_dev, float* out_obj_dev, void* stream) {
    if (!e) return SABER_ERR_INVALID;
    if (points_per_prompt < 1 || points_per_prompt > 9) return eng_fail(e, SABER_ERR_INVALID, "decode_prompts: 1..9 points per prompt (16 decoder tokens)");
    if (points_per_prompt > 1 && !labels_dev) return eng_fail(e, SABER_ERR_INVALID, "decode_prompts: labels are required with several points per prompt");
    ENG_DEVICE(e);
    e->decode_n_pts = points_per_prompt;
    const int st = eng_decode(e, slot, 0, pts_dev, labels_dev, n, multimask, mask_in_dev, 0.f, out_lowres_dev, out_iou_dev, out_obj_dev, (hipStream_t)stream);
    e->decode_n_pts = 1;
    return st;
}

// ------------------------------------------------------------------------------------------------ label plane
extern "C" int saber_label_plane(saber_engine* e, const uint32_t* bits_dev, const int* order_host, int n, int H, int W, uint16_t* plane_dev,
                                 void* stream) {
    if (!e) return SABER_ERR_INVALID;
    if (!plane_dev || H <= 0 || W <= 0 || n < 0 || (n > 0 && !bits_dev)) return eng_fail(e, SABER_ERR_INVALID, "label_plane: bad argument");
    if (n > 65535) return eng_fail(e, SABER_ERR_INVALID, "label_plane: more than 65535 masks do not fit a uint16 plane");
    ENG_DEVICE(e);
    hipStream_t s = (hipStream_t)stream;
    int* od = nullptr;
    if (order_host && n > 0) {
        if (e->order_cap < (size_t)n) { TRY(eng_regrow(e, &e->order_dev, (size_t)n, s)); e->order_cap = n; }
        ENG_HIP(e, hipMemcpyAsync(e->order_dev, order_host, sizeof(int) * n, hipMemcpyHostToDevice, s));
        od = e->order_dev;
    }
    ENG_KP(e, PC_ELEMENTWISE, 0.0, 0.0, launch_label_plane(bits_dev, od, n, H, W, plane_dev, s));
    ENG_HIP(e, hipGetLastError());
    return SABER_OK;
}

// ------------------------------------------------------------------------------------------------ duplicate-mask support
extern "C" int saber_mask_pair_intersections(saber_engine* e, const uint32_t* bits_dev, int n, int H, int W, int32_t* out_inter_dev, void* stream) {
    if (!e) return SABER_ERR_INVALID;
    if (n < 0 || H <= 0 || W <= 0 || (n > 0 && (!bits_dev || !out_inter_dev))) return eng_fail(e, SABER_ERR_INVALID, "mask_pair_intersections: bad argument");
    ENG_DEVICE(e);
    hipStream_t s = (hipStream_t)stream;
    ENG_KP(e, PC_MASK_POST, 0.0, (double)n * n * 0.5 * H * ((W + 31) / 32) * 8.0, launch_pair_intersections(bits_dev, n, (int64_t)H * ((W + 31) >> 5), out_inter_dev, s));
    ENG_HIP(e, hipGetLastError());
    return SABER_OK;
}

// ------------------------------------------------------------------------------------------------ work counters
extern "C" double saber_encoder_flops(const saber_engine* e) {
    if (!e) return 0.0;
    // SURVEY.md 8d formula (algorithmic, 2*MAC): per block qkv + shortcut + proj + attention + MLP, plus patch embed, neck, conv_s0/s1
    double fl = 65536.0 * 147 * e->embed_dim * 2;
    int tokens = 65536;
    for (const BlockSpec& b : e->blocks) {
        const double nin = tokens, nq = b.q_stride > 1 ? tokens / 4 : tokens;
        const double nk = b.window > 0 ? (double)b.window * b.window : (double)tokens;
        const double qpw = b.q_stride > 1 ? nk / 4 : nk;
        const double nwin = nin / nk;
        double mac = nin * b.din * 3.0 * b.dout + nq * (double)b.dout * b.dout + nwin * qpw * nk * b.dout * 2.0 + nq * 8.0 * b.dout * b.dout;
        if (b.din != b.dout) mac += nin * b.din * b.dout;
        fl += 2.0 * mac;
        if (b.q_stride > 1) tokens /= 4;
    }
    const double toks[4] = {65536, 16384, 4096, 1024};
    for (int s = 0; s < 4; ++s) fl += 2.0 * toks[s] * e->stage_dims[s] * 256.0;
    fl += 2.0 * (65536.0 * 256 * 32 + 16384.0 * 256 * 64);
    return fl;
}
extern "C" double saber_decoder_flops_per_prompt(void) { return 3.639e9; }
