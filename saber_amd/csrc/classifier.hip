// Domain-expert classifier filter on the engine's image embeddings.
// Reference: saber/classifier/models/predictor.py:117-175 (Predictor.predict: NormalizeIntensity -> one adaptive crop per mask ->
// area filter -> model -> softmax), saber/classifier/datasets/RandMaskCrop.py:44-171 (crop_and_resize_adaptive),
// saber/classifier/models/SAM2.py:118-197 (SAM2 image embedding of every crop, ROI / RONI masking, projection + classifier head).
//
// Device layout: the n crops of a call form one tall (n*320, 320) fp32 image, so a batch of them is ONE eng_encode call whose crop boxes
// are the 320-row bands (the reference re-runs the Hiera encoder per mask crop as well: that is 99.6 % of this path's FLOPs and runs on
// the engine's encoder kernels unchanged).  The head works on row-major (y, x) token rows: ROI/RONI select -> bf16 [4096][512];
// conv1x1 / conv3x3 as GEMMs over im2col rows (BatchNorm folded into the conv weights at finalize, PReLU and the 2x2 max-pools
// applied by the im2col gather of the NEXT layer), global average pool + Linear/LayerNorm/PReLU/Linear/softmax in one small kernel.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "common.h"
#include "engine.h"

#define TRY(x) do { int _r = (x); if (_r != SABER_OK) return _r; } while (0)
#define CROP 320
#define CROP_PIX (CROP * CROP)

struct ClsHostT { std::vector<int64_t> shape; std::vector<float> data; };

struct saber_classifier {
    saber_engine* e = nullptr;
    int num_classes = 0;
    bool finalized = false;
    std::map<std::string, ClsHostT> host;
    // folded weights
    bf16_t *w1 = nullptr, *w2 = nullptr, *w3 = nullptr;          // [256][512], [256][9*256], [128][9*256] (k = (ky*3+kx)*Cin + c)
    float *b1 = nullptr, *b2 = nullptr, *b3 = nullptr;
    float a1 = 0.25f, a2 = 0.25f, a3 = 0.25f, a4 = 0.25f;        // PReLU slopes
    float *fc1w = nullptr, *fc1b = nullptr, *lng = nullptr, *lnb = nullptr, *fc2w = nullptr, *fc2b = nullptr;
    // per-call workspaces (grown on demand)
    int *bbox = nullptr, *boxes = nullptr, *areas = nullptr, *sel = nullptr; size_t cap_n = 0;
    double* sums = nullptr;
    float* crops = nullptr; uint8_t* cmask = nullptr;
    bf16_t *A0 = nullptr, *A1 = nullptr, *A2 = nullptr; float *G1 = nullptr, *G2 = nullptr, *G3 = nullptr, *probs = nullptr; size_t cap_b = 0;
};

int eng_rm_tables(saber_engine* e);    // engine.hip

bf16_t saber_host_f2h(float f);          // engine.hip: fp32 -> IEEE half bits (RNE)
static inline bf16_t cls_f2bf(float f) {
    uint32_t u; memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (bf16_t)((u >> 16) | 0x40);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (bf16_t)(u >> 16);
}

// ------------------------------------------------------------------------------------------------ kernels
__global__ void cls_bbox_init_kernel(int* bbox, int n, int H, int W) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { bbox[4 * i] = H; bbox[4 * i + 1] = -1; bbox[4 * i + 2] = W; bbox[4 * i + 3] = -1; }
}
// one block per (row y, mask b): [y_min, y_max, x_min, x_max] of the non-zero pixels
__global__ __launch_bounds__(256) void cls_bbox_kernel(const uint8_t* __restrict__ masks, int H, int W, int* __restrict__ bbox) {
    const int b = blockIdx.y, y = blockIdx.x;
    const uint8_t* row = masks + ((int64_t)b * H + y) * W;
    int xmin = W, xmax = -1;
    for (int x = threadIdx.x; x < W; x += 256)
        if (row[x]) { xmin = min(xmin, x); xmax = max(xmax, x); }
    __shared__ int sx[2];
    if (threadIdx.x == 0) { sx[0] = W; sx[1] = -1; }
    __syncthreads();
    if (xmax >= 0) { atomicMin(&sx[0], xmin); atomicMax(&sx[1], xmax); }
    __syncthreads();
    if (threadIdx.x == 0 && sx[1] >= 0) {
        atomicMin(&bbox[4 * b + 0], y); atomicMax(&bbox[4 * b + 1], y);
        atomicMin(&bbox[4 * b + 2], sx[0]); atomicMax(&bbox[4 * b + 3], sx[1]);
    }
}
// sums[0] += sum x, sums[1] += sum x^2 (fp64)
__global__ __launch_bounds__(256) void cls_sums_kernel(const float* __restrict__ x, int64_t n, double* __restrict__ sums) {
    double s = 0.0, q = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) { const double v = x[i]; s += v; q += v * v; }
    __shared__ double ls[256], lq[256];
    ls[threadIdx.x] = s; lq[threadIdx.x] = q;
    __syncthreads();
    for (int k = 128; k > 0; k >>= 1) {
        if ((int)threadIdx.x < k) { ls[threadIdx.x] += ls[threadIdx.x + k]; lq[threadIdx.x] += lq[threadIdx.x + k]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { atomicAdd(&sums[0], ls[0]); atomicAdd(&sums[1], lq[0]); }
}
// ATen bilinear (align_corners=False, no antialias) source index / weight of one output coordinate
__device__ __forceinline__ void bil_tap(int o, int in, int out, int* i0, int* i1, float* l) {
    const float scale = (float)in / (float)out;
    float r = scale * ((float)o + 0.5f) - 0.5f;
    if (r < 0.f) r = 0.f;
    int i = min((int)r, in - 1);
    *l = fminf(fmaxf(r - (float)i, 0.f), 1.f);
    *i0 = i; *i1 = i + ((i + 1 > in - 1) ? 0 : 1);
}
// crop b = bilinear resize of the z-scored image window boxes[b] = (top, left, h, w) to 320x320; mask crop = nearest resize of the same
// window, binarised; areas[b] = pixels of the mask crop
__global__ __launch_bounds__(256) void cls_crop_kernel(const float* __restrict__ img, const uint8_t* __restrict__ masks, int H, int W,
                                                       const int* __restrict__ boxes, const double* __restrict__ sums, double npix,
                                                       float* __restrict__ crops, uint8_t* __restrict__ cmask, int* __restrict__ areas) {
    const int b = blockIdx.y;
    const int o = blockIdx.x * 256 + threadIdx.x;
    const int oy = o / CROP, ox = o - oy * CROP;
    const int top = boxes[4 * b], left = boxes[4 * b + 1], h = boxes[4 * b + 2], w = boxes[4 * b + 3];
    const double mean_d = sums[0] / npix;
    const double var_d = fmax(sums[1] / npix - mean_d * mean_d, 0.0);
    const float mean = (float)mean_d;
    float sd = (float)sqrt(var_d);
    if (sd == 0.f) sd = 1.f;
    int y0, y1, x0, x1; float ly, lx;
    bil_tap(oy, h, CROP, &y0, &y1, &ly);
    bil_tap(ox, w, CROP, &x0, &x1, &lx);
    const float* r0 = img + (int64_t)(top + y0) * W + left;
    const float* r1 = img + (int64_t)(top + y1) * W + left;
    const float v00 = (r0[x0] - mean) / sd, v01 = (r0[x1] - mean) / sd, v10 = (r1[x0] - mean) / sd, v11 = (r1[x1] - mean) / sd;
    const float wx0 = 1.f - lx, wy0 = 1.f - ly;
    crops[(int64_t)b * CROP_PIX + o] = wy0 * (wx0 * v00 + lx * v01) + ly * (wx0 * v10 + lx * v11);
    // nearest: src = min(floor(dst * in / out), in - 1)
    const int ny = min((int)floorf((float)oy * ((float)h / (float)CROP)), h - 1);
    const int nx = min((int)floorf((float)ox * ((float)w / (float)CROP)), w - 1);
    const int m = masks[((int64_t)b * H + top + ny) * W + left + nx] ? 1 : 0;
    cmask[(int64_t)b * CROP_PIX + o] = (uint8_t)m;
    const unsigned long long ball = __ballot(m);
    if ((threadIdx.x & 63) == 0 && ball) atomicAdd(&areas[b], __popcll(ball));
}
// A0[b][r = y*64+x][0..255] = emb * m, [256..511] = emb * (1 - m); m = mask crop sampled at (5y, 5x) (nearest 320 -> 64)
// (F16: the engine handle's 16-bit operand type - this file is compiled once, so the two element-wise kernels that write GEMM operands pick
// the conversion at compile time through a template flag; the GEMMs themselves dispatch through kernels.h)
template <bool F16> __device__ __forceinline__ uint32_t cls_pack(float lo, float hi) { return F16 ? pack_f16_rn(lo, hi) : pack_bf16_rn(lo, hi); }
template <bool F16>
__global__ __launch_bounds__(64) void cls_roi_kernel(const float* __restrict__ emb, const int* __restrict__ rm_to_eng, const uint8_t* __restrict__ cmask,
                                                     const int* __restrict__ sel, bf16_t* __restrict__ A0) {
    const int r = blockIdx.x, b = blockIdx.y, lane = threadIdx.x;
    const int y = r >> 6, x = r & 63;
    const int t = rm_to_eng[r];
    const bool m = cmask[(int64_t)sel[b] * CROP_PIX + (5 * y) * CROP + 5 * x] != 0;
    const float4 f = *reinterpret_cast<const float4*>(emb + ((int64_t)b * 4096 + t) * 256 + 4 * lane);
    const uint2 v = make_uint2(cls_pack<F16>(f.x, f.y), cls_pack<F16>(f.z, f.w)), z = make_uint2(0u, 0u);
    bf16_t* row = A0 + ((int64_t)b * 4096 + r) * 512 + 4 * lane;
    *reinterpret_cast<uint2*>(row) = m ? v : z;
    *reinterpret_cast<uint2*>(row + 256) = m ? z : v;
}
__device__ __forceinline__ float prelu(float v, float a) { return v >= 0.f ? v : a * v; }
// A[b][y*S+x][(ky*3+kx)*C + c] = act(G)[b][y+ky-1][x+kx-1][c] (zero outside); act = PReLU, followed by a 2x2 max-pool of the 2S x 2S grid
// when POOL.  One block per output pixel.
template <bool POOL, bool F16>
__global__ __launch_bounds__(256) void cls_im2col_kernel(const float* __restrict__ G, float alpha, int S, int C, bf16_t* __restrict__ A) {
    const int r = blockIdx.x, b = blockIdx.y;
    const int y = r / S, x = r - y * S;
    const int c4n = C >> 2;
    const int Sin = POOL ? 2 * S : S;
    const float* Gb = G + (int64_t)b * Sin * Sin * C;
    bf16_t* row = A + ((int64_t)b * S * S + r) * 9 * C;
    for (int idx = threadIdx.x; idx < 9 * c4n; idx += 256) {
        const int k = idx / c4n, c = (idx - k * c4n) * 4;
        const int yy = y + k / 3 - 1, xx = x + k % 3 - 1;
        uint2 o = make_uint2(0u, 0u);
        if (yy >= 0 && yy < S && xx >= 0 && xx < S) {
            float4 v;
            if (POOL) {
                const float* p = Gb + ((int64_t)(2 * yy) * Sin + 2 * xx) * C + c;
                const float4 a = *reinterpret_cast<const float4*>(p), bq = *reinterpret_cast<const float4*>(p + C);
                const float4 cq = *reinterpret_cast<const float4*>(p + (int64_t)Sin * C), d = *reinterpret_cast<const float4*>(p + (int64_t)Sin * C + C);
                v.x = fmaxf(fmaxf(prelu(a.x, alpha), prelu(bq.x, alpha)), fmaxf(prelu(cq.x, alpha), prelu(d.x, alpha)));
                v.y = fmaxf(fmaxf(prelu(a.y, alpha), prelu(bq.y, alpha)), fmaxf(prelu(cq.y, alpha), prelu(d.y, alpha)));
                v.z = fmaxf(fmaxf(prelu(a.z, alpha), prelu(bq.z, alpha)), fmaxf(prelu(cq.z, alpha), prelu(d.z, alpha)));
                v.w = fmaxf(fmaxf(prelu(a.w, alpha), prelu(bq.w, alpha)), fmaxf(prelu(cq.w, alpha), prelu(d.w, alpha)));
            } else {
                v = *reinterpret_cast<const float4*>(Gb + ((int64_t)yy * S + xx) * C + c);
                v.x = prelu(v.x, alpha); v.y = prelu(v.y, alpha); v.z = prelu(v.z, alpha); v.w = prelu(v.w, alpha);
            }
            o = make_uint2(cls_pack<F16>(v.x, v.y), cls_pack<F16>(v.z, v.w));
        }
        *reinterpret_cast<uint2*>(row + (int64_t)k * C + c) = o;
    }
}
// PReLU + 2x2 max-pool of the 32x32x128 map, global average, Linear(128,64) + LayerNorm + PReLU + Linear(64,nc) + softmax.  One block per crop.
__global__ __launch_bounds__(128) void cls_tail_kernel(const float* __restrict__ G3, float a3, const float* __restrict__ fc1w, const float* __restrict__ fc1b,
                                                       const float* __restrict__ lng, const float* __restrict__ lnb, float a4, const float* __restrict__ fc2w,
                                                       const float* __restrict__ fc2b, int nc, float* __restrict__ probs) {
    const int b = blockIdx.x, c = threadIdx.x;
    __shared__ float v[128], h[64], lg[64], st[2];
    const float* g = G3 + (int64_t)b * 1024 * 128 + c;
    float sum = 0.f;
    for (int py = 0; py < 16; ++py)
        for (int px = 0; px < 16; ++px) {
            const float* p = g + ((int64_t)(2 * py) * 32 + 2 * px) * 128;
            const float m = fmaxf(fmaxf(prelu(p[0], a3), prelu(p[128], a3)), fmaxf(prelu(p[32 * 128], a3), prelu(p[33 * 128], a3)));
            sum += m;
        }
    v[c] = sum * (1.0f / 256.0f);
    __syncthreads();
    if (c < 64) {
        float acc = fc1b[c];
        for (int k = 0; k < 128; ++k) acc = fmaf(fc1w[c * 128 + k], v[k], acc);
        h[c] = acc;
    }
    __syncthreads();
    if (c == 0) {
        float m = 0.f;
        for (int k = 0; k < 64; ++k) m += h[k];
        m *= (1.0f / 64.0f);
        float q = 0.f;
        for (int k = 0; k < 64; ++k) { const float d = h[k] - m; q += d * d; }
        st[0] = m; st[1] = rsqrtf(q * (1.0f / 64.0f) + 1e-5f);
    }
    __syncthreads();
    if (c < 64) h[c] = prelu((h[c] - st[0]) * st[1] * lng[c] + lnb[c], a4);
    __syncthreads();
    if (c < nc) {
        float acc = fc2b[c];
        for (int k = 0; k < 64; ++k) acc = fmaf(fc2w[c * 64 + k], h[k], acc);
        lg[c] = acc;
    }
    __syncthreads();
    if (c == 0) {
        float mx = lg[0];
        for (int k = 1; k < nc; ++k) mx = fmaxf(mx, lg[k]);
        float den = 0.f;
        for (int k = 0; k < nc; ++k) den += expf(lg[k] - mx);
        for (int k = 0; k < nc; ++k) probs[(int64_t)b * nc + k] = expf(lg[k] - mx) / den;
    }
}

static const char* launch_cls_roi(const float* emb, const int* rm, const uint8_t* cmask, const int* sel, bf16_t* A0, int k, hipStream_t s) {
    if (saber_op_is_f16()) hipLaunchKernelGGL(cls_roi_kernel<true>, dim3(4096, k), dim3(64), 0, s, emb, rm, cmask, sel, A0);
    else hipLaunchKernelGGL(cls_roi_kernel<false>, dim3(4096, k), dim3(64), 0, s, emb, rm, cmask, sel, A0);
    return nullptr;
}
static const char* launch_cls_im2col(bool pool, const float* G, float alpha, int S, int C, bf16_t* A, int k, hipStream_t s) {
    const bool f16 = saber_op_is_f16();
    if (pool && f16) hipLaunchKernelGGL((cls_im2col_kernel<true, true>), dim3(S * S, k), dim3(256), 0, s, G, alpha, S, C, A);
    else if (pool) hipLaunchKernelGGL((cls_im2col_kernel<true, false>), dim3(S * S, k), dim3(256), 0, s, G, alpha, S, C, A);
    else if (f16) hipLaunchKernelGGL((cls_im2col_kernel<false, true>), dim3(S * S, k), dim3(256), 0, s, G, alpha, S, C, A);
    else hipLaunchKernelGGL((cls_im2col_kernel<false, false>), dim3(S * S, k), dim3(256), 0, s, G, alpha, S, C, A);
    return nullptr;
}

// ------------------------------------------------------------------------------------------------ C-ABI
extern "C" int saber_classifier_create(saber_engine* e, int num_classes, saber_classifier** out) {
    if (!e || !out) return SABER_ERR_INVALID;
    if (num_classes < 1 || num_classes > 64) return eng_fail(e, SABER_ERR_INVALID, "classifier_create: num_classes must be 1..64");
    if (!e->finalized) return eng_fail(e, SABER_ERR_STATE, "classifier_create: engine not finalized");
    saber_classifier* c = new saber_classifier();
    c->e = e; c->num_classes = num_classes;
    *out = c;
    return SABER_OK;
}
extern "C" void saber_classifier_destroy(saber_classifier* c) { delete c; }   // device buffers belong to the engine handle and go with it

extern "C" int saber_classifier_set_weight(saber_classifier* c, const char* name, const float* host, const int64_t* shape, int ndim) {
    if (!c) return SABER_ERR_INVALID;
    if (!name || !host || !shape || ndim < 1 || ndim > 4) return eng_fail(c->e, SABER_ERR_INVALID, "classifier_set_weight: bad argument");
    if (c->finalized) return eng_fail(c->e, SABER_ERR_STATE, "classifier_set_weight after finalize");
    ClsHostT t; int64_t n = 1;
    for (int i = 0; i < ndim; ++i) { if (shape[i] <= 0) return eng_fail(c->e, SABER_ERR_INVALID, "classifier_set_weight: non-positive dim"); t.shape.push_back(shape[i]); n *= shape[i]; }
    t.data.assign(host, host + n);
    c->host[name] = std::move(t);
    return SABER_OK;
}

namespace {
struct ClsFinal {
    saber_classifier* c; int st = SABER_OK;
    const ClsHostT* get(const std::string& name, std::vector<int64_t> shape) {
        if (st != SABER_OK) return nullptr;
        auto it = c->host.find(name);
        if (it == c->host.end()) { st = eng_fail(c->e, SABER_ERR_INVALID, "classifier: missing weight tensor '" + name + "'"); return nullptr; }
        if (it->second.shape != shape) { st = eng_fail(c->e, SABER_ERR_INVALID, "classifier: weight tensor '" + name + "' has the wrong shape"); return nullptr; }
        return &it->second;
    }
    float* up_f32(const std::vector<float>& v) {
        if (st != SABER_OK) return nullptr;
        float* d = nullptr;
        st = eng_alloc(c->e, &d, v.size());
        if (st == SABER_OK && hipMemcpy(d, v.data(), v.size() * 4, hipMemcpyHostToDevice) != hipSuccess) st = eng_fail(c->e, SABER_ERR_HIP, "classifier: weight upload failed");
        return d;
    }
    bf16_t* up_bf16(const std::vector<float>& v) {
        if (st != SABER_OK) return nullptr;
        std::vector<bf16_t> h(v.size());
        for (size_t i = 0; i < v.size(); ++i) h[i] = c->e->op_f16 ? saber_host_f2h(v[i]) : cls_f2bf(v[i]);      // the engine handle's 16-bit operand type
        bf16_t* d = nullptr;
        st = eng_alloc(c->e, &d, h.size());
        if (st == SABER_OK && hipMemcpy(d, h.data(), h.size() * 2, hipMemcpyHostToDevice) != hipSuccess) st = eng_fail(c->e, SABER_ERR_HIP, "classifier: weight upload failed");
        return d;
    }
    // Conv2d(cin -> cout, k) followed by eval-mode BatchNorm2d: y = (conv(x) + b - rm) * g / sqrt(rv + eps) + beta, folded into the conv;
    // weight re-ordered from (cout, cin, k, k) to [cout][(ky*k + kx)*cin + c] (the im2col row order)
    void conv_bn(int ic, int ib, int cout, int cin, int k, bf16_t** w, float** b) {
        const std::string pc = "projection." + std::to_string(ic), pb = "projection." + std::to_string(ib);
        const ClsHostT* W = get(pc + ".weight", {cout, cin, k, k});
        const ClsHostT* B = get(pc + ".bias", {cout});
        const ClsHostT* g = get(pb + ".weight", {cout});
        const ClsHostT* be = get(pb + ".bias", {cout});
        const ClsHostT* rm = get(pb + ".running_mean", {cout});
        const ClsHostT* rv = get(pb + ".running_var", {cout});
        if (st != SABER_OK) return;
        std::vector<float> wf((size_t)cout * k * k * cin), bf(cout);
        for (int o = 0; o < cout; ++o) {
            const double sc = (double)g->data[o] / std::sqrt((double)rv->data[o] + 1e-5);
            bf[o] = (float)(((double)B->data[o] - rm->data[o]) * sc + be->data[o]);
            for (int ci = 0; ci < cin; ++ci)
                for (int q = 0; q < k * k; ++q)
                    wf[((size_t)o * k * k + q) * cin + ci] = (float)(W->data[((size_t)o * cin + ci) * k * k + q] * sc);
        }
        *w = up_bf16(wf); *b = up_f32(bf);
    }
    float scalar(const std::string& name) { const ClsHostT* t = get(name, {1}); return t ? t->data[0] : 0.f; }
};
}  // namespace

extern "C" int saber_classifier_finalize(saber_classifier* c) {
    if (!c) return SABER_ERR_INVALID;
    if (c->finalized) return SABER_OK;
    saber_engine* e = c->e;
    ENG_DEVICE(e);
    ClsFinal f{c};
    f.conv_bn(0, 1, 256, 512, 1, &c->w1, &c->b1);
    f.conv_bn(4, 5, 256, 256, 3, &c->w2, &c->b2);
    f.conv_bn(9, 10, 128, 256, 3, &c->w3, &c->b3);
    c->a1 = f.scalar("projection.2.weight"); c->a2 = f.scalar("projection.6.weight"); c->a3 = f.scalar("projection.11.weight");
    c->a4 = f.scalar("classifier.2.weight");
    const ClsHostT* t;
    if ((t = f.get("classifier.0.weight", {64, 128}))) c->fc1w = f.up_f32(t->data);
    if ((t = f.get("classifier.0.bias", {64}))) c->fc1b = f.up_f32(t->data);
    if ((t = f.get("classifier.1.weight", {64}))) c->lng = f.up_f32(t->data);
    if ((t = f.get("classifier.1.bias", {64}))) c->lnb = f.up_f32(t->data);
    if ((t = f.get("classifier.4.weight", {c->num_classes, 64}))) c->fc2w = f.up_f32(t->data);
    if ((t = f.get("classifier.4.bias", {c->num_classes}))) c->fc2b = f.up_f32(t->data);
    if (f.st != SABER_OK) return f.st;
    TRY(eng_alloc(e, &c->sums, 2));
    TRY(eng_rm_tables(e));
    c->host.clear();
    c->finalized = true;
    return SABER_OK;
}


static GemmParams cls_gemm(const bf16_t* A, int M, int K, const bf16_t* W, const float* bias, int N, float* out) {
    GemmParams p;
    p.A = A; p.lda = K; p.W = W; p.ldw = K; p.w_kpad = 1; p.bias = bias; p.M = M; p.N = N; p.K = K; p.Cf = out; p.ldcf = N;
    return p;
}

static const char* launch_cls_tail(saber_classifier* c, int k, hipStream_t s) {
    hipLaunchKernelGGL(cls_tail_kernel, dim3(k), dim3(128), 0, s, c->G3, c->a3, c->fc1w, c->fc1b, c->lng, c->lnb, c->a4, c->fc2w, c->fc2b, c->num_classes, c->probs);
    return nullptr;
}
static int cls_ensure_head_ws(saber_classifier* c, hipStream_t s) {
    saber_engine* e = c->e;
    const size_t B = e->max_images;
    if (c->cap_b >= B) return SABER_OK;
    TRY(eng_regrow(e, &c->A0, B * 4096 * 512, s)); TRY(eng_regrow(e, &c->G1, B * 4096 * 256, s));
    TRY(eng_regrow(e, &c->A1, B * 4096 * 2304, s)); TRY(eng_regrow(e, &c->G2, B * 4096 * 256, s));
    TRY(eng_regrow(e, &c->A2, B * 1024 * 2304, s)); TRY(eng_regrow(e, &c->G3, B * 1024 * 128, s));
    TRY(eng_regrow(e, &c->probs, B * 64, s));
    TRY(eng_regrow(e, &c->sel, std::max(B, c->cap_n), s));
    c->cap_b = B;
    return SABER_OK;
}
// head on the embeddings of engine slots 0..k-1; mask crop of slot b = cmask[sel[b]] (c->sel already on the device) -> c->probs [k][nc]
static int cls_head(saber_classifier* c, int k, const uint8_t* cmask, hipStream_t s) {
    saber_engine* e = c->e;
    ENG_KP(e, PC_ELEMENTWISE, 0.0, 0.0, launch_cls_roi(e->emb, e->rm_to_eng, cmask, c->sel, c->A0, k, s));
    GemmParams g = cls_gemm(c->A0, k * 4096, 512, c->w1, c->b1, 256, c->G1);
    ENG_KP(e, PC_GEMM, 2.0 * g.M * (double)g.N * g.K, 0.0, launch_gemm(g, s));
    ENG_KP(e, PC_ELEMENTWISE, 0.0, 0.0, launch_cls_im2col(false, c->G1, c->a1, 64, 256, c->A1, k, s));
    g = cls_gemm(c->A1, k * 4096, 2304, c->w2, c->b2, 256, c->G2);
    ENG_KP(e, PC_GEMM, 2.0 * g.M * (double)g.N * g.K, 0.0, launch_gemm(g, s));
    ENG_KP(e, PC_ELEMENTWISE, 0.0, 0.0, launch_cls_im2col(true, c->G2, c->a2, 32, 256, c->A2, k, s));
    g = cls_gemm(c->A2, k * 1024, 2304, c->w3, c->b3, 128, c->G3);
    ENG_KP(e, PC_GEMM, 2.0 * g.M * (double)g.N * g.K, 0.0, launch_gemm(g, s));
    ENG_KP(e, PC_ELEMENTWISE, 0.0, 0.0, launch_cls_tail(c, k, s));
    return SABER_OK;
}

// The window crop_and_resize_adaptive cuts for a mask with bounding box [y0, y1] x [x0, x1] (inclusive): RandMaskCrop.py:91-148
static void cls_crop_box(const int* bb, int H, int W, int* out) {
    const int y0 = bb[0], y1 = bb[1], x0 = bb[2], x1 = bb[3];
    out[0] = 0; out[1] = 0; out[2] = H; out[3] = W;                       // empty mask / nearly full mask: the whole image
    if (y1 < 0) return;
    const int bh = std::max(1, y1 - y0), bw = std::max(1, x1 - x0);
    if ((double)bh / H >= 0.9 && (double)bw / W >= 0.9) return;
    int ch = (int)(bh * (1 + 1.5)), cw = (int)(bw * (1 + 1.5));
    int top = (y0 + y1) / 2 - ch / 2, left = (x0 + x1) / 2 - cw / 2;
    top = std::max(0, std::min(top, H - ch));
    left = std::max(0, std::min(left, W - cw));
    out[0] = top; out[1] = left; out[2] = std::min(ch, H); out[3] = std::min(cw, W);
}

extern "C" int saber_classifier_predict(saber_classifier* c, const float* image_dev, int H, int W, const uint8_t* masks_dev, int n, int min_area,
                                        float* probs_host, void* stream) {
    if (!c) return SABER_ERR_INVALID;
    saber_engine* e = c->e;
    if (!c->finalized) return eng_fail(e, SABER_ERR_STATE, "classifier_predict: classifier not finalized");
    if (n < 0 || H <= 0 || W <= 0 || (n > 0 && (!image_dev || !masks_dev || !probs_host))) return eng_fail(e, SABER_ERR_INVALID, "classifier_predict: bad argument");
    if (n == 0) return SABER_OK;
    ENG_DEVICE(e);
    hipStream_t s = (hipStream_t)stream;
    const int nc = c->num_classes;
    std::fill(probs_host, probs_host + (size_t)n * nc, 0.0f);
    if (c->cap_n < (size_t)n) {
        TRY(eng_regrow(e, &c->bbox, (size_t)4 * n, s)); TRY(eng_regrow(e, &c->boxes, (size_t)4 * n, s));
        TRY(eng_regrow(e, &c->areas, (size_t)n, s));
        TRY(eng_regrow(e, &c->crops, (size_t)n * CROP_PIX, s)); TRY(eng_regrow(e, &c->cmask, (size_t)n * CROP_PIX, s));
        c->cap_n = n;
    }
    // 1. NormalizeIntensity statistics and the masks' bounding boxes
    ENG_HIP(e, hipMemsetAsync(c->sums, 0, 2 * sizeof(double), s));
    ENG_HIP(e, hipMemsetAsync(c->areas, 0, sizeof(int) * n, s));
    hipLaunchKernelGGL(cls_sums_kernel, dim3(256), dim3(256), 0, s, image_dev, (int64_t)H * W, c->sums);
    hipLaunchKernelGGL(cls_bbox_init_kernel, dim3((n + 255) / 256), dim3(256), 0, s, c->bbox, n, H, W);
    hipLaunchKernelGGL(cls_bbox_kernel, dim3(H, n), dim3(256), 0, s, masks_dev, H, W, c->bbox);
    std::vector<int> bb((size_t)4 * n), boxes((size_t)4 * n), areas(n);
    ENG_HIP(e, hipMemcpyAsync(bb.data(), c->bbox, sizeof(int) * 4 * n, hipMemcpyDeviceToHost, s));
    ENG_HIP(e, hipStreamSynchronize(s));
    for (int i = 0; i < n; ++i) cls_crop_box(&bb[4 * i], H, W, &boxes[4 * i]);
    // 2. crops + the area filter of Predictor.preprocess (on the resized mask)
    ENG_HIP(e, hipMemcpyAsync(c->boxes, boxes.data(), sizeof(int) * 4 * n, hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(cls_crop_kernel, dim3(CROP_PIX / 256, n), dim3(256), 0, s, image_dev, masks_dev, H, W, c->boxes, c->sums, (double)H * W, c->crops, c->cmask, c->areas);
    ENG_HIP(e, hipMemcpyAsync(areas.data(), c->areas, sizeof(int) * n, hipMemcpyDeviceToHost, s));
    ENG_HIP(e, hipStreamSynchronize(s));
    std::vector<int> valid;
    for (int i = 0; i < n; ++i) if (areas[i] >= min_area) valid.push_back(i);
    if (valid.empty()) return SABER_OK;
    // 3. encoder + head, max_images crops at a time
    const int B = e->max_images;
    TRY(cls_ensure_head_ws(c, s));
    std::vector<int> bands(4 * (size_t)B);
    std::vector<float> ph((size_t)B * nc);
    for (size_t v0 = 0; v0 < valid.size(); v0 += B) {
        const int k = (int)std::min((size_t)B, valid.size() - v0);
        for (int i = 0; i < k; ++i) { const int ci = valid[v0 + i]; bands[4 * i] = 0; bands[4 * i + 1] = ci * CROP; bands[4 * i + 2] = CROP; bands[4 * i + 3] = (ci + 1) * CROP; }
        TRY(eng_encode(e, c->crops, n * CROP, CROP, 1, bands.data(), k, 0, s));
        ENG_HIP(e, hipMemcpyAsync(c->sel, valid.data() + v0, sizeof(int) * k, hipMemcpyHostToDevice, s));
        TRY(cls_head(c, k, c->cmask, s));
        ENG_HIP(e, hipMemcpyAsync(ph.data(), c->probs, sizeof(float) * k * nc, hipMemcpyDeviceToHost, s));
        ENG_HIP(e, hipStreamSynchronize(s));
        for (int i = 0; i < k; ++i) std::copy(ph.begin() + (size_t)i * nc, ph.begin() + (size_t)(i + 1) * nc, probs_host + (size_t)valid[v0 + i] * nc);
    }
    ENG_HIP(e, hipGetLastError());
    return SABER_OK;
}

// The head alone on the embeddings the engine currently holds in slots 0..k-1 (saber_encode / saber_set_embed_tokens) with caller-provided
// 320x320 mask crops: SAM2Classifier.forward after the backbone (apply_mask_to_features -> projection -> pool -> classifier) + softmax.
extern "C" int saber_classifier_head(saber_classifier* c, const uint8_t* mask_crops_dev, int k, float* probs_host, void* stream) {
    if (!c) return SABER_ERR_INVALID;
    saber_engine* e = c->e;
    if (!c->finalized) return eng_fail(e, SABER_ERR_STATE, "classifier_head: classifier not finalized");
    if (k < 1 || k > e->max_images || !mask_crops_dev || !probs_host) return eng_fail(e, SABER_ERR_INVALID, "classifier_head: bad argument (k exceeds max_images?)");
    for (int i = 0; i < k; ++i) if (!e->slot_valid[i]) return eng_fail(e, SABER_ERR_STATE, "classifier_head: slot holds no encoded image");
    ENG_DEVICE(e);
    hipStream_t s = (hipStream_t)stream;
    TRY(cls_ensure_head_ws(c, s));
    std::vector<int> ident(k);
    for (int i = 0; i < k; ++i) ident[i] = i;
    ENG_HIP(e, hipMemcpyAsync(c->sel, ident.data(), sizeof(int) * k, hipMemcpyHostToDevice, s));
    TRY(cls_head(c, k, mask_crops_dev, s));
    std::vector<float> ph((size_t)k * c->num_classes);
    ENG_HIP(e, hipMemcpyAsync(probs_host, c->probs, sizeof(float) * k * c->num_classes, hipMemcpyDeviceToHost, s));
    ENG_HIP(e, hipStreamSynchronize(s));
    return SABER_OK;
}

// development / test access: the 320x320 crops and mask crops of the last predict call (n crops), and the head alone on caller-provided
// embeddings (row-major tokens) is covered through saber_set_embed_tokens + predict in the tests
extern "C" int saber_classifier_get_crops(saber_classifier* c, int n, float* crops_out_dev, uint8_t* masks_out_dev, void* stream) {
    if (!c) return SABER_ERR_INVALID;
    saber_engine* e = c->e;
    if (n < 0 || (size_t)n > c->cap_n) return eng_fail(e, SABER_ERR_INVALID, "classifier_get_crops: n exceeds the last predict call");
    ENG_DEVICE(e);
    hipStream_t s = (hipStream_t)stream;
    if (crops_out_dev) ENG_HIP(e, hipMemcpyAsync(crops_out_dev, c->crops, sizeof(float) * n * CROP_PIX, hipMemcpyDeviceToDevice, s));
    if (masks_out_dev) ENG_HIP(e, hipMemcpyAsync(masks_out_dev, c->cmask, (size_t)n * CROP_PIX, hipMemcpyDeviceToDevice, s));
    return SABER_OK;
}
