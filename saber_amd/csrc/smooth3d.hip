// Per-label adaptive 3-D Gaussian smoothing of the stitched label volume on the device.
// Replaces saber.filters.masks.fast_3d_gaussian_smoothing (saber/filters/masks.py:230-287) and the separable filter under it,
// saber.filters.gaussian.gaussian_smoothing_3d (saber/filters/gaussian.py:76-138): the step segment_tomogram_core applies to the
// segmenter's output before it is written (saber/entry_points/inference_core.py:68-74, scale = 0.05).  For every label value v != 0,
// ascending: mask = (volume == v); sigma = scale * 2 * (3 |mask| / 4 pi)^(1/3) (masks.py:289-309); taps = normalised
// exp(-i^2 / 2 sigma^2), i in [-r, r], 2r+1 = int(6 sigma + 1) made odd (gaussian.py:97-104); zero-padded 1-D convolutions along
// x, then y, then z in fp32 (gaussian.py:110-131); result[field > 0.5] = (uint8) v, later labels overwriting earlier ones.
//
// The reference convolves the WHOLE volume three times per label.  Here a label only costs its bounding box: outside the box
// (say beyond its last x) at most the taps of one side of the kernel can meet the mask, and those sum to (1 - tap0) / 2 < 0.5, so the
// field cannot pass the threshold there; and along an axis that has been convolved the other two still see zeros outside the box.
//   1. sm_max / sm_stats   largest label value; voxel count + bounding box per label value (one atomic set per wave and label)
//   2. host                sigma, radius, taps per present label (double / float arithmetic as the reference's numpy / torch lines)
//   3. sm_conv_x           T1 = conv_x(volume == v) on the box, all labels of a group in one launch
//   4. sm_conv_axis<0>     T2 = conv_y(T1)
//   5. sm_conv_axis<1>     field = conv_z(T2); field > 0.5 -> atomicMax(winner, v)   ("later label overwrites" = largest label)
//   6. sm_cast             out = (uint8) winner  (the reference's result array is uint8: label values wrap modulo 256)
// HBM-bound byte work: ~20 B per box voxel.  Each output sums its taps in ascending tap order with fma, as the CPU oracle does.
#include <algorithm>
#include <cmath>
#include <vector>

#include "engine.h"

struct SmLabel {
    uint32_t label;
    int z0, y0, x0, dz, dy, dx;
    int r;              // kernel radius, taps = 2r+1
    int tap_off;        // first tap in the tap table (preceded and followed by SM_OPT-1 zeros)
    int64_t ws_off;     // T1 at ws + ws_off, T2 at ws + ws_off + dz*dy*dx (floats)
};

#define SM_OPT 8        // outputs per thread along the convolved axis (sm_conv_axis)
#define SM_TL 32        // outputs per block along the convolved axis
#define SM_TX 64        // x columns per block

template <typename T>
__global__ __launch_bounds__(256) void sm_max_kernel(const T* __restrict__ lab, int64_t n, uint32_t* __restrict__ out) {
    uint32_t m = 0;
    for (int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x; v < n; v += (int64_t)gridDim.x * 256) m = max(m, (uint32_t)lab[v]);
    for (int o = 32; o > 0; o >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, o, 64));
    if ((threadIdx.x & 63) == 0 && m) atomicMax(out, m);
}

// stats[v] = {count, zmin, ymin, xmin, zmax, ymax, xmax, -}; one wave per 512-voxel piece of a row, 8 voxels per lane.
template <typename T>
__global__ __launch_bounds__(256) void sm_stats_kernel(const T* __restrict__ lab, int W, int64_t rows, int H, uint32_t* __restrict__ stats) {
    const int chunks = (W + 511) / 512;
    const int64_t piece = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (piece >= rows * chunks) return;
    const int64_t row = piece / chunks;
    const int lane = threadIdx.x & 63;
    const int xb = (int)(piece % chunks) * 512 + lane * 8;
    const T* p = lab + row * W;
    const uint32_t z = (uint32_t)(row / H), y = (uint32_t)(row % H);
    // runs inside this lane's 8 voxels; a run is flushed through the wave: lanes holding the same label combine first
    uint32_t cur = 0, cnt = 0, xlo = 0, xhi = 0;
    for (int i = 0; i <= 8; ++i) {
        const int x = xb + i;
        const uint32_t v = (i < 8 && x < W) ? (uint32_t)p[x] : 0u;
        const bool flush = (v != cur);
        uint32_t pend = (flush && cur) ? cur : 0u;              // label to flush now (0 = nothing)
        while (true) {
            const uint64_t any = __ballot(pend != 0);
            if (!any) break;
            const int leader = __ffsll((long long)any) - 1;
            const uint32_t lv = (uint32_t)__shfl((int)pend, leader, 64);
            const bool mine = (pend == lv);
            uint32_t c = mine ? cnt : 0u, lo = mine ? xlo : 0xffffffffu, hi = mine ? xhi : 0u;
            for (int o = 32; o > 0; o >>= 1) {
                c += (uint32_t)__shfl_xor((int)c, o, 64);
                lo = min(lo, (uint32_t)__shfl_xor((int)lo, o, 64));
                hi = max(hi, (uint32_t)__shfl_xor((int)hi, o, 64));
            }
            if (lane == leader) {
                uint32_t* s = stats + (size_t)lv * 8;
                atomicAdd(s + 0, c);
                atomicMin(s + 1, z); atomicMin(s + 2, y); atomicMin(s + 3, lo);
                atomicMax(s + 4, z); atomicMax(s + 5, y); atomicMax(s + 6, hi);
            }
            if (mine) pend = 0;
        }
        if (flush) { cur = v; cnt = 0; xlo = (uint32_t)x; }
        if (v) { ++cnt; xhi = (uint32_t)x; }
    }
}

// block -> (label of the group, tile): prefix[i] = tiles of labels < i (uniform binary search, scalar loads)
__device__ __forceinline__ int sm_find(const int* __restrict__ prefix, int n, int b) {
    int lo = 0, hi = n;                                         // prefix[lo] <= b < prefix[hi]
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (prefix[mid] <= b) lo = mid; else hi = mid;
    }
    return lo;
}

// T1[z][y][x] = sum_k taps[k] * (vol[z0+z][y0+y][x0+x+k-r] == label), x+k-r inside the box.  4 waves = 4 rows of the box, 64 x each.
template <typename T>
__global__ __launch_bounds__(256) void sm_conv_x_kernel(const T* __restrict__ vol, int H, int W, const SmLabel* __restrict__ labels,
                                                        const int* __restrict__ prefix, int n_labels, const float* __restrict__ taps,
                                                        float* __restrict__ ws) {
    extern __shared__ float sm_lds[];
    const int li = sm_find(prefix, n_labels, blockIdx.x);
    const SmLabel L = labels[li];
    const int t = blockIdx.x - prefix[li];
    const int xchunks = (L.dx + 63) / 64;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int row = (t / xchunks) * 4 + wave;                   // row of the box: z * dy + y
    const int xc = (t % xchunks) * 64;
    const int seg = 64 + 2 * L.r;
    float* s = sm_lds + wave * seg;
    const bool live = row < L.dz * L.dy;
    if (live) {
        const int z = row / L.dy, y = row % L.dy;
        const T* p = vol + ((int64_t)(L.z0 + z) * H + (L.y0 + y)) * W + L.x0;
        for (int i = lane; i < seg; i += 64) {
            const int x = xc - L.r + i;
            s[i] = (x >= 0 && x < L.dx && (uint32_t)p[x] == L.label) ? 1.0f : 0.0f;
        }
    }
    __syncthreads();
    if (!live || xc + lane >= L.dx) return;
    const float* tp = taps + L.tap_off;
    float acc = 0.0f;
    for (int k = 0; k <= 2 * L.r; ++k) acc = fmaf(tp[k], s[lane + k], acc);
    ws[L.ws_off + (int64_t)row * L.dx + xc + lane] = acc;
}

// 1-D convolution along y (AXIS 0: T1 -> T2) or z (AXIS 1: T2 -> field -> threshold / dense float output).
// Block = SM_TL outputs along the axis x SM_TX columns of one (outer) plane; thread = one column, SM_OPT consecutive outputs.
template <int AXIS, bool FLOAT_OUT>
__global__ __launch_bounds__(256) void sm_conv_axis_kernel(const SmLabel* __restrict__ labels, const int* __restrict__ prefix, int n_labels,
                                                           const float* __restrict__ taps, float* __restrict__ ws, int H, int W,
                                                           uint32_t* __restrict__ winner, float* __restrict__ dense) {
    extern __shared__ float sm_lds[];
    const int li = sm_find(prefix, n_labels, blockIdx.x);
    const SmLabel L = labels[li];
    int t = blockIdx.x - prefix[li];
    const int len = AXIS == 0 ? L.dy : L.dz;                    // convolved axis
    const int n_outer = AXIS == 0 ? L.dz : L.dy;
    const int64_t st_axis = AXIS == 0 ? L.dx : (int64_t)L.dy * L.dx;
    const int64_t st_outer = AXIS == 0 ? (int64_t)L.dy * L.dx : L.dx;
    const int xchunks = (L.dx + SM_TX - 1) / SM_TX, lchunks = (len + SM_TL - 1) / SM_TL;
    const int xc = (t % xchunks) * SM_TX; t /= xchunks;
    const int l0 = (t % lchunks) * SM_TL;
    const int outer = t / lchunks;
    (void)n_outer;
    const int64_t vol = (int64_t)L.dz * L.dy * L.dx;
    const float* src = ws + L.ws_off + (AXIS == 0 ? 0 : vol) + outer * st_outer;
    const int rows = SM_TL + 2 * L.r;
    const int xl = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const bool xok = xc + xl < L.dx;
    for (int j = grp; j < rows; j += 4) {
        const int l = l0 - L.r + j;
        sm_lds[j * SM_TX + xl] = (xok && l >= 0 && l < len) ? src[l * st_axis + xc + xl] : 0.0f;
    }
    __syncthreads();
    const int o0 = grp * SM_OPT;                                // this thread's outputs l0 + o0 .. + SM_OPT-1
    if (!xok || l0 + o0 >= len) return;
    // input j (relative to l0 + o0 - r) feeds output o with tap j - o; the tap table has SM_OPT-1 zeros on both sides
    const float* tp = taps + L.tap_off;
    float acc[SM_OPT];
#pragma unroll
    for (int o = 0; o < SM_OPT; ++o) acc[o] = 0.0f;
    const float* col = sm_lds + o0 * SM_TX + xl;
    for (int j = 0; j < 2 * L.r + SM_OPT; ++j) {
        const float v = col[j * SM_TX];
#pragma unroll
        for (int o = 0; o < SM_OPT; ++o) acc[o] = fmaf(tp[j - o], v, acc[o]);
    }
#pragma unroll
    for (int o = 0; o < SM_OPT; ++o) {
        const int l = l0 + o0 + o;
        if (l >= len) break;
        if (AXIS == 0) {
            ws[L.ws_off + vol + outer * st_outer + l * st_axis + xc + xl] = acc[o];
        } else {
            const int64_t g = ((int64_t)(L.z0 + l) * H + (L.y0 + outer)) * W + L.x0 + xc + xl;
            if (FLOAT_OUT) dense[g] = acc[o];
            else if (acc[o] > 0.5f) atomicMax(winner + g, L.label);
        }
    }
}

__global__ __launch_bounds__(256) void sm_cast_kernel(const uint32_t* __restrict__ winner, uint8_t* __restrict__ out, int64_t n) {
    const int64_t n4 = n >> 2;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        const uint4 w = reinterpret_cast<const uint4*>(winner)[i];
        reinterpret_cast<uint32_t*>(out)[i] = (w.x & 255u) | ((w.y & 255u) << 8) | ((w.z & 255u) << 16) | (w.w << 24);
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) out[(n4 << 2) + threadIdx.x] = (uint8_t)winner[(n4 << 2) + threadIdx.x];
}

// kernel size and taps exactly as gaussian.py:97-104 computes them (float32 taps, float32 normalisation)
static int sm_make_taps(double sigma, std::vector<float>& out) {
    int ks = (int)(2 * 3 * sigma + 1);
    if (ks % 2 == 0) ks += 1;
    const int r = ks / 2;
    const float den = (float)(2.0 * sigma * sigma);
    std::vector<float> t(ks);
    float sum = 0.0f;
    for (int i = -r; i <= r; ++i) { t[i + r] = expf(-(float)(i * i) / den); sum += t[i + r]; }
    for (int i = 0; i < ks; ++i) out.push_back(t[i] / sum);
    return r;
}

#define SM_HIP(e, call)                                                                                                  \
    do {                                                                                                                 \
        hipError_t _st = (call);                                                                                         \
        if (_st != hipSuccess) { cleanup(); return eng_fail((e), SABER_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(_st)); } \
    } while (0)

#define SM_MAX_LABEL (1u << 22)
#define SM_LDS_LIMIT (150 * 1024)

namespace {
struct SmScratch {
    uint32_t *maxv = nullptr, *stats = nullptr, *winner = nullptr;
    SmLabel* labels = nullptr;
    int* prefix = nullptr;
    float *taps = nullptr, *ws = nullptr;
    void release() {
        (void)hipFree(maxv); (void)hipFree(stats); (void)hipFree(winner); (void)hipFree(labels); (void)hipFree(prefix); (void)hipFree(taps); (void)hipFree(ws);
    }
};

template <typename T>
void sm_launch_stats(const T* lab, int Z, int H, int W, uint32_t* stats, hipStream_t s) {
    const int64_t rows = (int64_t)Z * H;
    const int64_t pieces = rows * ((W + 511) / 512);
    hipLaunchKernelGGL(sm_stats_kernel<T>, dim3((unsigned)((pieces + 3) / 4)), dim3(256), 0, s, lab, W, rows, H, stats);
}
template <typename T>
void sm_launch_max(const T* lab, int64_t n, uint32_t* out, hipStream_t s) {
    hipLaunchKernelGGL(sm_max_kernel<T>, dim3((unsigned)std::min<int64_t>((n + 255) / 256, 1 << 16)), dim3(256), 0, s, lab, n, out);
}
template <typename T>
void sm_launch_conv_x(const T* vol, int H, int W, const SmLabel* labels, const int* prefix, int n, int tiles, int rmax, const float* taps,
                      float* ws, hipStream_t s) {
    hipLaunchKernelGGL(sm_conv_x_kernel<T>, dim3(tiles), dim3(256), (size_t)4 * (64 + 2 * rmax) * sizeof(float), s, vol, H, W, labels, prefix,
                       n, taps, ws);
}
}  // namespace

// mode 0: label volume -> uint8 (fast_3d_gaussian_smoothing);  mode 1: binary mask (any non-zero) + fixed sigma -> float field
static int sm_run(saber_engine* e, const void* vol_dev, int elem_bytes, int Z, int H, int W, double scale, double fixed_sigma, int mode,
                  void* out_dev, int* out_n_labels, hipStream_t s) {
    const int64_t n = (int64_t)Z * H * W;
    SmScratch S;
    auto cleanup = [&]() { S.release(); };
    ENG_DEVICE(e);
    if (out_n_labels) *out_n_labels = 0;
    SM_HIP(e, hipMalloc(&S.maxv, 4));
    SM_HIP(e, hipMemsetAsync(S.maxv, 0, 4, s));
    if (elem_bytes == 1) sm_launch_max((const uint8_t*)vol_dev, n, S.maxv, s);
    else if (elem_bytes == 2) sm_launch_max((const uint16_t*)vol_dev, n, S.maxv, s);
    else sm_launch_max((const uint32_t*)vol_dev, n, S.maxv, s);
    uint32_t maxv = 0;
    SM_HIP(e, hipMemcpyAsync(&maxv, S.maxv, 4, hipMemcpyDeviceToHost, s));
    SM_HIP(e, hipStreamSynchronize(s));
    if (mode == 0) SM_HIP(e, hipMemsetAsync(out_dev, 0, (size_t)n, s));
    else SM_HIP(e, hipMemsetAsync(out_dev, 0, (size_t)n * 4, s));
    if (maxv == 0) { SM_HIP(e, hipStreamSynchronize(s)); cleanup(); return SABER_OK; }
    if (maxv > SM_MAX_LABEL) { cleanup(); return eng_fail(e, SABER_ERR_INVALID, "smooth_labels: label values above 2^22 are not supported"); }
    // ---- per-label count + bounding box
    const size_t n_stats = (size_t)maxv + 1;
    std::vector<uint32_t> st(n_stats * 8);
    for (size_t v = 0; v < n_stats; ++v) { uint32_t* p = &st[v * 8]; p[0] = 0; p[1] = p[2] = p[3] = 0xffffffffu; p[4] = p[5] = p[6] = p[7] = 0; }
    SM_HIP(e, hipMalloc(&S.stats, n_stats * 32));
    SM_HIP(e, hipMemcpyAsync(S.stats, st.data(), n_stats * 32, hipMemcpyHostToDevice, s));
    if (elem_bytes == 1) sm_launch_stats((const uint8_t*)vol_dev, Z, H, W, S.stats, s);
    else if (elem_bytes == 2) sm_launch_stats((const uint16_t*)vol_dev, Z, H, W, S.stats, s);
    else sm_launch_stats((const uint32_t*)vol_dev, Z, H, W, S.stats, s);
    SM_HIP(e, hipGetLastError());
    SM_HIP(e, hipMemcpyAsync(st.data(), S.stats, n_stats * 32, hipMemcpyDeviceToHost, s));
    SM_HIP(e, hipStreamSynchronize(s));
    // ---- host: sigma, radius, taps per present label, ascending label value (np.unique order, masks.py:253-262)
    std::vector<SmLabel> labs;
    std::vector<float> taps(SM_OPT - 1, 0.0f);
    int rmax = 0;
    int64_t biggest = 0;
    for (size_t v = 1; v < n_stats; ++v) {
        const uint32_t* p = &st[v * 8];
        if (!p[0]) continue;
        if (mode == 1 && v != 1) { cleanup(); return eng_fail(e, SABER_ERR_INVALID, "gaussian_smoothing_3d: the input must be a 0/1 mask"); }
        SmLabel L;
        L.label = (uint32_t)v;
        L.z0 = (int)p[1]; L.y0 = (int)p[2]; L.x0 = (int)p[3];
        L.dz = (int)(p[4] - p[1]) + 1; L.dy = (int)(p[5] - p[2]) + 1; L.dx = (int)(p[6] - p[3]) + 1;
        if (mode == 1) { L.z0 = L.y0 = L.x0 = 0; L.dz = Z; L.dy = H; L.dx = W; }      // the float field is wanted everywhere
        double sigma = fixed_sigma;
        if (mode == 0) {
            const double approx_diameter = 2.0 * std::pow((3.0 * (double)p[0]) / (4.0 * M_PI), 1.0 / 3.0);   // masks.py:303-308
            sigma = scale * approx_diameter;
        }
        L.tap_off = (int)taps.size();
        L.r = sm_make_taps(sigma, taps);
        taps.insert(taps.end(), SM_OPT - 1, 0.0f);
        L.ws_off = 0;
        rmax = std::max(rmax, L.r);
        biggest = std::max(biggest, (int64_t)L.dz * L.dy * L.dx);
        labs.push_back(L);
    }
    if ((size_t)(SM_TL + 2 * rmax) * SM_TX * sizeof(float) > SM_LDS_LIMIT) {
        cleanup();
        return eng_fail(e, SABER_ERR_INVALID, "smooth_labels: Gaussian radius " + std::to_string(rmax) + " exceeds the supported 284 voxels");
    }
    if (out_n_labels) *out_n_labels = (int)labs.size();
    // ---- groups of labels whose two fp32 boxes fit the workspace (>= the largest single label, at least 1 GiB of floats when useful)
    int64_t total = 0;
    for (auto& L : labs) total += 2 * (int64_t)L.dz * L.dy * L.dx;
    const int64_t ws_floats = std::min(total, std::max<int64_t>(2 * biggest, (int64_t)1 << 28));
    SM_HIP(e, hipMalloc(&S.ws, (size_t)ws_floats * 4));
    SM_HIP(e, hipMalloc(&S.taps, taps.size() * 4));
    SM_HIP(e, hipMemcpyAsync(S.taps, taps.data(), taps.size() * 4, hipMemcpyHostToDevice, s));
    SM_HIP(e, hipMalloc(&S.labels, labs.size() * sizeof(SmLabel)));
    SM_HIP(e, hipMalloc(&S.prefix, 3 * (labs.size() + 1) * sizeof(int)));
    if (mode == 0) {
        SM_HIP(e, hipMalloc(&S.winner, (size_t)n * 4));
        SM_HIP(e, hipMemsetAsync(S.winner, 0, (size_t)n * 4, s));
    }
    SM_HIP(e, hipFuncSetAttribute((const void*)sm_conv_axis_kernel<0, false>, hipFuncAttributeMaxDynamicSharedMemorySize, SM_LDS_LIMIT));
    SM_HIP(e, hipFuncSetAttribute((const void*)sm_conv_axis_kernel<1, false>, hipFuncAttributeMaxDynamicSharedMemorySize, SM_LDS_LIMIT));
    SM_HIP(e, hipFuncSetAttribute((const void*)sm_conv_axis_kernel<1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, SM_LDS_LIMIT));
    std::vector<int> prefix;
    size_t g0 = 0;
    while (g0 < labs.size()) {
        size_t g1 = g0;
        int64_t used = 0;
        int grmax = 0;
        int64_t tiles[3] = {0, 0, 0};
        prefix.assign(3 * (labs.size() + 1), 0);
        const size_t stride = labs.size() + 1;
        while (g1 < labs.size()) {
            SmLabel& L = labs[g1];
            const int64_t v2 = 2 * (int64_t)L.dz * L.dy * L.dx;
            if (g1 > g0 && used + v2 > ws_floats) break;
            L.ws_off = used;
            used += v2;
            grmax = std::max(grmax, L.r);
            const int64_t xch = (L.dx + 63) / 64;
            const int64_t tx = (((int64_t)L.dz * L.dy + 3) / 4) * xch;
            const int64_t ty = (int64_t)L.dz * ((L.dy + SM_TL - 1) / SM_TL) * xch;
            const int64_t tz = (int64_t)L.dy * ((L.dz + SM_TL - 1) / SM_TL) * xch;
            if (tiles[0] + tx > 0x7fffffff || tiles[1] + ty > 0x7fffffff || tiles[2] + tz > 0x7fffffff) { if (g1 == g0) { cleanup(); return eng_fail(e, SABER_ERR_INVALID, "smooth_labels: label too large"); } break; }
            const size_t i = g1 - g0;
            prefix[0 * stride + i] = (int)tiles[0]; prefix[1 * stride + i] = (int)tiles[1]; prefix[2 * stride + i] = (int)tiles[2];
            tiles[0] += tx; tiles[1] += ty; tiles[2] += tz;
            ++g1;
        }
        const int ng = (int)(g1 - g0);
        for (int a = 0; a < 3; ++a) prefix[a * stride + ng] = (int)tiles[a];
        SM_HIP(e, hipMemcpyAsync(S.labels, labs.data() + g0, (size_t)ng * sizeof(SmLabel), hipMemcpyHostToDevice, s));
        SM_HIP(e, hipMemcpyAsync(S.prefix, prefix.data(), prefix.size() * sizeof(int), hipMemcpyHostToDevice, s));
        if (elem_bytes == 1) sm_launch_conv_x((const uint8_t*)vol_dev, H, W, S.labels, S.prefix, ng, (int)tiles[0], grmax, S.taps + 0, S.ws, s);
        else if (elem_bytes == 2) sm_launch_conv_x((const uint16_t*)vol_dev, H, W, S.labels, S.prefix, ng, (int)tiles[0], grmax, S.taps + 0, S.ws, s);
        else sm_launch_conv_x((const uint32_t*)vol_dev, H, W, S.labels, S.prefix, ng, (int)tiles[0], grmax, S.taps + 0, S.ws, s);
        const size_t lds = (size_t)(SM_TL + 2 * grmax) * SM_TX * sizeof(float);
        hipLaunchKernelGGL((sm_conv_axis_kernel<0, false>), dim3((unsigned)tiles[1]), dim3(256), lds, s, (const SmLabel*)S.labels,
                           (const int*)(S.prefix + stride), ng, (const float*)S.taps, S.ws, H, W, (uint32_t*)nullptr, (float*)nullptr);
        if (mode == 0)
            hipLaunchKernelGGL((sm_conv_axis_kernel<1, false>), dim3((unsigned)tiles[2]), dim3(256), lds, s, (const SmLabel*)S.labels,
                               (const int*)(S.prefix + 2 * stride), ng, (const float*)S.taps, S.ws, H, W, S.winner, (float*)nullptr);
        else
            hipLaunchKernelGGL((sm_conv_axis_kernel<1, true>), dim3((unsigned)tiles[2]), dim3(256), lds, s, (const SmLabel*)S.labels,
                               (const int*)(S.prefix + 2 * stride), ng, (const float*)S.taps, S.ws, H, W, (uint32_t*)nullptr, (float*)out_dev);
        SM_HIP(e, hipGetLastError());
        SM_HIP(e, hipStreamSynchronize(s));                    // the descriptor / prefix host buffers are reused by the next group
        g0 = g1;
    }
    if (mode == 0) {
        hipLaunchKernelGGL(sm_cast_kernel, dim3((unsigned)std::min<int64_t>((n / 4 + 255) / 256 + 1, 1 << 16)), dim3(256), 0, s,
                           (const uint32_t*)S.winner, (uint8_t*)out_dev, n);
        SM_HIP(e, hipGetLastError());
    }
    SM_HIP(e, hipStreamSynchronize(s));
    cleanup();
    return SABER_OK;
}

extern "C" int saber_smooth_labels(saber_engine* e, const void* labels_dev, int elem_bytes, int Z, int H, int W, double scale,
                                   uint8_t* out_dev, int* out_n_labels, void* stream) {
    if (!e) return SABER_ERR_INVALID;
    if (!labels_dev || !out_dev || Z <= 0 || H <= 0 || W <= 0 || (elem_bytes != 1 && elem_bytes != 2 && elem_bytes != 4) || !(scale > 0.0))
        return eng_fail(e, SABER_ERR_INVALID, "smooth_labels: bad argument");
    if ((int64_t)Z * H * W >= (int64_t)0x7fffffff) return eng_fail(e, SABER_ERR_INVALID, "smooth_labels: volumes of 2^31 voxels or more are not supported");
    return sm_run(e, labels_dev, elem_bytes, Z, H, W, scale, 0.0, 0, out_dev, out_n_labels, (hipStream_t)stream);
}

extern "C" int saber_gaussian_smoothing_3d(saber_engine* e, const uint8_t* mask_dev, int Z, int H, int W, double sigma, float* out_dev,
                                           void* stream) {
    if (!e) return SABER_ERR_INVALID;
    if (!mask_dev || !out_dev || Z <= 0 || H <= 0 || W <= 0 || !(sigma > 0.0)) return eng_fail(e, SABER_ERR_INVALID, "gaussian_smoothing_3d: bad argument");
    if ((int64_t)Z * H * W >= (int64_t)0x7fffffff) return eng_fail(e, SABER_ERR_INVALID, "gaussian_smoothing_3d: volumes of 2^31 voxels or more are not supported");
    return sm_run(e, mask_dev, 1, Z, H, W, 0.0, sigma, 1, out_dev, nullptr, (hipStream_t)stream);
}
