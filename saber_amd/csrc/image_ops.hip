// Image-side kernels of the hot path:
//   K0  prepare_*        : reference prep.prepare (saber/utils/preprocessing.py:4-18,20-37,67-80):
//                          local contrast via uniform_filter(size=500, mode='reflect'), clip +-3 sigma,
//                          global min-max to [0,1].  uint16 / float32 slices.
//   K1a resize_normalize : SAM2Transforms (crop -> bilinear Resize(1024) -> ImageNet normalize), b1
//   K1b patch_embed      : Hiera PatchEmbed Conv2d(3->C, k7 s4 p3) + pos-embed add, written in the
//                          engine's token order (b2, b3)
#include "common.h"
#include "kernels.h"

// ------------------------------------------------------------------------------------------------ K0
// One block per line (row or column).  The line is reflect-extended by the filter footprint,
// prefix-summed in double (scipy accumulates in double too) and each output is
// (P[i+size] - P[i]) / size rounded to fp32 - exactly the value NI_UniformFilter1D stores.
__device__ __forceinline__ int reflect_index(int j, int L) {
    // scipy mode='reflect' (d c b a | a b c d | d c b a)
    const int period = 2 * L;
    j %= period;
    if (j < 0) j += period;
    return j < L ? j : period - 1 - j;
}

template <typename TIN, bool SQUARE_SECOND>
__global__ __launch_bounds__(256) void box_filter_lines_kernel(const TIN* __restrict__ in1, const float* __restrict__ in2,
                                                               float* __restrict__ out1, float* __restrict__ out2, int L,
                                                               int64_t line_stride, int64_t elem_stride, int size) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int n = L + size - 1;              // extended length
    const int npad = ((n + 255) / 256) * 256;
    double* e1 = reinterpret_cast<double*>(smem);
    double* e2 = e1 + npad + 1;
    __shared__ double wsum1[4], wsum2[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t base = (int64_t)blockIdx.x * line_stride;
    const int left = size / 2;
    // load reflect-extended line
    for (int j = tid; j < npad; j += 256) {
        double a = 0.0, b = 0.0;
        if (j < n) {
            const int src = reflect_index(j - left, L);
            const float v = (float)in1[base + (int64_t)src * elem_stride];
            a = (double)v;
            if (SQUARE_SECOND) b = (double)(v * v);  // image**2 is evaluated in float32 by numpy
            else b = (double)in2[base + (int64_t)src * elem_stride];
        }
        e1[j + 1] = a;
        e2[j + 1] = b;
    }
    if (tid == 0) { e1[0] = 0.0; e2[0] = 0.0; }
    __syncthreads();
    // inclusive prefix sums over e[1..npad]: thread owns a contiguous chunk
    const int per = npad / 256;
    double s1 = 0.0, s2 = 0.0;
    for (int k = 0; k < per; ++k) { s1 += e1[1 + tid * per + k]; s2 += e2[1 + tid * per + k]; }
    double p1 = s1, p2 = s2;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const double t1 = __shfl_up(p1, o, 64), t2 = __shfl_up(p2, o, 64);
        if (lane >= o) { p1 += t1; p2 += t2; }
    }
    if (lane == 63) { wsum1[wave] = p1; wsum2[wave] = p2; }
    __syncthreads();
    double off1 = p1 - s1, off2 = p2 - s2;
    for (int w = 0; w < wave; ++w) { off1 += wsum1[w]; off2 += wsum2[w]; }
    for (int k = 0; k < per; ++k) {
        off1 += e1[1 + tid * per + k]; e1[1 + tid * per + k] = off1;
        off2 += e2[1 + tid * per + k]; e2[1 + tid * per + k] = off2;
    }
    __syncthreads();
    const double inv = (double)size;
    for (int i = tid; i < L; i += 256) {
        out1[base + (int64_t)i * elem_stride] = (float)((e1[i + size] - e1[i]) / inv);
        out2[base + (int64_t)i * elem_stride] = (float)((e2[i + size] - e2[i]) / inv);
    }
}

__device__ __forceinline__ unsigned int float_to_ordered(float f) {
    const unsigned int u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ordered_to_float(unsigned int u) {
    return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u);
}

template <typename TIN>
__global__ __launch_bounds__(256) void contrast_kernel(const TIN* __restrict__ img, const float* __restrict__ mean,
                                                       const float* __restrict__ sq, float* __restrict__ z, int64_t n,
                                                       float cutoff, unsigned int* __restrict__ minmax) {
    float lo = 3.0e38f, hi = -3.0e38f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float x = (float)img[i];
        const float m = mean[i];
        const float var = fmaxf(sq[i] - m * m, 0.f);
        float v = (x - m) / (sqrtf(var) + 1e-8f);
        v = fminf(fmaxf(v, -cutoff), cutoff);
        z[i] = v;
        lo = fminf(lo, v);
        hi = fmaxf(hi, v);
    }
    lo = wave_min(lo);
    hi = wave_max(hi);
    if ((threadIdx.x & 63) == 0) {
        atomicMin(&minmax[0], float_to_ordered(lo));
        atomicMax(&minmax[1], float_to_ordered(hi));
    }
}

__global__ void minmax_init_kernel(unsigned int* minmax) {
    minmax[0] = 0xffffffffu;
    minmax[1] = 0u;
}

__global__ __launch_bounds__(256) void minmax_normalize_kernel(const float* __restrict__ z, float* __restrict__ out, int64_t n,
                                                               const unsigned int* __restrict__ minmax) {
    const float lo = ordered_to_float(minmax[0]), hi = ordered_to_float(minmax[1]);
    const float den = (hi - lo) + 1e-8f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) out[i] = (z[i] - lo) / den;
}

template <typename TIN>
static const char* prepare_impl(const TIN* img, int H, int W, float* out, float* ws, unsigned int* minmax, hipStream_t s) {
    // ws: 4 planes of H*W floats: t1, t2 (axis-0 pass), mean, sq (axis-1 pass); z reuses t1
    const int size = 500;
    const int64_t n = (int64_t)H * W;
    float *t1 = ws, *t2 = ws + n, *mean = ws + 2 * n, *sq = ws + 3 * n;
    auto lds_bytes = [&](int L) { const int npad = ((L + size - 1 + 255) / 256) * 256; return (size_t)2 * (npad + 1) * sizeof(double); };
    if (lds_bytes(H) > 152 * 1024 || lds_bytes(W) > 152 * 1024) return "prepare: image side too large for the LDS line buffer";
    // scipy filters axis 0 first (lines = columns), then axis 1 (lines = rows)
    hipLaunchKernelGGL((box_filter_lines_kernel<TIN, true>), dim3(W), dim3(256), lds_bytes(H), s, img, (const float*)nullptr, t1, t2, H,
                       (int64_t)1, (int64_t)W, size);
    hipLaunchKernelGGL((box_filter_lines_kernel<float, false>), dim3(H), dim3(256), lds_bytes(W), s, (const float*)t1, (const float*)t2,
                       mean, sq, W, (int64_t)W, (int64_t)1, size);
    hipLaunchKernelGGL(minmax_init_kernel, dim3(1), dim3(1), 0, s, minmax);
    hipLaunchKernelGGL((contrast_kernel<TIN>), dim3(1024), dim3(256), 0, s, img, (const float*)mean, (const float*)sq, t1, n, 3.0f, minmax);
    hipLaunchKernelGGL(minmax_normalize_kernel, dim3(1024), dim3(256), 0, s, (const float*)t1, out, n, (const unsigned int*)minmax);
    return nullptr;
}

// (H,W,3) input (reference: prep.prepare on an RGB array, saber/utils/preprocessing.py:67-80 via adapters/sam2/predictor.py:58-59):
// scipy's uniform_filter(size=500) runs over ALL three axes of the array, the channel axis included (axis order 0, 1, 2, each pass
// rounded to the array's dtype).  Along the 3-long channel axis the reflect-extended line has period 6 (a b c c b a), so a 500-wide
// window holds 83 periods plus two more samples: out[0] = (166 S + 2c) / 500, out[1] = (166 S + c + b) / 500, out[2] = (166 S + b + a) / 500
// with S = a + b + c (window = [i - 250, i + 249]).
__global__ __launch_bounds__(256) void channel_box3_kernel(float* __restrict__ m, float* __restrict__ q, int64_t npix) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < npix; i += (int64_t)gridDim.x * 256) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            float* p = (t ? q : m) + 3 * i;
            const double a = p[0], b = p[1], c = p[2], S = 166.0 * (a + b + c);
            p[0] = (float)((S + c + c) / 500.0);
            p[1] = (float)((S + c + b) / 500.0);
            p[2] = (float)((S + b + a) / 500.0);
        }
    }
}

const char* launch_prepare_rgb_f32(const float* img, int H, int W, float* out, float* ws, unsigned int* minmax, hipStream_t s) {
    // ws: 4 planes of 3*H*W floats
    const int size = 500;
    const int64_t n = (int64_t)H * W * 3;
    float *t1 = ws, *t2 = ws + n, *mean = ws + 2 * n, *sq = ws + 3 * n;
    auto lds_bytes = [&](int L) { const int npad = ((L + size - 1 + 255) / 256) * 256; return (size_t)2 * (npad + 1) * sizeof(double); };
    if (lds_bytes(H) > 152 * 1024 || lds_bytes(W) > 152 * 1024) return "prepare: image side too large for the LDS line buffer";
    hipLaunchKernelGGL((box_filter_lines_kernel<float, true>), dim3(3 * W), dim3(256), lds_bytes(H), s, img, (const float*)nullptr, t1, t2, H,
                       (int64_t)1, (int64_t)3 * W, size);
    for (int c = 0; c < 3; ++c)
        hipLaunchKernelGGL((box_filter_lines_kernel<float, false>), dim3(H), dim3(256), lds_bytes(W), s, (const float*)(t1 + c), (const float*)(t2 + c),
                           mean + c, sq + c, W, (int64_t)3 * W, (int64_t)3, size);
    hipLaunchKernelGGL(channel_box3_kernel, dim3(1024), dim3(256), 0, s, mean, sq, (int64_t)H * W);
    hipLaunchKernelGGL(minmax_init_kernel, dim3(1), dim3(1), 0, s, minmax);
    hipLaunchKernelGGL((contrast_kernel<float>), dim3(1024), dim3(256), 0, s, img, (const float*)mean, (const float*)sq, t1, n, 3.0f, minmax);
    hipLaunchKernelGGL(minmax_normalize_kernel, dim3(1024), dim3(256), 0, s, (const float*)t1, out, n, (const unsigned int*)minmax);
    return nullptr;
}

const char* launch_prepare_u16(const uint16_t* img, int H, int W, float* out, float* ws, unsigned int* minmax, hipStream_t s) {
    return prepare_impl<uint16_t>(img, H, W, out, ws, minmax, s);
}
const char* launch_prepare_f32(const float* img, int H, int W, float* out, float* ws, unsigned int* minmax, hipStream_t s) {
    return prepare_impl<float>(img, H, W, out, ws, minmax, s);
}

const char* image_ops_init_device() {
    hipError_t st = hipSuccess;
    if (st == hipSuccess) st = hipFuncSetAttribute(reinterpret_cast<const void*>(box_filter_lines_kernel<uint16_t, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024);
    if (st == hipSuccess) st = hipFuncSetAttribute(reinterpret_cast<const void*>(box_filter_lines_kernel<float, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024);
    if (st == hipSuccess) st = hipFuncSetAttribute(reinterpret_cast<const void*>(box_filter_lines_kernel<float, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024);
    return st == hipSuccess ? nullptr : hipGetErrorString(st);
}

// ------------------------------------------------------------------------------------------------ K1a
// out[b][c][oy][ox] = (resize(crop_b(image))[c] - mean[c]) / std[c]; separable triangle filter with the
// exact index/weight rule of ATen's antialiased bilinear (== plain bilinear when upsampling).
struct AxisTaps { int lo; int n; float scale; float center; float invscale; };
__device__ __forceinline__ AxisTaps axis_taps(int o, int in_size, int out_size) {
    AxisTaps t;
    t.scale = (float)in_size / (float)out_size;
    const float support = t.scale >= 1.0f ? t.scale : 1.0f;
    t.invscale = t.scale >= 1.0f ? 1.0f / t.scale : 1.0f;
    t.center = t.scale * ((float)o + 0.5f);
    t.lo = max((int)(t.center - support + 0.5f), 0);
    t.n = min((int)(t.center + support + 0.5f), in_size) - t.lo;
    return t;
}
__device__ __forceinline__ float tri(float x) { x = fabsf(x); return x < 1.0f ? 1.0f - x : 0.0f; }

__global__ __launch_bounds__(256) void resize_normalize_kernel(const float* __restrict__ img, int H, int W, int channels,
                                                               const int* __restrict__ crops, float* __restrict__ out, int res) {
    const int b = blockIdx.z;
    const int x0 = crops[4 * b + 0], y0 = crops[4 * b + 1], x1 = crops[4 * b + 2], y1 = crops[4 * b + 3];
    const int cw = x1 - x0, ch = y1 - y0;
    const int ox = blockIdx.x * 16 + (threadIdx.x & 15), oy = blockIdx.y * 16 + (threadIdx.x >> 4);
    if (ox >= res || oy >= res) return;
    const bool no_stats = channels < 0;              // channels == -1: one grey plane that already carries the model's normalisation
    if (no_stats) channels = 1;
    const AxisTaps tx = axis_taps(ox, cw, res), ty = axis_taps(oy, ch, res);
    float wxs = 0.f, wys = 0.f;
    for (int j = 0; j < tx.n; ++j) wxs += tri(((float)(j + tx.lo) - tx.center + 0.5f) * tx.invscale);
    for (int j = 0; j < ty.n; ++j) wys += tri(((float)(j + ty.lo) - ty.center + 0.5f) * ty.invscale);
    const float mean[3] = {0.485f, 0.456f, 0.406f}, stdv[3] = {0.229f, 0.224f, 0.225f};
    float acc[3] = {0.f, 0.f, 0.f};
    // ATen resizes horizontally first, then vertically (separable, fp32 intermediates)
    for (int jy = 0; jy < ty.n; ++jy) {
        const float wy = tri(((float)(jy + ty.lo) - ty.center + 0.5f) * ty.invscale) / wys;
        float row[3] = {0.f, 0.f, 0.f};
        for (int jx = 0; jx < tx.n; ++jx) {
            const float wx = tri(((float)(jx + tx.lo) - tx.center + 0.5f) * tx.invscale) / wxs;
            const int64_t pix = (int64_t)(y0 + ty.lo + jy) * W + (x0 + tx.lo + jx);
            if (channels == 1) {
                const float v = img[pix];
                row[0] += wx * v;
            } else {
#pragma unroll
                for (int c = 0; c < 3; ++c) row[c] += wx * img[pix * 3 + c];
            }
        }
        if (channels == 1) acc[0] += wy * row[0];
        else {
#pragma unroll
            for (int c = 0; c < 3; ++c) acc[c] += wy * row[c];
        }
    }
    const int64_t plane = (int64_t)res * res;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float v = channels == 1 ? acc[0] : acc[c];
        out[((int64_t)b * 3 + c) * plane + (int64_t)oy * res + ox] = no_stats ? v : (v - mean[c]) / stdv[c];
    }
}

const char* launch_resize_normalize(const float* img, int H, int W, int channels, const int* crops_dev, int n, float* out, int res,
                                    hipStream_t s) {
    if (channels != 1 && channels != 3 && channels != -1) return "resize_normalize: channels must be 1, 3 or -1 (grey, no statistics)";
    hipLaunchKernelGGL(resize_normalize_kernel, dim3(res / 16, res / 16, n), dim3(256), 0, s, img, H, W, channels, crops_dev, out, res);
    return nullptr;
}

// ------------------------------------------------------------------------------------------------ K1b
// One block per 8x8 window of output tokens (= 64 consecutive rows in the engine's token order).
// Lane = token, wave w = channel group [w*CPW, (w+1)*CPW); weights are wave-uniform (scalar loads),
// the 3x35x35 input patch sits in LDS.  fp32 VALU: this conv is 0.15 % of the encoder's FLOPs.
template <int CPW>
__global__ __launch_bounds__(256) void patch_embed_kernel(const float* __restrict__ pix, const float* __restrict__ wt,
                                                          const float* __restrict__ bias, const float* __restrict__ pos,
                                                          float* __restrict__ out, int res) {
    constexpr int C = 4 * CPW;
    __shared__ float patch[3][35][36];
    const int b = blockIdx.y, wi = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int wy, wx;
    perm_coords256(wi * 64, &wy, &wx);  // token coords of the window origin (multiples of 8)
    const int iy0 = wy * 4 - 3, ix0 = wx * 4 - 3;
    const int64_t plane = (int64_t)res * res;
    for (int idx = tid; idx < 3 * 35 * 35; idx += 256) {
        const int c = idx / 1225, r = idx - c * 1225;
        const int py = r / 35, px = r - py * 35;
        const int iy = iy0 + py, ix = ix0 + px;
        float v = 0.f;
        if (iy >= 0 && iy < res && ix >= 0 && ix < res) v = pix[((int64_t)b * 3 + c) * plane + (int64_t)iy * res + ix];
        patch[c][py][px] = v;
    }
    __syncthreads();
    const int ty = (((lane >> 5) & 1) << 2) | (((lane >> 3) & 1) << 1) | ((lane >> 1) & 1);
    const int tx = (((lane >> 4) & 1) << 2) | (((lane >> 2) & 1) << 1) | (lane & 1);
    float acc[CPW];
#pragma unroll
    for (int j = 0; j < CPW; ++j) acc[j] = 0.f;
    const float* wbase = wt + wave * CPW;
    for (int c = 0; c < 3; ++c)
        for (int ky = 0; ky < 7; ++ky)
#pragma unroll
            for (int kx = 0; kx < 7; ++kx) {
                const float v = patch[c][ty * 4 + ky][tx * 4 + kx];
                const float* wp = wbase + ((c * 7 + ky) * 7 + kx) * C;
#pragma unroll
                for (int j = 0; j < CPW; ++j) acc[j] = fmaf(v, wp[j], acc[j]);
            }
    const int64_t row = (int64_t)wi * 64 + lane;
    float* op = out + ((int64_t)b * 65536 + row) * C + wave * CPW;
    const float* pp = pos + row * C + wave * CPW;
    const float* bp = bias + wave * CPW;
#pragma unroll
    for (int j = 0; j < CPW; j += 4) {
        const float4 p4 = *reinterpret_cast<const float4*>(pp + j);
        const float4 b4 = *reinterpret_cast<const float4*>(bp + j);
        *reinterpret_cast<float4*>(op + j) =
            make_float4(acc[j] + b4.x + p4.x, acc[j + 1] + b4.y + p4.y, acc[j + 2] + b4.z + p4.z, acc[j + 3] + b4.w + p4.w);
    }
}

const char* launch_patch_embed(const float* pix, const float* wt, const float* bias, const float* pos, float* out, int n_images,
                               int C, int res, hipStream_t s) {
    if (res != 1024) return "patch_embed: only 1024x1024 model input is supported";
    const dim3 grid(1024, n_images);
    if (C == 144) hipLaunchKernelGGL(patch_embed_kernel<36>, grid, dim3(256), 0, s, pix, wt, bias, pos, out, res);
    else if (C == 112) hipLaunchKernelGGL(patch_embed_kernel<28>, grid, dim3(256), 0, s, pix, wt, bias, pos, out, res);
    else if (C == 96) hipLaunchKernelGGL(patch_embed_kernel<24>, grid, dim3(256), 0, s, pix, wt, bias, pos, out, res);
    else return "patch_embed: unsupported embed dim";
    return nullptr;
}

// ------------------------------------------------------------------------------------------------ non-finite sentinel
// Counts the values of an fp32 array that are NaN or +-inf into one device counter (the engine's overflow sentinel: with fp16 operands a
// stored activation beyond 65 504 becomes inf, and every consumer downstream of it - fp32 accumulators, the fp32 residual stream,
// LayerNorm and softmax statistics - carries the inf / NaN on to the tensors scanned here; engine.hip "sentinel").  HBM-bound, one pass.
__global__ __launch_bounds__(256) void nonfinite_scan_kernel(const float* __restrict__ p, int64_t n4, int64_t n, unsigned int* __restrict__ counter) {
    unsigned int bad = 0;
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
        typedef float f32x4v __attribute__((ext_vector_type(4)));
        const f32x4v v = __builtin_nontemporal_load(reinterpret_cast<const f32x4v*>(p) + i);
        // x * 0 is 0 for every finite x and NaN for NaN / inf: one fma chain per vector, one compare
        const float z = fmaf(v[0], 0.f, fmaf(v[1], 0.f, fmaf(v[2], 0.f, v[3] * 0.f)));
        bad += (z != z) ? 1u : 0u;
    }
    if (blockIdx.x == 0 && threadIdx.x < (int)(n - 4 * n4)) { const float x = p[4 * n4 + threadIdx.x] * 0.f; bad += (x != x) ? 1u : 0u; }
    const unsigned long long m = __ballot(bad != 0);
    if (m && (threadIdx.x & 63) == __ffsll((long long)m) - 1) atomicAdd(counter, (unsigned int)__popcll(m));
}

const char* launch_nonfinite_scan(const float* p, int64_t n, unsigned int* counter, hipStream_t s) {
    if (n <= 0) return nullptr;
    if (!p || !counter || (reinterpret_cast<uintptr_t>(p) & 15)) return "nonfinite_scan: null or unaligned pointer";
    const int64_t n4 = n / 4;
    const int64_t blocks = (n4 + 256 * 8 - 1) / (256 * 8);
    const int grid = (int)(blocks < 1 ? 1 : blocks > 2048 ? 2048 : blocks);
    hipLaunchKernelGGL(nonfinite_scan_kernel, dim3(grid), dim3(256), 0, s, p, n4, n, counter);
    return nullptr;
}
