// Kernels of the SAM2 video (memory) path, SURVEY.md 8f-1: what `SAM2Adapter.segment_volume` (saber/adapters/sam2/predictor.py:232-348)
// makes the third-party video predictor execute per tracked frame besides the image encoder and the mask decoder:
//   memory attention   axial RoPE on q / k (rope_kernel), row softmax of the score matrix (softmax_rows_kernel); the projections, the
//                      score and value products and the MLP are the engine's bf16 MFMA GEMM (gemm.hip)
//   memory encoder     mask down-sampler 3x3 stride-2 convolutions (conv3x3s2_kernel), the 7x7 depth-wise convolution of the fuser
//                      (dwconv7_kernel), layer scale + residual (axpy_kernel); LayerNorm2d / GELU / 1x1 convolutions are layernorm.hip / gemm.hip
//   mask plumbing      bilinear / antialiased resize with the fused "mask for memory" transform (resize_plane_kernel: ATen's
//                      upsample_bilinear2d rule, antialias = the triangle filter of _upsample_bilinear2d_aa), the 4x4 stride-4
//                      `mask_downsample` convolution (conv4x4s4_kernel)
// All tensors are channels-last ([pixels][C] fp32) with pixels in ROW-MAJOR (y, x) order: the spatial operators need neighbours, and
// the engine's bit-interleaved token order is entered / left with a row gather (saber_get_embed_tokens / saber_set_embed_tokens).
// First-correct kernels (one thread per output element, fp32 FMA): the sequential memory chain is a 'next' row whose cost is the
// per-frame Hiera encode (sharded over the ranks) - these kernels are not tuned.
#include <algorithm>

#include "common.h"
#include "kernels.h"

// ------------------------------------------------------------------------------------------------ RoPE
// x[row][2i], x[row][2i+1] <- (a cos - b sin, b cos + a sin); token = row % tokens_per_frame on a side x side grid; channel pairs
// i < C/4 rotate with the x coordinate, the rest with y; frequency = theta^(-4 (i mod C/4) / C).  Rows >= n_rot are copied.
__global__ __launch_bounds__(256) void rope_kernel(const float* __restrict__ x, int64_t rows, int n_rot, int C, int side, float theta,
                                                   float* __restrict__ out_f, bf16_t* __restrict__ out_bf) {
    const int half = C >> 1, quarter = C >> 2;
    const int64_t total = rows * half;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        const int64_t row = idx / half;
        const int i = (int)(idx - row * half);
        float a = x[row * C + 2 * i], b = x[row * C + 2 * i + 1];
        if (row < n_rot) {
            const int tok = (int)(row % ((int64_t)side * side));
            const int coord = i < quarter ? tok % side : tok / side;
            const int fi = i < quarter ? i : i - quarter;
            const float freq = 1.0f / powf(theta, (float)(4 * fi) / (float)C);
            float sn, cs;
            sincosf((float)coord * freq, &sn, &cs);
            const float ra = a * cs - b * sn, rb = b * cs + a * sn;
            a = ra; b = rb;
        }
        if (out_f) { out_f[row * C + 2 * i] = a; out_f[row * C + 2 * i + 1] = b; }
        if (out_bf) *reinterpret_cast<uint32_t*>(out_bf + row * C + 2 * i) = pack_op16(a, b);
    }
}
const char* launch_rope(const float* x, int64_t rows, int n_rot, int C, int side, float theta, float* out_f, bf16_t* out_bf, hipStream_t s) {
    if (rows <= 0) return nullptr;
    if (C & 3) return "rope: C must be a multiple of 4";
    const int64_t total = rows * (C >> 1);
    hipLaunchKernelGGL(rope_kernel, dim3((unsigned)std::min<int64_t>((total + 255) / 256, 8192)), dim3(256), 0, s, x, rows, n_rot, C, side, theta, out_f, out_bf);
    return nullptr;
}

// ------------------------------------------------------------------------------------------------ row softmax
// P[row][0..n) = softmax(scale * S[row][0..n)) as bf16, columns n..ldp are written as zeros (key padding for the PV product)
__global__ __launch_bounds__(256) void softmax_rows_kernel(const float* __restrict__ S, int64_t lds_, int n, float scale, bf16_t* __restrict__ P, int64_t ldp) {
    __shared__ float red[4];
    const int64_t row = blockIdx.x;
    const float* s = S + row * lds_;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float mx = -3.0e38f;
    for (int c = tid; c < n; c += 256) mx = fmaxf(mx, s[c]);
    mx = wave_max(mx);
    if (lane == 0) red[wave] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    float sum = 0.f;
    for (int c = tid; c < n; c += 256) sum += __expf((s[c] - mx) * scale);
    sum = wave_sum(sum);
    if (lane == 0) red[wave] = sum;
    __syncthreads();
    const float inv = 1.0f / ((red[0] + red[1]) + (red[2] + red[3]));
    bf16_t* p = P + row * ldp;
    for (int c = tid; c < ldp; c += 256) p[c] = c < n ? f2op(__expf((s[c] - mx) * scale) * inv) : (bf16_t)0;
}
const char* launch_softmax_rows(const float* S, int64_t lds_, int64_t rows, int n, float scale, bf16_t* P, int64_t ldp, hipStream_t s) {
    if (rows <= 0 || n <= 0) return nullptr;
    if (ldp < n) return "softmax_rows: ldp < n";
    hipLaunchKernelGGL(softmax_rows_kernel, dim3((unsigned)rows), dim3(256), 0, s, S, lds_, n, scale, P, ldp);
    return nullptr;
}

// ------------------------------------------------------------------------------------------------ convolutions (channels-last)
// out[(oy, ox)][co] = b[co] + sum_{ky,kx,ci} w[co][ci][ky][kx] in[(2 oy + ky - 1, 2 ox + kx - 1)][ci]   (3x3, stride 2, padding 1)
__global__ __launch_bounds__(256) void conv3x3s2_kernel(const float* __restrict__ in, int H, int W, int Cin, const float* __restrict__ w,
                                                        const float* __restrict__ b, int Cout, float* __restrict__ out) {
    const int Ho = H >> 1, Wo = W >> 1;
    const int64_t total = (int64_t)Ho * Wo * Cout;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        const int co = (int)(idx % Cout);
        const int64_t pix = idx / Cout;
        const int ox = (int)(pix % Wo), oy = (int)(pix / Wo);
        float acc = b[co];
        for (int ky = 0; ky < 3; ++ky) {
            const int iy = 2 * oy + ky - 1;
            if (iy < 0 || iy >= H) continue;
            for (int kx = 0; kx < 3; ++kx) {
                const int ix = 2 * ox + kx - 1;
                if (ix < 0 || ix >= W) continue;
                const float* ip = in + ((int64_t)iy * W + ix) * Cin;
                const float* wp = w + (int64_t)co * Cin * 9 + ky * 3 + kx;
                for (int ci = 0; ci < Cin; ++ci) acc = fmaf(wp[ci * 9], ip[ci], acc);
            }
        }
        out[idx] = acc;
    }
}
// the same convolution with the weights laid out [ky][kx][ci][co] (prepared once per model): four consecutive output channels per
// thread, so a wave's weight loads are whole lines and the input value is a broadcast (the (co, ci, ky, kx) layout reads one word per
// 36-byte stride: 2.5 ms per tracked frame on the four layers of the mask down-sampler)
__global__ __launch_bounds__(256) void conv3x3s2_t_kernel(const float* __restrict__ in, int H, int W, int Cin, const float* __restrict__ wt,
                                                          const float* __restrict__ b, int Cout, float* __restrict__ out) {
    const int Ho = H >> 1, Wo = W >> 1, c4n = Cout >> 2;
    const int64_t total = (int64_t)Ho * Wo * c4n;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        const int co = (int)(idx % c4n) * 4;
        const int64_t pix = idx / c4n;
        const int ox = (int)(pix % Wo), oy = (int)(pix / Wo);
        float4 acc = *reinterpret_cast<const float4*>(b + co);
        for (int ky = 0; ky < 3; ++ky) {
            const int iy = 2 * oy + ky - 1;
            if (iy < 0 || iy >= H) continue;
            for (int kx = 0; kx < 3; ++kx) {
                const int ix = 2 * ox + kx - 1;
                if (ix < 0 || ix >= W) continue;
                const float* ip = in + ((int64_t)iy * W + ix) * Cin;
                const float* wp = wt + (int64_t)(ky * 3 + kx) * Cin * Cout + co;
                for (int ci = 0; ci < Cin; ++ci) {
                    const float v = ip[ci];
                    const float4 w4 = *reinterpret_cast<const float4*>(wp + (int64_t)ci * Cout);
                    acc.x = fmaf(w4.x, v, acc.x); acc.y = fmaf(w4.y, v, acc.y); acc.z = fmaf(w4.z, v, acc.z); acc.w = fmaf(w4.w, v, acc.w);
                }
            }
        }
        *reinterpret_cast<float4*>(out + pix * Cout + co) = acc;
    }
}
const char* launch_conv3x3s2_t(const float* in, int H, int W, int Cin, const float* wt, const float* b, int Cout, float* out, hipStream_t s) {
    if ((H | W) & 1) return "conv3x3s2: H and W must be even";
    if (Cout & 3) return "conv3x3s2_t: Cout must be a multiple of 4";
    const int64_t total = (int64_t)(H / 2) * (W / 2) * (Cout / 4);
    hipLaunchKernelGGL(conv3x3s2_t_kernel, dim3((unsigned)std::min<int64_t>((total + 255) / 256, 65536)), dim3(256), 0, s, in, H, W, Cin, wt, b, Cout, out);
    return nullptr;
}
// depth-wise 7x7, padding 3: out[(y,x)][c] = b[c] + sum w[c][ky][kx] in[(y+ky-3, x+kx-3)][c]
__global__ __launch_bounds__(256) void dwconv7_kernel(const float* __restrict__ in, int H, int W, int C, const float* __restrict__ w,
                                                      const float* __restrict__ b, float* __restrict__ out) {
    const int64_t total = (int64_t)H * W * C;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        const int c = (int)(idx % C);
        const int64_t pix = idx / C;
        const int x = (int)(pix % W), y = (int)(pix / W);
        float acc = b[c];
        for (int ky = 0; ky < 7; ++ky) {
            const int iy = y + ky - 3;
            if (iy < 0 || iy >= H) continue;
            for (int kx = 0; kx < 7; ++kx) {
                const int ix = x + kx - 3;
                if (ix < 0 || ix >= W) continue;
                acc = fmaf(w[c * 49 + ky * 7 + kx], in[((int64_t)iy * W + ix) * C + c], acc);
            }
        }
        out[idx] = acc;
    }
}
// the same with the weights laid out [49][C] (coalesced across the channel lanes) and 4 channels per thread: the [C][49] form reads a
// different cache line per lane and tap (115 us for the fuser's 4096 x 256 outputs; this form: ~12 us)
__global__ __launch_bounds__(256) void dwconv7_t_kernel(const float* __restrict__ in, int H, int W, int C, const float* __restrict__ wt,
                                                        const float* __restrict__ b, float* __restrict__ out) {
    const int c4n = C >> 2;
    const int64_t total = (int64_t)H * W * c4n;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        const int c = (int)(idx % c4n) * 4;
        const int64_t pix = idx / c4n;
        const int x = (int)(pix % W), y = (int)(pix / W);
        float4 acc = *reinterpret_cast<const float4*>(b + c);
#pragma unroll
        for (int ky = 0; ky < 7; ++ky) {
            const int iy = y + ky - 3;
            if (iy < 0 || iy >= H) continue;
#pragma unroll
            for (int kx = 0; kx < 7; ++kx) {
                const int ix = x + kx - 3;
                if (ix < 0 || ix >= W) continue;
                const float4 v = *reinterpret_cast<const float4*>(in + ((int64_t)iy * W + ix) * C + c);
                const float4 w4 = *reinterpret_cast<const float4*>(wt + (int64_t)(ky * 7 + kx) * C + c);
                acc.x = fmaf(w4.x, v.x, acc.x); acc.y = fmaf(w4.y, v.y, acc.y); acc.z = fmaf(w4.z, v.z, acc.z); acc.w = fmaf(w4.w, v.w, acc.w);
            }
        }
        *reinterpret_cast<float4*>(out + pix * C + c) = acc;
    }
}
// single channel 4x4 stride 4 (the video predictor's `mask_downsample`)
__global__ __launch_bounds__(256) void conv4x4s4_kernel(const float* __restrict__ in, int H, int W, const float* __restrict__ w, const float* __restrict__ b,
                                                        float* __restrict__ out) {
    const int Ho = H >> 2, Wo = W >> 2;
    const int64_t total = (int64_t)Ho * Wo;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        const int ox = (int)(idx % Wo), oy = (int)(idx / Wo);
        float acc = b[0];
        for (int ky = 0; ky < 4; ++ky)
            for (int kx = 0; kx < 4; ++kx) acc = fmaf(w[ky * 4 + kx], in[(int64_t)(4 * oy + ky) * W + 4 * ox + kx], acc);
        out[idx] = acc;
    }
}
static unsigned grid_for(int64_t total) { return (unsigned)std::min<int64_t>((total + 255) / 256, 16384); }
const char* launch_conv3x3s2(const float* in, int H, int W, int Cin, const float* w, const float* b, int Cout, float* out, hipStream_t s) {
    if ((H & 1) || (W & 1) || H <= 0 || W <= 0) return "conv3x3s2: H and W must be even";
    hipLaunchKernelGGL(conv3x3s2_kernel, dim3(grid_for((int64_t)(H / 2) * (W / 2) * Cout)), dim3(256), 0, s, in, H, W, Cin, w, b, Cout, out);
    return nullptr;
}
const char* launch_dwconv7(const float* in, int H, int W, int C, const float* w, const float* b, float* out, hipStream_t s) {
    hipLaunchKernelGGL(dwconv7_kernel, dim3(grid_for((int64_t)H * W * C)), dim3(256), 0, s, in, H, W, C, w, b, out);
    return nullptr;
}
const char* launch_dwconv7_t(const float* in, int H, int W, int C, const float* wt, const float* b, float* out, hipStream_t s) {
    if (C & 3) return "dwconv7_t: C must be a multiple of 4";
    hipLaunchKernelGGL(dwconv7_t_kernel, dim3(grid_for((int64_t)H * W * (C / 4))), dim3(256), 0, s, in, H, W, C, wt, b, out);
    return nullptr;
}
const char* launch_conv4x4s4(const float* in, int H, int W, const float* w, const float* b, float* out, hipStream_t s) {
    if ((H & 3) || (W & 3)) return "conv4x4s4: H and W must be multiples of 4";
    hipLaunchKernelGGL(conv4x4s4_kernel, dim3(grid_for((int64_t)(H / 4) * (W / 4))), dim3(256), 0, s, in, H, W, w, b, out);
    return nullptr;
}

// ------------------------------------------------------------------------------------------------ plane resize
// ATen's separable rule (same index / weight arithmetic as resize_normalize_kernel in image_ops.hip): antialias = 0 is
// F.interpolate(mode="bilinear", align_corners=False), antialias = 1 adds the triangle filter of the down-sampling ratio.
// post: 0 none | 1 y = a * sigmoid(v) + c | 2 y = a * (v > 0) + c | 3 y = a * v + c | 4 y = (v >= a)
struct RTaps { int lo; int n; float center; float invscale; };
__device__ __forceinline__ RTaps rtaps(int o, int in_size, int out_size, int antialias) {
    RTaps t;
    const float scale = (float)in_size / (float)out_size;
    const float support = (antialias && scale >= 1.0f) ? scale : 1.0f;
    t.invscale = (antialias && scale >= 1.0f) ? 1.0f / scale : 1.0f;
    t.center = scale * ((float)o + 0.5f);
    t.lo = max((int)(t.center - support + 0.5f), 0);
    t.n = min((int)(t.center + support + 0.5f), in_size) - t.lo;
    return t;
}
__device__ __forceinline__ float rtri(float x) { x = fabsf(x); return x < 1.0f ? 1.0f - x : 0.0f; }
__global__ __launch_bounds__(256) void resize_plane_kernel(const float* __restrict__ in, int H, int W, float* __restrict__ out, int Ho, int Wo,
                                                           int antialias, int post, float a, float c, int64_t in_stride, int64_t out_stride) {
    const float* ip = in + (int64_t)blockIdx.y * in_stride;
    float* op = out + (int64_t)blockIdx.y * out_stride;
    const int64_t total = (int64_t)Ho * Wo;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        const int ox = (int)(idx % Wo), oy = (int)(idx / Wo);
        const RTaps tx = rtaps(ox, W, Wo, antialias), ty = rtaps(oy, H, Ho, antialias);
        float wxs = 0.f, wys = 0.f;
        for (int j = 0; j < tx.n; ++j) wxs += rtri(((float)(j + tx.lo) - tx.center + 0.5f) * tx.invscale);
        for (int j = 0; j < ty.n; ++j) wys += rtri(((float)(j + ty.lo) - ty.center + 0.5f) * ty.invscale);
        float acc = 0.f;
        for (int jy = 0; jy < ty.n; ++jy) {
            const float wy = rtri(((float)(jy + ty.lo) - ty.center + 0.5f) * ty.invscale) / wys;
            float row = 0.f;
            for (int jx = 0; jx < tx.n; ++jx)
                row += (rtri(((float)(jx + tx.lo) - tx.center + 0.5f) * tx.invscale) / wxs) * ip[(int64_t)(ty.lo + jy) * W + tx.lo + jx];
            acc += wy * row;
        }
        if (post == 1) acc = a / (1.0f + __expf(-acc)) + c;
        else if (post == 2) acc = (acc > 0.f ? a : 0.f) + c;
        else if (post == 3) acc = a * acc + c;
        else if (post == 4) acc = acc >= a ? 1.0f : 0.0f;
        op[idx] = acc;
    }
}
const char* launch_resize_plane(const float* in, int n_planes, int H, int W, float* out, int Ho, int Wo, int antialias, int post, float a, float c, hipStream_t s) {
    if (n_planes <= 0) return nullptr;
    if (H <= 0 || W <= 0 || Ho <= 0 || Wo <= 0) return "resize_plane: empty plane";
    hipLaunchKernelGGL(resize_plane_kernel, dim3(grid_for((int64_t)Ho * Wo), n_planes), dim3(256), 0, s, in, H, W, out, Ho, Wo, antialias, post, a, c,
                       (int64_t)H * W, (int64_t)Ho * Wo);
    return nullptr;
}

// ------------------------------------------------------------------------------------------------ elementwise
// out[row][c] = x[row][c] + alpha * g[c] * y[row][c]   (g may be NULL = 1; x may be NULL = 0)
__global__ __launch_bounds__(256) void axpy_kernel(const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ g, float alpha,
                                                   int64_t rows, int C, float* __restrict__ out) {
    const int64_t total = rows * C;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        const int c = (int)(idx % C);
        out[idx] = (x ? x[idx] : 0.f) + alpha * (g ? g[c] : 1.0f) * y[idx];
    }
}
const char* launch_axpy(const float* x, const float* y, const float* g, float alpha, int64_t rows, int C, float* out, hipStream_t s) {
    if (rows <= 0) return nullptr;
    hipLaunchKernelGGL(axpy_kernel, dim3(grid_for(rows * C)), dim3(256), 0, s, x, y, g, alpha, rows, C, out);
    return nullptr;
}

__global__ __launch_bounds__(256) void bf16_to_f32_kernel(const bf16_t* __restrict__ x, int64_t n, float* __restrict__ out) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) out[i] = op2f(x[i]);
}
const char* launch_bf16_to_f32(const bf16_t* x, int64_t n, float* out, hipStream_t s) {
    if (n <= 0) return nullptr;
    hipLaunchKernelGGL(bf16_to_f32_kernel, dim3(grid_for(n)), dim3(256), 0, s, x, n, out);
    return nullptr;
}

// ------------------------------------------------------------------------------------------------ painting a tracked mask into the label volume
// plane[y][x] = label where logits[ys][xs] > thr, (ys, xs) = nearest source pixel of the output pixel centre (skimage resize order 0:
// floor((i + 0.5) * in / out)); other pixels keep their value.  SAM2Adapter.segment_volume's _apply (predictor.py:288-298) per object.
__global__ __launch_bounds__(256) void paint_nearest_kernel(const float* __restrict__ logits, int Hv, int Wv, float thr, int label,
                                                            uint16_t* __restrict__ plane, int H, int W, int* __restrict__ any_flag) {
    const int64_t total = (int64_t)H * W;
    bool hit = false;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        const int y = (int)(idx / W), x = (int)(idx - (int64_t)y * W);
        const int ys = min(max((int)floor(((double)y + 0.5) * Hv / H), 0), Hv - 1);
        const int xs = min(max((int)floor(((double)x + 0.5) * Wv / W), 0), Wv - 1);
        if (logits[(int64_t)ys * Wv + xs] > thr) { plane[idx] = (uint16_t)label; hit = true; }
    }
    if (any_flag && __any(hit) && (threadIdx.x & 63) == 0) atomicOr(any_flag, 1);
}
const char* launch_paint_nearest(const float* logits, int Hv, int Wv, float thr, int label, uint16_t* plane, int H, int W, int* any_flag, hipStream_t s) {
    if (Hv <= 0 || Wv <= 0 || H <= 0 || W <= 0) return "paint_nearest: bad shape";
    if (label < 0 || label > 65535) return "paint_nearest: label does not fit uint16";
    const int64_t total = (int64_t)H * W;
    hipLaunchKernelGGL(paint_nearest_kernel, dim3((unsigned)std::min<int64_t>((total + 255) / 256, 4096)), dim3(256), 0, s, logits, Hv, Wv, thr, label, plane, H, W, any_flag);
    return nullptr;
}

// ------------------------------------------------------------------------------------------------ Gaussian anti-aliasing filter
// One axis of scipy.ndimage.gaussian_filter(mode="mirror") on n planes of H x W: out[y][x] = sum_k w[k] in[mirror(y + k - r)][x] (axis 0)
// or along x (axis 1); mirror = reflection about the centre of the edge pixel (index -1 -> 1).  skimage.transform.resize(anti_aliasing=True)
// applies it with sigma = (factor - 1) / 2 before a down-sampling interpolation (saber/adapters/preprocessing.py:21).
#define GM_MAX_R 64
struct GaussTaps { float w[2 * GM_MAX_R + 1]; };
__global__ __launch_bounds__(256) void gauss_mirror_kernel(const float* __restrict__ in, float* __restrict__ out, int n_planes, int H, int W, int axis, int r, GaussTaps taps) {
    const int64_t total = (int64_t)n_planes * H * W;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int x = (int)(i % W);
        const int y = (int)((i / W) % H);
        const float* pl = in + (i / ((int64_t)H * W)) * H * W;
        const int n = axis == 0 ? H : W, c = axis == 0 ? y : x;
        float acc = 0.f;
        for (int k = -r; k <= r; ++k) {
            int j = c + k;
            // mirror (period 2 n - 2); n == 1: the only pixel
            if (n == 1) j = 0;
            else {
                const int per = 2 * n - 2;
                j %= per; if (j < 0) j += per;
                if (j >= n) j = per - j;
            }
            acc = fmaf(taps.w[k + r], axis == 0 ? pl[(int64_t)j * W + x] : pl[(int64_t)y * W + j], acc);
        }
        out[i] = acc;
    }
}
const char* launch_gauss_mirror(const float* in, float* out, int n_planes, int H, int W, int axis, double sigma, hipStream_t s) {
    if (n_planes <= 0 || H <= 0 || W <= 0 || (axis != 0 && axis != 1) || !(sigma > 0.0)) return "gauss_mirror: bad argument";
    const int r = (int)(4.0 * sigma + 0.5);                  // scipy: truncate = 4.0
    if (r > GM_MAX_R) return "gauss_mirror: sigma too large (radius > 64)";
    GaussTaps t;
    double sum = 0.0;
    for (int k = -r; k <= r; ++k) sum += exp(-0.5 * k * k / (sigma * sigma));
    for (int k = -r; k <= r; ++k) t.w[k + r] = (float)(exp(-0.5 * k * k / (sigma * sigma)) / sum);
    const int64_t total = (int64_t)n_planes * H * W;
    hipLaunchKernelGGL(gauss_mirror_kernel, dim3((unsigned)std::min<int64_t>((total + 255) / 256, 65536 * 4)), dim3(256), 0, s, in, out, n_planes, H, W, axis, r, t);
    return nullptr;
}
