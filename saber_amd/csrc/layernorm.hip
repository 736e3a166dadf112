// Row LayerNorm (one wave per row) and small elementwise helpers.
// Replaces nn.LayerNorm / LayerNorm2d calls of the sam2 image model.
// fp32 statistics (two-pass in registers), outputs: optional fp32, optional bf16,
// optional bf16 of (y + addvec[row % add_mod]) used for "keys + image_pe".
#include "common.h"
#include "kernels.h"

#define LN_MAX_CHUNKS 5  // 5 * 64 lanes * 4 floats = 1280 channels

__global__ __launch_bounds__(256) void layernorm_kernel(LayerNormParams p) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= p.rows) return;
    const float* x = p.x + (int64_t)row * p.ldx;
    const int nvec = p.C >> 2;
    float4 v[LN_MAX_CHUNKS];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAX_CHUNKS; ++i) {
        const int c = lane + 64 * i;
        if (c < nvec) {
            v[i] = *reinterpret_cast<const float4*>(x + 4 * c);
            s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
        } else v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    const float mean = wave_sum(s) / (float)p.C;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAX_CHUNKS; ++i) {
        const int c = lane + 64 * i;
        if (c < nvec) {
            const float a = v[i].x - mean, b = v[i].y - mean, cc = v[i].z - mean, d = v[i].w - mean;
            q += (a * a + b * b) + (cc * cc + d * d);
        }
    }
    const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)p.C + p.eps);
    const float* addv = p.addvec ? p.addvec + (int64_t)(row % p.add_mod) * p.C : nullptr;
    // window-padding rows (tiny/small/base+ trunks): the reference pads the NORMALISED tokens with zeros
    const bool zero_row = p.row_valid && !p.row_valid[row % p.valid_mod];
#pragma unroll
    for (int i = 0; i < LN_MAX_CHUNKS; ++i) {
        const int c = lane + 64 * i;
        if (c < nvec) {
            const float4 g = *reinterpret_cast<const float4*>(p.gamma + 4 * c);
            const float4 b = *reinterpret_cast<const float4*>(p.beta + 4 * c);
            float y0 = (v[i].x - mean) * rstd * g.x + b.x;
            float y1 = (v[i].y - mean) * rstd * g.y + b.y;
            float y2 = (v[i].z - mean) * rstd * g.z + b.z;
            float y3 = (v[i].w - mean) * rstd * g.w + b.w;
            if (p.act == ACT_GELU) { y0 = gelu_erf(y0); y1 = gelu_erf(y1); y2 = gelu_erf(y2); y3 = gelu_erf(y3); }
            if (zero_row) { y0 = 0.f; y1 = 0.f; y2 = 0.f; y3 = 0.f; }
            if (p.out_f) *reinterpret_cast<float4*>(p.out_f + (int64_t)row * p.ldo + 4 * c) = make_float4(y0, y1, y2, y3);
            if (p.out_bf)
                *reinterpret_cast<uint2*>(p.out_bf + (int64_t)row * p.ldo + 4 * c) = make_uint2(pack_bf16(y0, y1), pack_bf16(y2, y3));
            if (p.out_bf_add) {
                const float4 a = *reinterpret_cast<const float4*>(addv + 4 * c);
                *reinterpret_cast<uint2*>(p.out_bf_add + (int64_t)row * p.ldo + 4 * c) =
                    make_uint2(pack_bf16(y0 + a.x, y1 + a.y), pack_bf16(y2 + a.z, y3 + a.w));
            }
        }
    }
}

const char* launch_layernorm(const LayerNormParams& p, hipStream_t s) {
    if (p.rows <= 0) return nullptr;
    if ((p.C & 3) || p.C > LN_MAX_CHUNKS * 256) return "layernorm: C must be a multiple of 4 and <= 1280";
    if ((p.ldx & 3) || (p.ldo & 3)) return "layernorm: strides must be multiples of 4";
    if (p.out_bf_add && (!p.addvec || p.add_mod <= 0)) return "layernorm: addvec missing";
    if (p.row_valid && p.valid_mod <= 0) return "layernorm: valid_mod";
    hipLaunchKernelGGL(layernorm_kernel, dim3((p.rows + 3) / 4), dim3(256), 0, s, p);
    return nullptr;
}

// out_bf[r][c] = bf16(x[r][c] + y[(r % ymod)][c]); optional fp32 copy of the sum.
__global__ __launch_bounds__(256) void add_to_bf16_kernel(const float* __restrict__ x, const float* __restrict__ y, int ymod,
                                                        bf16_t* __restrict__ out_bf, float* __restrict__ out_f, int64_t rows, int C) {
    const int nvec = C >> 2;
    const int64_t total = rows * nvec;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = idx / nvec;
        const int c = (int)(idx - r * nvec);
        float4 a = *reinterpret_cast<const float4*>(x + r * C + 4 * c);
        if (y) {
            const float4 b = *reinterpret_cast<const float4*>(y + (r % ymod) * C + 4 * c);
            a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
        }
        if (out_bf) *reinterpret_cast<uint2*>(out_bf + r * C + 4 * c) = make_uint2(pack_bf16(a.x, a.y), pack_bf16(a.z, a.w));
        if (out_f) *reinterpret_cast<float4*>(out_f + r * C + 4 * c) = a;
    }
}

__global__ __launch_bounds__(256) void gather_rows_kernel(const float* __restrict__ in, int64_t in_rows, float* __restrict__ out, int64_t out_rows,
                                                          const int* __restrict__ idx, int C, int64_t total) {
    const int nvec = C >> 2;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / nvec;
        const int c = (int)(i - r * nvec);
        const int64_t img = r / out_rows;
        const int src = idx[r - img * out_rows];
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (src >= 0) v = *reinterpret_cast<const float4*>(in + (img * in_rows + src) * C + 4 * c);
        *reinterpret_cast<float4*>(out + r * C + 4 * c) = v;
    }
}

const char* launch_gather_rows(const float* in, int64_t in_rows, float* out, int64_t out_rows, const int* idx, int C, int n_images, hipStream_t s) {
    if (n_images <= 0 || out_rows <= 0) return nullptr;
    if (C & 3) return "gather_rows: C must be a multiple of 4";
    const int64_t total = (int64_t)n_images * out_rows * (C >> 2);
    int blocks = (int)((total + 255) / 256);
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(gather_rows_kernel, dim3(blocks), dim3(256), 0, s, in, in_rows, out, out_rows, idx, C, total);
    return nullptr;
}

const char* launch_add_to_bf16(const float* x, const float* y, int ymod, bf16_t* out_bf, float* out_f, int64_t rows, int C,
                               hipStream_t s) {
    if (rows <= 0) return nullptr;
    if (C & 3) return "add_to_bf16: C must be a multiple of 4";
    if (y && ymod <= 0) return "add_to_bf16: ymod";
    const int64_t total = rows * (C >> 2);
    int blocks = (int)((total + 255) / 256);
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(add_to_bf16_kernel, dim3(blocks), dim3(256), 0, s, x, y, ymod, out_bf, out_f, rows, C);
    return nullptr;
}
