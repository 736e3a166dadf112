// Row LayerNorm (one wave per row) and small elementwise helpers.
// Replaces nn.LayerNorm / LayerNorm2d calls of the sam2 image model.
// fp32 statistics (two-pass in registers), outputs: optional fp32, optional bf16,
// optional bf16 of (y + addvec[row % add_mod]) used for "keys + image_pe".
#include <algorithm>

#include "common.h"
#include "kernels.h"

#define LN_MAX_CHUNKS 5  // 5 * 64 lanes * 4 floats = 1280 channels

// One wave per row; a wave walks rows wave_id, wave_id + n_waves, ... of a grid sized to ONE resident set of workgroups (<= 8 per CU):
// gamma / beta (and the row's x) live in registers, so a row costs one load and one store instruction per 1-KB chunk instead of three
// loads, and there is no second, partly filled round of workgroups (86 016 rows in chunks of 8 per wave were 2 688 workgroups on 2 048
// slots: a 31 %-full tail round, 70 us where the bytes need 47).
template <int NCH>
__global__ __launch_bounds__(256) void layernorm_kernel(LayerNormParams p) {
    const int lane = threadIdx.x & 63;
    const int64_t n_waves = (int64_t)gridDim.x * 4;
    const int64_t row0 = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row0 >= p.rows) return;
    const int nvec = p.C >> 2;
    float4 g[NCH], b[NCH];
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int c = lane + 64 * i;
        if (c < nvec) { g[i] = *reinterpret_cast<const float4*>(p.gamma + 4 * c); b[i] = *reinterpret_cast<const float4*>(p.beta + 4 * c); }
        else { g[i] = make_float4(0.f, 0.f, 0.f, 0.f); b[i] = g[i]; }
    }
    const float invC = 1.0f / (float)p.C;
    float4 v[NCH], vn[NCH];
    auto load_row = [&](int64_t row, float4 (&dst)[NCH]) {
        const float* x = p.x + row * p.ldx;
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int c = lane + 64 * i;
            dst[i] = c < nvec ? *reinterpret_cast<const float4*>(x + 4 * c) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    load_row(row0, v);
    for (int64_t row = row0; row < p.rows; row += n_waves) {
        if (row + n_waves < p.rows) load_row(row + n_waves, vn);             // next row in flight while this one is reduced
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < NCH; ++i) s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
        const float mean = wave_sum(s) * invC;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int c = lane + 64 * i;
            if (c < nvec) {
                const float a = v[i].x - mean, bb = v[i].y - mean, cc = v[i].z - mean, d = v[i].w - mean;
                q += (a * a + bb * bb) + (cc * cc + d * d);
            }
        }
        const float rstd = 1.0f / sqrtf(wave_sum(q) * invC + p.eps);
        const float* addv = p.addvec ? p.addvec + (int64_t)(row % p.add_mod) * p.C : nullptr;
        // window-padding rows (tiny/small/base+ trunks): the reference pads the NORMALISED tokens with zeros
        const bool zero_row = p.row_valid && !p.row_valid[row % p.valid_mod];
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int c = lane + 64 * i;
            if (c < nvec) {
                float y0 = (v[i].x - mean) * rstd * g[i].x + b[i].x;
                float y1 = (v[i].y - mean) * rstd * g[i].y + b[i].y;
                float y2 = (v[i].z - mean) * rstd * g[i].z + b[i].z;
                float y3 = (v[i].w - mean) * rstd * g[i].w + b[i].w;
                if (p.act == ACT_GELU) { y0 = gelu_erf(y0); y1 = gelu_erf(y1); y2 = gelu_erf(y2); y3 = gelu_erf(y3); }
                if (zero_row) { y0 = 0.f; y1 = 0.f; y2 = 0.f; y3 = 0.f; }
                if (p.out_f) *reinterpret_cast<float4*>(p.out_f + row * p.ldo + 4 * c) = make_float4(y0, y1, y2, y3);
                if (p.out_bf)
                    *reinterpret_cast<uint2*>(p.out_bf + row * p.ldo + 4 * c) = make_uint2(pack_op16(y0, y1), pack_op16(y2, y3));
                if (p.out_bf_add) {
                    const float4 a = *reinterpret_cast<const float4*>(addv + 4 * c);
                    *reinterpret_cast<uint2*>(p.out_bf_add + row * p.ldo + 4 * c) =
                        make_uint2(pack_op16(y0 + a.x, y1 + a.y), pack_op16(y2 + a.z, y3 + a.w));
                }
            }
        }
#pragma unroll
        for (int i = 0; i < NCH; ++i) v[i] = vn[i];
    }
}

// Narrow rows (C <= 192: the 144-channel stage-0 tokens of Hiera-L, 1.4 M rows per slice): FOUR rows per wave, one per 16-lane DPP
// row, 3 float4 per lane.  With one wave per row only 36 of 64 lanes held data and every 576-byte row paid two whole-wave
// reductions (2.9 TB/s); here a row's statistics never leave its DPP row (row16_sum: four DPP adds, no permlane swap).
__global__ __launch_bounds__(256) void layernorm_rows4_kernel(LayerNormParams p) {
    const int lane = threadIdx.x & 63, sub = lane >> 4, li = lane & 15;
    const int64_t n_groups = (int64_t)gridDim.x * 4;                       // wave-sized groups of 4 rows in flight
    const int64_t g0 = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int nvec = p.C >> 2;
    float4 g[3], b[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int c = li + 16 * i;
        if (c < nvec) { g[i] = *reinterpret_cast<const float4*>(p.gamma + 4 * c); b[i] = *reinterpret_cast<const float4*>(p.beta + 4 * c); }
        else { g[i] = make_float4(0.f, 0.f, 0.f, 0.f); b[i] = g[i]; }
    }
    const float invC = 1.0f / (float)p.C;
    const int64_t n_row_groups = (p.rows + 3) >> 2;
    float4 v[3], vn[3];
    auto load_row = [&](int64_t grp, float4 (&dst)[3]) {
        const int64_t row = grp * 4 + sub;
        const float* x = p.x + (row < p.rows ? row : (int64_t)p.rows - 1) * p.ldx;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int c = li + 16 * i;
            dst[i] = c < nvec ? *reinterpret_cast<const float4*>(x + 4 * c) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    if (g0 >= n_row_groups) return;
    load_row(g0, v);
    for (int64_t grp = g0; grp < n_row_groups; grp += n_groups) {
        if (grp + n_groups < n_row_groups) load_row(grp + n_groups, vn);
        const int64_t row = grp * 4 + sub;
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 3; ++i) s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
        const float mean = row16_sum(s) * invC;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            if (li + 16 * i < nvec) {
                const float a = v[i].x - mean, bb = v[i].y - mean, cc = v[i].z - mean, d = v[i].w - mean;
                q += (a * a + bb * bb) + (cc * cc + d * d);
            }
        }
        const float rstd = 1.0f / sqrtf(row16_sum(q) * invC + p.eps);
        if (row < p.rows) {
            const float* addv = p.addvec ? p.addvec + (int64_t)(row % p.add_mod) * p.C : nullptr;
            const bool zero_row = p.row_valid && !p.row_valid[row % p.valid_mod];
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const int c = li + 16 * i;
                if (c < nvec) {
                    float y0 = (v[i].x - mean) * rstd * g[i].x + b[i].x;
                    float y1 = (v[i].y - mean) * rstd * g[i].y + b[i].y;
                    float y2 = (v[i].z - mean) * rstd * g[i].z + b[i].z;
                    float y3 = (v[i].w - mean) * rstd * g[i].w + b[i].w;
                    if (p.act == ACT_GELU) { y0 = gelu_erf(y0); y1 = gelu_erf(y1); y2 = gelu_erf(y2); y3 = gelu_erf(y3); }
                    if (zero_row) { y0 = 0.f; y1 = 0.f; y2 = 0.f; y3 = 0.f; }
                    if (p.out_f) *reinterpret_cast<float4*>(p.out_f + row * p.ldo + 4 * c) = make_float4(y0, y1, y2, y3);
                    if (p.out_bf)
                        *reinterpret_cast<uint2*>(p.out_bf + row * p.ldo + 4 * c) = make_uint2(pack_op16(y0, y1), pack_op16(y2, y3));
                    if (p.out_bf_add) {
                        const float4 a = *reinterpret_cast<const float4*>(addv + 4 * c);
                        *reinterpret_cast<uint2*>(p.out_bf_add + row * p.ldo + 4 * c) =
                            make_uint2(pack_op16(y0 + a.x, y1 + a.y), pack_op16(y2 + a.z, y3 + a.w));
                    }
                }
            }
        }
#pragma unroll
        for (int i = 0; i < 3; ++i) v[i] = vn[i];
    }
}

const char* launch_layernorm(const LayerNormParams& p, hipStream_t s) {
    if (p.rows <= 0) return nullptr;
    if ((p.C & 3) || p.C > LN_MAX_CHUNKS * 256) return "layernorm: C must be a multiple of 4 and <= 1280";
    if ((p.ldx & 3) || (p.ldo & 3)) return "layernorm: strides must be multiples of 4";
    if (p.out_bf_add && (!p.addvec || p.add_mod <= 0)) return "layernorm: addvec missing";
    if (p.row_valid && p.valid_mod <= 0) return "layernorm: valid_mod";
    if (p.C <= 192 && p.rows >= 4096) {        // narrow rows: four per wave
        static int resident4 = 0;
        if (!resident4) {
            int nb = 0;
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void*)layernorm_rows4_kernel, 256, 0) != hipSuccess || nb < 1) nb = 4;
            resident4 = std::min(nb, 8);
        }
        const int64_t groups = (p.rows + 3) / 4;
        hipLaunchKernelGGL(layernorm_rows4_kernel, dim3((unsigned)std::min<int64_t>((groups + 3) / 4, (int64_t)256 * resident4)), dim3(256), 0, s, p);
        return nullptr;
    }
    const int nch = (p.C / 4 + 63) / 64;
    const int slot = nch <= 1 ? 0 : nch == 2 ? 1 : nch == 3 ? 2 : 3;
    static int resident[4] = {0, 0, 0, 0};             // workgroups of each instantiation that fit one CU (register budget)
    if (!resident[slot]) {
        int nb = 0;
        const void* fn = slot == 0 ? (const void*)layernorm_kernel<1> : slot == 1 ? (const void*)layernorm_kernel<2>
                       : slot == 2 ? (const void*)layernorm_kernel<3> : (const void*)layernorm_kernel<LN_MAX_CHUNKS>;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, fn, 256, 0) != hipSuccess || nb < 1) nb = 4;
        resident[slot] = std::min(nb, 8);
    }
    const dim3 grid((unsigned)std::min<int64_t>((p.rows + 3) / 4, (int64_t)256 * resident[slot]));
    if (slot == 0) hipLaunchKernelGGL(layernorm_kernel<1>, grid, dim3(256), 0, s, p);
    else if (slot == 1) hipLaunchKernelGGL(layernorm_kernel<2>, grid, dim3(256), 0, s, p);
    else if (slot == 2) hipLaunchKernelGGL(layernorm_kernel<3>, grid, dim3(256), 0, s, p);
    else hipLaunchKernelGGL(layernorm_kernel<LN_MAX_CHUNKS>, grid, dim3(256), 0, s, p);
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
        if (out_bf) *reinterpret_cast<uint2*>(out_bf + r * C + 4 * c) = make_uint2(pack_op16(a.x, a.y), pack_op16(a.z, a.w));
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
