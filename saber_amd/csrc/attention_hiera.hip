// Hiera MultiScaleAttention core: softmax(q k^T / sqrt(72)) v per (window, head), head_dim 72,
// with the stage-transition 2x2 max-pool of q fused into the q load.
// Replaces `MultiScaleAttention.forward` of the third-party sam2 trunk.
//
// Token order (common.h) makes every window a contiguous run of rows of the qkv matrix and every
// 2x2 pooling group 4 consecutive rows, so no window partition / unpartition is ever materialised.
//
// MFMA formulation (v_mfma_f32_16x16x32_bf16), everything kept in the "query on lane&15" layout:
//   S^T = K . Q^T   A-operand = K fragment (row = key),  B-operand = Q fragment (col = query)
//                   -> lane (i = lane&15, g = lane>>4) holds S[q=i][key = 16*kt + 4g + r]
//   O^T = V^T . P^T A-operand = V^T fragment read with ds_read_b64_tr_b16 from row-major V in LDS,
//                   B-operand = P straight from the S accumulators (no lane movement)
//                   -> lane holds O[q=i][d = 16*dt + 4g + r]: 4 consecutive d = one 8-byte store.
// head_dim 72 is zero-padded to 96 for the QK^T contraction and to 80 for the PV output.
#include "common.h"
#include "kernels.h"

#define HD 72
#define VSTRIDE 160  // bytes per V row in LDS (80 bf16): 8 rows x 32 B land on disjoint banks for the tr read

typedef s16x4 __attribute__((address_space(3))) * lds_s16x4_ptr;

__device__ __forceinline__ bf16x4 tr_read(const char* p) {
    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(p));
    return __builtin_bit_cast(bf16x4, v);
}
__device__ __forceinline__ bf16x8 cat4(bf16x4 a, bf16x4 b) {
    return __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
}
__device__ __forceinline__ bf16x8 pack8(float a0, float a1, float a2, float a3, float b0, float b1, float b2, float b3) {
    uint4 u = make_uint4(pack_bf16(a0, a1), pack_bf16(a2, a3), pack_bf16(b0, b1), pack_bf16(b2, b3));
    return __builtin_bit_cast(bf16x8, u);
}
__device__ __forceinline__ uint4 bf16max4(uint4 a, uint4 b) {
    // elementwise max of 8 packed bf16 (exact: compare as floats)
    uint32_t* pa = reinterpret_cast<uint32_t*>(&a);
    const uint32_t* pb = reinterpret_cast<const uint32_t*>(&b);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float alo = __uint_as_float(pa[i] << 16), ahi = __uint_as_float(pa[i] & 0xffff0000u);
        const float blo = __uint_as_float(pb[i] << 16), bhi = __uint_as_float(pb[i] & 0xffff0000u);
        const uint32_t lo = __float_as_uint(fmaxf(alo, blo)) >> 16;
        const uint32_t hi = __float_as_uint(fmaxf(ahi, bhi)) & 0xffff0000u;
        pa[i] = hi | lo;
    }
    return a;
}

// q fragment for 16 rows starting at pooled/unpooled row index `row` (already validated by caller)
__device__ __forceinline__ bf16x8 load_q_frag(const bf16_t* qkv, int64_t rs, int64_t tok0, int row, bool valid, int hdoff,
                                              int hoff, int q_pool) {
    uint4 v = make_uint4(0, 0, 0, 0);
    if (valid && hdoff < HD) {
        if (q_pool) {
            const bf16_t* p = qkv + (tok0 + 4 * (int64_t)row) * rs + hoff + hdoff;
            v = *reinterpret_cast<const uint4*>(p);
            v = bf16max4(v, *reinterpret_cast<const uint4*>(p + rs));
            v = bf16max4(v, *reinterpret_cast<const uint4*>(p + 2 * rs));
            v = bf16max4(v, *reinterpret_cast<const uint4*>(p + 3 * rs));
        } else {
            v = *reinterpret_cast<const uint4*>(qkv + (tok0 + row) * rs + hoff + hdoff);
        }
    }
    return __builtin_bit_cast(bf16x8, v);
}

// ------------------------------------------------------------------------------------------------
// small windows (NK = 16 or 64 keys): one wave per (window, head); K fragments live in registers,
// V is staged into a wave-private LDS region (no workgroup barrier anywhere).
template <int NK>
__global__ __launch_bounds__(256) void hiera_attn_small_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ out,
                                                               int n_windows, int heads, int q_pool) {
    constexpr int NKP = NK < 32 ? 32 : NK;
    constexpr int KT = NKP / 16;
    constexpr int KS = NKP / 32;
    __shared__ __attribute__((aligned(16))) char vlds[4 * NKP * VSTRIDE];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int task = blockIdx.x * 4 + wave;
    if (task >= n_windows * heads) return;  // wave-uniform
    const int w = task / heads, h = task - w * heads;
    const int64_t rs = 3 * (int64_t)heads * HD;
    const int64_t os = (int64_t)heads * HD;
    const int64_t tok0 = (int64_t)w * NK;
    const int fi = lane & 15, fg = lane >> 4;
    char* vs = vlds + wave * NKP * VSTRIDE;

    // stage V (rows >= NK and cols 72..79 are zero)
    for (int idx = lane; idx < NKP * 10; idx += 64) {
        const int row = idx / 10, ch = idx - row * 10;
        uint4 val = make_uint4(0, 0, 0, 0);
        if (row < NK && ch < 9) val = *reinterpret_cast<const uint4*>(qkv + (tok0 + row) * rs + 2 * os + h * HD + ch * 8);
        *reinterpret_cast<uint4*>(vs + row * VSTRIDE + ch * 16) = val;
    }
    // K fragments straight from global
    bf16x8 kf[KT][3];
#pragma unroll
    for (int kt = 0; kt < KT; ++kt)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const int key = kt * 16 + fi, hdoff = 32 * c + 8 * fg;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (key < NK && hdoff < HD) v = *reinterpret_cast<const uint4*>(qkv + (tok0 + key) * rs + os + h * HD + hdoff);
            kf[kt][c] = __builtin_bit_cast(bf16x8, v);
        }
    __builtin_amdgcn_wave_barrier();

    const int nq = q_pool ? NK / 4 : NK;
    const int64_t orow0 = (int64_t)w * nq;
    const float sc = 0.11785113019775793f * 1.4426950408889634f;  // 72^-0.5 * log2(e)
    for (int qt = 0; qt * 16 < nq; ++qt) {
        const int row = qt * 16 + fi;
        const bool rvalid = row < nq;
        bf16x8 qf[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) qf[c] = load_q_frag(qkv, rs, tok0, row, rvalid, 32 * c + 8 * fg, h * HD, q_pool);
        f32x4 s[KT];
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) {
            s[kt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int c = 0; c < 3; ++c) s[kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[kt][c], qf[c], s[kt], 0, 0, 0);
        }
        float mx = -3.0e38f;
#pragma unroll
        for (int kt = 0; kt < KT; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const bool kvalid = (kt * 16 + fg * 4 + r) < NK;
                s[kt][r] = kvalid ? s[kt][r] * sc : -3.0e38f;
                mx = fmaxf(mx, s[kt][r]);
            }
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        float sum = 0.f;
#pragma unroll
        for (int kt = 0; kt < KT; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float e = exp2f(s[kt][r] - mx);
                s[kt][r] = e;
                sum += e;
            }
        sum += __shfl_xor(sum, 16, 64);
        sum += __shfl_xor(sum, 32, 64);
        const float inv = 1.0f / sum;
        bf16x8 pf[KS];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
            pf[ks] = pack8(s[2 * ks][0], s[2 * ks][1], s[2 * ks][2], s[2 * ks][3], s[2 * ks + 1][0], s[2 * ks + 1][1],
                           s[2 * ks + 1][2], s[2 * ks + 1][3]);
#pragma unroll
        for (int dt = 0; dt < 5; ++dt) {
            f32x4 o = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const char* base = vs + (32 * ks + 4 * fg + (fi >> 2)) * VSTRIDE + (16 * dt + 4 * (fi & 3)) * 2;
                const bf16x8 vf = cat4(tr_read(base), tr_read(base + 16 * VSTRIDE));
                o = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf[ks], o, 0, 0, 0);
            }
            const int d = 16 * dt + 4 * fg;
            if (rvalid && d < HD)
                *reinterpret_cast<uint2*>(out + (orow0 + row) * os + h * HD + d) =
                    make_uint2(pack_bf16(o[0] * inv, o[1] * inv), pack_bf16(o[2] * inv, o[3] * inv));
        }
    }
}

// ------------------------------------------------------------------------------------------------
// large windows (nk % 128 == 0, incl. global attention): 4 waves share 128-key K/V blocks in LDS,
// online softmax across blocks, QT query tiles (16 rows each) per wave.
#define KB 128
#define K_LDS_BYTES (KB * 256)
#define V_LDS_BYTES (KB * VSTRIDE)

template <int QT>
__global__ __launch_bounds__(256) void hiera_attn_large_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ out,
                                                               int n_windows, int nk, int heads, int q_pool) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* ks_ = smem;
    char* vs = smem + K_LDS_BYTES;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fi = lane & 15, fg = lane >> 4;
    const int nq = q_pool ? nk / 4 : nk;
    const int qblocks = nq / (64 * QT);
    // XCD-aware block -> (window, head, q-block) map: blocks b and b + 8 share an XCD (and its L2).  Every block of a window
    // runs on XCD (w % 8), heads and q-blocks in consecutive dispatch slots, so a window's K/V rows (whose 128-B lines are
    // shared by the 8 heads) are fetched from HBM once instead of once per XCD that touches them.
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int per_w = heads * qblocks;
    const int w = (slot / per_w) * 8 + xcd;
    if (w >= n_windows) return;                      // padding blocks (block-uniform)
    const int h = (slot % per_w) / qblocks, qb = slot % qblocks;
    const int64_t rs = 3 * (int64_t)heads * HD;
    const int64_t os = (int64_t)heads * HD;
    const int64_t tok0 = (int64_t)w * nk;
    const int64_t orow0 = (int64_t)w * nq;
    const int q0 = qb * 64 * QT + wave * 16 * QT;

    // zero the whole K/V region once: pad chunks are never overwritten afterwards
    for (int i = tid; i < (K_LDS_BYTES + V_LDS_BYTES) / 16; i += 256) reinterpret_cast<uint4*>(smem)[i] = make_uint4(0, 0, 0, 0);

    bf16x8 qf[QT][3];
#pragma unroll
    for (int t = 0; t < QT; ++t)
#pragma unroll
        for (int c = 0; c < 3; ++c) qf[t][c] = load_q_frag(qkv, rs, tok0, q0 + 16 * t + fi, true, 32 * c + 8 * fg, h * HD, q_pool);

    float m[QT], l[QT];
    f32x4 o[QT][5];
#pragma unroll
    for (int t = 0; t < QT; ++t) {
        m[t] = -3.0e38f;
        l[t] = 0.f;
#pragma unroll
        for (int dt = 0; dt < 5; ++dt) o[t][dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    const float sc = 0.11785113019775793f * 1.4426950408889634f;

    // staging: 128 rows x 9 chunks = 1152 16-B chunks per operand, 4.5 per thread
    u32x4 rk[5], rv[5];
    auto gload = [&](int kb) {
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            const int idx = min(tid + 256 * j, KB * 9 - 1);  // unconditional (clamped) loads keep rk/rv in registers
            const int row = idx / 9, ch = idx - row * 9;
            const bf16_t* p = qkv + (tok0 + (int64_t)kb * KB + row) * rs + h * HD + ch * 8;
            rk[j] = *reinterpret_cast<const u32x4*>(p + os);
            rv[j] = *reinterpret_cast<const u32x4*>(p + 2 * os);
        }
    };
    auto lstore = [&]() {
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            const int idx = tid + 256 * j;
            if (idx < KB * 9) {
                const int row = idx / 9, ch = idx - row * 9;
                *reinterpret_cast<u32x4*>(ks_ + row * 256 + ((ch ^ (row & 15)) << 4)) = rk[j];
                *reinterpret_cast<u32x4*>(vs + row * VSTRIDE + ch * 16) = rv[j];
            }
        }
    };

    const int nkb = nk / KB;
    gload(0);
    for (int kb = 0; kb < nkb; ++kb) {
        __syncthreads();  // previous block fully consumed (and, first time, zero-fill done)
        lstore();
        __syncthreads();
        if (kb + 1 < nkb) gload(kb + 1);

        f32x4 s[QT][8];
#pragma unroll
        for (int kt = 0; kt < 8; ++kt) {
            bf16x8 kf[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const int row = 16 * kt + fi;
                kf[c] = *reinterpret_cast<const bf16x8*>(ks_ + row * 256 + (((4 * c + fg) ^ (row & 15)) << 4));
            }
#pragma unroll
            for (int t = 0; t < QT; ++t) {
                s[t][kt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int c = 0; c < 3; ++c) s[t][kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[c], qf[t][c], s[t][kt], 0, 0, 0);
            }
        }
        bf16x8 pf[QT][4];
#pragma unroll
        for (int t = 0; t < QT; ++t) {
            float mx = -3.0e38f;
#pragma unroll
            for (int kt = 0; kt < 8; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    s[t][kt][r] *= sc;
                    mx = fmaxf(mx, s[t][kt][r]);
                }
            mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            const float mn = fmaxf(m[t], mx);
            const float alpha = exp2f(m[t] - mn);
            m[t] = mn;
            float sum = 0.f;
#pragma unroll
            for (int kt = 0; kt < 8; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float e = exp2f(s[t][kt][r] - mn);
                    s[t][kt][r] = e;
                    sum += e;
                }
            sum += __shfl_xor(sum, 16, 64);
            sum += __shfl_xor(sum, 32, 64);
            l[t] = l[t] * alpha + sum;
#pragma unroll
            for (int dt = 0; dt < 5; ++dt)
#pragma unroll
                for (int r = 0; r < 4; ++r) o[t][dt][r] *= alpha;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
                pf[t][ks] = pack8(s[t][2 * ks][0], s[t][2 * ks][1], s[t][2 * ks][2], s[t][2 * ks][3], s[t][2 * ks + 1][0],
                                  s[t][2 * ks + 1][1], s[t][2 * ks + 1][2], s[t][2 * ks + 1][3]);
        }
#pragma unroll
        for (int dt = 0; dt < 5; ++dt)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const char* base = vs + (32 * ks + 4 * fg + (fi >> 2)) * VSTRIDE + (16 * dt + 4 * (fi & 3)) * 2;
                const bf16x8 vf = cat4(tr_read(base), tr_read(base + 16 * VSTRIDE));
#pragma unroll
                for (int t = 0; t < QT; ++t) o[t][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf[t][ks], o[t][dt], 0, 0, 0);
            }
    }
#pragma unroll
    for (int t = 0; t < QT; ++t) {
        const float inv = 1.0f / l[t];
        const int row = q0 + 16 * t + fi;
#pragma unroll
        for (int dt = 0; dt < 5; ++dt) {
            const int d = 16 * dt + 4 * fg;
            if (d < HD)
                *reinterpret_cast<uint2*>(out + (orow0 + row) * os + h * HD + d) =
                    make_uint2(pack_bf16(o[t][dt][0] * inv, o[t][dt][1] * inv), pack_bf16(o[t][dt][2] * inv, o[t][dt][3] * inv));
        }
    }
}

const char* hiera_attention_init_device() {
    hipError_t st = hipSuccess;
    if (st == hipSuccess) st = hipFuncSetAttribute(reinterpret_cast<const void*>(hiera_attn_large_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize,
                        K_LDS_BYTES + V_LDS_BYTES);
    if (st == hipSuccess) st = hipFuncSetAttribute(reinterpret_cast<const void*>(hiera_attn_large_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize,
                        K_LDS_BYTES + V_LDS_BYTES);
    return st == hipSuccess ? nullptr : hipGetErrorString(st);
}

const char* launch_hiera_attention(const bf16_t* qkv, bf16_t* out, int n_windows, int nk, int heads, int q_pool,
                                   hipStream_t s) {
    if (n_windows <= 0) return nullptr;
    if (((uintptr_t)qkv & 15) || ((uintptr_t)out & 7)) return "hiera_attention: pointer alignment";
    if (nk == 16 || nk == 64) {
        const int tasks = n_windows * heads;
        const dim3 grid((tasks + 3) / 4);
        if (nk == 16) hipLaunchKernelGGL(hiera_attn_small_kernel<16>, grid, dim3(256), 0, s, qkv, out, n_windows, heads, q_pool);
        else hipLaunchKernelGGL(hiera_attn_small_kernel<64>, grid, dim3(256), 0, s, qkv, out, n_windows, heads, q_pool);
        return nullptr;
    }
    if (nk % KB != 0) return "hiera_attention: nk must be 16, 64 or a multiple of 128";
    const int nq = q_pool ? nk / 4 : nk;
    if (false && nq % 128 == 0) {
        const dim3 grid(((n_windows + 7) / 8) * 8 * heads * (nq / 128));
        hipLaunchKernelGGL(hiera_attn_large_kernel<2>, grid, dim3(256), K_LDS_BYTES + V_LDS_BYTES, s, qkv, out, n_windows, nk, heads, q_pool);
    } else if (nq % 64 == 0) {
        const dim3 grid(((n_windows + 7) / 8) * 8 * heads * (nq / 64));
        hipLaunchKernelGGL(hiera_attn_large_kernel<1>, grid, dim3(256), K_LDS_BYTES + V_LDS_BYTES, s, qkv, out, n_windows, nk, heads, q_pool);
    } else return "hiera_attention: nq must be a multiple of 64 for large windows";
    return nullptr;
}
