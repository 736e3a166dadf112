// Flash-style attention for SAM2's memory attention (upstream MemoryAttentionLayer: d_model 256, ONE head of 256 channels,
// 4 096 image tokens as queries, 4 096 (self attention) or ~8 200 (cross attention to the memory bank) keys).
// Round 2 composed it from GEMMs: scores materialised as fp32 (4096, N_k), a softmax kernel, a P.V GEMM - 2 x 134 MB of score traffic per
// attention and three launches.  Here the structure of dec_t2i_kernel (decoder_fused.hip; 64 query rows of 256 channels over thousands
// of keys is exactly its shape): a workgroup owns 64 query rows, 8 waves = 4 query tiles x 2 key halves, K and V tiles of 64 keys go
// global -> LDS directly (2-stage ring, XOR-swizzled 512-B rows), S^T = K.Q^T and O^T = V^T.P^T keep the query on lane & 15, V^T
// fragments come from the row-major V tile through ds_read_b64_tr_b16, online softmax in the exp2 domain, the two key halves are merged
// through LDS at the end.  The keys can be split over several workgroups (flash256_combine_kernel merges the partial results).
// bf16 operands (Q, K, V, P), fp32 accumulation / statistics, bf16 output = bf16(O / l + bias_v) (the value bias is added after the
// product: softmax rows sum to one).
#include "common.h"
#include "kernels.h"

typedef s16x4 __attribute__((address_space(3))) * f2_lds_s16x4_ptr;
typedef unsigned int f2_u32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(1))) const void* f2_gptr;
typedef __attribute__((address_space(3))) void* f2_lptr;
__device__ __forceinline__ op16x8 f2_cat4(op16x4 a, op16x4 b) { return __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7); }
__device__ __forceinline__ op16x8 f2_pack8(float a0, float a1, float a2, float a3, float b0, float b1, float b2, float b3) {
    uint4 u = make_uint4(pack_op16(a0, a1), pack_op16(a2, a3), pack_op16(b0, b1), pack_op16(b2, b3));
    return __builtin_bit_cast(op16x8, u);
}
#define F2_ROWB 512
#define F2_KB 64
#define F2_STAGE (2 * F2_KB * F2_ROWB)         // K tile + V tile
#define F2_LDS (2 * F2_STAGE)

// Opart [q-block][split][64][256] fp32 un-normalised, ML [q-block][split][64][2] (running max in the exp2 domain, sum)
__global__ __launch_bounds__(512) void flash256_kernel(const bf16_t* __restrict__ Q, const bf16_t* __restrict__ K, const bf16_t* __restrict__ V, int n_keys,
                                                      float qscale, int split, const float* __restrict__ bias_v, bf16_t* __restrict__ out,
                                                      float* __restrict__ Opart, float* __restrict__ ML) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int qt = wave & 3, kh = wave >> 2;
    const int fi = lane & 15, fg = lane >> 4;
    const int qb = blockIdx.x / split, sp = blockIdx.x - qb * split;
    const int nkb_all = (n_keys + F2_KB - 1) / F2_KB;
    const int per = (nkb_all + split - 1) / split;
    const int kb0 = sp * per, nkb = max(0, min(per, nkb_all - kb0));
    const int q0 = qb * 64;

    op16x8 qf[8];
    {
        const bf16_t* qrow = Q + (int64_t)(q0 + qt * 16 + fi) * 256;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) qf[ks] = __builtin_bit_cast(op16x8, *reinterpret_cast<const uint4*>(qrow + 32 * ks + 8 * fg));
    }
    float m = -3.0e38f, l = 0.f;
    f32x4 o[16];
#pragma unroll
    for (int dt = 0; dt < 16; ++dt) o[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // direct-to-LDS: K tile = 32 pieces of 1 KB (2 rows each), V tile the same; wave w issues pieces 4 w .. 4 w + 3 of each.  Rows beyond
    // the last key are clamped (their scores are masked below; V rows of masked keys meet P = 0)
    int srow[4], schunk[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        srow[i] = 2 * (wave * 4 + i) + (lane >> 5);
        schunk[i] = (lane & 31) ^ (srow[i] & 15);
    }
    auto issue = [&](int kb, int stage) {
        char* sx = smem + stage * F2_STAGE;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int64_t row = min((int64_t)(kb0 + kb) * F2_KB + srow[i], (int64_t)n_keys - 1);
            __builtin_amdgcn_global_load_lds((f2_gptr)(K + row * 256 + schunk[i] * 8), (f2_lptr)(sx + (wave * 4 + i) * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((f2_gptr)(V + row * 256 + schunk[i] * 8), (f2_lptr)(sx + F2_KB * F2_ROWB + (wave * 4 + i) * 1024), 16, 0, 0);
        }
    };
    int koff[8];
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) koff[ks] = fi * F2_ROWB + (((4 * ks + fg) ^ fi) << 4);
    const int vrow = 4 * fg + (fi >> 2);
    const int vsel = (fi & 3) >> 1, vlow = (fi & 1) * 8;

    if (nkb > 0) issue(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    int stage = 0;
    for (int kb = 0; kb < nkb; ++kb) {
        if (kb + 1 < nkb) issue(kb + 1, stage ^ 1);
        const char* ks_ = smem + stage * F2_STAGE + kh * 32 * F2_ROWB;
        const char* vs_ = smem + stage * F2_STAGE + F2_KB * F2_ROWB + kh * 32 * F2_ROWB;
        f32x4 s[2];
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
            s[kt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                const op16x8 kx = *reinterpret_cast<const op16x8*>(ks_ + kt * 16 * F2_ROWB + koff[ks]);
                s[kt] = MFMA_16x16x32(kx, qf[ks], s[kt], 0, 0, 0);
            }
        }
        // scale into the exp2 domain; keys beyond n_keys take no part
        const int key_base = (kb0 + kb) * F2_KB + kh * 32 + 4 * fg;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) s[kt][r] = (key_base + kt * 16 + r < n_keys) ? s[kt][r] * qscale : -INFINITY;     // (m starts finite: exp2(-inf - m) = 0)
        float mx = fmaxf(fmaxf(fmaxf(s[0][0], s[0][1]), fmaxf(s[0][2], s[0][3])), fmaxf(fmaxf(s[1][0], s[1][1]), fmaxf(s[1][2], s[1][3])));
        mx = xor32_max(xor16_max(mx));
        if (__any(mx > m)) {
            const float mn = fmaxf(m, mx);
            const float alpha = __builtin_amdgcn_exp2f(m - mn);
            m = mn;
            l *= alpha;
#pragma unroll
            for (int dt = 0; dt < 16; ++dt)
#pragma unroll
                for (int r = 0; r < 4; ++r) o[dt][r] *= alpha;
        }
        float sum = 0.f;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float e = __builtin_amdgcn_exp2f(s[kt][r] - m);
                s[kt][r] = e;
                sum += e;
            }
        sum = xor32_sum(xor16_sum(sum));
        l += sum;
        const op16x8 pf = f2_pack8(s[0][0], s[0][1], s[0][2], s[0][3], s[1][0], s[1][1], s[1][2], s[1][3]);
        {
            const uint32_t va = (uint32_t)(uintptr_t)(f2_lptr)(vs_ + vrow * F2_ROWB + vlow);
#pragma unroll
            for (int d4 = 0; d4 < 16; d4 += 4) {
                f2_u32x2 lo[4], hi[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int sw = ((2 * (d4 + j) + vsel) ^ (vrow & 15)) << 4;
                    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(lo[j]) : "v"(va + sw) : "memory");
                    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:8192" : "=v"(hi[j]) : "v"(va + sw) : "memory");
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const op16x8 vf = f2_cat4(__builtin_bit_cast(op16x4, lo[j]), __builtin_bit_cast(op16x4, hi[j]));
                    o[d4 + j] = MFMA_16x16x32(vf, pf, o[d4 + j], 0, 0, 0);
                }
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        stage ^= 1;
    }
    // merge the two key halves of each query tile: waves 4..7 park (m, l, O) in LDS, waves 0..3 combine
    float* mo = reinterpret_cast<float*>(smem) + (size_t)qt * 16 * 260;
    if (kh == 1) {
#pragma unroll
        for (int dt = 0; dt < 16; ++dt) *reinterpret_cast<float4*>(mo + fi * 260 + 16 * dt + 4 * fg) = make_float4(o[dt][0], o[dt][1], o[dt][2], o[dt][3]);
        if (fg == 0) { mo[fi * 260 + 256] = m; mo[fi * 260 + 257] = l; }
    }
    __syncthreads();
    if (kh == 0) {
        const float m2 = mo[fi * 260 + 256], l2 = mo[fi * 260 + 257];
        const float mn = fmaxf(m, m2);
        const float a1 = exp2f(m - mn), a2 = exp2f(m2 - mn);
        const int q = q0 + qt * 16 + fi;
        if (split > 1) {
            float* op = Opart + (((int64_t)qb * split + sp) * 64 + qt * 16 + fi) * 256;
#pragma unroll
            for (int dt = 0; dt < 16; ++dt) {
                const float4 t = *reinterpret_cast<const float4*>(mo + fi * 260 + 16 * dt + 4 * fg);
                *reinterpret_cast<float4*>(op + 16 * dt + 4 * fg) = make_float4(o[dt][0] * a1 + t.x * a2, o[dt][1] * a1 + t.y * a2, o[dt][2] * a1 + t.z * a2, o[dt][3] * a1 + t.w * a2);
            }
            if (fg == 0) {
                float* mlp = ML + (((int64_t)qb * split + sp) * 64 + qt * 16 + fi) * 2;
                mlp[0] = mn;
                mlp[1] = l * a1 + l2 * a2;
            }
        } else {
            const float inv = 1.0f / (l * a1 + l2 * a2);
#pragma unroll
            for (int dt = 0; dt < 16; ++dt) {
                const float4 t = *reinterpret_cast<const float4*>(mo + fi * 260 + 16 * dt + 4 * fg);
                const float4 b = *reinterpret_cast<const float4*>(bias_v + 16 * dt + 4 * fg);
                *reinterpret_cast<uint2*>(out + (int64_t)q * 256 + 16 * dt + 4 * fg) =
                    make_uint2(pack_op16((o[dt][0] * a1 + t.x * a2) * inv + b.x, (o[dt][1] * a1 + t.y * a2) * inv + b.y),
                               pack_op16((o[dt][2] * a1 + t.z * a2) * inv + b.z, (o[dt][3] * a1 + t.w * a2) * inv + b.w));
            }
        }
    }
}

// out[q] = bf16(sum_s w_s Opart[q][s] / L + bias), w_s = 2^(m_s - max m), L = sum_s w_s l_s; one block per 4 query rows
__global__ __launch_bounds__(256) void flash256_combine_kernel(const float* __restrict__ Opart, const float* __restrict__ ML, int split, const float* __restrict__ bias_v,
                                                              bf16_t* __restrict__ out, int n_q) {
    const int q = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (q >= n_q) return;
    const int qb = q >> 6, r = q & 63;
    float mm = -3.0e38f;
    for (int s = 0; s < split; ++s) mm = fmaxf(mm, ML[(((int64_t)qb * split + s) * 64 + r) * 2]);
    float L = 0.f;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int s = 0; s < split; ++s) {
        const float* mlp = ML + (((int64_t)qb * split + s) * 64 + r) * 2;
        const float w = exp2f(mlp[0] - mm);
        L += w * mlp[1];
        const float4 t = *reinterpret_cast<const float4*>(Opart + (((int64_t)qb * split + s) * 64 + r) * 256 + 4 * lane);
        acc.x += w * t.x; acc.y += w * t.y; acc.z += w * t.z; acc.w += w * t.w;
    }
    const float inv = 1.0f / L;
    const float4 b = *reinterpret_cast<const float4*>(bias_v + 4 * lane);
    *reinterpret_cast<uint2*>(out + (int64_t)q * 256 + 4 * lane) = make_uint2(pack_op16(acc.x * inv + b.x, acc.y * inv + b.y), pack_op16(acc.z * inv + b.z, acc.w * inv + b.w));
}

const char* launch_flash256(const bf16_t* Q, const bf16_t* K, const bf16_t* V, int n_q, int n_keys, float scale, const float* bias_v, bf16_t* out, float* ws,
                            size_t ws_floats, hipStream_t s) {
    if (n_q <= 0 || (n_q & 63) || n_keys <= 0 || !Q || !K || !V || !bias_v || !out) return "flash256: bad argument (n_q must be a multiple of 64)";
    const int qblocks = n_q / 64, nkb = (n_keys + F2_KB - 1) / F2_KB;
    int split = 1;
    while (qblocks * split < 256 && split * 2 <= nkb && split < 8) split *= 2;
    if (split > 1 && (!ws || ws_floats < (size_t)qblocks * split * 64 * 258)) split = 1;
    float* Opart = ws;
    float* ML = ws ? ws + (size_t)qblocks * split * 64 * 256 : nullptr;
    static bool attr = false;
    if (!attr) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(flash256_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, F2_LDS) != hipSuccess) return "flash256: cannot reserve LDS";
        attr = true;
    }
    hipLaunchKernelGGL(flash256_kernel, dim3(qblocks * split), dim3(512), F2_LDS, s, Q, K, V, n_keys, scale * 1.4426950408889634f, split, bias_v, out, Opart, ML);
    if (split > 1) hipLaunchKernelGGL(flash256_combine_kernel, dim3((n_q + 3) / 4), dim3(256), 0, s, (const float*)Opart, (const float*)ML, split, bias_v, out, n_q);
    return nullptr;
}
