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
// head_dim (72 for Hiera-L, 96 for tiny/small, 56 for base+) is zero-padded to a multiple of 32 for the QK^T contraction
// and to a multiple of 16 for the PV output.
//
// Windows whose key count is not a multiple of the key tile (the 14x14 = 196-key and 7x7 = 49-key padded windows of the
// tiny/small/base+ trunks) and the global blocks of those trunks (whose token matrix carries the window padding rows, which
// must not act as keys) are handled by masking: scores of keys >= nk, or of keys whose kmask byte is 0, are set to -3e38.
#include "common.h"
#include "kernels.h"

template <int HD_> struct HdTraits {
    static constexpr int HD = HD_;
    static constexpr int NCH = HD_ / 8;              // 16-B chunks per row
    static constexpr int CK = (HD_ + 31) / 32;       // k-steps of the QK^T contraction
    static constexpr int DT = (HD_ + 15) / 16;       // 16-wide tiles of the PV output
    // bytes per V row in LDS: >= 32 * DT, and 8 consecutive rows x 32 B land on disjoint banks for the tr read
    static constexpr int VSTRIDE = HD_ > 80 ? 224 : 160;
    static constexpr int VCH = VSTRIDE / 16;
    // HD^-0.5 * log2(e): scores live in the exp2 domain
    static constexpr float SCALE = (HD_ == 72 ? 0.11785113019775793f : HD_ == 96 ? 0.10206207261596575f : 0.1336306209562122f) * 1.4426950408889634f;
};

typedef s16x4 __attribute__((address_space(3))) * lds_s16x4_ptr;

__device__ __forceinline__ op16x4 tr_read(const char* p) {
    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(p));
    return __builtin_bit_cast(op16x4, v);
}
__device__ __forceinline__ op16x8 cat4(op16x4 a, op16x4 b) {
    return __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
}
__device__ __forceinline__ op16x8 pack8(float a0, float a1, float a2, float a3, float b0, float b1, float b2, float b3) {
    uint4 u = make_uint4(pack_op16(a0, a1), pack_op16(a2, a3), pack_op16(b0, b1), pack_op16(b2, b3));
    return __builtin_bit_cast(op16x8, u);
}
__device__ __forceinline__ uint4 bf16max4(uint4 a, uint4 b) {
    // elementwise max of 8 packed 16-bit operands (exact: compare as floats)
    uint32_t* pa = reinterpret_cast<uint32_t*>(&a);
    const uint32_t* pb = reinterpret_cast<const uint32_t*>(&b);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        // (the maximum of two representable values is representable: the re-pack is exact in either operand type)
        pa[i] = pack_op16(fmaxf(op16_lo(pa[i]), op16_lo(pb[i])), fmaxf(op16_hi(pa[i]), op16_hi(pb[i])));
    }
    return a;
}

// q fragment for 16 rows starting at pooled/unpooled row index `row` (already validated by caller)
template <int HD>
__device__ __forceinline__ op16x8 load_q_frag(const bf16_t* qkv, int64_t rs, int64_t tok0, int row, bool valid, int hdoff,
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
    return __builtin_bit_cast(op16x8, v);
}

// ------------------------------------------------------------------------------------------------
// small windows (nk <= NK = 16 or 64 keys): one wave per (window, head); K fragments live in registers,
// V is staged into a wave-private LDS region (no workgroup barrier anywhere).
template <int NK, int HD>
__global__ __launch_bounds__(256) void hiera_attn_small_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ out,
                                                               int n_windows, int nk, int heads, int q_pool) {
    using TR = HdTraits<HD>;
    constexpr int VSTRIDE = TR::VSTRIDE, VCH = TR::VCH, NCH = TR::NCH, CK = TR::CK, DT = TR::DT;
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
    const int64_t tok0 = (int64_t)w * nk;
    const int fi = lane & 15, fg = lane >> 4;
    char* vs = vlds + wave * NKP * VSTRIDE;

    // stage V (rows >= nk and the pad columns are zero)
    for (int idx = lane; idx < NKP * VCH; idx += 64) {
        const int row = idx / VCH, ch = idx - row * VCH;
        uint4 val = make_uint4(0, 0, 0, 0);
        if (row < nk && ch < NCH) val = *reinterpret_cast<const uint4*>(qkv + (tok0 + row) * rs + 2 * os + h * HD + ch * 8);
        *reinterpret_cast<uint4*>(vs + row * VSTRIDE + ch * 16) = val;
    }
    // K fragments straight from global
    op16x8 kf[KT][CK];
#pragma unroll
    for (int kt = 0; kt < KT; ++kt)
#pragma unroll
        for (int c = 0; c < CK; ++c) {
            const int key = kt * 16 + fi, hdoff = 32 * c + 8 * fg;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (key < nk && hdoff < HD) v = *reinterpret_cast<const uint4*>(qkv + (tok0 + key) * rs + os + h * HD + hdoff);
            kf[kt][c] = __builtin_bit_cast(op16x8, v);
        }
    __builtin_amdgcn_wave_barrier();

    const int nq = q_pool ? nk / 4 : nk;
    const int64_t orow0 = (int64_t)w * nq;
    const float sc = TR::SCALE;
    for (int qt = 0; qt * 16 < nq; ++qt) {
        const int row = qt * 16 + fi;
        const bool rvalid = row < nq;
        op16x8 qf[CK];
#pragma unroll
        for (int c = 0; c < CK; ++c) qf[c] = load_q_frag<HD>(qkv, rs, tok0, row, rvalid, 32 * c + 8 * fg, h * HD, q_pool);
        f32x4 s[KT];
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) {
            s[kt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int c = 0; c < CK; ++c) s[kt] = MFMA_16x16x32(kf[kt][c], qf[c], s[kt], 0, 0, 0);
        }
        // softmax in the exp2 domain with the scale folded into the exponent's fma: exp2(s * sc - max * sc), raw v_exp_f32
        // (exp2f() adds a denormal-range fix-up of 4 more VALU instructions per element; results below 2^-126 flush to 0 either way)
        float mx = -3.0e38f;
#pragma unroll
        for (int kt = 0; kt < KT; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const bool kvalid = (kt * 16 + fg * 4 + r) < nk;
                s[kt][r] = kvalid ? s[kt][r] : -3.0e38f;
                mx = fmaxf(mx, s[kt][r]);
            }
        mx = xor32_max(xor16_max(mx));
        const float nmx = -mx * sc;
        float sum = 0.f;
#pragma unroll
        for (int kt = 0; kt < KT; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float e = __builtin_amdgcn_exp2f(fmaf(s[kt][r], sc, nmx));
                s[kt][r] = e;
                sum += e;
            }
        sum = xor32_sum(xor16_sum(sum));
        const float inv = __builtin_amdgcn_rcpf(sum);
        op16x8 pf[KS];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
            pf[ks] = pack8(s[2 * ks][0], s[2 * ks][1], s[2 * ks][2], s[2 * ks][3], s[2 * ks + 1][0], s[2 * ks + 1][1],
                           s[2 * ks + 1][2], s[2 * ks + 1][3]);
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            f32x4 o = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const char* base = vs + (32 * ks + 4 * fg + (fi >> 2)) * VSTRIDE + (16 * dt + 4 * (fi & 3)) * 2;
                const op16x8 vf = cat4(tr_read(base), tr_read(base + 16 * VSTRIDE));
                o = MFMA_16x16x32(vf, pf[ks], o, 0, 0, 0);
            }
            const int d = 16 * dt + 4 * fg;
            if (rvalid && d < HD)
                *reinterpret_cast<uint2*>(out + (orow0 + row) * os + h * HD + d) =
                    make_uint2(pack_op16(o[0] * inv, o[1] * inv), pack_op16(o[2] * inv, o[3] * inv));
        }
    }
}

// ------------------------------------------------------------------------------------------------
// large windows (incl. global attention): 4 waves share 128-key K/V blocks in LDS, online softmax across blocks,
// QT query tiles (16 rows each) per wave.  MASKED = the key count is not a multiple of 128 and/or a key mask is given.
#define KB 128
#define K_LDS_BYTES (KB * 256)

template <int QT, int HD, bool MASKED>
__global__ __launch_bounds__(256) void hiera_attn_large_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ out,
                                                               int n_windows, int nk, int heads, int q_pool,
                                                               const uint8_t* __restrict__ kmask) {
    using TR = HdTraits<HD>;
    constexpr int VSTRIDE = TR::VSTRIDE, NCH = TR::NCH, CK = TR::CK, DT = TR::DT;
    constexpr int V_LDS_BYTES = KB * VSTRIDE;
    constexpr int NJ = (KB * NCH + 255) / 256;     // staging chunks per thread and operand
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* ks_ = smem;
    char* vs = smem + K_LDS_BYTES;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fi = lane & 15, fg = lane >> 4;
    const int nq = q_pool ? nk / 4 : nk;
    const int qblocks = (nq + 64 * QT - 1) / (64 * QT);
    // XCD-aware block -> (window, head, q-block) map: blocks b and b + 8 share an XCD (and its L2).  Every block of a window
    // runs on XCD (w % 8), heads and q-blocks in consecutive dispatch slots, so a window's K/V rows (whose 128-B lines are
    // shared by the heads) are fetched from HBM once instead of once per XCD that touches them.
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

    op16x8 qf[QT][CK];
#pragma unroll
    for (int t = 0; t < QT; ++t)
#pragma unroll
        for (int c = 0; c < CK; ++c)
            qf[t][c] = load_q_frag<HD>(qkv, rs, tok0, q0 + 16 * t + fi, !MASKED || (q0 + 16 * t + fi) < nq, 32 * c + 8 * fg, h * HD, q_pool);

    float m[QT], l[QT];
    f32x4 o[QT][DT];
#pragma unroll
    for (int t = 0; t < QT; ++t) {
        m[t] = -3.0e38f;
        l[t] = 0.f;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) o[t][dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    const float sc = TR::SCALE;

    // staging: 128 rows x NCH 16-B chunks per operand
    u32x4 rk[NJ], rv[NJ];
    auto gload = [&](int kb) {
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int idx = min(tid + 256 * j, KB * NCH - 1);  // unconditional (clamped) loads keep rk/rv in registers
            const int row = idx / NCH, ch = idx - row * NCH;
            int key = kb * KB + row;
            if (MASKED) key = min(key, nk - 1);                // rows past the window: any finite data, their scores are masked
            const bf16_t* p = qkv + (tok0 + key) * rs + h * HD + ch * 8;
            rk[j] = *reinterpret_cast<const u32x4*>(p + os);
            rv[j] = *reinterpret_cast<const u32x4*>(p + 2 * os);
        }
    };
    auto lstore = [&]() {
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int idx = tid + 256 * j;
            if (idx < KB * NCH) {
                const int row = idx / NCH, ch = idx - row * NCH;
                *reinterpret_cast<u32x4*>(ks_ + row * 256 + ((ch ^ (row & 15)) << 4)) = rk[j];
                *reinterpret_cast<u32x4*>(vs + row * VSTRIDE + ch * 16) = rv[j];
            }
        }
    };

    const int nkb = (nk + KB - 1) / KB;
    gload(0);
    for (int kb = 0; kb < nkb; ++kb) {
        __syncthreads();  // previous block fully consumed (and, first time, zero-fill done)
        lstore();
        __syncthreads();
        if (kb + 1 < nkb) gload(kb + 1);

        f32x4 s[QT][8];
#pragma unroll
        for (int kt = 0; kt < 8; ++kt) {
            op16x8 kf[CK];
#pragma unroll
            for (int c = 0; c < CK; ++c) {
                const int row = 16 * kt + fi;
                kf[c] = *reinterpret_cast<const op16x8*>(ks_ + row * 256 + (((4 * c + fg) ^ (row & 15)) << 4));
            }
#pragma unroll
            for (int t = 0; t < QT; ++t) {
                s[t][kt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int c = 0; c < CK; ++c) s[t][kt] = MFMA_16x16x32(kf[c], qf[t][c], s[t][kt], 0, 0, 0);
            }
        }
        uint32_t mbits = 0xffffffffu;                      // bit 4*kt + r: key kb*128 + 16*kt + 4*fg + r takes part
        if (MASKED) {
            mbits = 0u;
#pragma unroll
            for (int kt = 0; kt < 8; ++kt) {
                const int key = kb * KB + 16 * kt + 4 * fg;
                uint32_t mv = 0x01010101u;
                if (kmask) mv = *reinterpret_cast<const uint32_t*>(kmask + key);   // kmask is zero-padded to a multiple of 128
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (key + r < nk && ((mv >> (8 * r)) & 0xffu)) mbits |= 1u << (4 * kt + r);
            }
        }
        op16x8 pf[QT][4];
#pragma unroll
        for (int t = 0; t < QT; ++t) {
            // online softmax in the exp2 domain; the scale rides in the exponent's fma and v_exp_f32 is used raw (see the small kernel)
            float mx = -3.0e38f;
#pragma unroll
            for (int kt = 0; kt < 8; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (MASKED && !((mbits >> (4 * kt + r)) & 1u)) s[t][kt][r] = -3.0e38f;
                    mx = fmaxf(mx, s[t][kt][r]);
                }
            mx = xor32_max(xor16_max(mx));
            const float mn = fmaxf(m[t], mx * sc);
            const float alpha = __builtin_amdgcn_exp2f(m[t] - mn);
            m[t] = mn;
            float sum = 0.f;
#pragma unroll
            for (int kt = 0; kt < 8; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float e = __builtin_amdgcn_exp2f(fmaf(s[t][kt][r], sc, -mn));
                    if (MASKED && !((mbits >> (4 * kt + r)) & 1u)) e = 0.f;   // also covers a block with no valid key at all
                    s[t][kt][r] = e;
                    sum += e;
                }
            sum = xor32_sum(xor16_sum(sum));
            l[t] = l[t] * alpha + sum;
#pragma unroll
            for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                for (int r = 0; r < 4; ++r) o[t][dt][r] *= alpha;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
                pf[t][ks] = pack8(s[t][2 * ks][0], s[t][2 * ks][1], s[t][2 * ks][2], s[t][2 * ks][3], s[t][2 * ks + 1][0],
                                  s[t][2 * ks + 1][1], s[t][2 * ks + 1][2], s[t][2 * ks + 1][3]);
        }
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const char* base = vs + (32 * ks + 4 * fg + (fi >> 2)) * VSTRIDE + (16 * dt + 4 * (fi & 3)) * 2;
                const op16x8 vf = cat4(tr_read(base), tr_read(base + 16 * VSTRIDE));
#pragma unroll
                for (int t = 0; t < QT; ++t) o[t][dt] = MFMA_16x16x32(vf, pf[t][ks], o[t][dt], 0, 0, 0);
            }
    }
#pragma unroll
    for (int t = 0; t < QT; ++t) {
        const float inv = l[t] > 0.f ? __builtin_amdgcn_rcpf(l[t]) : 0.f;
        const int row = q0 + 16 * t + fi;
        if (MASKED && row >= nq) continue;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            const int d = 16 * dt + 4 * fg;
            if (d < HD)
                *reinterpret_cast<uint2*>(out + (orow0 + row) * os + h * HD + d) =
                    make_uint2(pack_op16(o[t][dt][0] * inv, o[t][dt][1] * inv), pack_op16(o[t][dt][2] * inv, o[t][dt][3] * inv));
        }
    }
}

// ------------------------------------------------------------------------------------------------
// 256-key windows (the 16x16 windows of Hiera-L stage 2: 32 of the 48 blocks).  Persistent workgroups (one per CU, 8 waves x
// 32 queries) walk over (window, head) tasks: the task's whole K and V (256 x HD each) go global -> LDS in one burst of
// direct-to-LDS loads into DENSE rows of HD * 2 bytes (no staging registers, no padding), double-buffered, so the next task's
// image and Q fragments are in flight while the current task runs its two 128-key halves out of LDS: one barrier per task.  Dense 144-B rows are conflict-free for the K fragment reads (36 dwords per row: 16 rows hit 16 distinct
// bank quads); the chunks a fragment reads past a row's HD columns belong to the next row - finite data that meets the zero
// padding of the Q fragments (QK^T) or lands in output columns >= HD that are never stored (PV).
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(1))) const void* gptr_a;
typedef __attribute__((address_space(3))) void* lptr_a;

template <int HD>
__global__ __launch_bounds__(512) void hiera_attn_win256_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ out, int n_windows, int heads) {
    using TR = HdTraits<HD>;
    constexpr int CK = TR::CK, DT = TR::DT;
    constexpr int ROWB = HD * 2;                    // dense row
    constexpr int KBYTES = 256 * ROWB;              // 36 KB for HD = 72
    constexpr int NINST = KBYTES / 1024;            // wave-instructions per operand
    constexpr int BUF = 2 * KBYTES + 256;           // K image, V image, slack for the over-reads behind V's last rows
    static_assert(KBYTES % 1024 == 0 && NINST % 4 == 0, "dense K/V image must split into whole 1-KB pieces");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fi = lane & 15, fg = lane >> 4;
    // PERSISTENT: blocks b and b + 8 share an XCD; the (window, head) tasks of the windows w % 8 == b % 8 are dealt round-robin
    // to the gridDim.x / 8 blocks of that XCD, so the 128-B lines that neighbouring heads share are fetched into one L2 only.
    const int xcd = blockIdx.x & 7, j0 = blockIdx.x >> 3, nj = gridDim.x >> 3;
    const int nw_local = n_windows > xcd ? (n_windows - xcd + 7) >> 3 : 0;
    const int ntask = nw_local * heads;
    const int64_t rs = 3 * (int64_t)heads * HD;
    const int64_t os = (int64_t)heads * HD;
    const int q0 = wave * 32;
    if (j0 >= ntask) return;
    if (tid < 32) reinterpret_cast<uint4*>(smem + (tid >> 4) * BUF + 2 * KBYTES)[tid & 15] = make_uint4(0, 0, 0, 0);

    auto task_ptr = [&](int u, int* h) -> int64_t {     // first token of the task's window
        const int wl = u / heads;
        *h = u - wl * heads;
        return (int64_t)(wl * 8 + xcd) * 256;
    };
    // the window's K and V rows of head h, global -> LDS: 2 * NINST 1-KB pieces, NINST / 4 of each operand per wave
    // j0 .. j1: which of this wave's (NINST + 7) / 8 piece pairs (the next task's image is issued in four parts BETWEEN the MFMA groups
    // of the current task: an LDS-DMA issue stalls the issuing wave for 60-185 cycles, which a burst of 10 at the task's start pays in
    // full while the matrix cores idle - behind a wave's own MFMAs in flight the stall is hidden)
    auto issue = [&](int64_t tok0, int h, char* buf, int j0 = 0, int j1 = 1000) {
#pragma unroll
        for (int j = 0; j < (NINST + 7) / 8; ++j) {
            if (j < j0 || j >= j1) continue;
            const int piece = wave + 8 * j;
            if (piece < NINST) {
                const int o = piece * 1024 + lane * 16;
                const int row = o / ROWB, ch = (o - row * ROWB) >> 4;
                const bf16_t* src = qkv + (tok0 + row) * rs + h * HD + ch * 8;
                __builtin_amdgcn_global_load_lds((gptr_a)(src + os), (lptr_a)(buf + piece * 1024), 16, 0, 0);
                __builtin_amdgcn_global_load_lds((gptr_a)(src + 2 * os), (lptr_a)(buf + KBYTES + piece * 1024), 16, 0, 0);
            }
        }
    };
    // Q fragments of a task: asynchronous 16-B loads in inline asm, so that hipcc neither waits for them at the (distant) first
    // use with vmcnt(0) nor orders them against the LDS-DMA; columns >= HD are zeroed after the hand-placed wait
    auto q_issue = [&](int64_t tok0, int h, u32x4 (&q)[2][CK]) {
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int c = 0; c < CK; ++c) {
                const int hdoff = 32 * c + 8 * fg;
                const bf16_t* src = qkv + (tok0 + q0 + 16 * t + fi) * rs + h * HD + (hdoff < HD ? hdoff : 0);
                asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(q[t][c]) : "v"(src) : "memory");
            }
    };
    int h;
    int64_t tok0 = task_ptr(j0, &h);
    issue(tok0, h, smem);
    u32x4 qraw[2][CK];
    q_issue(tok0, h, qraw);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const float sc = TR::SCALE;
    int it = 0;
    for (int u = j0; u < ntask; u += nj, ++it) {
        char* buf = smem + (it & 1) * BUF;
        // this wave's pieces of the task's K/V image and its Q fragments have landed (wait at the end of the previous iteration);
        // the barrier makes the whole image visible and says that every wave is done reading the other buffer, which is re-filled next
        // (raw s_barrier: __syncthreads() would also drain the output stores still in flight with vmcnt(0))
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        op16x8 qc[2][CK];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int c = 0; c < CK; ++c) {
                const u32x4 z4 = {0u, 0u, 0u, 0u};
                qc[t][c] = __builtin_bit_cast(op16x8, (32 * c + 8 * fg) < HD ? qraw[t][c] : z4);
            }
        const int64_t tok_c = tok0;
        const int h_c = h;
        const bool more = u + nj < ntask;
        char* nbuf = smem + ((it + 1) & 1) * BUF;
        if (more) {                                   // next task: Q fragments into registers now, the K/V image in parts inside the key loop
            tok0 = task_ptr(u + nj, &h);
            q_issue(tok0, h, qraw);
        }
        float m[2], l[2];
        f32x4 o[2][DT];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            m[t] = -3.0e38f;
            l[t] = 0.f;
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) o[t][dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll 1
        for (int kb = 0; kb < 2; ++kb) {
            const char* kbase = buf + kb * 128 * ROWB;
            const char* vbase = buf + KBYTES + kb * 128 * ROWB;
            f32x4 s[2][8];
#pragma unroll
            for (int kt = 0; kt < 8; ++kt) {
                op16x8 kf[CK];
#pragma unroll
                for (int c = 0; c < CK; ++c) kf[c] = *reinterpret_cast<const op16x8*>(kbase + (16 * kt + fi) * ROWB + (4 * c + fg) * 16);
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    s[t][kt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int c = 0; c < CK; ++c) s[t][kt] = MFMA_16x16x32(kf[c], qc[t][c], s[t][kt], 0, 0, 0);
                }
            }
            constexpr int NJ = (NINST + 7) / 8;
            // (placement measured on the stage-2 shape: 125 us per launch against 140 with all pairs at the task's start; pairs 1 | 1 | 2 | rest)
            if (more) { if (kb == 0) issue(tok0, h, nbuf, 0, 1); else issue(tok0, h, nbuf, 2, NJ > 4 ? 4 : NJ); }
            op16x8 pf[2][4];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                float mx = -3.0e38f;
#pragma unroll
                for (int kt = 0; kt < 8; ++kt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) mx = fmaxf(mx, s[t][kt][r]);
                mx = xor32_max(xor16_max(mx));
                const float mn = fmaxf(m[t], mx * sc);
                const float alpha = __builtin_amdgcn_exp2f(m[t] - mn);
                m[t] = mn;
                float sum = 0.f;
#pragma unroll
                for (int kt = 0; kt < 8; ++kt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float e = __builtin_amdgcn_exp2f(fmaf(s[t][kt][r], sc, -mn));
                        s[t][kt][r] = e;
                        sum += e;
                    }
                sum = xor32_sum(xor16_sum(sum));
                l[t] = l[t] * alpha + sum;
#pragma unroll
                for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) o[t][dt][r] *= alpha;
#pragma unroll
                for (int ks = 0; ks < 4; ++ks)
                    pf[t][ks] = pack8(s[t][2 * ks][0], s[t][2 * ks][1], s[t][2 * ks][2], s[t][2 * ks][3], s[t][2 * ks + 1][0],
                                      s[t][2 * ks + 1][1], s[t][2 * ks + 1][2], s[t][2 * ks + 1][3]);
            }
            if (more) { if (kb == 0) issue(tok0, h, nbuf, 1, 2); else issue(tok0, h, nbuf, NJ > 4 ? 4 : NJ, NJ); }
            // V^T fragments through inline asm: a compiler-visible ds_read_b64_tr_b16 is ordered behind the direct-to-LDS loads
            // of the NEXT task with s_waitcnt vmcnt(0), which would serialise the prefetch with this task's PV products
            const uint32_t va = (uint32_t)(uintptr_t)(lptr_a)vbase + (4 * fg + (fi >> 2)) * ROWB + 8 * (fi & 3);
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                u32x2 lo[4], hi[4];
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(lo[ks]) : "v"(va + 32 * ks * ROWB + 32 * dt) : "memory");
                    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(hi[ks]) : "v"(va + (32 * ks + 16) * ROWB + 32 * dt) : "memory");
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    const op16x8 vf = cat4(__builtin_bit_cast(op16x4, lo[ks]), __builtin_bit_cast(op16x4, hi[ks]));
#pragma unroll
                    for (int t = 0; t < 2; ++t) o[t][dt] = MFMA_16x16x32(vf, pf[t][ks], o[t][dt], 0, 0, 0);
                }
            }
        }
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const float inv = __builtin_amdgcn_rcpf(l[t]);
            const int row = q0 + 16 * t + fi;
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                const int d = 16 * dt + 4 * fg;
                if (d < HD)
                    *reinterpret_cast<uint2*>(out + (tok_c + row) * os + h_c * HD + d) =
                        make_uint2(pack_op16(o[t][dt][0] * inv, o[t][dt][1] * inv), pack_op16(o[t][dt][2] * inv, o[t][dt][3] * inv));
            }
        }
        // the next task's loads were issued before this task's 2 * DT output stores: all but the youngest 2 * DT operations done
        // = image and Q fragments landed, stores still draining
        if (more) {
            if (DT == 5) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    }
}
// ------------------------------------------------------------------------------------------------
// Global attention of Hiera-L (blocks 23, 33, 43: every image is ONE window of 4096 keys) and any other window whose key
// count is a multiple of 256.  Same building blocks as the 256-key kernel - dense direct-to-LDS K/V images, 8 waves x 32
// queries, asynchronous Q loads, inline-asm V^T reads - arranged as ONE CONTINUOUS STREAM of 128-key blocks through a
// 3-stage ring: a task is (window, head, 256-query chunk), its key blocks follow each other in the stream and the first
// blocks of the next task are already in flight while the current task finishes (online softmax across blocks, one raw
// barrier per block).  The register-staged kernel above keeps 64 queries per workgroup, re-stages K/V through VGPRs with two
// barriers per block and runs at ~340 TFLOP/s on this shape.
#define ST_KB 128
template <int HD>
__global__ __launch_bounds__(512) void hiera_attn_stream_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ out, int n_windows, int nk, int heads) {
    using TR = HdTraits<HD>;
    constexpr int CK = TR::CK, DT = TR::DT;
    constexpr int ROWB = HD * 2;                    // dense row
    constexpr int KBYTES = ST_KB * ROWB;            // 18 KB for HD = 72
    constexpr int NPIECE = 2 * KBYTES / 1024;       // 1-KB pieces of one stream item (K image then V image)
    constexpr int PPW = (NPIECE + 7) / 8;           // pieces per wave (the surplus ones repeat the last piece: same bytes, same place)
    constexpr int STAGE = 2 * KBYTES + 256;         // + slack for the over-reads behind V's last rows
    constexpr int NQL = 2 * CK;                     // Q loads per wave and task
    static_assert((2 * KBYTES) % 1024 == 0, "K/V image must split into whole 1-KB pieces");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fi = lane & 15, fg = lane >> 4;
    const int chunks = nk / 256, nkb = nk / ST_KB;
    // tasks of this XCD (blocks b, b + 8, ... share an L2): with one chunk per (window, head) the heads of a window stay together
    // (their 144-B slices share 128-B lines); with several chunks all chunks of a (window, head) pair do (they share its K/V)
    const int xcd = blockIdx.x & 7, j0 = blockIdx.x >> 3, nj = gridDim.x >> 3;
    const int npairs = n_windows * heads;
    const int nloc = chunks == 1 ? (n_windows > xcd ? (n_windows - xcd + 7) >> 3 : 0) * heads : (npairs > xcd ? (npairs - xcd + 7) >> 3 : 0);
    const int ntask = nloc * chunks;
    if (j0 >= ntask) return;
    const int64_t rs = 3 * (int64_t)heads * HD;
    const int64_t os = (int64_t)heads * HD;
    const int q0 = wave * 32;
    if (tid < 48) reinterpret_cast<uint4*>(smem + (tid >> 4) * STAGE + 2 * KBYTES)[tid & 15] = make_uint4(0, 0, 0, 0);

    auto task_of = [&](int t, int* h, int* qbase) -> int64_t {      // first token of the task's window, head, first query of the chunk
        const int ul = t / chunks, c = t - ul * chunks;
        int w;
        if (chunks == 1) { const int wl = ul / heads; *h = ul - wl * heads; w = wl * 8 + xcd; }
        else { const int u = ul * 8 + xcd; w = u / heads; *h = u - w * heads; }
        *qbase = c * 256;
        return (int64_t)w * nk;
    };
    auto issue_item = [&](int64_t tok0, int h, int kb, int stage) {
        char* buf = smem + stage * STAGE;
#pragma unroll
        for (int j = 0; j < PPW; ++j) {
            const int piece = min(wave + 8 * j, NPIECE - 1);
            const int o = (piece * 1024 + lane * 16) % KBYTES;
            const int row = o / ROWB, ch = (o - row * ROWB) >> 4;
            const bf16_t* src = qkv + (tok0 + kb * ST_KB + row) * rs + h * HD + ch * 8 + (piece * 1024 < KBYTES ? os : 2 * os);
            __builtin_amdgcn_global_load_lds((gptr_a)src, (lptr_a)(buf + piece * 1024), 16, 0, 0);
        }
    };
    auto q_issue = [&](int64_t tok0, int h, int qbase, u32x4 (&q)[2][CK]) {
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int c = 0; c < CK; ++c) {
                const int hdoff = 32 * c + 8 * fg;
                const bf16_t* src = qkv + (tok0 + qbase + q0 + 16 * t + fi) * rs + h * HD + (hdoff < HD ? hdoff : 0);
                asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(q[t][c]) : "v"(src) : "memory");
            }
    };
    // issue cursor
    int ti = j0, kbi = 0, hi, qbi;
    int64_t toki = task_of(ti, &hi, &qbi);
    int si = 0;
    u32x4 qraw[2][CK];
    // ops (this wave) of the stream items in flight, oldest first: PPW, or PPW + NQL for the first block of a task
    int n1 = 0, n2 = 0;                               // items i+1 and i+2 relative to the item being computed
    auto issue_next = [&]() -> int {                  // issue the item under the cursor, advance, return its op count (0 if the stream has ended)
        if (ti >= ntask) return 0;
        issue_item(toki, hi, kbi, si);
        int ops = PPW;
        if (kbi == 0) { q_issue(toki, hi, qbi, qraw); ops += NQL; }
        si = si == 2 ? 0 : si + 1;
        if (++kbi == nkb) { kbi = 0; ti += nj; if (ti < ntask) toki = task_of(ti, &hi, &qbi); }
        return ops;
    };
    auto wait_younger = [&](int younger) {            // all operations except the `younger` most recent ones have completed
        if (younger == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (younger == PPW) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPW) : "memory");
        else if (younger == PPW + NQL) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPW + NQL) : "memory");
        else if (younger == 2 * PPW) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PPW) : "memory");
        else if (younger == 2 * PPW + NQL) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PPW + NQL) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // not reachable (a task has >= 2 blocks): stay safe
    };
    // compute cursor
    int tc = j0, kbc = 0, hc, qbc, sc_ = 0;
    int64_t tokc = task_of(tc, &hc, &qbc);
    (void)issue_next();                               // item 0 (with the first task's Q)
    n1 = issue_next();                                // item 1
    wait_younger(n1);
    op16x8 qc[2][CK];
    float m[2], l[2];
    f32x4 o[2][DT];
    const float sc = TR::SCALE;
    while (tc < ntask) {
        // item (tc, kbc) has landed for this wave (wait at the end of the previous iteration); the barrier makes it visible to
        // everyone and says every wave is done with the stage that is re-filled next
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (kbc == 0) {                               // a new task starts: its Q fragments arrived together with its first block
#pragma unroll
            for (int t = 0; t < 2; ++t) {
#pragma unroll
                for (int c = 0; c < CK; ++c) {
                    const u32x4 z4 = {0u, 0u, 0u, 0u};
                    qc[t][c] = __builtin_bit_cast(op16x8, (32 * c + 8 * fg) < HD ? qraw[t][c] : z4);
                }
                m[t] = -3.0e38f;
                l[t] = 0.f;
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) o[t][dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
            }
        }
        n2 = issue_next();                            // item i + 2 into the stage of item i - 1
        const char* kbase = smem + sc_ * STAGE;
        const char* vbase = kbase + KBYTES;
        {
            f32x4 s[2][8];
#pragma unroll
            for (int kt = 0; kt < 8; ++kt) {
                op16x8 kf[CK];
#pragma unroll
                for (int c = 0; c < CK; ++c) kf[c] = *reinterpret_cast<const op16x8*>(kbase + (16 * kt + fi) * ROWB + (4 * c + fg) * 16);
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    s[t][kt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int c = 0; c < CK; ++c) s[t][kt] = MFMA_16x16x32(kf[c], qc[t][c], s[t][kt], 0, 0, 0);
                }
            }
            op16x8 pf[2][4];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                float mx = -3.0e38f;
#pragma unroll
                for (int kt = 0; kt < 8; ++kt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) mx = fmaxf(mx, s[t][kt][r]);
                mx = xor32_max(xor16_max(mx));
                const float mn = fmaxf(m[t], mx * sc);
                const float alpha = __builtin_amdgcn_exp2f(m[t] - mn);
                m[t] = mn;
                float sum = 0.f;
#pragma unroll
                for (int kt = 0; kt < 8; ++kt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float e = __builtin_amdgcn_exp2f(fmaf(s[t][kt][r], sc, -mn));
                        s[t][kt][r] = e;
                        sum += e;
                    }
                sum = xor32_sum(xor16_sum(sum));
                l[t] = l[t] * alpha + sum;
#pragma unroll
                for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) o[t][dt][r] *= alpha;
#pragma unroll
                for (int ks = 0; ks < 4; ++ks)
                    pf[t][ks] = pack8(s[t][2 * ks][0], s[t][2 * ks][1], s[t][2 * ks][2], s[t][2 * ks][3], s[t][2 * ks + 1][0],
                                      s[t][2 * ks + 1][1], s[t][2 * ks + 1][2], s[t][2 * ks + 1][3]);
            }
            const uint32_t va = (uint32_t)(uintptr_t)(lptr_a)vbase + (4 * fg + (fi >> 2)) * ROWB + 8 * (fi & 3);
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                u32x2 lo[4], hi2[4];
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(lo[ks]) : "v"(va + 32 * ks * ROWB + 32 * dt) : "memory");
                    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(hi2[ks]) : "v"(va + (32 * ks + 16) * ROWB + 32 * dt) : "memory");
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    const op16x8 vf = cat4(__builtin_bit_cast(op16x4, lo[ks]), __builtin_bit_cast(op16x4, hi2[ks]));
#pragma unroll
                    for (int t = 0; t < 2; ++t) o[t][dt] = MFMA_16x16x32(vf, pf[t][ks], o[t][dt], 0, 0, 0);
                }
            }
        }
        // the next item must have landed before the next barrier; the one after it (and its Q loads) may stay in flight.  The wait
        // precedes the output stores of a finished task so that they are not waited for here.
        wait_younger(n2);
        n1 = n2;
        sc_ = sc_ == 2 ? 0 : sc_ + 1;
        if (++kbc == nkb) {
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const float inv = __builtin_amdgcn_rcpf(l[t]);
                const int row = qbc + q0 + 16 * t + fi;
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) {
                    const int d = 16 * dt + 4 * fg;
                    if (d < HD)
                        *reinterpret_cast<uint2*>(out + (tokc + row) * os + hc * HD + d) =
                            make_uint2(pack_op16(o[t][dt][0] * inv, o[t][dt][1] * inv), pack_op16(o[t][dt][2] * inv, o[t][dt][3] * inv));
                }
            }
            kbc = 0;
            tc += nj;
            if (tc < ntask) tokc = task_of(tc, &hc, &qbc);
        }
    }
}
#define STREAM_LDS(HD) (3 * (2 * ST_KB * (HD) * 2 + 256))

#define WIN256_LDS(HD) (2 * (2 * 256 * (HD) * 2 + 256))

template <int HD> static hipError_t attn_attrs() {
    constexpr int lds = K_LDS_BYTES + KB * HdTraits<HD>::VSTRIDE;
    hipError_t st = hipFuncSetAttribute(reinterpret_cast<const void*>(hiera_attn_large_kernel<1, HD, false>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (st == hipSuccess) st = hipFuncSetAttribute(reinterpret_cast<const void*>(hiera_attn_large_kernel<1, HD, true>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if constexpr (WIN256_LDS(HD) <= 160 * 1024) {
        if (st == hipSuccess) st = hipFuncSetAttribute(reinterpret_cast<const void*>(hiera_attn_win256_kernel<HD>), hipFuncAttributeMaxDynamicSharedMemorySize, WIN256_LDS(HD));
        if constexpr (HD == 72)
            if (st == hipSuccess) st = hipFuncSetAttribute(reinterpret_cast<const void*>(hiera_attn_stream_kernel<HD>), hipFuncAttributeMaxDynamicSharedMemorySize, STREAM_LDS(HD));
    }
    return st;
}

const char* hiera_attention_init_device() {
    hipError_t st = attn_attrs<72>();
    if (st == hipSuccess) st = attn_attrs<96>();
    if (st == hipSuccess) st = attn_attrs<56>();
    return st == hipSuccess ? nullptr : hipGetErrorString(st);
}

template <int HD>
static const char* launch_hd(const bf16_t* qkv, bf16_t* out, int n_windows, int nk, int heads, int q_pool, const uint8_t* kmask, hipStream_t s) {
    if (nk <= 64 && !kmask) {
        if (q_pool && (nk & 3)) return "hiera_attention: q_pool needs nk % 4 == 0";
        const int tasks = n_windows * heads;
        const dim3 grid((tasks + 3) / 4);
        if (nk <= 16) hipLaunchKernelGGL((hiera_attn_small_kernel<16, HD>), grid, dim3(256), 0, s, qkv, out, n_windows, nk, heads, q_pool);
        else hipLaunchKernelGGL((hiera_attn_small_kernel<64, HD>), grid, dim3(256), 0, s, qkv, out, n_windows, nk, heads, q_pool);
        return nullptr;
    }
    if (q_pool && (nk & 3)) return "hiera_attention: q_pool needs nk % 4 == 0";
    const int nq = q_pool ? nk / 4 : nk;
    if constexpr (WIN256_LDS(HD) <= 160 * 1024) if (nk == 256 && !q_pool && !kmask && !(g_saber_debug_flags & (32 | 2))) {
        const int tasks_per_xcd = ((n_windows + 7) / 8) * heads;       // persistent: one workgroup per CU, fewer when there is less work
        hipLaunchKernelGGL((hiera_attn_win256_kernel<HD>), dim3(8 * (tasks_per_xcd < 32 ? tasks_per_xcd : 32)), dim3(512), WIN256_LDS(HD), s, qkv, out, n_windows, heads);
        return nullptr;
    }
    if constexpr (HD == 72) if (nk >= 256 && (nk % 256) == 0 && !q_pool && !kmask && !(g_saber_debug_flags & 32)) {
        const int chunks = nk / 256;
        const int loc = ((n_windows * heads + 7) / 8) * chunks;          // tasks of the busiest XCD
        hipLaunchKernelGGL((hiera_attn_stream_kernel<HD>), dim3(8 * (loc < 32 ? loc : 32)), dim3(512), STREAM_LDS(HD), s, qkv, out, n_windows, nk, heads);
        return nullptr;
    }
    constexpr int lds = K_LDS_BYTES + KB * HdTraits<HD>::VSTRIDE;
    const dim3 grid(((n_windows + 7) / 8) * 8 * heads * ((nq + 63) / 64));
    if (kmask || (nk % KB) != 0 || (nq % 64) != 0)
        hipLaunchKernelGGL((hiera_attn_large_kernel<1, HD, true>), grid, dim3(256), lds, s, qkv, out, n_windows, nk, heads, q_pool, kmask);
    else
        hipLaunchKernelGGL((hiera_attn_large_kernel<1, HD, false>), grid, dim3(256), lds, s, qkv, out, n_windows, nk, heads, q_pool, kmask);
    return nullptr;
}

const char* launch_hiera_attention(const bf16_t* qkv, bf16_t* out, int n_windows, int nk, int heads, int hd, int q_pool,
                                   const uint8_t* kmask, hipStream_t s) {
    if (n_windows <= 0) return nullptr;
    if (nk <= 0 || heads <= 0) return "hiera_attention: bad shape";
    if (((uintptr_t)qkv & 15) || ((uintptr_t)out & 7)) return "hiera_attention: pointer alignment";
    if (kmask && ((uintptr_t)kmask & 3)) return "hiera_attention: key mask must be 4-byte aligned";
    if (hd == 72) return launch_hd<72>(qkv, out, n_windows, nk, heads, q_pool, kmask, s);
    if (hd == 96) return launch_hd<96>(qkv, out, n_windows, nk, heads, q_pool, kmask, s);
    if (hd == 56) return launch_hd<56>(qkv, out, n_windows, nk, heads, q_pool, kmask, s);
    return "hiera_attention: head_dim must be 72 (large), 96 (tiny/small) or 56 (base+)";
}
