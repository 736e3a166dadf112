// Fused image-side kernels of the two-way transformer.
//
// The cross attentions between the 8 prompt tokens and the 4096 image tokens are algebraically folded so
// that the 4096x256 image-token matrix X of a prompt is streamed from HBM once per attention and never
// projected to q/k/v in memory:
//
//  tokens -> image  (cross_attn_token_to_image, final_attn_token_to_image), head h, token t, c = 8h + t:
//      score[c][n] = q_{t,h} . (W_k (x_n + pe_n) + b_k)_h = Qt[c] . (x_n + pe_n) + const(c)    Qt[c] = s W_k,h^T q_{t,h}
//      (the constant is softmax-invariant over n and dropped);  out_{t,h} = W_v,h (sum_n p[c][n] x_n) + b_v,h
//      => one "attention" with 64 query rows of dimension 256 over keys X+PE and values X      (dec_t2i_kernel)
//  image -> tokens  (cross_attn_image_to_token) + residual + LayerNorm (norm4):
//      score[n][c] = (x_n + pe_n) . Kt[c] + cb[c],  softmax over the 8 tokens of each head,
//      y_n = sum_c p[n][c] Vt[c] + b_o,   Kt[c] = s W_q,h^T k_{t,h},  cb[c] = s b_q,h . k_{t,h},  Vt[c] = W_o[:,h] v_{t,h}
//      x_n <- LN(x_n + y_n)                                                                      (dec_i2t_kernel)
//
// Same arithmetic as the unfolded form up to fp reassociation; 1.9 instead of 3.6 GFLOP per prompt and ~17 MB
// instead of ~60 MB of HBM traffic per prompt.  MFMA layouts follow attention_hiera.hip (S^T = K.Q^T,
// O^T = V^T.P^T, V^T fragments via ds_read_b64_tr_b16).
#include "common.h"
#include "kernels.h"
#include <utility>

// cache policy of dec_i2t's read of a prompt's image-token state, its LAST use (2 = nontemporal: same-box A/B 20.55 -> 19.86 ms per slice;
// the same hint on dec_upscale's reads cost it 0.5 ms)
#ifndef I2T_X_AUX
#define I2T_X_AUX 2
#endif
// the same hint on dec_t2i's reads (the 2 GB of a decode batch never survive in the caches until dec_i2t reads them again): 16.38 -> 16.02 ms;
// nontemporal STORES of dec_upscale's logits: +2 ms (dropped)
#ifndef T2I_X_AUX
#define T2I_X_AUX 2
#endif

typedef s16x4 __attribute__((address_space(3))) * lds_s16x4_ptr;
typedef unsigned int u32x2_d __attribute__((ext_vector_type(2)));
__device__ __forceinline__ op16x4 tr_read_d(const char* p) {
    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(p));
    return __builtin_bit_cast(op16x4, v);
}
__device__ __forceinline__ op16x8 cat4_d(op16x4 a, op16x4 b) { return __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7); }
__device__ __forceinline__ op16x8 pack8_d(float a0, float a1, float a2, float a3, float b0, float b1, float b2, float b3) {
    uint4 u = make_uint4(pack_op16(a0, a1), pack_op16(a2, a3), pack_op16(b0, b1), pack_op16(b2, b3));
    return __builtin_bit_cast(op16x8, u);
}
// 8 packed bf16 + 8 packed bf16 -> 8 packed bf16 (fp32 add, RNE)
__device__ __forceinline__ u32x4 add_bf16x8(u32x4 a, u32x4 b) {
    u32x4 r;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float lo = op16_lo(a[i]) + op16_lo(b[i]);
        const float hi = op16_hi(a[i]) + op16_hi(b[i]);
        r[i] = pack_op16(lo, hi);
    }
    return r;
}

#define DC 256           // channels of the decoder
#define ROW_B 512        // bytes per 256-channel bf16 row
#define VT_STRIDE 544    // padded LDS row stride for tr reads: 8 rows x 32 B hit disjoint banks
__device__ __forceinline__ int kswz(int row, int chunk) { return row * ROW_B + ((chunk ^ (row & 15)) << 4); }

// ------------------------------------------------------------------------------------------------ fold
// out[p][8h+t][d] = scale * sum_j a[p][t][16h+j] * W(16h+j, d);  W row-major [128][256] (mode 0: k_proj / q_proj
// weight, out [p][64][256]) or [256][128] indexed W[d][16h+j] (mode 1: out_proj weight, out TRANSPOSED [p][256][64]).  cb[p][8h+t] = scale * sum_j a . bias[16h+j].
__global__ __launch_bounds__(256) void dec_fold_kernel(const float* __restrict__ a, const bf16_t* __restrict__ W, const float* __restrict__ bias,
                                                       int mode, float scale, bf16_t* __restrict__ out, float* __restrict__ cb) {
    __shared__ __attribute__((aligned(16))) float as[8 * 128];
    const int p = blockIdx.x, d = threadIdx.x;
    reinterpret_cast<float4*>(as)[d] = reinterpret_cast<const float4*>(a + (int64_t)p * 1024)[d];
    __syncthreads();
    if (mode == 0) {
        // thread = (head h, 8 consecutive output columns): 16 coalesced 16-byte loads of W and 8 16-byte stores per thread (one thread
        // per column did 128 two-byte loads and 64 two-byte stores: the kernel was bound by their issue, 25-40 us per launch)
        const int h = d >> 5, c0 = (d & 31) * 8;
        float acc[8][8];
#pragma unroll
        for (int t = 0; t < 8; ++t)
#pragma unroll
            for (int c = 0; c < 8; ++c) acc[t][c] = 0.f;
#pragma unroll
        for (int j4 = 0; j4 < 4; ++j4) {
            float w[4][8];
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const uint4 u = *reinterpret_cast<const uint4*>(W + (16 * h + 4 * j4 + jj) * 256 + c0);
                const uint32_t uu[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
                for (int k = 0; k < 4; ++k) { w[jj][2 * k] = op16_lo(uu[k]); w[jj][2 * k + 1] = op16_hi(uu[k]); }
            }
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const float4 av = *reinterpret_cast<const float4*>(as + t * 128 + 16 * h + 4 * j4);      // broadcast read
#pragma unroll
                for (int c = 0; c < 8; ++c) acc[t][c] += av.x * w[0][c] + av.y * w[1][c] + av.z * w[2][c] + av.w * w[3][c];
            }
        }
#pragma unroll
        for (int t = 0; t < 8; ++t)
            *reinterpret_cast<uint4*>(out + ((int64_t)p * 64 + 8 * h + t) * 256 + c0) =
                make_uint4(pack_op16(acc[t][0] * scale, acc[t][1] * scale), pack_op16(acc[t][2] * scale, acc[t][3] * scale),
                           pack_op16(acc[t][4] * scale, acc[t][5] * scale), pack_op16(acc[t][6] * scale, acc[t][7] * scale));
    } else {
        for (int h = 0; h < 8; ++h) {
            float w[16];
            {           // thread d owns row d of W: 32 contiguous bytes per head
                const uint4 w0 = *reinterpret_cast<const uint4*>(W + d * 128 + 16 * h), w1 = *reinterpret_cast<const uint4*>(W + d * 128 + 16 * h + 8);
                const uint32_t ww[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
#pragma unroll
                for (int j = 0; j < 8; ++j) { w[2 * j] = op16_lo(ww[j]); w[2 * j + 1] = op16_hi(ww[j]); }
            }
            float r[8];
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                float acc = 0.f;
#pragma unroll
                for (int j4 = 0; j4 < 4; ++j4) {       // broadcast reads, 16 bytes at a time
                    const float4 av = *reinterpret_cast<const float4*>(as + t * 128 + 16 * h + 4 * j4);
                    acc += av.x * w[4 * j4] + av.y * w[4 * j4 + 1] + av.z * w[4 * j4 + 2] + av.w * w[4 * j4 + 3];
                }
                r[t] = acc * scale;
            }
            // Vt^T: rows = channel d, 64 folded columns -> the 8 tokens of head h are 16 contiguous bytes of row d
            *reinterpret_cast<uint4*>(out + ((int64_t)p * 256 + d) * 64 + 8 * h) =
                make_uint4(pack_op16(r[0], r[1]), pack_op16(r[2], r[3]), pack_op16(r[4], r[5]), pack_op16(r[6], r[7]));
        }
    }
    if (cb && d < 64) {
        const int h = d >> 3, t = d & 7;
        float acc = 0.f;
        for (int j = 0; j < 16; ++j) acc += as[t * 128 + 16 * h + j] * bias[16 * h + j];
        cb[(int64_t)p * 64 + d] = acc * scale;
    }
}

const char* launch_dec_fold(const float* a, const bf16_t* W, const float* bias, int mode, float scale, bf16_t* out, float* cb, int P,
                            hipStream_t s) {
    if (P <= 0) return nullptr;
    hipLaunchKernelGGL(dec_fold_kernel, dim3(P), dim3(256), 0, s, a, W, bias, mode, scale, out, cb);
    return nullptr;
}

// ------------------------------------------------------------------------------------------------ tokens -> image
// grid = P * split, 512 threads.  Each block: 64 folded query rows over 4096/split keys of prompt p, in 64-key blocks.
//  * score = Qt.x + Qt.pe.  Qt = W_k^T q (folded, per head) so the second term is q_h . (W_k pe)_h: a contraction of 16 per head
//    against PEK = pe W_k^T [4096][128], which depends on the model only and is computed once at finalize.  A q tile's 16 rows
//    are the 8 tokens of heads 2qt and 2qt+1, so ONE MFMA per 16 keys with the block-diagonal operand [q_2qt | 0 ; 0 | q_2qt+1]
//    (k = 32) adds the positional term: 18 MFMAs, 18 fragment reads and 48 KB of tile per 64-key block instead of 32, 32 and 64.
//  * both operand tiles are plain copies and go global -> LDS directly (global_load_lds_dwordx4, 2-stage ring, no staging
//    registers, no ds_write).
//  * one LDS image of the X tile (XOR-swizzled 512-B rows, source-side swizzle) serves the K operand (ds_read_b128 rows)
//    and the V operand (ds_read_b64_tr_b16 with the same XOR).
//  * 8 waves: wave = (key half kh) * 4 + (q tile); each wave keeps its own online-softmax partial over its 32 keys of
//    every block; the two halves are merged through LDS at the end.  Two waves per SIMD hide each other's LDS latency.
// Writes un-normalised partial O [P][split][64][256] and (m, l) [P][split][64][2] (log2 domain).
// NW = waves per workgroup.  NW = 8 (default): 64-key blocks, wave = (key half) * 4 + (q tile), 2-stage ring, the halves merged through LDS at
// the end.  NW = 4: 32-key blocks, wave = q tile over ALL keys (no merge), 3-stage ring of 24-KB stages, TWO workgroups per
// CU that share no barrier and drift apart (one's MFMA phases overlap the other's softmax / LDS-DMA issue phases).
#define T2I_PEK_ROWB 256                      // one PEK row: 128 bf16
template <int NW> struct T2ICfg {
    static constexpr int KB = 8 * NW;                                   // keys per block iteration
    static constexpr int STAGE = KB * ROW_B + KB * T2I_PEK_ROWB;        // X tile + PEK tile
    static constexpr int NST = NW == 4 ? 3 : 2;
    static constexpr int LDS = NST * STAGE;                             // NW = 4: 72 KB >= the 66.5 KB the v_proj tail needs
};
typedef __attribute__((address_space(1))) const void* gptr_d;
typedef __attribute__((address_space(3))) void* lptr_d;
__device__ __forceinline__ void lds_write_b64(uint32_t addr, uint32_t lo, uint32_t hi) {
    const uint64_t v = ((uint64_t)hi << 32) | lo;
    asm volatile("ds_write_b64 %0, %1" ::"v"(addr), "v"(v) : "memory");
}

// BUILD: the X tile of a key block is not copied from HBM but assembled by the waves themselves as bf16(embb + h2 . W3^T) (XBuild in
// kernels.h: the m2m prompts' src = image_embed + mask-prompt embedding, which then never exists in memory).  Wave w owns channels
// 32 w .. 32 w + 31 of all 64 keys: 2 x 4 MFMAs (k = 16 zero-padded to 32) whose C operand is the fp32 embb tile, loaded into registers at
// the top of the iteration for the NEXT block and written (bf16, the tile's swizzle) after this block's PV products.
template <int NW, bool STAMPS, bool BUILD>
__global__ __launch_bounds__(64 * NW) void dec_t2i_kernel(const bf16_t* __restrict__ X, int64_t x_bs, int x_div, int x_off,
                                                      const bf16_t* __restrict__ pek, const bf16_t* __restrict__ Qt,
                                                      const float* __restrict__ tq, float qscale,
                                                      float* __restrict__ Opart, float* __restrict__ ML, int split,
                                                      const bf16_t* __restrict__ Wv, const float* __restrict__ bv, bf16_t* __restrict__ out,
                                                      unsigned long long* __restrict__ stamps, const float* __restrict__ embb, const bf16_t* __restrict__ h2,
                                                      const float* __restrict__ w3) {
    static_assert(!BUILD || NW == 8, "the tile builder is written for the 8-wave form");
    constexpr int T2I_KB = T2ICfg<NW>::KB, T2I_STAGE = T2ICfg<NW>::STAGE, NST = T2ICfg<NW>::NST;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    unsigned long long ts[6] = {0, 0, 0, 0, 0, 0}, tprev = 0;   // development only (stamps != nullptr), see tools/dec_stamps.py
#define T2I_STAMP(k) do { if (STAMPS) { const unsigned long long _n = __builtin_amdgcn_s_memtime(); ts[k] += _n - tprev; tprev = _n; } } while (0)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int qt = wave & 3, kh = wave >> 2;
    const int fi = lane & 15, fg = lane >> 4;
    const int p = blockIdx.x / split, sp = blockIdx.x - p * split;
    const int nkeys = 4096 / split, key0 = sp * nkeys, nkb = nkeys / T2I_KB;
    const bf16_t* Xp = X + (BUILD ? 0 : (int64_t)((p + x_off) / x_div) * x_bs + (int64_t)key0 * DC);   // image tokens of prompt p (see kernels.h XMap)
    const bf16_t* Pp = pek + (int64_t)key0 * 128;
    // BUILD: embb rows of the prompt's slot (x_bs / x_div / x_off describe the slot map then), this wave's channels; h2 rows of the prompt
    // embb is stored in the builder's own order (launch_embb_tiles): [16-row tile][16-channel tile][lane][4 floats], so that one load
    // instruction of a wave reads one contiguous KB.  This wave: channel tiles 2 wave, 2 wave + 1 of the block's four row tiles.
    const float* Ep = BUILD ? embb + (int64_t)((p + x_off) / x_div) * x_bs + ((int64_t)(key0 / 16) * 16 + 2 * wave) * 256 + lane * 4 : nullptr;
    const bf16_t* Hp = BUILD ? h2 + ((int64_t)p * 4096 + key0 + fi) * 16 + 8 * (fg & 1) : nullptr;
    op16x8 w3f[2];
    if (BUILD) {
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
            float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            if (fg < 2) {
                const float* wr = w3 + (32 * wave + 16 * ct + fi) * 16 + 8 * fg;
                const float4 a = *reinterpret_cast<const float4*>(wr), b = *reinterpret_cast<const float4*>(wr + 4);
                v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
            }
            w3f[ct] = pack8_d(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7]);
        }
    }
    f32x4 eb[2][4];
    u32x4 hb[4];
    // The loads go through inline asm and are waited for by hand (build_write): left to the compiler, its wait lands wherever it
    // schedules the first register copy, and across the loop's back edge it is always vmcnt(0).
    auto load_regs = [&](int kb) {
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) {
            const float* ea = Ep + (int64_t)(kb * 4 + tt) * 16 * 256;
            asm volatile("global_load_dwordx4 %0, %2, off\n\tglobal_load_dwordx4 %1, %2, off offset:1024" : "=&v"(eb[0][tt]), "=&v"(eb[1][tt]) : "v"(ea) : "memory");
            const bf16_t* ha = Hp + (int64_t)(kb * T2I_KB + 16 * tt) * 16;
            asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(hb[tt]) : "v"(ha) : "memory");
        }
    };
    auto build_write = [&](int stage) {
        const uint32_t sx = (uint32_t)(uintptr_t)(lptr_d)(smem + stage * T2I_STAGE);
        // every load of this wave has landed (the registers are operands of the wait so that no use can be scheduled above it)
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(eb[0][0]), "+v"(eb[0][1]), "+v"(eb[0][2]), "+v"(eb[0][3]), "+v"(eb[1][0]), "+v"(eb[1][1]), "+v"(eb[1][2]), "+v"(eb[1][3]),
                     "+v"(hb[0]), "+v"(hb[1]), "+v"(hb[2]), "+v"(hb[3]) :: "memory");
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) {
            const op16x8 hf = __builtin_bit_cast(op16x8, fg < 2 ? hb[tt] : (u32x4){0u, 0u, 0u, 0u});
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) {
                const f32x4 a = MFMA_16x16x32(w3f[ct], hf, eb[ct][tt], 0, 0, 0);     // a[r] = X0[key 16 tt + fi][channel 32 wave + 16 ct + 4 fg + r]
                lds_write_b64(sx + (16 * tt + fi) * ROW_B + (((4 * wave + 2 * ct + (fg >> 1)) ^ fi) << 4) + (fg & 1) * 8, pack_op16(a[0], a[1]), pack_op16(a[2], a[3]));
            }
        }
    };

    op16x8 qf[8], pq;
    {
        const bf16_t* qrow = Qt + ((int64_t)p * 64 + qt * 16 + fi) * DC;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) qf[ks] = __builtin_bit_cast(op16x8, *reinterpret_cast<const uint4*>(qrow + 32 * ks + 8 * fg));
        // block-diagonal projected query: row fi = token (fi & 7) of head 2 qt + (fi >> 3); k = 8 fg .. 8 fg + 7 of [head A 16 | head B 16]
        const int hsel = fg >> 1;
        const float* qp = tq + ((int64_t)p * 8 + (fi & 7)) * 128 + 16 * (2 * qt + hsel) + 8 * (fg & 1);
        const float4 a = *reinterpret_cast<const float4*>(qp), b = *reinterpret_cast<const float4*>(qp + 4);
        const float z = ((fi >> 3) == hsel) ? qscale : 0.f;
        pq = pack8_d(a.x * z, a.y * z, a.z * z, a.w * z, b.x * z, b.y * z, b.z * z, b.w * z);
    }
    float m = -3.0e38f, l = 0.f;
    f32x4 o[16];
#pragma unroll
    for (int dt = 0; dt < 16; ++dt) o[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // this wave's 4 + 4 direct-to-LDS instructions per stage: instruction i covers rows 2i, 2i+1 (1 KB) of a 64-row tile
    int srow[4], schunk[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        srow[i] = 2 * (wave * 4 + i) + (lane >> 5);
        schunk[i] = (lane & 31) ^ (srow[i] & 15);          // logical chunk that must land at physical slot lane & 31
    }
    // PEK tile: 64 rows x 256 B = 16 pieces of 4 rows; wave w issues pieces 2w, 2w+1; 16-B chunks XOR-swizzled by (row & 15)
    int prow[2], pchunk[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        prow[i] = 4 * (wave * 2 + i) + (lane >> 4);
        pchunk[i] = (lane & 15) ^ (prow[i] & 15);
    }
    auto issue = [&](int kb, int stage) {
        char* sx = smem + stage * T2I_STAGE;
        if (!BUILD)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int64_t off = (int64_t)(kb * T2I_KB + srow[i]) * DC + schunk[i] * 8;
            // (a state shared by the prompts of a crop - x_div > 1, layer 0 of the first pass - is re-read from L2 by all of them: no hint)
            if (x_div > 1) __builtin_amdgcn_global_load_lds((gptr_d)(Xp + off), (lptr_d)(sx + (wave * 4 + i) * 1024), 16, 0, 0);
            else __builtin_amdgcn_global_load_lds((gptr_d)(Xp + off), (lptr_d)(sx + (wave * 4 + i) * 1024), 16, 0, T2I_X_AUX);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
            __builtin_amdgcn_global_load_lds((gptr_d)(Pp + (int64_t)(kb * T2I_KB + prow[i]) * 128 + pchunk[i] * 8),
                                             (lptr_d)(sx + T2I_KB * ROW_B + (wave * 2 + i) * 1024), 16, 0, 0);
    };
    // per-lane LDS offsets that do not depend on the block
    int koff[8];                                        // K-operand fragment of key row (16kt + fi), k-step ks
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) koff[ks] = fi * ROW_B + (((4 * ks + fg) ^ fi) << 4);
    const int vrow = 4 * fg + (fi >> 2);                // tr-read row within the 32-key half (+16 for the second read)
    const int vsel = (fi & 3) >> 1, vlow = (fi & 1) * 8;

    // NST - 1 tiles in flight; a tile = 6 operations per wave (4 X pieces + 2 PEK pieces)
    issue(0, 0);
    if (BUILD) { load_regs(0); build_write(0); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
    if (NST == 3 && nkb > 1) { issue(1, 1); asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); }
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (STAMPS) tprev = __builtin_amdgcn_s_memtime();
    int stage = 0;
    for (int kb = 0; kb < nkb; ++kb) {
        {
            const int kn = kb + NST - 1;                    // the stage it goes into was read in the previous iteration
            if (kn < nkb) { issue(kn, kn % NST); if (BUILD) load_regs(kn); }
        }
        T2I_STAMP(0);
        const char* xs = smem + stage * T2I_STAGE + kh * 32 * ROW_B;     // this wave's 32 keys
        const char* ps = smem + stage * T2I_STAGE + T2I_KB * ROW_B + kh * 32 * T2I_PEK_ROWB;
        f32x4 s[2];
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
            const op16x8 kp = *reinterpret_cast<const op16x8*>(ps + (kt * 16 + fi) * T2I_PEK_ROWB + (((4 * qt + fg) ^ fi) << 4));
            s[kt] = MFMA_16x16x32(kp, pq, (f32x4){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                const op16x8 kx = *reinterpret_cast<const op16x8*>(xs + kt * 16 * ROW_B + koff[ks]);
                s[kt] = MFMA_16x16x32(kx, qf[ks], s[kt], 0, 0, 0);
            }
        }
        T2I_STAMP(1);
        float mx = fmaxf(fmaxf(fmaxf(s[0][0], s[0][1]), fmaxf(s[0][2], s[0][3])), fmaxf(fmaxf(s[1][0], s[1][1]), fmaxf(s[1][2], s[1][3])));
        mx = xor32_max(xor16_max(mx));
        if (__any(mx > m)) {        // rescale only when some row's running maximum grows (wave-uniform branch)
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
        const op16x8 pf = pack8_d(s[0][0], s[0][1], s[0][2], s[0][3], s[1][0], s[1][1], s[1][2], s[1][3]);
        T2I_STAMP(2);
        // V^T fragments through inline asm, four output tiles per wait: a compiler-visible ds_read_b64_tr_b16 is ordered behind the
        // direct-to-LDS loads of the NEXT key block with s_waitcnt vmcnt(0) (the prefetch then ended between the softmax and
        // the PV products instead of at the end of the iteration; measured effect small: the block is bound by the issue of its two
        // waves per SIMD, ~1 500 cycles each per 64-key block, not by the wait)
        {
            const uint32_t va = (uint32_t)(uintptr_t)(lptr_d)(xs + vrow * ROW_B + vlow);
#pragma unroll
            for (int d4 = 0; d4 < 16; d4 += 4) {
                u32x2_d lo[4], hi[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int sw = ((2 * (d4 + j) + vsel) ^ (vrow & 15)) << 4;       // rows vrow and vrow+16 share (row & 15)
                    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(lo[j]) : "v"(va + sw) : "memory");
                    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:8192" : "=v"(hi[j]) : "v"(va + sw) : "memory");
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const op16x8 vf = cat4_d(__builtin_bit_cast(op16x4, lo[j]), __builtin_bit_cast(op16x4, hi[j]));
                    o[d4 + j] = MFMA_16x16x32(vf, pf, o[d4 + j], 0, 0, 0);
                }
            }
        }
        T2I_STAMP(3);
        if (BUILD && kb + 1 < nkb) build_write((kb + 1) % NST);      // (the compiler waits for the registers loaded at the top)
        // next block landed (this wave's part); with 3 stages the one after it stays in flight
        if (NST == 3 && kb + 2 < nkb) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        T2I_STAMP(4);
        __builtin_amdgcn_s_barrier();
        T2I_STAMP(5);
        stage = stage + 1 == NST ? 0 : stage + 1;
    }
    if (STAMPS && stamps && lane == 0)
        for (int k = 0; k < 6; ++k) stamps[((int64_t)blockIdx.x * NW + wave) * 6 + k] = ts[k];
    // merge the two key halves of each q tile: waves 4..7 park (m, l, O) in LDS, waves 0..3 combine and store
    float* mo = reinterpret_cast<float*>(smem) + (size_t)qt * 16 * 260;      // [16 q][256 + 4] floats per q tile
    if (NW == 8 && kh == 1) {
#pragma unroll
        for (int dt = 0; dt < 16; ++dt) *reinterpret_cast<float4*>(mo + fi * 260 + 16 * dt + 4 * fg) = make_float4(o[dt][0], o[dt][1], o[dt][2], o[dt][3]);
        if (fg == 0) { mo[fi * 260 + 256] = m; mo[fi * 260 + 257] = l; }
    }
    __syncthreads();
    if (NW == 4) {            // no second key half: the "other half" is empty
#pragma unroll
        for (int dt = 0; dt < 16; ++dt) *reinterpret_cast<float4*>(mo + fi * 260 + 16 * dt + 4 * fg) = make_float4(0.f, 0.f, 0.f, 0.f);
        if (fg == 0) { mo[fi * 260 + 256] = -3.0e38f; mo[fi * 260 + 257] = 0.f; }
        __builtin_amdgcn_wave_barrier();
    }
    if (kh == 0) {
        const float m2 = mo[fi * 260 + 256], l2 = mo[fi * 260 + 257];
        const float mn = fmaxf(m, m2);
        const float a1 = exp2f(m - mn), a2 = exp2f(m2 - mn);
        const int q = qt * 16 + fi;
        if (split > 1) {
            float* op = Opart + (((int64_t)p * split + sp) * 64 + q) * DC;
#pragma unroll
            for (int dt = 0; dt < 16; ++dt) {
                const float4 t = *reinterpret_cast<const float4*>(mo + fi * 260 + 16 * dt + 4 * fg);
                *reinterpret_cast<float4*>(op + 16 * dt + 4 * fg) = make_float4(o[dt][0] * a1 + t.x * a2, o[dt][1] * a1 + t.y * a2, o[dt][2] * a1 + t.z * a2, o[dt][3] * a1 + t.w * a2);
            }
            if (fg == 0) {
                float* mlp = ML + (((int64_t)p * split + sp) * 64 + q) * 2;
                mlp[0] = mn;
                mlp[1] = l * a1 + l2 * a2;
            }
        } else {
            // whole key range in this block: normalise and apply v_proj here (out[p][t][16h+i] = Wv[16h+i] . Z[8h+t] + bv[16h+i])
            // instead of a round trip of the partials through HBM and a second kernel.  The wave's 16 query rows are the 8 tokens
            // of heads 2qt and 2qt+1: Z goes back to LDS in fp32, is re-read as the bf16 B operand (k = channel d, n = row q),
            // and both heads' 16x256 slices of Wv multiply all 16 rows; each row keeps the tile of its own head.
            const float inv = __builtin_amdgcn_rcpf(l * a1 + l2 * a2);
#pragma unroll
            for (int dt = 0; dt < 16; ++dt) {
                const float4 t = *reinterpret_cast<const float4*>(mo + fi * 260 + 16 * dt + 4 * fg);
                *reinterpret_cast<float4*>(mo + fi * 260 + 16 * dt + 4 * fg) =
                    make_float4((o[dt][0] * a1 + t.x * a2) * inv, (o[dt][1] * a1 + t.y * a2) * inv, (o[dt][2] * a1 + t.z * a2) * inv, (o[dt][3] * a1 + t.w * a2) * inv);
            }
            __builtin_amdgcn_wave_barrier();
            f32x4 r0 = (f32x4){0.f, 0.f, 0.f, 0.f}, r1 = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                const float4 z0 = *reinterpret_cast<const float4*>(mo + fi * 260 + 32 * ks + 8 * fg), z1 = *reinterpret_cast<const float4*>(mo + fi * 260 + 32 * ks + 8 * fg + 4);
                const op16x8 zf = pack8_d(z0.x, z0.y, z0.z, z0.w, z1.x, z1.y, z1.z, z1.w);
                const op16x8 w0 = __builtin_bit_cast(op16x8, *reinterpret_cast<const uint4*>(Wv + (int64_t)(32 * qt + fi) * DC + 32 * ks + 8 * fg));
                const op16x8 w1 = __builtin_bit_cast(op16x8, *reinterpret_cast<const uint4*>(Wv + (int64_t)(32 * qt + 16 + fi) * DC + 32 * ks + 8 * fg));
                r0 = MFMA_16x16x32(w0, zf, r0, 0, 0, 0);
                r1 = MFMA_16x16x32(w1, zf, r1, 0, 0, 0);
            }
            const int hsel = fi >> 3, hh = 2 * qt + hsel, t = fi & 7;      // lane: row q = fi -> head hh, token t; outputs 16 hh + 4 fg + r
            const float4 b4 = *reinterpret_cast<const float4*>(bv + 16 * hh + 4 * fg);
            const f32x4 r = hsel ? r1 : r0;
            *reinterpret_cast<uint2*>(out + (int64_t)p * 1024 + t * 128 + 16 * hh + 4 * fg) =
                make_uint2(pack_op16(r[0] + b4.x, r[1] + b4.y), pack_op16(r[2] + b4.z, r[3] + b4.w));
        }
    }
}

// combine the split partials and apply v_proj per head: out[p][t][16h+i] = W_v[16h+i] . Z[p][8h+t] + b_v[16h+i]  (bf16)
// one block per (prompt, head)
__global__ __launch_bounds__(256) void dec_t2i_finish_kernel(const float* __restrict__ Opart, const float* __restrict__ ML, int split,
                                                             const bf16_t* __restrict__ Wv, const float* __restrict__ bv,
                                                             bf16_t* __restrict__ out) {
    __shared__ float z[8][DC + 1];
    __shared__ float wgt[8][8];
    const int p = blockIdx.x >> 3, h = blockIdx.x & 7, tid = threadIdx.x;
    if (tid < 8) {
        const int c = 8 * h + tid;
        float mm = -3.0e38f;
        for (int s = 0; s < split; ++s) mm = fmaxf(mm, ML[(((int64_t)p * split + s) * 64 + c) * 2]);
        float L = 0.f;
        for (int s = 0; s < split; ++s) {
            const float* mlp = ML + (((int64_t)p * split + s) * 64 + c) * 2;
            const float w = exp2f(mlp[0] - mm);
            wgt[tid][s] = w;
            L += w * mlp[1];
        }
        const float inv = 1.0f / L;
        for (int s = 0; s < split; ++s) wgt[tid][s] *= inv;
    }
    __syncthreads();
    for (int t = 0; t < 8; ++t) {
        float acc = 0.f;
        for (int s = 0; s < split; ++s) acc += wgt[t][s] * Opart[(((int64_t)p * split + s) * 64 + 8 * h + t) * DC + tid];
        z[t][tid] = acc;
    }
    __syncthreads();
    if (tid < 128) {  // (t, i): out feature 16h + i of token t
        const int t = tid >> 4, oo = 16 * h + (tid & 15);
        const bf16_t* w = Wv + oo * 256;
        float acc = bv[oo];
        for (int d = 0; d < DC; ++d) acc += op2f(w[d]) * z[t][d];
        out[(int64_t)p * 1024 + t * 128 + oo] = f2op(acc);
    }
}

// ------------------------------------------------------------------------------------------------ tokens -> image, one wave per SIMD (round 5)
// dec_t2i_kernel reads every key fragment once per QUERY TILE (its waves are 4 query tiles x 2 key halves): 272 KB of LDS reads and 100
// LDS instructions per 64-key block for 272 MFMAs, and the chain read -> wait -> MFMA of each wave is short (34 MFMAs per step).  Here a
// wave owns a key QUARTER and all FOUR query tiles: a key / value fragment is read once and feeds four MFMAs (136 per 32-key step), the
// partial sums of the whole 64 x 256 output (256 registers) live in the wave, and the query fragments are re-read from a 32-KB LDS image
// (they would be another 128 registers).  That needs ~400 registers: ONE wave per SIMD, so the latencies are hidden inside the wave
// (fragments of the next k-step / value group are requested before the MFMAs of the current one).
//  * wave w streams keys 128 kb + 32 w .. + 31 of every 128-key block through a PRIVATE two-stage ring (2 x 16 KB, global -> LDS directly):
//    no workgroup barrier in the key loop, counted vmcnt waits only;
//  * the positional operand rows (PEK, 1 MB for all prompts: L2) come through registers, requested one step ahead;
//  * the four key quarters are merged pairwise through the (then dead) rings, then v_proj as in dec_t2i_kernel's tail.
// grid = P (whole key range per workgroup; the engine's split > 1 cases - fewer than 512 prompts - stay on dec_t2i_kernel).
#ifdef SABER_OP_F16
#define T4_MFMA_OP "v_mfma_f32_16x16x32_f16"
#else
#define T4_MFMA_OP "v_mfma_f32_16x16x32_bf16"
#endif
// The 64 x 256 partial sums of a wave are PHYSICAL AccVGPRs named in the instruction strings: tile (q, dt) = a[(16 q + dt) * 4 .. + 3].  As
// compiler-allocated values they were moved through VGPRs and scratch every iteration (332 spilled registers: the conditional rescale is VALU
// work on them, so the allocator wants them in VGPRs and then splits their live ranges) - the lesson of tools/experiments/gemm_w1e.hip.
#define T4_ACC(q, dt) (((q) * 16 + (dt)) * 4)
#define T4_ALL_AGPRS "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15", "a16", "a17", "a18", "a19", "a20", "a21", "a22", "a23", "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31", "a32", "a33", "a34", "a35", "a36", "a37", "a38", "a39", "a40", "a41", "a42", "a43", "a44", "a45", "a46", "a47", "a48", "a49", "a50", "a51", "a52", "a53", "a54", "a55", "a56", "a57", "a58", "a59", "a60", "a61", "a62", "a63", "a64", "a65", "a66", "a67", "a68", "a69", "a70", "a71", "a72", "a73", "a74", "a75", "a76", "a77", "a78", "a79", "a80", "a81", "a82", "a83", "a84", "a85", "a86", "a87", "a88", "a89", "a90", "a91", "a92", "a93", "a94", "a95", "a96", "a97", "a98", "a99", "a100", "a101", "a102", "a103", "a104", "a105", "a106", "a107", "a108", "a109", "a110", "a111", "a112", "a113", "a114", "a115", "a116", "a117", "a118", "a119", "a120", "a121", "a122", "a123", "a124", "a125", "a126", "a127", "a128", "a129", "a130", "a131", "a132", "a133", "a134", "a135", "a136", "a137", "a138", "a139", "a140", "a141", "a142", "a143", "a144", "a145", "a146", "a147", "a148", "a149", "a150", "a151", "a152", "a153", "a154", "a155", "a156", "a157", "a158", "a159", "a160", "a161", "a162", "a163", "a164", "a165", "a166", "a167", "a168", "a169", "a170", "a171", "a172", "a173", "a174", "a175", "a176", "a177", "a178", "a179", "a180", "a181", "a182", "a183", "a184", "a185", "a186", "a187", "a188", "a189", "a190", "a191", "a192", "a193", "a194", "a195", "a196", "a197", "a198", "a199", "a200", "a201", "a202", "a203", "a204", "a205", "a206", "a207", "a208", "a209", "a210", "a211", "a212", "a213", "a214", "a215", "a216", "a217", "a218", "a219", "a220", "a221", "a222", "a223", "a224", "a225", "a226", "a227", "a228", "a229", "a230", "a231", "a232", "a233", "a234", "a235", "a236", "a237", "a238", "a239", "a240", "a241", "a242", "a243", "a244", "a245", "a246", "a247", "a248", "a249", "a250", "a251", "a252", "a253", "a254", "a255"
template <int N> __device__ __forceinline__ void t4_acc_mfma(const op16x8& a, const op16x8& b) {
    asm volatile(T4_MFMA_OP " a[%0:%1], %2, %3, a[%0:%1]" : : "n"(N), "n"(N + 3), "v"(a), "v"(b));
}
template <int N> __device__ __forceinline__ f32x4 t4_acc_get() {
    float x0, x1, x2, x3;
    asm volatile("v_accvgpr_read_b32 %0, a[%4]\n\tv_accvgpr_read_b32 %1, a[%5]\n\tv_accvgpr_read_b32 %2, a[%6]\n\tv_accvgpr_read_b32 %3, a[%7]"
                 : "=v"(x0), "=v"(x1), "=v"(x2), "=v"(x3) : "n"(N), "n"(N + 1), "n"(N + 2), "n"(N + 3));
    return (f32x4){x0, x1, x2, x3};
}
template <int N> __device__ __forceinline__ void t4_acc_set(const f32x4& v) {
    asm volatile("v_accvgpr_write_b32 a[%4], %0\n\tv_accvgpr_write_b32 a[%5], %1\n\tv_accvgpr_write_b32 a[%6], %2\n\tv_accvgpr_write_b32 a[%7], %3"
                 : : "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "n"(N), "n"(N + 1), "n"(N + 2), "n"(N + 3));
}
template <int... I, class F> __device__ __forceinline__ void t4_for(std::integer_sequence<int, I...>, F&& f) { (f(std::integral_constant<int, I>{}), ...); }
#define T4_SEQ(n) std::make_integer_sequence<int, n>{}
#define T4_LAZY 8.0f                              // log2 units
#define T4_STAGE (32 * ROW_B)                    // 32 keys
#define T4_RING (2 * T4_STAGE)
#define T4_QOFF (4 * T4_RING)                    // the prompt's 64 folded query rows, dec_t2i's tile format
#define T4_LDS (T4_QOFF + 64 * ROW_B)            // 160 KB
// SHARED: the image tokens belong to the crop, not to the prompt (x_div > 1: first pass, layer 0) - re-read from L2 by its prompts, no streaming hint
// MFMAs on compiler-allocated VGPR accumulators, as asm too: the compiler must not emit an MFMA of its own in this kernel, or it would place
// that MFMA's result in the AccVGPRs the kernel has taken (seen: the score tiles in a[0:3], a[20:23]).  It cannot see these are MFMAs, so
// the wait states between them and the VALU that reads their results are the kernel's business (s_nop below).
#define T4_MFMA_V0(acc, a, b) asm volatile(T4_MFMA_OP " %0, %1, %2, 0" : "=&v"(acc) : "v"(a), "v"(b))
#define T4_MFMA_V(acc, a, b) asm volatile(T4_MFMA_OP " %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b))
template <bool STAMPS, bool SHARED>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
void dec_t2i_w1_kernel(const bf16_t* __restrict__ X, int64_t x_bs, int x_div, int x_off, const bf16_t* __restrict__ pek, const bf16_t* __restrict__ Qt,
                       const float* __restrict__ tq, float qscale, const bf16_t* __restrict__ Wv, const float* __restrict__ bv, bf16_t* __restrict__ out,
                       unsigned long long* __restrict__ stamps) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    unsigned long long ts[4] = {0, 0, 0, 0}, tprev = 0;
#define T4_STAMP(k) do { if (STAMPS) { const unsigned long long _n = __builtin_amdgcn_s_memtime(); ts[k] += _n - tprev; tprev = _n; } } while (0)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fi = lane & 15, fg = lane >> 4;
    const int p = blockIdx.x;
    const bf16_t* Xp = X + (int64_t)((p + x_off) / x_div) * x_bs;
    constexpr int NKB = 4096 / 128;

    // the query rows -> LDS (512-B rows, 16-byte chunks XOR-swizzled by row & 15)
    for (int idx = tid; idx < 64 * 32; idx += 256) {
        const int row = idx >> 5, ch = idx & 31;
        *reinterpret_cast<u32x4*>(smem + T4_QOFF + row * ROW_B + ((ch ^ (row & 15)) << 4)) = *reinterpret_cast<const u32x4*>(Qt + ((int64_t)p * 64 + row) * DC + ch * 8);
    }
    op16x8 pq[4];
#pragma unroll
    for (int qt = 0; qt < 4; ++qt) {
        const int hsel = fg >> 1;
        const float* qp = tq + ((int64_t)p * 8 + (fi & 7)) * 128 + 16 * (2 * qt + hsel) + 8 * (fg & 1);
        const float4 a = *reinterpret_cast<const float4*>(qp), b = *reinterpret_cast<const float4*>(qp + 4);
        const float z = ((fi >> 3) == hsel) ? qscale : 0.f;
        pq[qt] = pack8_d(a.x * z, a.y * z, a.z * z, a.w * z, b.x * z, b.y * z, b.z * z, b.w * z);
    }
    float m[4], l[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) { m[q] = -3.0e38f; l[q] = 0.f; }
    // the AccVGPR file is taken (see above)
    asm volatile("" ::: T4_ALL_AGPRS);
    t4_for(T4_SEQ(64), [&](auto I) { t4_acc_set<decltype(I)::value * 4>((f32x4){0.f, 0.f, 0.f, 0.f}); });

    // LDS-DMA: a stage = 16 pieces of 2 rows; lanes 0..31 fill row 2 i, lanes 32..63 row 2 i + 1; source-side swizzle
    const int hrow = lane >> 5, lsw = (lane & 31) ^ hrow;         // (2 i + h) & 15 = ((2 i) & 15) ^ h
    char* ring = smem + wave * T4_RING;
    auto issue = [&](int kb) {
        const bf16_t* src = Xp + (int64_t)(kb * 128 + 32 * wave + hrow) * DC;
        char* dst = ring + (kb & 1) * T4_STAGE;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const bf16_t* g = src + (2 * i) * DC + ((lsw ^ ((2 * i) & 15)) << 3);
            __builtin_amdgcn_global_load_lds((gptr_d)g, (lptr_d)(dst + i * 1024), 16, 0, SHARED ? 0 : T2I_X_AUX);
        }
    };
    // PEK fragments of a step: [kt][q tile]: row = key 16 kt + fi, chunk 4 qt + fg.  Requested by asm and awaited by the counted wait at the
    // top of the step that uses them (tied to that wait through "+v"): as compiler-visible loads their first use got s_waitcnt vmcnt(0) - with
    // the 16 LDS-DMA pieces of the refill and the next step's PEK loads in flight behind them.
    auto load_kp = [&](int kb, op16x8 (&kp)[2][4]) {
        const bf16_t* pr = pek + (int64_t)(kb * 128 + 32 * wave + fi) * 128 + fg * 8;
        const bf16_t* pr1 = pr + 16 * 128;
        asm volatile("global_load_dwordx4 %0, %4, off\n\tglobal_load_dwordx4 %1, %4, off offset:64\n\tglobal_load_dwordx4 %2, %4, off offset:128\n\tglobal_load_dwordx4 %3, %4, off offset:192"
                     : "=&v"(kp[0][0]), "=&v"(kp[0][1]), "=&v"(kp[0][2]), "=&v"(kp[0][3]) : "v"(pr) : "memory");
        asm volatile("global_load_dwordx4 %0, %4, off\n\tglobal_load_dwordx4 %1, %4, off offset:64\n\tglobal_load_dwordx4 %2, %4, off offset:128\n\tglobal_load_dwordx4 %3, %4, off offset:192"
                     : "=&v"(kp[1][0]), "=&v"(kp[1][1]), "=&v"(kp[1][2]), "=&v"(kp[1][3]) : "v"(pr1) : "memory");
    };
    const uint32_t ring_a = (uint32_t)(uintptr_t)(lptr_d)ring;
    const uint32_t q_a = (uint32_t)(uintptr_t)(lptr_d)(smem + T4_QOFF);
    uint32_t koff[4];                 // fragment (row fi, chunk 4 ks + fg): ks and ks + 4 differ by 256 bytes
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) koff[ks] = fi * ROW_B + (((4 * ks + fg) ^ fi) << 4);
    const int vrow = 4 * fg + (fi >> 2);
    const uint32_t vbase = vrow * ROW_B + (fi & 1) * 8;
    const uint32_t vx = (((fi & 3) >> 1) ^ (vrow & 15)) << 4;          // chunk (2 d + vsel) ^ (vrow & 15) = (vsel ^ (vrow & 15)) ^ 2 d

    op16x8 kpa[2][4], kpb[2][4];
    issue(0); issue(1);
    load_kp(0, kpa);
    __syncthreads();                  // the query image
    if (STAMPS) tprev = __builtin_amdgcn_s_memtime();
    // AHEAD: another step follows (its PEK rows are requested, its stage is in flight); REFILL: the step after that exists too.  Compile-time so
    // that the number of vector-memory operations in flight is the same on every path (the compiler's own waits for the PEK registers
    // become vmcnt(0) otherwise).
    auto step = [&](int kb, op16x8 (&kp)[2][4], op16x8 (&kpn)[2][4], auto AHEAD_, auto REFILL_) {
        constexpr bool AHEAD = decltype(AHEAD_)::value, REFILL = decltype(REFILL_)::value;
        if constexpr (AHEAD) load_kp(kb + 1, kpn);
        // this step's stage has landed: behind it in the queue are the next stage's 16 pieces and the 8 PEK loads just requested
#define T4_KP_TIED "+v"(kp[0][0]), "+v"(kp[0][1]), "+v"(kp[0][2]), "+v"(kp[0][3]), "+v"(kp[1][0]), "+v"(kp[1][1]), "+v"(kp[1][2]), "+v"(kp[1][3])
        if constexpr (AHEAD) asm volatile("s_waitcnt vmcnt(24)" : T4_KP_TIED : : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" : T4_KP_TIED : : "memory");
#undef T4_KP_TIED
        const uint32_t xs = ring_a + (kb & 1) * T4_STAGE;
        // ---- scores: s[q][kt] = PEK part + sum over 8 k-steps; fragments of k-step ks + 1 are requested before the MFMAs of ks
        f32x4 s[4][2];
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int kt = 0; kt < 2; ++kt) T4_MFMA_V0(s[q][kt], kp[kt][q], pq[q]);
        op16x8 fr[2][6];              // [buffer][K kt 0, K kt 1, Q 0..3]
        auto request = [&](int ks, op16x8 (&f)[6]) {
            const uint32_t ka = xs + koff[ks & 3] + (ks >> 2) * 256, qa = q_a + koff[ks & 3] + (ks >> 2) * 256;
            asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:8192" : "=&v"(f[0]), "=&v"(f[1]) : "v"(ka) : "memory");
            asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:8192\n\tds_read_b128 %2, %4 offset:16384\n\tds_read_b128 %3, %4 offset:24576"
                         : "=&v"(f[2]), "=&v"(f[3]), "=&v"(f[4]), "=&v"(f[5]) : "v"(qa) : "memory");
        };
        request(0, fr[0]);
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            op16x8 (&f)[6] = fr[ks & 1];
            if (ks + 1 < 8) {
                request(ks + 1, fr[(ks + 1) & 1]);
                asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]), "+v"(f[4]), "+v"(f[5]));
            } else {
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]), "+v"(f[4]), "+v"(f[5]));
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                T4_MFMA_V(s[q][0], f[0], f[2 + q]);
                T4_MFMA_V(s[q][1], f[1], f[2 + q]);
            }
        }
        T4_STAMP(0);
        // ---- value fragments of the first group are requested before the softmax arithmetic
        u32x2_d vl[2][4], vh[2][4];
        auto request_v = [&](int d4, u32x2_d (&lo)[4], u32x2_d (&hi)[4]) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint32_t va = xs + vbase + (vx ^ ((2 * (d4 + j)) << 4));
                asm volatile("ds_read_b64_tr_b16 %0, %2\n\tds_read_b64_tr_b16 %1, %2 offset:8192" : "=&v"(lo[j]), "=&v"(hi[j]) : "v"(va) : "memory");
            }
        };
        request_v(0, vl[0], vh[0]);
        asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");          // the last score MFMAs -> the VALU that reads them
        op16x8 pf[4];
        t4_for(T4_SEQ(4), [&](auto Q) {
            constexpr int q = decltype(Q)::value;
            float mx = fmaxf(fmaxf(fmaxf(s[q][0][0], s[q][0][1]), fmaxf(s[q][0][2], s[q][0][3])), fmaxf(fmaxf(s[q][1][0], s[q][1][1]), fmaxf(s[q][1][2], s[q][1][3])));
            mx = xor32_max(xor16_max(mx));
            // The partial sums are rescaled only when a row's maximum has grown by more than 2^T4_LAZY since its reference m was set (the
            // rescale is 64 AccVGPR reads, multiplies and writes per query tile here; with a test per step it ran in ~60 % of the steps on
            // random data and was half of the kernel's time - stamped).  Until then the weights are exp2(s - m) <= 2^T4_LAZY: exact
            // arithmetic all the same (the reference cancels in O / l), and far inside the 16-bit operand's and fp32's range.
            if (__any(mx > m[q] + T4_LAZY)) {
                const float mn = fmaxf(m[q], mx);
                const float alpha = __builtin_amdgcn_exp2f(m[q] - mn);
                m[q] = mn;
                l[q] *= alpha;
                t4_for(T4_SEQ(16), [&](auto D) {
                    constexpr int n = T4_ACC(q, decltype(D)::value);
                    const f32x4 v = t4_acc_get<n>();
                    t4_acc_set<n>(v * alpha);
                });
            }
            float sum = 0.f;
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float e = __builtin_amdgcn_exp2f(s[q][kt][r] - m[q]);
                    s[q][kt][r] = e;
                    sum += e;
                }
            sum = xor32_sum(xor16_sum(sum));
            l[q] += sum;
            pf[q] = pack8_d(s[q][0][0], s[q][0][1], s[q][0][2], s[q][0][3], s[q][1][0], s[q][1][1], s[q][1][2], s[q][1][3]);
        });
        T4_STAMP(1);
        // ---- O += V^T P: 4 groups of 4 channel tiles, each value fragment feeds the four query tiles
        asm volatile("s_nop 3" ::: "memory");          // (v_accvgpr_write of a rescale -> MFMA SrcC)
        t4_for(T4_SEQ(4), [&](auto G) {
            constexpr int g = decltype(G)::value;
            u32x2_d (&lo)[4] = vl[g & 1];
            u32x2_d (&hi)[4] = vh[g & 1];
            // (this group was requested one group earlier; the next one is requested only now: LGKM_CNT is a 4-bit counter, and 16 reads in
            // flight would make every counted wait meaningless)
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(lo[0]), "+v"(lo[1]), "+v"(lo[2]), "+v"(lo[3]), "+v"(hi[0]), "+v"(hi[1]), "+v"(hi[2]), "+v"(hi[3]));
            if constexpr (g + 1 < 4) request_v(4 * (g + 1), vl[(g + 1) & 1], vh[(g + 1) & 1]);
            t4_for(T4_SEQ(4), [&](auto J) {
                constexpr int j = decltype(J)::value;
                const op16x8 vfr = cat4_d(__builtin_bit_cast(op16x4, lo[j]), __builtin_bit_cast(op16x4, hi[j]));
                t4_acc_mfma<T4_ACC(0, 4 * g + j)>(vfr, pf[0]);
                t4_acc_mfma<T4_ACC(1, 4 * g + j)>(vfr, pf[1]);
                t4_acc_mfma<T4_ACC(2, 4 * g + j)>(vfr, pf[2]);
                t4_acc_mfma<T4_ACC(3, 4 * g + j)>(vfr, pf[3]);
            });
        });
        T4_STAMP(2);
        // every read of this stage has returned (the last wait above): refill it
        if constexpr (REFILL) issue(kb + 2);
        T4_STAMP(3);
    };
    // two steps per trip: the PEK registers alternate between two sets (a copy would make the compiler wait for everything in flight)
    for (int kb = 0; kb < NKB - 2; kb += 2) {
        step(kb, kpa, kpb, std::true_type{}, std::true_type{});
        step(kb + 1, kpb, kpa, std::true_type{}, std::true_type{});
    }
    step(NKB - 2, kpa, kpb, std::true_type{}, std::false_type{});
    step(NKB - 1, kpb, kpa, std::false_type{}, std::false_type{});
    if (STAMPS && stamps && lane == 0)
        for (int k = 0; k < 4; ++k) stamps[((int64_t)blockIdx.x * 4 + wave) * 4 + k] = ts[k];

    // ---- merge the four key quarters pairwise through the LDS (two regions of [64 q][260] floats), then v_proj per query tile
    float* reg0 = reinterpret_cast<float*>(smem);
    float* reg1 = reg0 + 64 * 260;
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");          // the last MFMAs' results -> v_accvgpr_read
    auto park = [&](float* r, auto Q) {
        constexpr int q = decltype(Q)::value;
        t4_for(T4_SEQ(16), [&](auto D) {
            constexpr int dt = decltype(D)::value;
            const f32x4 v = t4_acc_get<T4_ACC(q, dt)>();
            *reinterpret_cast<float4*>(r + (16 * q + fi) * 260 + 16 * dt + 4 * fg) = make_float4(v[0], v[1], v[2], v[3]);
        });
        if (fg == 0) { r[(16 * q + fi) * 260 + 256] = m[q]; r[(16 * q + fi) * 260 + 257] = l[q]; }
    };
    auto merge = [&](const float* r, auto Q) {
        constexpr int q = decltype(Q)::value;
        const float m2 = r[(16 * q + fi) * 260 + 256], l2 = r[(16 * q + fi) * 260 + 257];
        const float mn = fmaxf(m[q], m2);
        const float a1 = exp2f(m[q] - mn), a2 = exp2f(m2 - mn);
        t4_for(T4_SEQ(16), [&](auto D) {
            constexpr int dt = decltype(D)::value;
            const float4 t = *reinterpret_cast<const float4*>(r + (16 * q + fi) * 260 + 16 * dt + 4 * fg);
            const f32x4 v = t4_acc_get<T4_ACC(q, dt)>();
            t4_acc_set<T4_ACC(q, dt)>((f32x4){v[0] * a1 + t.x * a2, v[1] * a1 + t.y * a2, v[2] * a1 + t.z * a2, v[3] * a1 + t.w * a2});
        });
        m[q] = mn;
        l[q] = l[q] * a1 + l2 * a2;
    };
    using Q0 = std::integral_constant<int, 0>; using Q1 = std::integral_constant<int, 1>; using Q2 = std::integral_constant<int, 2>; using Q3 = std::integral_constant<int, 3>;
    __syncthreads();                          // every wave has left its ring
    if (wave & 1) { float* r = (wave >> 1) ? reg1 : reg0; park(r, Q0{}); park(r, Q1{}); park(r, Q2{}); park(r, Q3{}); }
    __syncthreads();
    if (!(wave & 1)) { const float* r = (wave >> 1) ? reg1 : reg0; merge(r, Q0{}); merge(r, Q1{}); merge(r, Q2{}); merge(r, Q3{}); }
    __syncthreads();
    // waves 0 and 2 hold the two halves: wave 0 keeps query tiles 0, 1 and hands 2, 3 over; wave 2 the other way round (region 0)
    if (wave == 0) { park(reg0, Q2{}); park(reg0, Q3{}); }
    if (wave == 2) { park(reg0, Q0{}); park(reg0, Q1{}); }
    __syncthreads();
    if (wave == 0) { merge(reg0, Q0{}); merge(reg0, Q1{}); }
    if (wave == 2) { merge(reg0, Q2{}); merge(reg0, Q3{}); }
    // normalised rows -> region 1
    t4_for(T4_SEQ(4), [&](auto Q) {
        constexpr int q = decltype(Q)::value;
        if (wave == 2 * (q >> 1)) {
            const float inv = __builtin_amdgcn_rcpf(l[q]);
            t4_for(T4_SEQ(16), [&](auto D) {
                constexpr int dt = decltype(D)::value;
                const f32x4 v = t4_acc_get<T4_ACC(q, dt)>();
                *reinterpret_cast<float4*>(reg1 + (16 * q + fi) * 260 + 16 * dt + 4 * fg) = make_float4(v[0] * inv, v[1] * inv, v[2] * inv, v[3] * inv);
            });
        }
    });
    __syncthreads();
    {   // dec_t2i_kernel's v_proj tail, query tile = wave
        const int qt = wave;
        const float* mo = reg1 + (size_t)qt * 16 * 260;
        f32x4 r0 = (f32x4){0.f, 0.f, 0.f, 0.f}, r1 = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            const float4 z0 = *reinterpret_cast<const float4*>(mo + fi * 260 + 32 * ks + 8 * fg), z1 = *reinterpret_cast<const float4*>(mo + fi * 260 + 32 * ks + 8 * fg + 4);
            const op16x8 zf = pack8_d(z0.x, z0.y, z0.z, z0.w, z1.x, z1.y, z1.z, z1.w);
            const op16x8 w0 = __builtin_bit_cast(op16x8, *reinterpret_cast<const uint4*>(Wv + (int64_t)(32 * qt + fi) * DC + 32 * ks + 8 * fg));
            const op16x8 w1 = __builtin_bit_cast(op16x8, *reinterpret_cast<const uint4*>(Wv + (int64_t)(32 * qt + 16 + fi) * DC + 32 * ks + 8 * fg));
            T4_MFMA_V(r0, w0, zf);
            T4_MFMA_V(r1, w1, zf);
        }
        asm volatile("s_nop 15\n\ts_nop 7" : "+v"(r0), "+v"(r1));
        const int hsel = fi >> 3, hh = 2 * qt + hsel, t = fi & 7;
        const float4 b4 = *reinterpret_cast<const float4*>(bv + 16 * hh + 4 * fg);
        const f32x4 r = hsel ? r1 : r0;
        *reinterpret_cast<uint2*>(out + (int64_t)p * 1024 + t * 128 + 16 * hh + 4 * fg) =
            make_uint2(pack_op16(r[0] + b4.x, r[1] + b4.y), pack_op16(r[2] + b4.z, r[3] + b4.w));
    }
#undef T4_STAMP
}

// dec_t2i_w1_kernel is the route for split == 1 (-16 % against dec_t2i_kernel<8> in the engine, profiles/r05_t2i_w1_*); SABER_AMD_T2I_W1=0 (read per
// call) or debug flag 0x20000000 selects the 8-wave kernel for A/Bs, debug flag 0x10000000 forces the new one.
static bool t2i_w1_route() {
    if (g_saber_debug_flags & 0x10000000) return true;
    if (g_saber_debug_flags & 0x20000000) return false;
    const char* e = getenv("SABER_AMD_T2I_W1");
    return !(e && e[0] == '0');
}
const char* launch_dec_t2i(const bf16_t* X, XMap xm, const bf16_t* pek, const bf16_t* Qt, const float* tq, float qscale, float* Opart, float* ML,
                           int P, int split, const bf16_t* Wv, const float* bv, bf16_t* out, hipStream_t s, const XBuild* build) {
    if (P <= 0) return nullptr;
    if (split != 1 && split != 2 && split != 4 && split != 8) return "dec_t2i: split must be 1, 2, 4 or 8";
    if (xm.div <= 0) return "dec_t2i: XMap.div must be positive";
    const float* nf = nullptr; const bf16_t* nb = nullptr;
    if (build) {
        if (!build->embb || !build->h2 || !build->w3 || build->map.div <= 0) return "dec_t2i: incomplete XBuild";
        hipLaunchKernelGGL((dec_t2i_kernel<8, false, true>), dim3(P * split), dim3(512), T2ICfg<8>::LDS, s, (const bf16_t*)nullptr, build->map.stride, build->map.div, build->map.off,
                           pek, Qt, tq, qscale, Opart, ML, split, Wv, bv, out, (unsigned long long*)nullptr, build->embb, build->h2, build->w3);
    } else if (split == 1 && t2i_w1_route()) {      // round 5: one wave per SIMD, four query tiles per wave (whole key range per workgroup: P >= 512 in the engine)
#define T4_LAUNCH(ST, SH) hipLaunchKernelGGL((dec_t2i_w1_kernel<ST, SH>), dim3(P), dim3(256), T4_LDS, s, X, xm.stride, xm.div, xm.off, pek, Qt, tq, qscale, Wv, bv, out, g_saber_stamp_buf)
        if (g_saber_stamp_buf) { if (xm.div > 1) T4_LAUNCH(true, true); else T4_LAUNCH(true, false); }
        else { if (xm.div > 1) T4_LAUNCH(false, true); else T4_LAUNCH(false, false); }
#undef T4_LAUNCH
    } else if (!(g_saber_debug_flags & 4)) {  // 4-wave workgroups (two per CU, no key-half merge) measure the same as one 8-wave workgroup: kept as an option
        if (g_saber_stamp_buf)
            hipLaunchKernelGGL((dec_t2i_kernel<8, true, false>), dim3(P * split), dim3(512), T2ICfg<8>::LDS, s, X, xm.stride, xm.div, xm.off, pek, Qt, tq, qscale, Opart, ML, split, Wv, bv, out, g_saber_stamp_buf, nf, nb, nf);
        else
            hipLaunchKernelGGL((dec_t2i_kernel<8, false, false>), dim3(P * split), dim3(512), T2ICfg<8>::LDS, s, X, xm.stride, xm.div, xm.off, pek, Qt, tq, qscale, Opart, ML, split, Wv, bv, out, g_saber_stamp_buf, nf, nb, nf);
    }
    else
        hipLaunchKernelGGL((dec_t2i_kernel<4, false, false>), dim3(P * split), dim3(256), T2ICfg<4>::LDS, s, X, xm.stride, xm.div, xm.off, pek, Qt, tq, qscale, Opart, ML, split, Wv, bv, out, g_saber_stamp_buf, nf, nb, nf);
    if (split > 1) hipLaunchKernelGGL(dec_t2i_finish_kernel, dim3(P * 8), dim3(256), 0, s, (const float*)Opart, (const float*)ML, split, Wv, bv, out);
    return nullptr;
}

// ------------------------------------------------------------------------------------------------ image -> tokens
// X_out[p][n] = LN(x_n + softmax_heads((x_n + pe_n).Kt + cb).Vt + b_o), one block per prompt, 32-row tiles.
// Pure streaming kernel: the folded operands of the prompt live in REGISTERS (each wave owns one 16-column slice of Kt
// for GEMM1 and a 64-channel slice of Vt^T for GEMM2), the X and PEQ tiles go global -> LDS directly through a 4-stage
// ring (3 tiles in flight).  (x + pe).Kt = x.Kt + pe.Kt, and since Kt = W_q^T k (folded, per head) the positional term is
// k_h . (W_q pe)_h: a contraction of 16 per head against PEQ = pe W_q^T [4096][128] (model constant, computed at finalize).
// A wave's 16 score columns are the 8 tokens of heads 2qr and 2qr+1, so one MFMA with the block-diagonal operand
// [k_2qr | 0 ; 0 | k_2qr+1] (k = 32) adds it: 9 MFMAs and fragment reads per tile for GEMM1 instead of 16, 24 KB per stage instead of 32.
// 8 waves: wave = (row tile rt) * 4 + (quarter qr).  Per tile: GEMM1 -> softmax -> P via LDS -> GEMM2 -> residual +
// LayerNorm (row statistics exchanged through LDS) -> bf16 rows.  Normalisation of tile t is deferred until after the
// barrier of tile t+1, so each tile costs ONE workgroup barrier.
// RT = 16-row tiles per workgroup (4 waves each).  RT = 1: 4-wave workgroups of 62 KB LDS, TWO per CU: they share no barrier, drift
// apart, and one's MFMA phases overlap the other's softmax / LayerNorm / store phases (RT = 2, one 8-wave workgroup per CU, keeps
// both halves of a SIMD in the same phase all the time).
#define I2T_PEQ_ROWB 256                          // one PEQ row: 128 bf16
#define I2T_NSTAGE 4
#define I2T_PSTRIDE 144                           // bytes per P row (64 bf16 + pad)
template <int RT> struct I2TCfg {
    static constexpr int ROWS = 16 * RT;
    static constexpr int STAGE = ROWS * ROW_B + ROWS * I2T_PEQ_ROWB;   // X tile + PEQ tile
    static constexpr int PBUF_B = RT * 16 * I2T_PSTRIDE;                // one P buffer: [RT][16 rows][144 B]
    static constexpr int STAT_B = RT * 16 * 4 * 2 * 4;                  // one statistics buffer: [RT][16 rows][4 quarters][sum, sumsq]
    static constexpr int OSCR_B = 4 * RT * 2048;                        // per wave: 16 rows x 128 B of the output tile, transposed into full-line stores
    static constexpr int LDS = I2T_NSTAGE * STAGE + 2 * PBUF_B + 2 * STAT_B + OSCR_B;
    static constexpr int W3_B = 256 * 32;                               // BUILD: mask_downscaling.6.weight as bf16 [256][16]
};

// BUILD (RT = 1): the X tile is assembled by the waves as bf16(embb + h2 . W3^T) instead of being copied from HBM (see dec_t2i_kernel):
// wave qr owns channels 64 qr .. 64 qr + 63 of the 16 rows (the channels it later reads back as the residual): 4 MFMAs per tile.  The
// registers of tile t + 3 are loaded right after the barrier of iteration t, tile t + 2 is written from the registers loaded one
// iteration earlier; it becomes visible with the barrier of iteration t + 1 and is consumed in iteration t + 2.
template <int RT, bool STAMPS, bool BUILD>
__global__ __launch_bounds__(256 * RT, BUILD ? 2 : 1) void dec_i2t_kernel(const bf16_t* __restrict__ X, int64_t x_bs, int x_div, int x_off, const bf16_t* __restrict__ peq,
                                                      const bf16_t* __restrict__ Kt, const float* __restrict__ tk, float kscale, const float* __restrict__ cb,
                                                      const bf16_t* __restrict__ VtT, const float* __restrict__ bo,
                                                      const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                                      bf16_t* __restrict__ Xout, int nsplit, int dbg, unsigned long long* __restrict__ stamps,
                                                      const float* __restrict__ embb, const bf16_t* __restrict__ h2, const float* __restrict__ w3) {
    static_assert(!BUILD || RT == 1, "the tile builder is written for the 4-wave form");
    using CF = I2TCfg<RT>;
    constexpr int I2T_ROWS = CF::ROWS, I2T_STAGE = CF::STAGE, I2T_PBUF_B = CF::PBUF_B, I2T_STAT_B = CF::STAT_B;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // development only (stamps != nullptr): per-wave cycle sums of the phases of the tile loop, see tools/dec_stamps.py
    unsigned long long ts[6] = {0, 0, 0, 0, 0, 0}, tprev = 0;
#define I2T_STAMP(k) do { if (STAMPS) { const unsigned long long _n = __builtin_amdgcn_s_memtime(); ts[k] += _n - tprev; tprev = _n; } } while (0)
    char* pbuf = smem + I2T_NSTAGE * I2T_STAGE;                         // [2 rt][16 rows][144 B]
    float* stat = reinterpret_cast<float*>(pbuf + 2 * I2T_PBUF_B);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int rt = wave >> 2, qr = wave & 3;
    const int fi = lane & 15, fg = lane >> 4;
    const int p = blockIdx.x / nsplit;
    const int NT = (4096 / I2T_ROWS) / nsplit;                  // tiles of this block
    const int64_t row0 = (int64_t)(blockIdx.x % nsplit) * NT * I2T_ROWS;
    const bf16_t* Xp = X + (BUILD ? 0 : (int64_t)((p + x_off) / x_div) * x_bs + row0 * DC);
    const bf16_t* pep = peq + row0 * 128;
    // (embb in the builder's order, see dec_t2i_kernel: this wave reads channel tiles 4 qr .. 4 qr + 3 of a row tile = 4 contiguous KB)
    const float* Ep = BUILD ? embb + (int64_t)((p + x_off) / x_div) * x_bs + ((row0 / 16) * 16 + 4 * qr) * 256 + lane * 4 : nullptr;
    const bf16_t* Hp = BUILD ? h2 + ((int64_t)p * 4096 + row0 + fi) * 16 + 8 * (fg & 1) : nullptr;
    // W3 (bf16) lives in LDS behind the kernel's own regions: 16 registers per lane would push the kernel past 256 and cost the second
    // workgroup of the CU.  A-operand fragment of channel tile ct: row 64 qr + 16 ct + fi, k = 8 (fg & 1) .. + 7; the lanes of k >= 16
    // (fg >= 2) read the same bytes as their k - 16 twins: their products vanish against the zeros of the h2 operand.
    const uint32_t w3_a = (uint32_t)(uintptr_t)(lptr_d)(smem + CF::LDS) + (64 * qr + fi) * 32 + (fg & 1) * 16;
    if (BUILD) {
        const int ch = tid;        // 256 threads: one W3 row each
        const float* wr = w3 + ch * 16;
        const float4 a = *reinterpret_cast<const float4*>(wr), b = *reinterpret_cast<const float4*>(wr + 4), c = *reinterpret_cast<const float4*>(wr + 8), d = *reinterpret_cast<const float4*>(wr + 12);
        *reinterpret_cast<uint4*>(smem + CF::LDS + ch * 32) = make_uint4(pack_op16(a.x, a.y), pack_op16(a.z, a.w), pack_op16(b.x, b.y), pack_op16(b.z, b.w));
        *reinterpret_cast<uint4*>(smem + CF::LDS + ch * 32 + 16) = make_uint4(pack_op16(c.x, c.y), pack_op16(c.z, c.w), pack_op16(d.x, d.y), pack_op16(d.z, d.w));
        __syncthreads();
    }
    f32x4 eb[4];
    u32x4 hb;
    // 5 loads per tile through inline asm, waited for by hand in build_write (counted: the younger PEQ piece and output stores stay in flight)
    auto load_regs = [&](int t) {
        const float* ea = Ep + (int64_t)t * 16 * 256;
        asm volatile("global_load_dwordx4 %0, %4, off\n\tglobal_load_dwordx4 %1, %4, off offset:1024\n\tglobal_load_dwordx4 %2, %4, off offset:2048\n\tglobal_load_dwordx4 %3, %4, off offset:3072"
                     : "=&v"(eb[0]), "=&v"(eb[1]), "=&v"(eb[2]), "=&v"(eb[3]) : "v"(ea) : "memory");
        const bf16_t* ha = Hp + (int64_t)t * I2T_ROWS * 16;
        asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(hb) : "v"(ha) : "memory");
    };
    bf16_t* Xo = Xout + ((int64_t)p * 4096 + row0) * DC;

    // folded operands of this prompt, straight into registers
    op16x8 kf[8], vf[4][2];
#pragma unroll
    for (int ks = 0; ks < 8; ++ks)
        kf[ks] = __builtin_bit_cast(op16x8, *reinterpret_cast<const uint4*>(Kt + ((int64_t)p * 64 + 16 * qr + fi) * DC + 32 * ks + 8 * fg));
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
            vf[t][ks] = __builtin_bit_cast(op16x8, *reinterpret_cast<const uint4*>(VtT + ((int64_t)p * 256 + 64 * qr + 16 * t + fi) * 64 + 32 * ks + 8 * fg));
    op16x8 kq;   // block-diagonal projected keys: row fi = token (fi & 7) of head 2 qr + (fi >> 3); k = 8 fg .. + 7 of [head A 16 | head B 16]
    {
        const int hsel = fg >> 1;
        const float* kp = tk + ((int64_t)p * 8 + (fi & 7)) * 128 + 16 * (2 * qr + hsel) + 8 * (fg & 1);
        const float4 a = *reinterpret_cast<const float4*>(kp), b = *reinterpret_cast<const float4*>(kp + 4);
        const float z = ((fi >> 3) == hsel) ? kscale : 0.f;
        kq = pack8_d(a.x * z, a.y * z, a.z * z, a.w * z, b.x * z, b.y * z, b.z * z, b.w * z);
    }
    const float4 cb4 = *reinterpret_cast<const float4*>(cb + (int64_t)p * 64 + 16 * qr + 4 * fg);
    float4 bo4[4], g4[4], be4[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        bo4[t] = *reinterpret_cast<const float4*>(bo + 64 * qr + 16 * t + 4 * fg);
        g4[t] = *reinterpret_cast<const float4*>(gamma + 64 * qr + 16 * t + 4 * fg);
        be4[t] = *reinterpret_cast<const float4*>(beta + 64 * qr + 16 * t + 4 * fg);
    }
    // direct-to-LDS: per stage 16 X wave-instructions (1 KB = 2 rows each; wave w issues 2w, 2w+1) and 8 PEQ ones (4 rows of
    // 256 B each, 16-B chunks XOR-swizzled by row & 15; wave w issues piece w)
    int srow[2], schunk[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        srow[i] = 2 * (wave * 2 + i) + (lane >> 5);
        schunk[i] = (lane & 31) ^ (srow[i] & 15);
    }
    const int prow = 4 * wave + (lane >> 4), pchunk = (lane & 15) ^ (prow & 15);
    auto issue = [&](int t) {
        char* sx = smem + (t & (I2T_NSTAGE - 1)) * I2T_STAGE;
        if (!BUILD)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int64_t off = (int64_t)(t * I2T_ROWS + srow[i]) * DC + schunk[i] * 8;
            if (x_div > 1) __builtin_amdgcn_global_load_lds((gptr_d)(Xp + off), (lptr_d)(sx + (wave * 2 + i) * 1024), 16, 0, 0);
            else __builtin_amdgcn_global_load_lds((gptr_d)(Xp + off), (lptr_d)(sx + (wave * 2 + i) * 1024), 16, 0, I2T_X_AUX);
        }
        __builtin_amdgcn_global_load_lds((gptr_d)(pep + (int64_t)(t * I2T_ROWS + prow) * 128 + pchunk * 8), (lptr_d)(sx + I2T_ROWS * ROW_B + wave * 1024), 16, 0, 0);
    };
    const int poff = (16 * rt + fi) * I2T_PEQ_ROWB + (((4 * qr + fg) ^ fi) << 4);   // B fragment of the PEQ tile: row 16 rt + fi, columns 32 qr + 8 fg ..
    int xoff[8];      // B-operand fragment of row (16 rt + fi), k-step ks, in the swizzled tile
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) xoff[ks] = (16 * rt + fi) * ROW_B + (((4 * ks + fg) ^ fi) << 4);
    int roff[4];      // residual: x[m][64 qr + 16 t + 4 fg .. +3]
#pragma unroll
    for (int t = 0; t < 4; ++t) roff[t] = (16 * rt + fi) * ROW_B + (((8 * qr + 2 * t + (fg >> 1)) ^ fi) << 4) + (fg & 1) * 8;
    // P and the row statistics are double-buffered by tile parity so that one barrier per tile suffices
    const uint32_t prow_a = (uint32_t)(uintptr_t)(lptr_d)(pbuf + (rt * 16 + fi) * I2T_PSTRIDE);
    const uint32_t stat_a = (uint32_t)(uintptr_t)(lptr_d)(stat + ((rt * 16 + fi) * 4) * 2);
    const uint32_t smem_a = (uint32_t)(uintptr_t)(lptr_d)smem;
    const uint32_t oscr_a = smem_a + I2T_NSTAGE * I2T_STAGE + 2 * I2T_PBUF_B + 2 * I2T_STAT_B + wave * 2048;
    f32x2 g2[8], be2[8], bo2[8];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        g2[2 * t] = (f32x2){g4[t].x, g4[t].y}; g2[2 * t + 1] = (f32x2){g4[t].z, g4[t].w};
        be2[2 * t] = (f32x2){be4[t].x, be4[t].y}; be2[2 * t + 1] = (f32x2){be4[t].z, be4[t].w};
        bo2[2 * t] = (f32x2){bo4[t].x, bo4[t].y}; bo2[2 * t + 1] = (f32x2){bo4[t].z, bo4[t].w};
    }
    f32x2 y2[8];                      // y of the previous tile (bias + residual added), normalised one barrier later
    // normalise + store tile tp from y2 and the exchanged statistics
    auto finish_tile = [&](int tp) {
        f32x4 a, b;
        asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:16\n\ts_waitcnt lgkmcnt(0)" : "=&v"(a), "=&v"(b) : "v"(stat_a + (tp & 1) * I2T_STAT_B) : "memory");
        const float tot = (a.x + a.z) + (b.x + b.z), tsq = (a.y + a.w) + (b.y + b.w);
        const float mean = tot * (1.0f / DC);
        const float rstd = __builtin_amdgcn_rsqf(fmaxf(tsq * (1.0f / DC) - mean * mean, 0.f) + eps);
        const f32x2 mean2 = (f32x2){mean, mean}, rstd2 = (f32x2){rstd, rstd};
        // The lane owns 4 x 8 B of row (16 rt + fi); stored as such, every store instruction touches 16 rows with 32 B each and the
        // CU's store path (~7 B/cycle for such row-per-lane stores) becomes the longest phase of the tile.  The wave's 16 x 128 B go
        // through 2 KB of LDS instead (XOR-swizzled 16-B chunks) and leave as two instructions of 8 full 128-B lines each.
        const uint32_t tb = oscr_a + fi * 128 + (fg & 1) * 8;
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) {
            const f32x2 v0 = ((y2[2 * tt] - mean2) * rstd2) * g2[2 * tt] + be2[2 * tt];
            const f32x2 v1 = ((y2[2 * tt + 1] - mean2) * rstd2) * g2[2 * tt + 1] + be2[2 * tt + 1];
            lds_write_b64(tb + (((2 * tt + (fg >> 1)) ^ (fi & 7)) << 4), pack_op16(v0.x, v0.y), pack_op16(v1.x, v1.y));
        }
        u32x4 o0, o1;
        asm volatile("s_waitcnt lgkmcnt(0)\n\tds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:1024\n\ts_waitcnt lgkmcnt(0)"
                     : "=&v"(o0), "=&v"(o1) : "v"(oscr_a + (lane >> 3) * 128 + (((lane & 7) ^ ((lane >> 3) & 7)) << 4)) : "memory");
        bf16_t* orow = Xo + (int64_t)(tp * I2T_ROWS + 16 * rt + (lane >> 3)) * DC + 64 * qr + 8 * (lane & 7);
        __builtin_nontemporal_store(o0, reinterpret_cast<u32x4*>(orow));
        __builtin_nontemporal_store(o1, reinterpret_cast<u32x4*>(orow + 8 * DC));       // streamed: the 2 MB per prompt are read back by the next kernel long after they left the caches
    };

    // BUILD: X0 tile t from the registers load_regs(t) filled: a[r] = X0[row fi][channel 64 qr + 16 ct + 4 fg + r], written where the wave
    // reads its residual back (roff)
    auto build_write = [&](int t, int younger) {
        const uint32_t sx = smem_a + (t & (I2T_NSTAGE - 1)) * I2T_STAGE;
        // `younger` = this wave's vector-memory operations issued after the tile's loads that may stay in flight
#define I2T_WAITR(N) asm volatile("s_waitcnt vmcnt(" #N ")" : "+v"(eb[0]), "+v"(eb[1]), "+v"(eb[2]), "+v"(eb[3]), "+v"(hb) :: "memory")
        if (younger >= 3) I2T_WAITR(3); else if (younger == 2) I2T_WAITR(2); else if (younger == 1) I2T_WAITR(1); else I2T_WAITR(0);
        const op16x8 hf = __builtin_bit_cast(op16x8, fg < 2 ? hb : (u32x4){0u, 0u, 0u, 0u});
        op16x8 w3f[4];
        asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:512\n\tds_read_b128 %2, %4 offset:1024\n\tds_read_b128 %3, %4 offset:1536\n\ts_waitcnt lgkmcnt(0)"
                     : "=&v"(w3f[0]), "=&v"(w3f[1]), "=&v"(w3f[2]), "=&v"(w3f[3]) : "v"(w3_a) : "memory");
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) {
            const f32x4 a = MFMA_16x16x32(w3f[ct], hf, eb[ct], 0, 0, 0);
            lds_write_b64(sx + roff[ct], pack_op16(a[0], a[1]), pack_op16(a[2], a[3]));
        }
    };
    issue(0); issue(1); issue(2);
    if (BUILD) {
        load_regs(0); build_write(0, 0);
        if (NT > 1) { load_regs(1); build_write(1, 0); }
        if (NT > 2) load_regs(2);
        asm volatile("s_waitcnt vmcnt(0)\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
    } else
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (STAMPS) tprev = __builtin_amdgcn_s_memtime();
    for (int t = 0; t < NT; ++t) {
        const char* xs = smem + (t & (I2T_NSTAGE - 1)) * I2T_STAGE;
        const char* ps = xs + I2T_ROWS * ROW_B;
        // GEMM1 (swapped): S^T[c][m] = Kt[c].x[m] + Kt[c].pe[m] + cb[c] for this wave's 16 columns c
        f32x4 s = (f32x4){cb4.x, cb4.y, cb4.z, cb4.w}, s1;
        {
            const op16x8 pf = *reinterpret_cast<const op16x8*>(ps + poff);
            s1 = MFMA_16x16x32(kq, pf, (f32x4){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
        }
#pragma unroll
        for (int ks = 0; ks < 8; ks += 2) {       // two accumulation chains
            const op16x8 xf0 = *reinterpret_cast<const op16x8*>(xs + xoff[ks]);
            const op16x8 xf1 = *reinterpret_cast<const op16x8*>(xs + xoff[ks + 1]);
            s = MFMA_16x16x32(kf[ks], xf0, s, 0, 0, 0);
            s1 = MFMA_16x16x32(kf[ks + 1], xf1, s1, 0, 0, 0);
        }
        s += s1;
        // softmax over the 8 tokens of a head: this lane's 4 values + lane ^ 16
        {
            float mx = fmaxf(fmaxf(s[0], s[1]), fmaxf(s[2], s[3]));
            mx = xor16_max(mx);
            float sum = 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) { s[r] = __builtin_amdgcn_exp2f(s[r] - mx); sum += s[r]; }
            sum = xor16_sum(sum);
            const float inv = __builtin_amdgcn_rcpf(sum);
            // LDS stores go through inline asm: a compiler-visible ds_write makes it wait vmcnt(0) for the direct-to-LDS loads in flight
            lds_write_b64(prow_a + (t & 1) * I2T_PBUF_B + (16 * qr + 4 * fg) * 2, pack_op16(s[0] * inv, s[1] * inv), pack_op16(s[2] * inv, s[3] * inv));
        }
        // tile t+1 (issued two iterations ago) must have landed before the barrier makes it visible to everyone; the younger
        // loads and the bf16 stores stay in flight.  Queue behind L(t+1): [S(t-3)] L(t+2) [S(t-2)]; a load group is 3 ops, a store group 2.
        I2T_STAMP(0);
        if (BUILD) {
            // the wave's PEQ piece of tile t+1 is older than the registers of tile t+1, which were waited for when that tile was written
            // (iteration t-1, or the prologue): nothing to wait for here but the LDS writes
            if (t + 3 >= NT) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        else if (t + 3 >= NT) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (t < 2) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
        else if (t == 2) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        I2T_STAMP(1);
        __builtin_amdgcn_s_barrier();      // the ONE barrier per tile: P(t) and stat(t-1) complete, tile t+1 visible, slot of tile t-1 free
        I2T_STAMP(2);
        if (t + 3 < NT) issue(t + 3);
        if (BUILD) {
            // behind the loads of tile t+2 (issued in iteration t-1, in the prologue for t = 0) this wave has issued: the 2 output stores of
            // iteration t-1 (t >= 2) and this iteration's PEQ piece (t + 3 < NT)
            if (t + 2 < NT) build_write(t + 2, (t >= 2 ? 2 : 0) + (t + 3 < NT ? 1 : 0));
            if (t + 3 < NT) load_regs(t + 3);
        }
        I2T_STAMP(3);
        if (t > 0) finish_tile(t - 1);
        I2T_STAMP(4);
        // GEMM2: Y^T[d][m] for this wave's 64 channels
        f32x4 y[4];
        {
            op16x8 p0, p1;
            asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:64\n\ts_waitcnt lgkmcnt(0)" : "=&v"(p0), "=&v"(p1) : "v"(prow_a + (t & 1) * I2T_PBUF_B + 16 * fg) : "memory");
#pragma unroll
            for (int tt = 0; tt < 4; ++tt) {
                y[tt] = (f32x4){0.f, 0.f, 0.f, 0.f};
                y[tt] = MFMA_16x16x32(vf[tt][0], p0, y[tt], 0, 0, 0);
                y[tt] = MFMA_16x16x32(vf[tt][1], p1, y[tt], 0, 0, 0);
            }
        }
        // residual + bias, partial LayerNorm statistics over this wave's 64 channels of row m.  The residual reads of the
        // DMA-written tile also go through inline asm (a visible ds_read of that region again forces vmcnt(0)).
        uint64_t xr0, xr1, xr2, xr3;
        {
            const uint32_t xa = smem_a + (t & (I2T_NSTAGE - 1)) * I2T_STAGE;
            asm volatile("ds_read_b64 %0, %4\n\tds_read_b64 %1, %5\n\tds_read_b64 %2, %6\n\tds_read_b64 %3, %7\n\ts_waitcnt lgkmcnt(0)"
                         : "=&v"(xr0), "=&v"(xr1), "=&v"(xr2), "=&v"(xr3)
                         : "v"(xa + roff[0]), "v"(xa + roff[1]), "v"(xa + roff[2]), "v"(xa + roff[3])
                         : "memory");
        }
        const uint64_t xrs[4] = {xr0, xr1, xr2, xr3};
        f32x2 sum2 = (f32x2){0.f, 0.f}, sq2 = (f32x2){0.f, 0.f};
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) {
            const uint32_t xlo = (uint32_t)xrs[tt], xhi = (uint32_t)(xrs[tt] >> 32);
            const f32x2 r0 = (f32x2){op16_lo(xlo), op16_hi(xlo)};
            const f32x2 r1 = (f32x2){op16_lo(xhi), op16_hi(xhi)};
            y2[2 * tt] = ((f32x2){y[tt][0], y[tt][1]} + bo2[2 * tt]) + r0;
            y2[2 * tt + 1] = ((f32x2){y[tt][2], y[tt][3]} + bo2[2 * tt + 1]) + r1;
            sum2 += y2[2 * tt]; sum2 += y2[2 * tt + 1];
            sq2 = __builtin_elementwise_fma(y2[2 * tt], y2[2 * tt], sq2);
            sq2 = __builtin_elementwise_fma(y2[2 * tt + 1], y2[2 * tt + 1], sq2);
        }
        float sum = sum2.x + sum2.y, sq = sq2.x + sq2.y;
        sum = xor16_sum(sum); sq = xor16_sum(sq);
        sum = xor32_sum(sum); sq = xor32_sum(sq);
        if (fg == 0) lds_write_b64(stat_a + (t & 1) * I2T_STAT_B + qr * 8, __float_as_uint(sum), __float_as_uint(sq));
        I2T_STAMP(5);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    finish_tile(NT - 1);
    if (STAMPS && stamps && lane == 0)
        for (int k = 0; k < 6; ++k) stamps[((int64_t)blockIdx.x * (4 * RT) + wave) * 6 + k] = ts[k];
}

// ------------------------------------------------------------------------------------------------ image -> tokens, one wave per SIMD (round 5)
// dec_i2t_kernel splits a 16-row tile over four waves (score quarters, then channel quarters): the softmax weights and the LayerNorm
// statistics cross the waves through LDS and the tile costs a workgroup barrier; each wave re-reads the whole X tile.  Here a WAVE owns a
// tile: all 64 score columns (the folded K operand of the prompt: 32 fragments) and all 256 output channels (the folded V operand: 32
// fragments) - 256 registers of MFMA A operands, held in the AccVGPRs (gfx950 MFMAs read A / B from either file) and named in the
// instruction strings.  The softmax weights never leave the registers: the P fragment of k-step j is the lane's own score registers of the
// quarters 2 j and 2 j + 1 (k = 8 fg + e  <->  column 16 (2 j + (e >> 2)) + 4 fg + (e & 3)), and the V fragments are gathered in that order
// when they are loaded; the LayerNorm statistics are in-wave reductions.  No barrier, no cross-wave traffic: wave w streams tiles w, w + 4, ...
// through a private two-stage ring (global -> LDS directly) and transposes its finished tile through 16 KB of private LDS into full 512-B
// rows.  The wave's VALU work (softmax, residual, LayerNorm: ~450 instructions per tile) does not overlap its own MFMAs (section 8.2), but
// four independent waves keep four SIMDs busy all the time, where the four-wave form waits 36 % of its cycles.
#define I4_XST (16 * ROW_B)                       // X tile
#define I4_STAGE (I4_XST + 16 * I2T_PEQ_ROWB)     // + PEQ tile: 12 KB
#define I4_RING (2 * I4_STAGE)
#define I4_SCR (16 * 512)                         // fp32 transpose scratch: 16 rows x 128 channels (a tile leaves in two channel halves)
#define I4_CONST 8192                             // the prompt's small operands: b_o [256] fp32 | score bias [64] fp32 | positional fragments [4][64 lanes][16 B]
#define I4_WAVE (I4_RING + I4_SCR + I4_CONST)     // 40 KB per wave
#define I4_LDS (4 * I4_WAVE)                      // 160 KB
#define I4_KF(q, ks) ((((q) * 8 + (ks)) * 4))                 // AccVGPR of the folded-K fragment (quarter q, k-step ks)
#define I4_VF(dt, j) (128 + (((dt) * 2 + (j)) * 4))           // AccVGPR of the folded-V fragment (channel tile dt, k-step j)
template <int N> __device__ __forceinline__ void i4_load_a16(const void* p) {
    asm volatile("global_load_dwordx4 a[%0:%1], %2, off" : : "n"(N), "n"(N + 3), "v"(p) : "memory");
}
template <int N> __device__ __forceinline__ void i4_load_a8(const void* p) {
    asm volatile("global_load_dwordx2 a[%0:%1], %2, off" : : "n"(N), "n"(N + 1), "v"(p) : "memory");
}
template <int N> __device__ __forceinline__ void i4_mfma_a(f32x4& acc, const op16x8& b) {          // acc += A(a[N..N+3]) . b
    asm volatile(T4_MFMA_OP " %0, a[%1:%2], %3, %0" : "+v"(acc) : "n"(N), "n"(N + 3), "v"(b));
}
template <bool SHARED>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
void dec_i2t_w1_kernel(const bf16_t* __restrict__ X, int64_t x_bs, int x_div, int x_off, const bf16_t* __restrict__ peq,
                       const bf16_t* __restrict__ Kt, const float* __restrict__ tk, float kscale, const float* __restrict__ cb,
                       const bf16_t* __restrict__ VtT, const float* __restrict__ bo,
                       const float* __restrict__ gamma, const float* __restrict__ beta, float eps, bf16_t* __restrict__ Xout) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fi = lane & 15, fg = lane >> 4;
    const int p = blockIdx.x;
    const bf16_t* Xp = X + (int64_t)((p + x_off) / x_div) * x_bs;
    bf16_t* Xo = Xout + (int64_t)p * 4096 * DC;
    constexpr int NT = 4096 / 16 / 4;                 // tiles per wave
    char* ring = smem + wave * I4_WAVE;
    const uint32_t ring_a = (uint32_t)(uintptr_t)(lptr_d)ring;
    const uint32_t scr_a = ring_a + I4_RING;
    const uint32_t cst_a = scr_a + I4_SCR;

    // ---- the prompt's folded operands -> AccVGPRs.  The file is taken: the compiler must not place anything of its own there - neither
    // MFMA results (there is no compiler-visible MFMA in this kernel) nor VGPR spills (it spills to free AccVGPRs first: the kernel has to
    // stay clear of 256 VGPRs, which is why b_o, the score bias and the positional fragments live in LDS and not in 96 registers).
    asm volatile("" ::: T4_ALL_AGPRS);
    t4_for(T4_SEQ(4), [&](auto Q) {
        constexpr int q = decltype(Q)::value;
        t4_for(T4_SEQ(8), [&](auto KS) {
            constexpr int ks = decltype(KS)::value;
            i4_load_a16<I4_KF(q, ks)>(Kt + ((int64_t)p * 64 + 16 * q + fi) * DC + 32 * ks + 8 * fg);
        });
    });
    t4_for(T4_SEQ(16), [&](auto D) {
        constexpr int dt = decltype(D)::value;
        const bf16_t* vr = VtT + ((int64_t)p * 256 + 16 * dt + fi) * 64 + 4 * fg;
        // k-step j: elements 0..3 = columns 32 j + 4 fg .. + 3 (quarter 2 j), elements 4..7 = columns 32 j + 16 + 4 fg .. + 3 (quarter 2 j + 1)
        i4_load_a8<I4_VF(dt, 0)>(vr);          i4_load_a8<I4_VF(dt, 0) + 2>(vr + 16);
        i4_load_a8<I4_VF(dt, 1)>(vr + 32);     i4_load_a8<I4_VF(dt, 1) + 2>(vr + 48);
    });
    // b_o, the score bias, and the positional operand (block-diagonal: the 8 tokens of heads 2 q, 2 q + 1 against their 16 PEQ channels
    // each) -> the wave's constants in LDS (written and read by this wave only)
    {
        float* cst = reinterpret_cast<float*>(ring + I4_RING + I4_SCR);
        *reinterpret_cast<float4*>(cst + 4 * lane) = *reinterpret_cast<const float4*>(bo + 4 * lane);
        if (lane < 16) *reinterpret_cast<float4*>(cst + 256 + 4 * lane) = *reinterpret_cast<const float4*>(cb + (int64_t)p * 64 + 4 * lane);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int hsel = fg >> 1;
            const float* kp = tk + ((int64_t)p * 8 + (fi & 7)) * 128 + 16 * (2 * q + hsel) + 8 * (fg & 1);
            const float4 a = *reinterpret_cast<const float4*>(kp), b = *reinterpret_cast<const float4*>(kp + 4);
            const float z = ((fi >> 3) == hsel) ? kscale : 0.f;
            const op16x8 kq = pack8_d(a.x * z, a.y * z, a.z * z, a.w * z, b.x * z, b.y * z, b.z * z, b.w * z);
            *reinterpret_cast<op16x8*>(reinterpret_cast<char*>(cst) + 2048 + q * 1024 + lane * 16) = kq;
        }
    }
    // LayerNorm gain / bias of the lane's channels in the ROW-MAJOR (store) layout: channels 128 h + 4 (lane & 31) .. + 3 of half h
    const int k32 = lane & 31;
    float gr[2][4], br[2][4];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const float4 g0 = *reinterpret_cast<const float4*>(gamma + 128 * h + 4 * k32), b0 = *reinterpret_cast<const float4*>(beta + 128 * h + 4 * k32);
        gr[h][0] = g0.x; gr[h][1] = g0.y; gr[h][2] = g0.z; gr[h][3] = g0.w;
        br[h][0] = b0.x; br[h][1] = b0.y; br[h][2] = b0.z; br[h][3] = b0.w;
    }

    // ---- LDS-DMA of a tile: 8 X pieces (2 rows of 512 B) + 4 PEQ pieces (4 rows of 256 B), source-side swizzle (dec_i2t_kernel's images)
    const int hrow = lane >> 5, lsw = k32 ^ hrow;
    const int prow = lane >> 4;
    auto issue = [&](int n) {                       // the wave's n-th tile
        const int row0 = (wave + 4 * n) * 16;
        char* dst = ring + (n & 1) * I4_STAGE;
        const bf16_t* src = Xp + (int64_t)(row0 + hrow) * DC;
#pragma unroll
        for (int i = 0; i < 8; ++i)
            __builtin_amdgcn_global_load_lds((gptr_d)(src + (2 * i) * DC + ((lsw ^ ((2 * i) & 15)) << 3)), (lptr_d)(dst + i * 1024), 16, 0, SHARED ? 0 : I2T_X_AUX);
        const bf16_t* psrc = peq + (int64_t)(row0 + prow) * 128;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_global_load_lds((gptr_d)(psrc + (4 * i) * 128 + (((lane & 15) ^ (4 * i + prow)) << 3)), (lptr_d)(dst + I4_XST + i * 1024), 16, 0, 0);
    };
    uint32_t xoff[4];                 // X fragment (row fi, chunk 4 ks + fg): ks and ks + 4 differ by 256 bytes
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) xoff[ks] = fi * ROW_B + (((4 * ks + fg) ^ fi) << 4);
    const uint32_t roff0 = fi * ROW_B + (fg & 1) * 8;       // residual: row fi, channels 16 dt + 4 fg .. + 3 = chunk 2 dt + (fg >> 1), half fg & 1
    const int rch = fg >> 1;
    // transpose scratch: fp32 [16 rows][32 chunks of 16 B] (one channel half), chunk c of row r at ((c ^ r) << 4)
    const uint32_t swr = scr_a + fi * 512;                  // write: row fi, chunk 4 (dt & 7) + fg
    const uint32_t srd = scr_a + hrow * 512;                // read: row 2 i + hrow, chunk k32

    issue(0); issue(1);
    asm volatile("s_waitcnt vmcnt(24) lgkmcnt(0)" ::: "memory");       // the 96 operand loads, the constants (the two tiles' 24 pieces may stay in flight)
    // queue of a wave at the top of step n >= 2, behind the pieces of tile n: S(n - 2) 16, D(n + 1) 12, S(n - 1) 16 = 44 (S = a tile's 16
    // stores, D = its 12 LDS-DMA pieces).  ONE loop body (wave-uniform branches pick the wait): peeled copies of the step kept a dozen
    // loop-invariant addresses alive, and the compiler parked them in "free" AccVGPRs - on top of the folded K operand.
#pragma unroll 1
    for (int n = 0; n < NT; ++n) {
        if (n == 0) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        else if (n >= 2 && n < NT - 2) asm volatile("s_waitcnt vmcnt(44)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const uint32_t xs = ring_a + (n & 1) * I4_STAGE;
        // ---- scores: s[q] = cb + PEQ part + sum over 8 k-steps of (folded K quarter q) . (X fragment)
        // (LGKM_CNT is a 4-bit counter: never more than 15 LDS operations in flight before a wait - with 16 or 20 the counted waits below
        // let the first MFMAs through before their operands had landed: NaN rows, a wrong first channel tile)
        op16x8 xf[8], pf4[4], kq4[4];
        f32x4 s[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) asm volatile("ds_read_b128 %0, %1" : "=v"(s[q]) : "v"(cst_a + 1024 + (16 * q + 4 * fg) * 4) : "memory");
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) asm volatile("ds_read_b128 %0, %1" : "=v"(xf[ks]) : "v"(xs + xoff[ks & 3] + (ks >> 2) * 256) : "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(xf[0]), "+v"(xf[1]), "+v"(xf[2]), "+v"(xf[3]), "+v"(xf[4]), "+v"(xf[5]), "+v"(xf[6]), "+v"(xf[7]),
                     "+v"(s[0]), "+v"(s[1]), "+v"(s[2]), "+v"(s[3]));
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            asm volatile("ds_read_b128 %0, %1" : "=v"(pf4[q]) : "v"(xs + I4_XST + fi * I2T_PEQ_ROWB + (((4 * q + fg) ^ fi) << 4)) : "memory");
            asm volatile("ds_read_b128 %0, %1" : "=v"(kq4[q]) : "v"(cst_a + 2048 + q * 1024 + lane * 16) : "memory");
        }
        t4_for(T4_SEQ(8), [&](auto KS) {
            constexpr int ks = decltype(KS)::value;
            i4_mfma_a<I4_KF(0, ks)>(s[0], xf[ks]); i4_mfma_a<I4_KF(1, ks)>(s[1], xf[ks]);
            i4_mfma_a<I4_KF(2, ks)>(s[2], xf[ks]); i4_mfma_a<I4_KF(3, ks)>(s[3], xf[ks]);
        });
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(pf4[0]), "+v"(pf4[1]), "+v"(pf4[2]), "+v"(pf4[3]), "+v"(kq4[0]), "+v"(kq4[1]), "+v"(kq4[2]), "+v"(kq4[3]));
#pragma unroll
        for (int q = 0; q < 4; ++q) T4_MFMA_V(s[q], kq4[q], pf4[q]);
        // y starts as b_o (the C operand of the first PV product): the first eight channel tiles requested here, under the last score MFMAs
        f32x4 y[16];
#pragma unroll
        for (int dt = 0; dt < 8; ++dt) asm volatile("ds_read_b128 %0, %1" : "=v"(y[dt]) : "v"(cst_a + (16 * dt + 4 * fg) * 4) : "memory");
        asm volatile("s_nop 15\n\ts_nop 7" : "+v"(s[0]), "+v"(s[1]), "+v"(s[2]), "+v"(s[3]));          // the last score MFMAs -> VALU
        // ---- softmax over the 8 tokens of each head: a quarter's 16 columns = heads 2 q (fg 0, 1) and 2 q + 1 (fg 2, 3)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float mx = fmaxf(fmaxf(s[q][0], s[q][1]), fmaxf(s[q][2], s[q][3]));
            mx = xor16_max(mx);
            float sum = 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) { s[q][r] = __builtin_amdgcn_exp2f(s[q][r] - mx); sum += s[q][r]; }
            sum = xor16_sum(sum);
            const float inv = __builtin_amdgcn_rcpf(sum);
#pragma unroll
            for (int r = 0; r < 4; ++r) s[q][r] *= inv;
        }
        const op16x8 p0 = pack8_d(s[0][0], s[0][1], s[0][2], s[0][3], s[1][0], s[1][1], s[1][2], s[1][3]);
        const op16x8 p1 = pack8_d(s[2][0], s[2][1], s[2][2], s[2][3], s[3][0], s[3][1], s[3][2], s[3][3]);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_nop 3" : "+v"(y[0]), "+v"(y[1]), "+v"(y[2]), "+v"(y[3]), "+v"(y[4]), "+v"(y[5]), "+v"(y[6]), "+v"(y[7]));
#pragma unroll
        for (int dt = 8; dt < 16; ++dt) asm volatile("ds_read_b128 %0, %1" : "=v"(y[dt]) : "v"(cst_a + (16 * dt + 4 * fg) * 4) : "memory");
        // ---- y = b_o + (folded V) . P; the residual rows (the last reads of this stage) are requested under the second k-step's products
        t4_for(T4_SEQ(8), [&](auto D) { i4_mfma_a<I4_VF(decltype(D)::value, 0)>(y[decltype(D)::value], p0); });
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(y[8]), "+v"(y[9]), "+v"(y[10]), "+v"(y[11]), "+v"(y[12]), "+v"(y[13]), "+v"(y[14]), "+v"(y[15]));
        u32x2_d res[16];
#pragma unroll
        for (int dt = 0; dt < 8; ++dt)
            asm volatile("ds_read_b64 %0, %1" : "=v"(res[dt]) : "v"(xs + roff0 + (((2 * dt + rch) ^ fi) << 4)) : "memory");
        t4_for(T4_SEQ(8), [&](auto D) { i4_mfma_a<I4_VF(decltype(D)::value + 8, 0)>(y[decltype(D)::value + 8], p0); });
        t4_for(T4_SEQ(8), [&](auto D) { i4_mfma_a<I4_VF(decltype(D)::value, 1)>(y[decltype(D)::value], p1); });
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(res[0]), "+v"(res[1]), "+v"(res[2]), "+v"(res[3]), "+v"(res[4]), "+v"(res[5]), "+v"(res[6]), "+v"(res[7]));
#pragma unroll
        for (int dt = 8; dt < 16; ++dt)
            asm volatile("ds_read_b64 %0, %1" : "=v"(res[dt]) : "v"(xs + roff0 + (((2 * dt + rch) ^ fi) << 4)) : "memory");
        t4_for(T4_SEQ(8), [&](auto D) { i4_mfma_a<I4_VF(decltype(D)::value + 8, 1)>(y[decltype(D)::value + 8], p1); });
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(res[8]), "+v"(res[9]), "+v"(res[10]), "+v"(res[11]), "+v"(res[12]), "+v"(res[13]), "+v"(res[14]), "+v"(res[15]) : : "memory");
        // every read of this stage has returned: refill it with the tile after next
        if (n + 2 < NT) issue(n + 2);
        asm volatile("s_nop 15\n\ts_nop 7" : "+v"(y[0]), "+v"(y[1]), "+v"(y[2]), "+v"(y[3]), "+v"(y[4]), "+v"(y[5]), "+v"(y[6]), "+v"(y[7]),
                     "+v"(y[8]), "+v"(y[9]), "+v"(y[10]), "+v"(y[11]), "+v"(y[12]), "+v"(y[13]), "+v"(y[14]), "+v"(y[15]) : : "memory");
        // ---- residual + LayerNorm statistics of row fi (64 channels in the lane, the other 192 in the lanes fi + 16, + 32, + 48)
        f32x2 sum2 = (f32x2){0.f, 0.f}, sq2 = (f32x2){0.f, 0.f};
#pragma unroll
        for (int dt = 0; dt < 16; ++dt) {
            const uint32_t xlo = res[dt][0], xhi = res[dt][1];
            y[dt][0] += op16_lo(xlo); y[dt][1] += op16_hi(xlo); y[dt][2] += op16_lo(xhi); y[dt][3] += op16_hi(xhi);
            const f32x2 a = (f32x2){y[dt][0], y[dt][1]}, b = (f32x2){y[dt][2], y[dt][3]};
            sum2 += a; sum2 += b;
            sq2 = __builtin_elementwise_fma(a, a, sq2);
            sq2 = __builtin_elementwise_fma(b, b, sq2);
        }
        float sum = sum2.x + sum2.y, sq = sq2.x + sq2.y;
        sum = xor32_sum(xor16_sum(sum)); sq = xor32_sum(xor16_sum(sq));
        const float mean = sum * (1.0f / DC);
        const float rstd = __builtin_amdgcn_rsqf(fmaxf(sq * (1.0f / DC) - mean * mean, 0.f) + eps);
        // ---- (y - mean) rstd -> scratch (fp32), back as rows, gain / bias, 16-bit, 256-B runs; one channel half at a time.  The eight row
        // reads of a half are requested together (one LDS round trip, not eight), and the second half is written while the first is stored.
        bf16_t* orow = Xo + (int64_t)((wave + 4 * n) * 16 + hrow) * DC + 4 * k32;
        auto put_half = [&](int h) {
#pragma unroll
            for (int d8 = 0; d8 < 8; ++d8) {
                const int dt = 8 * h + d8;
                const f32x4 t = (f32x4){(y[dt][0] - mean) * rstd, (y[dt][1] - mean) * rstd, (y[dt][2] - mean) * rstd, (y[dt][3] - mean) * rstd};
                asm volatile("ds_write_b128 %0, %1" : : "v"(swr + (((4 * d8 + fg) ^ fi) << 4)), "v"(t) : "memory");
            }
        };
        f32x4 a[8];
        auto get_half = [&]() {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // the writes
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("ds_read_b128 %0, %1" : "=v"(a[i]) : "v"(srd + i * 1024 + ((k32 ^ (2 * i + hrow)) << 4)) : "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : : "memory");
        };
        auto store_half = [&](int h) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                u32x2_d o;
                o[0] = pack_op16(a[i][0] * gr[h][0] + br[h][0], a[i][1] * gr[h][1] + br[h][1]);
                o[1] = pack_op16(a[i][2] * gr[h][2] + br[h][2], a[i][3] * gr[h][3] + br[h][3]);
                __builtin_nontemporal_store(o, reinterpret_cast<u32x2_d*>(orow + (int64_t)(2 * i) * DC + 128 * h));
            }
        };
        put_half(0);
        get_half();
        put_half(1);            // (the scratch is free again: every read of half 0 has returned)
        store_half(0);
        get_half();
        store_half(1);
    }
}

// dec_i2t_w1_kernel is the route for whole-prompt launches (P >= 512, no XBuild); SABER_AMD_I2T_W1=0 (read per call) or debug flag 0x40000000
// selects the four-wave kernel for A/Bs.
static bool i2t_w1_route() {
    if (g_saber_debug_flags & 0x40000000) return false;
    const char* e = getenv("SABER_AMD_I2T_W1");
    return !(e && e[0] == '0');
}
const char* launch_dec_i2t(const bf16_t* X, XMap xm, const bf16_t* peq, const bf16_t* Kt, const float* tk, float kscale, const float* cb, const bf16_t* VtT,
                           const float* bo, const float* gamma, const float* beta, float eps, bf16_t* Xout, int P, hipStream_t s, const XBuild* build) {
    if (P <= 0) return nullptr;
    const float* nf = nullptr; const bf16_t* nb = nullptr;
    if (build) {
        if (!build->embb || !build->h2 || !build->w3 || build->map.div <= 0) return "dec_i2t: incomplete XBuild";
        int ns = 1;
        while (P * ns < 512 && ns < 8) ns *= 2;
        hipLaunchKernelGGL((dec_i2t_kernel<1, false, true>), dim3(P * ns), dim3(256), I2TCfg<1>::LDS + I2TCfg<1>::W3_B, s, (const bf16_t*)nullptr, build->map.stride, build->map.div, build->map.off, peq, Kt, tk,
                           kscale, cb, VtT, bo, gamma, beta, eps, Xout, ns, 0, (unsigned long long*)nullptr, build->embb, build->h2, build->w3);
        return nullptr;
    }
    int nsplit = 1;
    while (P * nsplit < 512 && nsplit < 8) nsplit *= 2;   // small crops: split a prompt's tiles over several blocks
    if ((g_saber_debug_flags >> 20) & 0xff) nsplit = (g_saber_debug_flags >> 20) & 0xff;      // (bits 8-19 belong to the GEMM kernels, 28-30 to the kernel routes below)
    if (xm.div <= 0) return "dec_i2t: XMap.div must be positive";
    if (nsplit == 1 && !g_saber_stamp_buf && !(g_saber_debug_flags & 1) && i2t_w1_route()) {
        if (xm.div > 1) hipLaunchKernelGGL((dec_i2t_w1_kernel<true>), dim3(P), dim3(256), I4_LDS, s, X, xm.stride, xm.div, xm.off, peq, Kt, tk, kscale, cb, VtT, bo, gamma, beta, eps, Xout);
        else hipLaunchKernelGGL((dec_i2t_w1_kernel<false>), dim3(P), dim3(256), I4_LDS, s, X, xm.stride, xm.div, xm.off, peq, Kt, tk, kscale, cb, VtT, bo, gamma, beta, eps, Xout);
        return nullptr;
    }
    if (g_saber_debug_flags & 1)
        hipLaunchKernelGGL((dec_i2t_kernel<2, false, false>), dim3(P * nsplit), dim3(512), I2TCfg<2>::LDS, s, X, xm.stride, xm.div, xm.off, peq, Kt, tk, kscale, cb, VtT, bo, gamma, beta, eps, Xout, nsplit, 0, g_saber_stamp_buf, nf, nb, nf);
    else
        if (g_saber_stamp_buf)
            hipLaunchKernelGGL((dec_i2t_kernel<1, true, false>), dim3(P * nsplit), dim3(256), I2TCfg<1>::LDS, s, X, xm.stride, xm.div, xm.off, peq, Kt, tk, kscale, cb, VtT, bo, gamma, beta, eps, Xout, nsplit, 0, g_saber_stamp_buf, nf, nb, nf);
        else
            hipLaunchKernelGGL((dec_i2t_kernel<1, false, false>), dim3(P * nsplit), dim3(256), I2TCfg<1>::LDS, s, X, xm.stride, xm.div, xm.off, peq, Kt, tk, kscale, cb, VtT, bo, gamma, beta, eps, Xout, nsplit, 0, g_saber_stamp_buf, nf, nb, nf);
    return nullptr;
}


// ------------------------------------------------------------------------------------------------ image -> tokens of layer l FUSED with the
// tokens -> image attention that follows it (round 5; VERDICT r01-r04 "i2t + next-t2i")
// X' = LN(X + softmax((X + pe).Kt).Vt + b_o) is written to HBM as before, and every pair of finished 16-row tiles is at once the next
// attention's key / value block: the 2 MB of X' per prompt are NOT read back by a separate dec_t2i launch.  The queries of that
// attention depend on the tokens only (dec_tokens_kernel has produced fold_q / tq before this kernel runs).
// Structure: dec_i2t_kernel<1>'s tile loop unchanged (4 waves = the four 16-column quarters of the score matrix, one barrier per tile),
// plus, per wave, ONE 16-row query tile of the next attention (wave = q tile): finish_tile also drops the normalised rows into `xbuf`
// (dec_t2i's tile format: 512-B rows, 16-byte chunks XOR-swizzled by row & 15), four tiles deep, and every second iteration the wave
// runs dec_t2i's block step on the 32 keys of the pair finished two iterations earlier (18 + 16 MFMAs, online softmax).  The PEK rows of
// a pair come global -> LDS directly, two wave-instructions per wave and pair, two stages.
// Registers: dec_i2t's ~200 + 32 (Q fragments) + 64 (partial sums) + ~16: ONE workgroup of four waves per CU (one wave per SIMD) - the
// price of the fusion, see DESIGN.md section 8.2.
#define FZ_XBUF (4 * 16 * ROW_B)                 // 32 KB: four X' tiles
#define FZ_PEK_STAGE (32 * T2I_PEK_ROWB)         // 8 KB: the PEK rows of one pair of tiles
struct I2TFuse {
    const bf16_t* pek; const bf16_t* Qt; const float* tq; float qscale; const bf16_t* Wv; const float* bv; bf16_t* out;
};
template <bool STAMPS>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
void dec_i2t_t2i_kernel(const bf16_t* __restrict__ X, int64_t x_bs, int x_div, int x_off, const bf16_t* __restrict__ peq,
                        const bf16_t* __restrict__ Kt, const float* __restrict__ tk, float kscale, const float* __restrict__ cb,
                        const bf16_t* __restrict__ VtT, const float* __restrict__ bo,
                        const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                        bf16_t* __restrict__ Xout, I2TFuse f, unsigned long long* __restrict__ stamps) {
    using CF = I2TCfg<1>;
    constexpr int I2T_ROWS = CF::ROWS, I2T_STAGE = CF::STAGE, I2T_PBUF_B = CF::PBUF_B, I2T_STAT_B = CF::STAT_B;
    constexpr int NT = 4096 / I2T_ROWS;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    unsigned long long ts[4] = {0, 0, 0, 0}, tprev = 0;
#define FZ_STAMP(k) do { if (STAMPS) { const unsigned long long _n = __builtin_amdgcn_s_memtime(); ts[k] += _n - tprev; tprev = _n; } } while (0)
    char* pbuf = smem + I2T_NSTAGE * I2T_STAGE;
    float* stat = reinterpret_cast<float*>(pbuf + 2 * I2T_PBUF_B);
    char* xbuf = smem + CF::LDS;
    char* pkbuf = xbuf + FZ_XBUF;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int qr = wave;                  // i2t: score quarter;  t2i: query tile
    const int fi = lane & 15, fg = lane >> 4;
    const int p = blockIdx.x;
    const bf16_t* Xp = X + (int64_t)((p + x_off) / x_div) * x_bs;
    bf16_t* Xo = Xout + (int64_t)p * 4096 * DC;

    // ---- i2t operands (dec_i2t_kernel)
    op16x8 kf[8], vf[4][2];
#pragma unroll
    for (int ks = 0; ks < 8; ++ks)
        kf[ks] = __builtin_bit_cast(op16x8, *reinterpret_cast<const uint4*>(Kt + ((int64_t)p * 64 + 16 * qr + fi) * DC + 32 * ks + 8 * fg));
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
            vf[t][ks] = __builtin_bit_cast(op16x8, *reinterpret_cast<const uint4*>(VtT + ((int64_t)p * 256 + 64 * qr + 16 * t + fi) * 64 + 32 * ks + 8 * fg));
    op16x8 kq;
    {
        const int hsel = fg >> 1;
        const float* kp = tk + ((int64_t)p * 8 + (fi & 7)) * 128 + 16 * (2 * qr + hsel) + 8 * (fg & 1);
        const float4 a = *reinterpret_cast<const float4*>(kp), b = *reinterpret_cast<const float4*>(kp + 4);
        const float z = ((fi >> 3) == hsel) ? kscale : 0.f;
        kq = pack8_d(a.x * z, a.y * z, a.z * z, a.w * z, b.x * z, b.y * z, b.z * z, b.w * z);
    }
    const float4 cb4 = *reinterpret_cast<const float4*>(cb + (int64_t)p * 64 + 16 * qr + 4 * fg);
    f32x2 g2[8], be2[8], bo2[8];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const float4 b4 = *reinterpret_cast<const float4*>(bo + 64 * qr + 16 * t + 4 * fg);
        const float4 g4 = *reinterpret_cast<const float4*>(gamma + 64 * qr + 16 * t + 4 * fg);
        const float4 e4 = *reinterpret_cast<const float4*>(beta + 64 * qr + 16 * t + 4 * fg);
        g2[2 * t] = (f32x2){g4.x, g4.y}; g2[2 * t + 1] = (f32x2){g4.z, g4.w};
        be2[2 * t] = (f32x2){e4.x, e4.y}; be2[2 * t + 1] = (f32x2){e4.z, e4.w};
        bo2[2 * t] = (f32x2){b4.x, b4.y}; bo2[2 * t + 1] = (f32x2){b4.z, b4.w};
    }
    // ---- t2i operands of this wave's query tile (dec_t2i_kernel, qt = wave)
    op16x8 qf[8], pq;
    {
        const bf16_t* qrow = f.Qt + ((int64_t)p * 64 + qr * 16 + fi) * DC;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) qf[ks] = __builtin_bit_cast(op16x8, *reinterpret_cast<const uint4*>(qrow + 32 * ks + 8 * fg));
        const int hsel = fg >> 1;
        const float* qp = f.tq + ((int64_t)p * 8 + (fi & 7)) * 128 + 16 * (2 * qr + hsel) + 8 * (fg & 1);
        const float4 a = *reinterpret_cast<const float4*>(qp), b = *reinterpret_cast<const float4*>(qp + 4);
        const float z = ((fi >> 3) == hsel) ? f.qscale : 0.f;
        pq = pack8_d(a.x * z, a.y * z, a.z * z, a.w * z, b.x * z, b.y * z, b.z * z, b.w * z);
    }
    float m = -3.0e38f, l = 0.f;
    f32x4 o[16];
#pragma unroll
    for (int dt = 0; dt < 16; ++dt) o[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // ---- LDS-DMA: per tile 2 X pieces + 1 PEQ piece per wave (dec_i2t_kernel), per PAIR of tiles 2 PEK pieces per wave
    int srow[2], schunk[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        srow[i] = 2 * (wave * 2 + i) + (lane >> 5);
        schunk[i] = (lane & 31) ^ (srow[i] & 15);
    }
    const int prow = 4 * wave + (lane >> 4), pchunk = (lane & 15) ^ (prow & 15);
    auto issue = [&](int t) {
        char* sx = smem + (t & (I2T_NSTAGE - 1)) * I2T_STAGE;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int64_t off = (int64_t)(t * I2T_ROWS + srow[i]) * DC + schunk[i] * 8;
            if (x_div > 1) __builtin_amdgcn_global_load_lds((gptr_d)(Xp + off), (lptr_d)(sx + (wave * 2 + i) * 1024), 16, 0, 0);
            else __builtin_amdgcn_global_load_lds((gptr_d)(Xp + off), (lptr_d)(sx + (wave * 2 + i) * 1024), 16, 0, I2T_X_AUX);
        }
        __builtin_amdgcn_global_load_lds((gptr_d)(peq + (int64_t)(t * I2T_ROWS + prow) * 128 + pchunk * 8), (lptr_d)(sx + I2T_ROWS * ROW_B + wave * 1024), 16, 0, 0);
    };
    int krow[2], kchunk[2];          // PEK: 32 rows x 256 B = 8 pieces of 4 rows; wave w issues pieces 2w, 2w + 1
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        krow[i] = 4 * (wave * 2 + i) + (lane >> 4);
        kchunk[i] = (lane & 15) ^ (krow[i] & 15);
    }
    auto issue_pek = [&](int pair) {             // the PEK rows of tiles 2 pair, 2 pair + 1
        char* sk = pkbuf + (pair & 1) * FZ_PEK_STAGE;
#pragma unroll
        for (int i = 0; i < 2; ++i)
            __builtin_amdgcn_global_load_lds((gptr_d)(f.pek + (int64_t)(pair * 32 + krow[i]) * 128 + kchunk[i] * 8), (lptr_d)(sk + (wave * 2 + i) * 1024), 16, 0, 0);
    };
    const int poff = fi * I2T_PEQ_ROWB + (((4 * qr + fg) ^ fi) << 4);
    int xoff[8];
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) xoff[ks] = fi * ROW_B + (((4 * ks + fg) ^ fi) << 4);
    int roff[4];      // this lane's 8-byte pieces of row fi: channels 64 qr + 16 t + 4 fg .. + 3, in the swizzled tile format
#pragma unroll
    for (int t = 0; t < 4; ++t) roff[t] = fi * ROW_B + (((8 * qr + 2 * t + (fg >> 1)) ^ fi) << 4) + (fg & 1) * 8;
    const uint32_t prow_a = (uint32_t)(uintptr_t)(lptr_d)(pbuf + fi * I2T_PSTRIDE);
    const uint32_t stat_a = (uint32_t)(uintptr_t)(lptr_d)(stat + (fi * 4) * 2);
    const uint32_t smem_a = (uint32_t)(uintptr_t)(lptr_d)smem;
    const uint32_t oscr_a = smem_a + I2T_NSTAGE * I2T_STAGE + 2 * I2T_PBUF_B + 2 * I2T_STAT_B + wave * 2048;
    const uint32_t xbuf_a = (uint32_t)(uintptr_t)(lptr_d)xbuf;
    f32x2 y2[8];
    auto finish_tile = [&](int tp) {
        f32x4 a, b;
        asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:16\n\ts_waitcnt lgkmcnt(0)" : "=&v"(a), "=&v"(b) : "v"(stat_a + (tp & 1) * I2T_STAT_B) : "memory");
        const float tot = (a.x + a.z) + (b.x + b.z), tsq = (a.y + a.w) + (b.y + b.w);
        const float mean = tot * (1.0f / DC);
        const float rstd = __builtin_amdgcn_rsqf(fmaxf(tsq * (1.0f / DC) - mean * mean, 0.f) + eps);
        const f32x2 mean2 = (f32x2){mean, mean}, rstd2 = (f32x2){rstd, rstd};
        const uint32_t tb = oscr_a + fi * 128 + (fg & 1) * 8;
        const uint32_t xb = xbuf_a + (tp & 3) * (16 * ROW_B);
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) {
            const f32x2 v0 = ((y2[2 * tt] - mean2) * rstd2) * g2[2 * tt] + be2[2 * tt];
            const f32x2 v1 = ((y2[2 * tt + 1] - mean2) * rstd2) * g2[2 * tt + 1] + be2[2 * tt + 1];
            const uint32_t lo = pack_op16(v0.x, v0.y), hi = pack_op16(v1.x, v1.y);
            lds_write_b64(tb + (((2 * tt + (fg >> 1)) ^ (fi & 7)) << 4), lo, hi);
            lds_write_b64(xb + roff[tt], lo, hi);            // the next attention's key / value tile
        }
        u32x4 o0, o1;
        asm volatile("s_waitcnt lgkmcnt(0)\n\tds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:1024\n\ts_waitcnt lgkmcnt(0)"
                     : "=&v"(o0), "=&v"(o1) : "v"(oscr_a + (lane >> 3) * 128 + (((lane & 7) ^ ((lane >> 3) & 7)) << 4)) : "memory");
        bf16_t* orow = Xo + (int64_t)(tp * I2T_ROWS + (lane >> 3)) * DC + 64 * qr + 8 * (lane & 7);
        __builtin_nontemporal_store(o0, reinterpret_cast<u32x4*>(orow));
        __builtin_nontemporal_store(o1, reinterpret_cast<u32x4*>(orow + 8 * DC));
    };
    // dec_t2i_kernel's block step on the 32 keys of tiles 2 pair, 2 pair + 1 (this wave's 16 query rows)
    int koff[8];
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) koff[ks] = fi * ROW_B + (((4 * ks + fg) ^ fi) << 4);
    const int vrow = 4 * fg + (fi >> 2);
    const int vsel = (fi & 3) >> 1, vlow = (fi & 1) * 8;
    auto t2i_step = [&](int pair) {
        const uint32_t xs = xbuf_a + (pair & 1) * (32 * ROW_B);
        const uint32_t ps = (uint32_t)(uintptr_t)(lptr_d)pkbuf + (pair & 1) * FZ_PEK_STAGE;
        f32x4 s[2];
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
            op16x8 kp, kx[8];
            asm volatile("ds_read_b128 %0, %1" : "=v"(kp) : "v"(ps + (kt * 16 + fi) * T2I_PEK_ROWB + (((4 * qr + fg) ^ fi) << 4)) : "memory");
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) asm volatile("ds_read_b128 %0, %1" : "=v"(kx[ks]) : "v"(xs + kt * 16 * ROW_B + koff[ks]) : "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(kp), "+v"(kx[0]), "+v"(kx[1]), "+v"(kx[2]), "+v"(kx[3]), "+v"(kx[4]), "+v"(kx[5]), "+v"(kx[6]), "+v"(kx[7]));
            s[kt] = MFMA_16x16x32(kp, pq, (f32x4){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) s[kt] = MFMA_16x16x32(kx[ks], qf[ks], s[kt], 0, 0, 0);
        }
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
        const op16x8 pf = pack8_d(s[0][0], s[0][1], s[0][2], s[0][3], s[1][0], s[1][1], s[1][2], s[1][3]);
        const uint32_t va = xs + vrow * ROW_B + vlow;
#pragma unroll
        for (int d4 = 0; d4 < 16; d4 += 4) {
            u32x2_d lo[4], hi[4];
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
                const op16x8 vfr = cat4_d(__builtin_bit_cast(op16x4, lo[j]), __builtin_bit_cast(op16x4, hi[j]));
                o[d4 + j] = MFMA_16x16x32(vfr, pf, o[d4 + j], 0, 0, 0);
            }
        }
    };

    issue(0); issue(1); issue(2);
    issue_pek(0);
    // queue of this wave: L(0) L(1) L(2) K(0); tile 0 must have landed
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (STAMPS) tprev = __builtin_amdgcn_s_memtime();
    for (int t = 0; t < NT; ++t) {
        const char* xs = smem + (t & (I2T_NSTAGE - 1)) * I2T_STAGE;
        const char* ps = xs + I2T_ROWS * ROW_B;
        f32x4 s = (f32x4){cb4.x, cb4.y, cb4.z, cb4.w}, s1;
        {
            const op16x8 pf = *reinterpret_cast<const op16x8*>(ps + poff);
            s1 = MFMA_16x16x32(kq, pf, (f32x4){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
        }
#pragma unroll
        for (int ks = 0; ks < 8; ks += 2) {
            const op16x8 xf0 = *reinterpret_cast<const op16x8*>(xs + xoff[ks]);
            const op16x8 xf1 = *reinterpret_cast<const op16x8*>(xs + xoff[ks + 1]);
            s = MFMA_16x16x32(kf[ks], xf0, s, 0, 0, 0);
            s1 = MFMA_16x16x32(kf[ks + 1], xf1, s1, 0, 0, 0);
        }
        s += s1;
        {
            float mx = fmaxf(fmaxf(s[0], s[1]), fmaxf(s[2], s[3]));
            mx = xor16_max(mx);
            float sum = 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) { s[r] = __builtin_amdgcn_exp2f(s[r] - mx); sum += s[r]; }
            sum = xor16_sum(sum);
            const float inv = __builtin_amdgcn_rcpf(sum);
            lds_write_b64(prow_a + (t & 1) * I2T_PBUF_B + (16 * qr + 4 * fg) * 2, pack_op16(s[0] * inv, s[1] * inv), pack_op16(s[2] * inv, s[3] * inv));
        }
        FZ_STAMP(0);
        // Tile t + 1 must have landed before the barrier publishes it.  After its barrier iteration i issues L(i + 3) [3 operations],
        // S(i - 1) [2 stores] and, when i is even and >= 2, K(i / 2) [2] - the PEK rows of the pair that is consumed in iteration i + 3, into
        // the stage that the step of iteration i - 1 has read (every wave is past that step: this iteration's barrier).  Behind L(t + 1) in
        // this wave's queue: S(t - 3), L(t + 2), S(t - 2) and exactly one K: 9 operations may stay in flight.  The PEK rows an odd
        // iteration consumes were issued BEFORE L(t + 1) (iteration t - 3), so the same wait covers them.
        if (t + 3 >= NT) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (t < 3) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");            // t = 0: L(2) K(0);  t = 1: K(0) L(3);  t = 2: L(4) S(0)
        else asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        FZ_STAMP(1);
        if (t + 3 < NT) issue(t + 3);
        if (t > 0) finish_tile(t - 1);
        // the pair (t - 3, t - 2): both tiles were dropped into xbuf before this iteration's barrier
        if ((t & 1) && t >= 3) t2i_step((t - 3) >> 1);
        else if (!(t & 1) && t >= 2) issue_pek(t >> 1);
        FZ_STAMP(2);
        f32x4 y[4];
        {
            op16x8 p0, p1;
            asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:64\n\ts_waitcnt lgkmcnt(0)" : "=&v"(p0), "=&v"(p1) : "v"(prow_a + (t & 1) * I2T_PBUF_B + 16 * fg) : "memory");
#pragma unroll
            for (int tt = 0; tt < 4; ++tt) {
                y[tt] = (f32x4){0.f, 0.f, 0.f, 0.f};
                y[tt] = MFMA_16x16x32(vf[tt][0], p0, y[tt], 0, 0, 0);
                y[tt] = MFMA_16x16x32(vf[tt][1], p1, y[tt], 0, 0, 0);
            }
        }
        uint64_t xr0, xr1, xr2, xr3;
        {
            const uint32_t xa = smem_a + (t & (I2T_NSTAGE - 1)) * I2T_STAGE;
            asm volatile("ds_read_b64 %0, %4\n\tds_read_b64 %1, %5\n\tds_read_b64 %2, %6\n\tds_read_b64 %3, %7\n\ts_waitcnt lgkmcnt(0)"
                         : "=&v"(xr0), "=&v"(xr1), "=&v"(xr2), "=&v"(xr3)
                         : "v"(xa + roff[0]), "v"(xa + roff[1]), "v"(xa + roff[2]), "v"(xa + roff[3])
                         : "memory");
        }
        const uint64_t xrs[4] = {xr0, xr1, xr2, xr3};
        f32x2 sum2 = (f32x2){0.f, 0.f}, sq2 = (f32x2){0.f, 0.f};
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) {
            const uint32_t xlo = (uint32_t)xrs[tt], xhi = (uint32_t)(xrs[tt] >> 32);
            const f32x2 r0 = (f32x2){op16_lo(xlo), op16_hi(xlo)};
            const f32x2 r1 = (f32x2){op16_lo(xhi), op16_hi(xhi)};
            y2[2 * tt] = ((f32x2){y[tt][0], y[tt][1]} + bo2[2 * tt]) + r0;
            y2[2 * tt + 1] = ((f32x2){y[tt][2], y[tt][3]} + bo2[2 * tt + 1]) + r1;
            sum2 += y2[2 * tt]; sum2 += y2[2 * tt + 1];
            sq2 = __builtin_elementwise_fma(y2[2 * tt], y2[2 * tt], sq2);
            sq2 = __builtin_elementwise_fma(y2[2 * tt + 1], y2[2 * tt + 1], sq2);
        }
        float sum = sum2.x + sum2.y, sq = sq2.x + sq2.y;
        sum = xor16_sum(sum); sq = xor16_sum(sq);
        sum = xor32_sum(sum); sq = xor32_sum(sq);
        if (fg == 0) lds_write_b64(stat_a + (t & 1) * I2T_STAT_B + qr * 8, __float_as_uint(sum), __float_as_uint(sq));
        FZ_STAMP(3);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    finish_tile(NT - 1);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    // the loop's last step (t = NT - 1) took the pair (NT - 4, NT - 3); the last pair is left
    t2i_step(NT / 2 - 1);
    if (STAMPS && stamps && lane == 0)
        for (int k = 0; k < 4; ++k) stamps[((int64_t)blockIdx.x * 4 + wave) * 4 + k] = ts[k];
    // ---- normalise and apply v_proj (dec_t2i_kernel's tail for a whole key range in one workgroup): out[p][t][16 h + i]
    __syncthreads();
    float* mo = reinterpret_cast<float*>(smem) + (size_t)qr * 16 * 260;
    {
        const float inv = __builtin_amdgcn_rcpf(l);
#pragma unroll
        for (int dt = 0; dt < 16; ++dt)
            *reinterpret_cast<float4*>(mo + fi * 260 + 16 * dt + 4 * fg) = make_float4(o[dt][0] * inv, o[dt][1] * inv, o[dt][2] * inv, o[dt][3] * inv);
        __builtin_amdgcn_wave_barrier();
        f32x4 r0 = (f32x4){0.f, 0.f, 0.f, 0.f}, r1 = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            const float4 z0 = *reinterpret_cast<const float4*>(mo + fi * 260 + 32 * ks + 8 * fg), z1 = *reinterpret_cast<const float4*>(mo + fi * 260 + 32 * ks + 8 * fg + 4);
            const op16x8 zf = pack8_d(z0.x, z0.y, z0.z, z0.w, z1.x, z1.y, z1.z, z1.w);
            const op16x8 w0 = __builtin_bit_cast(op16x8, *reinterpret_cast<const uint4*>(f.Wv + (int64_t)(32 * qr + fi) * DC + 32 * ks + 8 * fg));
            const op16x8 w1 = __builtin_bit_cast(op16x8, *reinterpret_cast<const uint4*>(f.Wv + (int64_t)(32 * qr + 16 + fi) * DC + 32 * ks + 8 * fg));
            r0 = MFMA_16x16x32(w0, zf, r0, 0, 0, 0);
            r1 = MFMA_16x16x32(w1, zf, r1, 0, 0, 0);
        }
        const int hsel = fi >> 3, hh = 2 * qr + hsel, tt = fi & 7;
        const float4 b4 = *reinterpret_cast<const float4*>(f.bv + 16 * hh + 4 * fg);
        const f32x4 r = hsel ? r1 : r0;
        *reinterpret_cast<uint2*>(f.out + (int64_t)p * 1024 + tt * 128 + 16 * hh + 4 * fg) =
            make_uint2(pack_op16(r[0] + b4.x, r[1] + b4.y), pack_op16(r[2] + b4.z, r[3] + b4.w));
    }
#undef FZ_STAMP
}

const char* launch_dec_i2t_t2i(const bf16_t* X, XMap xm, const bf16_t* peq, const bf16_t* Kt, const float* tk, float kscale, const float* cb, const bf16_t* VtT,
                               const float* bo, const float* gamma, const float* beta, float eps, bf16_t* Xout, int P,
                               const bf16_t* pek, const bf16_t* Qt, const float* tq, float qscale, const bf16_t* Wv, const float* bv, bf16_t* out, hipStream_t s) {
    if (P <= 0) return nullptr;
    if (xm.div <= 0) return "dec_i2t_t2i: XMap.div must be positive";
    const I2TFuse f{pek, Qt, tq, qscale, Wv, bv, out};
    constexpr int lds = I2TCfg<1>::LDS + FZ_XBUF + 2 * FZ_PEK_STAGE;
    static_assert(lds >= 4 * 16 * 260 * 4, "the v_proj tail re-uses the front of the LDS");
    if (g_saber_stamp_buf)
        hipLaunchKernelGGL((dec_i2t_t2i_kernel<true>), dim3(P), dim3(256), lds, s, X, xm.stride, xm.div, xm.off, peq, Kt, tk, kscale, cb, VtT, bo, gamma, beta, eps, Xout, f, g_saber_stamp_buf);
    else
        hipLaunchKernelGGL((dec_i2t_t2i_kernel<false>), dim3(P), dim3(256), lds, s, X, xm.stride, xm.div, xm.off, peq, Kt, tk, kscale, cb, VtT, bo, gamma, beta, eps, Xout, f, (unsigned long long*)nullptr);
    return nullptr;
}

// ------------------------------------------------------------------------------------------------ upscaling head
// Fused output_upscaling + hypernetwork product:
//   u1 = GELU(LN64(ConvT1(x) + feat_s1));  u2 = GELU(ConvT2(u1) + feat_s0);  masks[k] = hyper[k] . u2
// Persistent blocks: one block owns a tile of 32 consecutive tokens (2x16 tokens = 8x64 output pixels) and walks over
// the prompts.  Both ConvTranspose weights stay resident in LDS (128 KB + 16 KB), the tile's feat_s1 / feat_s0 rows and
// the LayerNorm parameters stay in registers.  The X rows of a wave's 16 tokens are loaded straight into MFMA operand registers
// (next prompt prefetched while the current one is computed) and every lane stores its own 2x2 output pixels, so the prompt
// loop contains NO workgroup barrier: the two waves of a SIMD drift apart and one's MFMA phases overlap the other's
// LayerNorm / GELU / hyper-product VALU phases (in lock-step the VALU work was 60 % of the time with the matrix cores idle).
// Phase A: [32 tok] x [256 = pos*64+ch] over K=256; wave w owns pos = w, so its LayerNorm groups are wave-local and its
// GELU outputs feed phase B straight from registers (k-slot permutation, W2 pre-permuted on the host).
// Phase B: [(pos w, 32 tok)] x [128 = pos2*32+ch2] over K=64, epilogue = +feat_s0, GELU, 4 dot products with hyper.
// The 32-channel 256x256 upscaled embedding (8 MB fp32 per prompt) never exists in memory.
// Round 4: THREE waves per SIMD.  SQ counters under the 8-wave kernel (profiles/r04_decoder_sq_counters_8wave.txt, tools/sq_counters.sh): a wave
// spends 46 % of its cycles executing VALU instructions (703 per prompt at 4.6 cycles each: a single wave issues one VALU instruction per
// 4 cycles, the SIMD-32 pipe takes one per 2), 29 % waiting to issue and 17 % parked at a waitcnt; compile-time ablations (UP_ABL,
// tools/upscale_ablate.sh): GELU -33 %, hypernetwork product -13 %, X loads -3 %, LayerNorm reductions / stores / MFMAs ~0, everything
// arithmetic off: still 49 % (the 48 weight-fragment reads per wave and prompt, 384 KB of LDS traffic per prompt and CU).  So the kernel is
// bound by instruction latency at two waves per SIMD, not by a pipe: the fix is a third wave, which needs <= 168 registers (220 before):
//   * the hypernetwork product runs on the matrix cores (the round-3 experiment UP_MFMA_HYPER: 8 operand registers instead of 32; u2 is
//     rounded to the 16-bit operand type there),
//   * the LayerNorm gain / bias live in 512 B of LDS instead of 32 registers,
//   * X rows are addressed through a per-prompt buffer descriptor (one 32-bit lane offset).
// 12 waves = 4 ConvT1 positions x UP_NTG = 3 groups of 16 tokens: a tile is 48 tokens, 4096 = 85 x 48 + 16, so the (tile, prompt) units are
// dealt to ONE resident workgroup per CU as equal contiguous ranges (tile-major) instead of a tile per workgroup.
// make EXTRA="-DUP_NTG=2 -DUP_MFMA_HYPER=0" rebuilds the round-3 configuration (8 waves, VALU hypernetwork product) for A/Bs.
#ifndef UP_NTG
#define UP_NTG 3
#endif
#ifndef UP_MFMA_HYPER
#define UP_MFMA_HYPER 1
#endif
#ifndef UP_NS
#define UP_NS 1                       // prompts per iteration of a wave ("streams"): 2 halves the weight-fragment reads per prompt at 2 x the per-prompt registers
#endif
#define UP_TOK (16 * UP_NTG)
#define UP_TILES ((4096 + UP_TOK - 1) / UP_TOK)
#define UP_THREADS (256 * UP_NTG)
// Development build (make EXTRA=-DUP_DEV=1 BUILD=build_dev LIB=../libsaber_amd_dev.so; tools/upscale_ablate.py): run-time switches that
// take single pieces of the prompt loop away (results are garbage) and per-phase s_memtime stamps, to see what the kernel's time is made
// of.  0 in the shipped library: every UPD(..) below folds to false and the stamps disappear.
// -DUP_ABL=<mask> instead: the same switches fixed at COMPILE time (no run-time branches: the register allocation and schedule of the
// shipped kernel minus the piece; tools/upscale_ablate.sh builds one library variant per mask).
#ifndef UP_DEV
#define UP_DEV 0
#endif
#ifndef UP_ABL
#define UP_ABL 0
#endif
#define UPD(bit) ((UP_ABL & (bit)) || (UP_DEV && (dbg & (bit))))
enum { UPD_NO_GELU = 1, UPD_NO_W1_READ = 2, UPD_NO_W2_READ = 4, UPD_NO_STORE = 8, UPD_NO_HYPER = 16, UPD_NO_XLOAD = 32, UPD_NO_LN = 64, UPD_NO_MFMA_A = 128, UPD_NO_MFMA_B = 256 };
#define UP_W1S (256 * ROW_B)          // W1 [256 n][256 k] bf16, kswz
#define UP_W2S (128 * 128)            // W2p [128 n2][64] bf16 (k-slots pre-permuted), swz128
#define UP_LNS 512                    // LayerNorm gain [64] + bias [64], fp32
#define UP_LDS (UP_W1S + UP_W2S + UP_LNS)
__device__ __forceinline__ int swz128(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }

__global__ __launch_bounds__(UP_THREADS) void dec_upscale_kernel(const bf16_t* __restrict__ X, const bf16_t* __restrict__ W1, const float* __restrict__ b1,
                                                                 const float* __restrict__ ln_g, const float* __restrict__ ln_b,
                                                                 const bf16_t* __restrict__ W2p, const float* __restrict__ b2,
                                                                 const float* __restrict__ fs1, const float* __restrict__ fs0, int s_div, int s_off,
                                                                 const float* __restrict__ hyper, float* __restrict__ masks4, int P,
                                                                 const uint8_t* __restrict__ live, const float* __restrict__ iou4, int multimask, int dbg,
                                                                 unsigned long long* __restrict__ stamps, unsigned int* __restrict__ sentinel) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* w1s = smem;
    char* w2s = w1s + UP_W1S;
    float* lnp = reinterpret_cast<float*>(w2s + UP_W2S);
    unsigned long long ts[5] = {0, 0, 0, 0, 0}, tprev = 0;
#define UP_STAMP(k) do { if (UP_DEV && stamps) { const unsigned long long _n = __builtin_amdgcn_s_memtime(); ts[k] += _n - tprev; tprev = _n; } } while (0)
    // wave = (token group tg) * 4 + (pos = ConvT1 output position); UP_NTG waves per SIMD (the epilogues are bound by VALU latency: see above)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pos = wave & 3, tg = wave >> 2;
    const int fi = lane & 15, fg = lane >> 4;

    // resident weights
    for (int idx = tid; idx < 256 * 32; idx += UP_THREADS) {
        const int row = idx >> 5, ch = idx & 31;
        *reinterpret_cast<u32x4*>(w1s + kswz(row, ch)) = *reinterpret_cast<const u32x4*>(W1 + row * 256 + ch * 8);
    }
    for (int idx = tid; idx < 128 * 8; idx += UP_THREADS) {
        const int row = idx >> 3, ch = idx & 7;
        *reinterpret_cast<u32x4*>(w2s + swz128(row, ch)) = *reinterpret_cast<const u32x4*>(W2p + row * 64 + ch * 8);
    }
    if (tid < 64) { lnp[tid] = ln_g[tid]; lnp[64 + tid] = ln_b[tid]; }
    __syncthreads();   // resident weights visible; the only workgroup barrier of the kernel

    // this workgroup's share of the (tile, prompt) units, tile-major: [u0, u1)
    const long long U = (long long)UP_TILES * P;
    const long long u0 = U * blockIdx.x / gridDim.x, u1 = U * (blockIdx.x + 1) / gridDim.x;
    const int dy1 = pos >> 1, dx1 = pos & 1;
    const bool fb0 = fg & 1, fb1 = fg >> 1;
    // live (optional): prompts whose flag is 0 are skipped altogether (their masks are never read: engine.hip decode_chunk, IoU pruning)
    auto next_live = [&](int q, int qe) { while (live && q < qe && !live[q]) ++q; return q; };     // block-uniform

    // overflow sentinel (engine.hip "sentinel"): x * 0 is NaN for a NaN / inf logit and 0 otherwise, so one fma per stored pixel keeps a
    // sticky flag in `chk`; one atomic per wave at the end, and only when something was seen
    float chk = 0.f;
    for (long long u = u0; u < u1;) {
        const int tile = (int)(u / P);
        const int p_begin = (int)(u - (long long)tile * P);
        const int p_end = (int)((u1 - (long long)tile * P) < (long long)P ? (u1 - (long long)tile * P) : (long long)P);
        u = (long long)tile * P + p_end;
        const int tok = tile * UP_TOK + tg * 16 + fi;
        if (tile * UP_TOK + tg * 16 >= 4096) continue;          // wave-uniform: the last tile holds 16 tokens (4096 = 85 x 48 + 16)
        // per-lane constants of this tile: feat_s1 + b1 and feat_s0 + b2 (fp32: they are the C operands of the first MFMA of their
        // accumulator, so adding them costs nothing)
        float4 f1[4];
        f32x4 f0[4][2];
        // the high-resolution features belong to the crop (slot) of the prompt: prompts of one slot are contiguous, so a wave
        // re-reads them only when its prompt sequence crosses into the next slot
        int cur_slot = -1;
        auto load_feats = [&](int slot) {
            const float* s1 = fs1 + (int64_t)slot * 16384 * 64;
            const float* s0 = fs0 + (int64_t)slot * 65536 * 32;
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) {
                const float4 r = *reinterpret_cast<const float4*>(s1 + ((int64_t)tok * 4 + pos) * 64 + ni * 16 + 4 * fg);
                const float4 b = *reinterpret_cast<const float4*>(b1 + pos * 64 + ni * 16 + 4 * fg);
                f1[ni] = make_float4(r.x + b.x, r.y + b.y, r.z + b.z, r.w + b.w);
            }
#pragma unroll
            for (int pos2 = 0; pos2 < 4; ++pos2)
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) {
                    const float4 r = *reinterpret_cast<const float4*>(s0 + (((int64_t)tok * 4 + pos) * 4 + pos2) * 32 + hh * 16 + 4 * fg);
                    const float4 b = *reinterpret_cast<const float4*>(b2 + pos2 * 32 + hh * 16 + 4 * fg);
                    f0[pos2][hh] = (f32x4){r.x + b.x, r.y + b.y, r.z + b.z, r.w + b.w};
                }
        };
        // B-operand fragments of the wave's 16 tokens: lane (fi, fg) holds X[tok][32 ks + 8 fg .. +7], ks = 0..7; one 32-bit lane offset into
        // a per-prompt buffer descriptor (the prompt's 2-MB state)
        const uint32_t xoff = (uint32_t)((tok * DC + 8 * fg) * 2);
        op16x8 xf[UP_NS][8];
        auto xload = [&](int p, op16x8 (&dst)[8]) {
            const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)(X + (int64_t)p * 4096 * DC), 0, 4096 * DC * 2, 0x00020000);
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) dst[ks] = __builtin_bit_cast(op16x8, __builtin_amdgcn_raw_buffer_load_b128(xr, xoff + 64 * ks, 0, 0));
        };
        // output pixels of this lane: token (gy, gx) on the 64x64 grid, ConvT1 position (dy1, dx1); mask k = fg
        int gy, gx;
        perm_coords(tok, 2, &gy, &gx);
        const int64_t obase = ((int64_t)fg * 256 + gy * 4 + dy1 * 2) * 256 + gx * 4 + dx1 * 2;

        // UP_NS prompts per iteration (streams): every weight fragment read from LDS feeds one MFMA per stream, and the streams' epilogues are
        // independent instruction chains the scheduler interleaves.  A stream-1 prompt must belong to the slot (crop) of its stream-0 prompt
        // (the tile's feature registers are per slot); where none is left (odd tail, slot boundary) stream 1 idles on stale data, nothing stored.
        auto slot_of = [&](int q) { return (q + s_off) / s_div; };
        auto pick = [&](int pa, int (&pp)[UP_NS]) {       // fills pp[] from the first live prompt at or after pa; returns where the next iteration starts
            pp[0] = next_live(pa, p_end);
            int q = pp[0] < p_end ? next_live(pp[0] + 1, p_end) : p_end;
#pragma unroll
            for (int sI = 1; sI < UP_NS; ++sI) {
                if (q < p_end && slot_of(q) == slot_of(pp[0])) { pp[sI] = q; q = next_live(q + 1, p_end); }
                else pp[sI] = -1;
            }
            return q;
        };
        int cur[UP_NS], nxt[UP_NS];
        int q_next = pick(p_begin, cur);
#pragma unroll
        for (int sI = 0; sI < UP_NS; ++sI) if (cur[sI] >= 0 && cur[sI] < p_end) xload(cur[sI], xf[sI]);
        if (UP_DEV && stamps) tprev = __builtin_amdgcn_s_memtime();
        while (cur[0] < p_end) {
            const int q_after = pick(q_next, nxt);
            // compiler fence: without it the loop-invariant weight fragments (32 + 16 ds_read_b128 per wave) are hoisted out of the
            // prompt loop and spill
            asm volatile("" ::: "memory");
            {
                const int slot = slot_of(cur[0]);     // block-uniform
                if (slot != cur_slot) { load_feats(slot); cur_slot = slot; }
            }
            // Planes anybody reads afterwards (iou4 given): a multimask decode returns masks 1-3; a single-mask decode returns mask 0 or, when
            // that one is unstable, the best of 1-3 by predicted IoU (first maximum: mask_pick_kernel / mask_select_dynamic_kernel) - the other
            // planes are not written (with the hypernetwork product on the VALU their dot products are skipped too).  Block-uniform.
            int need[UP_NS];
            float* orow[UP_NS];
#pragma unroll
            for (int sI = 0; sI < UP_NS; ++sI) {
                const int p = cur[sI];
                need[sI] = p < 0 ? 0 : 0xF;
                if (iou4 && p >= 0) {
                    if (multimask) need[sI] = 0xE;
                    else {
                        int best = 1;
                        float bv = iou4[p * 4 + 1];
                        if (iou4[p * 4 + 2] > bv) { bv = iou4[p * 4 + 2]; best = 2; }
                        if (iou4[p * 4 + 3] > bv) best = 3;
                        need[sI] = 1 | (1 << best);
                    }
                }
                need[sI] = __builtin_amdgcn_readfirstlane(need[sI]);
                orow[sI] = masks4 + (int64_t)(p < 0 ? cur[0] : p) * 4 * 65536 + obase;
            }
#if UP_MFMA_HYPER
            // The hypernetwork product masks[k] = hyper[k] . u2 on the matrix cores (instead of 128 FMAs + a 3-step lane transpose per lane and
            // prompt).  A operand:
            // row fi = mask fi >> 2 (each mask's row four times), k-slot 8 fg + j = channel 4 fg + j (j < 4) | 16 + 4 fg + (j - 4): the order
            // phase B leaves its GELU outputs in, so they are packed into the B operand as they stand.  Every row 4 fg + r of the result is
            // mask fg of token fi: the lane reads its own mask from register 0, no lane movement.  hyper enters as a hi + lo pair of the 16-bit
            // operand type (two MFMAs, exact to 2^-17 / 2^-23); u2 is rounded to that type (2^-9 / 2^-12 per element, averaged over the 32-term sum).
            op16x8 hy_hi[UP_NS], hy_lo[UP_NS];
#pragma unroll
            for (int sI = 0; sI < UP_NS; ++sI) {
                const float* hp = hyper + (int64_t)(cur[sI] < 0 ? cur[0] : cur[sI]) * 128 + (fi >> 2) * 32 + 4 * fg;
                const float4 a = *reinterpret_cast<const float4*>(hp), b = *reinterpret_cast<const float4*>(hp + 16);
                const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
                float l[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) l[j] = v[j] - op2f(f2op(v[j]));
                hy_hi[sI] = pack8_d(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7]);
                hy_lo[sI] = pack8_d(l[0], l[1], l[2], l[3], l[4], l[5], l[6], l[7]);
            }
#else
            static_assert(UP_NS == 1, "the VALU hypernetwork product is built for one prompt per iteration");
            float4 hy[2][4];
#pragma unroll
            for (int hh = 0; hh < 2; ++hh)
#pragma unroll
                for (int k = 0; k < 4; ++k) hy[hh][k] = *reinterpret_cast<const float4*>(hyper + (int64_t)cur[0] * 128 + k * 32 + hh * 16 + 4 * fg);
#endif
            // ---------------- phase A: [16 tok of this wave] x [64 outputs of pos] over K = 256, per stream
            f32x4 acc[UP_NS][4];
#pragma unroll
            for (int sI = 0; sI < UP_NS; ++sI)
#pragma unroll
                for (int ni = 0; ni < 4; ++ni) acc[sI][ni] = (f32x4){f1[ni].x, f1[ni].y, f1[ni].z, f1[ni].w};       // bias + feat_s1 enter as the C operand
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
#pragma unroll
                for (int ni = 0; ni < 4; ++ni) {
                    const op16x8 wf = *reinterpret_cast<const op16x8*>(w1s + (UPD(UPD_NO_W1_READ) ? kswz(fi, fg) : kswz(pos * 64 + ni * 16 + fi, ks * 4 + fg)));
#pragma unroll
                    for (int sI = 0; sI < UP_NS; ++sI) {
                        if (!UPD(UPD_NO_MFMA_A)) acc[sI][ni] = MFMA_16x16x32(wf, xf[sI][ks], acc[sI][ni], 0, 0, 0);
                        else acc[sI][ni][0] += __builtin_bit_cast(f32x4, wf)[0] + __builtin_bit_cast(f32x4, xf[sI][ks])[0];
                    }
                }
            }
            UP_STAMP(0);
            // the operand registers are free again: the next prompts' rows load while both epilogues and phase B run
            if (!UPD(UPD_NO_XLOAD)) {
#pragma unroll
                for (int sI = 0; sI < UP_NS; ++sI) if (nxt[sI] >= 0 && nxt[sI] < p_end) xload(nxt[sI], xf[sI]);
            }
            // epilogue A: + (bias + feat_s1), LayerNorm over the 64 channels of (tok, pos), GELU, pack as phase-B operand
            op16x8 uf[UP_NS][2];
#pragma unroll
            for (int sI = 0; sI < UP_NS; ++sI) {
                float v[4][4], sum = 0.f;
#pragma unroll
                for (int ni = 0; ni < 4; ++ni) {
                    v[ni][0] = acc[sI][ni][0]; v[ni][1] = acc[sI][ni][1]; v[ni][2] = acc[sI][ni][2]; v[ni][3] = acc[sI][ni][3];
                    sum += (v[ni][0] + v[ni][1]) + (v[ni][2] + v[ni][3]);
                }
                if (!UPD(UPD_NO_LN)) sum = xor32_sum(xor16_sum(sum));
                const float mean = sum * (1.0f / 64.0f);
                float var = 0.f;
#pragma unroll
                for (int ni = 0; ni < 4; ++ni)
#pragma unroll
                    for (int r = 0; r < 4; ++r) { const float d = v[ni][r] - mean; var += d * d; }
                if (!UPD(UPD_NO_LN)) var = xor32_sum(xor16_sum(var));
                const float rstd = __builtin_amdgcn_rsqf(var * (1.0f / 64.0f) + 1e-6f);
                const f32x2 mean2 = {mean, mean}, rstd2 = {rstd, rstd};
#pragma unroll
                for (int ni = 0; ni < 4; ++ni) {
                    const float4 g4 = *reinterpret_cast<const float4*>(lnp + ni * 16 + 4 * fg), b4 = *reinterpret_cast<const float4*>(lnp + 64 + ni * 16 + 4 * fg);
                    f32x2 a = ((f32x2){v[ni][0], v[ni][1]} - mean2) * rstd2 * (f32x2){g4.x, g4.y} + (f32x2){b4.x, b4.y};
                    f32x2 b = ((f32x2){v[ni][2], v[ni][3]} - mean2) * rstd2 * (f32x2){g4.z, g4.w} + (f32x2){b4.z, b4.w};
                    if (!UPD(UPD_NO_GELU)) { a = gelu_erf2(a); b = gelu_erf2(b); }
                    v[ni][0] = a.x; v[ni][1] = a.y; v[ni][2] = b.x; v[ni][3] = b.y;
                }
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
                    uf[sI][ks] = pack8_d(v[2 * ks][0], v[2 * ks][1], v[2 * ks][2], v[2 * ks][3], v[2 * ks + 1][0], v[2 * ks + 1][1], v[2 * ks + 1][2],
                                         v[2 * ks + 1][3]);
            }
            UP_STAMP(1);
            // ---------------- phase B (two halves of the 128 outputs: pos2 in {0,1} then {2,3})
#pragma unroll
            for (int hb = 0; hb < 2; ++hb) {
                f32x4 c2[UP_NS][4];
#pragma unroll
                for (int sI = 0; sI < UP_NS; ++sI)
#pragma unroll
                    for (int nl = 0; nl < 4; ++nl) c2[sI][nl] = f0[2 * hb + (nl >> 1)][nl & 1];                       // bias + feat_s0 enter as the C operand
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                    for (int nl = 0; nl < 4; ++nl) {
                        const op16x8 w2f = *reinterpret_cast<const op16x8*>(w2s + (UPD(UPD_NO_W2_READ) ? swz128(fi, fg) : swz128((hb * 4 + nl) * 16 + fi, ks * 4 + fg)));
#pragma unroll
                        for (int sI = 0; sI < UP_NS; ++sI) {
                            if (!UPD(UPD_NO_MFMA_B)) c2[sI][nl] = MFMA_16x16x32(w2f, uf[sI][ks], c2[sI][nl], 0, 0, 0);
                            else c2[sI][nl][0] += __builtin_bit_cast(f32x4, w2f)[0] + __builtin_bit_cast(f32x4, uf[sI][ks])[0];
                        }
                    }
                UP_STAMP(2);
#pragma unroll
                for (int sI = 0; sI < UP_NS; ++sI) {
                    float2 px2;                                // the two pixels (dx2 = 0, 1) of output row dy2 = hb
#pragma unroll
                    for (int pp = 0; pp < 2; ++pp) {           // pos2 = 2*hb + pp
#if UP_MFMA_HYPER
                        f32x2 ua0 = (f32x2){c2[sI][2 * pp][0], c2[sI][2 * pp][1]}, ub0 = (f32x2){c2[sI][2 * pp][2], c2[sI][2 * pp][3]};
                        f32x2 ua1 = (f32x2){c2[sI][2 * pp + 1][0], c2[sI][2 * pp + 1][1]}, ub1 = (f32x2){c2[sI][2 * pp + 1][2], c2[sI][2 * pp + 1][3]};
                        if (!UPD(UPD_NO_GELU)) { ua0 = gelu_erf2(ua0); ub0 = gelu_erf2(ub0); ua1 = gelu_erf2(ua1); ub1 = gelu_erf2(ub1); }
                        float mine;
                        if (UPD(UPD_NO_HYPER)) mine = ua0.x + ub0.y + ua1.x + ub1.y;
                        else {
                            const op16x8 uop = pack8_d(ua0.x, ua0.y, ub0.x, ub0.y, ua1.x, ua1.y, ub1.x, ub1.y);
                            f32x4 dm = MFMA_16x16x32(hy_hi[sI], uop, (f32x4){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                            dm = MFMA_16x16x32(hy_lo[sI], uop, dm, 0, 0, 0);
                            mine = dm[0];
                        }
                        if (pp == 0) px2.x = mine; else px2.y = mine;
#else
                        float part[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                        for (int hh = 0; hh < 2; ++hh) {
                            const int nl = 2 * pp + hh;
                            f32x2 ua = (f32x2){c2[sI][nl][0], c2[sI][nl][1]}, ub = (f32x2){c2[sI][nl][2], c2[sI][nl][3]};
                            if (!UPD(UPD_NO_GELU)) { ua = gelu_erf2(ua); ub = gelu_erf2(ub); }
                            const float u0 = ua.x, u1 = ua.y, u2 = ub.x, u3 = ub.y;
                            if (UPD(UPD_NO_HYPER)) { part[hh] += u0 + u1 + u2 + u3; continue; }
#pragma unroll
                            for (int k = 0; k < 4; ++k)
                                if ((need[sI] >> k) & 1) part[k] = fmaf(u0, hy[hh][k].x, fmaf(u1, hy[hh][k].y, fmaf(u2, hy[hh][k].z, fmaf(u3, hy[hh][k].w, part[k]))));
                        }
                        // transpose-reduce over the four fg lanes of a token: lane fg ends with the complete sum of mask k = fg
                        // (3 shuffles and selects instead of 8 shuffles and a 4-way branch)
                        float k0 = fb0 ? part[1] : part[0], k1 = fb0 ? part[3] : part[2];
                        k0 += shfl_xor16(fb0 ? part[0] : part[1], (lane >> 4) & 1);
                        k1 += shfl_xor16(fb0 ? part[2] : part[3], (lane >> 4) & 1);
                        float mine = fb1 ? k1 : k0;
                        mine += shfl_xor32(fb1 ? k0 : k1, lane >= 32);
                        if (pp == 0) px2.x = mine; else px2.y = mine;
#endif
                    }
                    if (((need[sI] >> fg) & 1) && !UPD(UPD_NO_STORE)) {
                        *reinterpret_cast<float2*>(orow[sI] + hb * 256) = px2;
                        chk = fmaf(px2.x, 0.f, fmaf(px2.y, 0.f, chk));
                    }
                    if (UPD(UPD_NO_STORE) && px2.x + px2.y == 1.2345e30f) *reinterpret_cast<float2*>(orow[sI] + hb * 256) = px2;     // (keeps the values alive)
                }
                UP_STAMP(3);
            }
            UP_STAMP(4);
#pragma unroll
            for (int sI = 0; sI < UP_NS; ++sI) cur[sI] = nxt[sI];
            q_next = q_after;
        }
    }
    (void)fb0; (void)fb1;
    if (sentinel) {
        const unsigned long long bad = __ballot(chk != chk);
        if (bad && lane == __ffsll((long long)bad) - 1) atomicAdd(sentinel, (unsigned int)__popcll(bad));
    }
    if (UP_DEV && stamps && lane == 0)
        for (int k = 0; k < 5; ++k) stamps[((int64_t)blockIdx.x * (4 * UP_NTG) + wave) * 5 + k] = ts[k];
#undef UP_STAMP
}

const char* launch_dec_upscale(const bf16_t* X, const bf16_t* W1, const float* b1, const float* ln_g, const float* ln_b, const bf16_t* W2p,
                               const float* b2, const float* fs1, const float* fs0, XMap sm, const float* hyper, float* masks4, int P,
                               hipStream_t s, const uint8_t* live, const float* iou4, int multimask, unsigned int* sentinel) {
    if (P <= 0) return nullptr;
    if (sm.div <= 0) return "dec_upscale: XMap.div must be positive";
    // one resident workgroup per CU; each takes an equal contiguous share of the UP_TILES x P (tile, prompt) units
    const int n_cu = saber_cu_count();
    const long long units = (long long)UP_TILES * P;
    const int grid = (int)(units < n_cu ? units : n_cu);
    hipLaunchKernelGGL(dec_upscale_kernel, dim3(grid), dim3(UP_THREADS), UP_LDS, s, X, W1, b1, ln_g, ln_b, W2p, b2, fs1, fs0, sm.div, sm.off, hyper, masks4, P, live, iou4, multimask,
                       UP_DEV ? (g_saber_debug_flags >> 8) & 0x1ff : 0, UP_DEV ? g_saber_stamp_buf : nullptr, sentinel);
    return nullptr;
}

const char* decoder_fused_init_device() {
    hipError_t st = hipFuncSetAttribute(reinterpret_cast<const void*>(dec_t2i_kernel<4, false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, T2ICfg<4>::LDS);
    if (st == hipSuccess) st = hipFuncSetAttribute(reinterpret_cast<const void*>(dec_t2i_kernel<8, false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, T2ICfg<8>::LDS);
    if (st == hipSuccess) st = hipFuncSetAttribute(reinterpret_cast<const void*>(dec_t2i_kernel<8, true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, T2ICfg<8>::LDS);
    if (st == hipSuccess) st = hipFuncSetAttribute(reinterpret_cast<const void*>(dec_t2i_kernel<8, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, T2ICfg<8>::LDS);
    if (st == hipSuccess) st = hipFuncSetAttribute(reinterpret_cast<const void*>(dec_i2t_kernel<1, false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, I2TCfg<1>::LDS);
    if (st == hipSuccess) st = hipFuncSetAttribute(reinterpret_cast<const void*>(dec_i2t_kernel<1, true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, I2TCfg<1>::LDS);
    if (st == hipSuccess) st = hipFuncSetAttribute(reinterpret_cast<const void*>(dec_i2t_kernel<1, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, I2TCfg<1>::LDS + I2TCfg<1>::W3_B);
    if (st == hipSuccess) st = hipFuncSetAttribute(reinterpret_cast<const void*>(dec_i2t_kernel<2, false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, I2TCfg<2>::LDS);
    if (st == hipSuccess) st = hipFuncSetAttribute(reinterpret_cast<const void*>(dec_upscale_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, UP_LDS);
    if (st == hipSuccess) st = hipFuncSetAttribute(reinterpret_cast<const void*>(dec_i2t_w1_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, I4_LDS);
    if (st == hipSuccess) st = hipFuncSetAttribute(reinterpret_cast<const void*>(dec_i2t_w1_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, I4_LDS);
    if (st == hipSuccess) st = hipFuncSetAttribute(reinterpret_cast<const void*>(dec_t2i_w1_kernel<false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, T4_LDS);
    if (st == hipSuccess) st = hipFuncSetAttribute(reinterpret_cast<const void*>(dec_t2i_w1_kernel<false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, T4_LDS);
    if (st == hipSuccess) st = hipFuncSetAttribute(reinterpret_cast<const void*>(dec_t2i_w1_kernel<true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, T4_LDS);
    if (st == hipSuccess) st = hipFuncSetAttribute(reinterpret_cast<const void*>(dec_t2i_w1_kernel<true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, T4_LDS);
    if (st == hipSuccess) st = hipFuncSetAttribute(reinterpret_cast<const void*>(dec_i2t_t2i_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, I2TCfg<1>::LDS + FZ_XBUF + 2 * FZ_PEK_STAGE);
    if (st == hipSuccess) st = hipFuncSetAttribute(reinterpret_cast<const void*>(dec_i2t_t2i_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, I2TCfg<1>::LDS + FZ_XBUF + 2 * FZ_PEK_STAGE);
    return st == hipSuccess ? nullptr : hipGetErrorString(st);
}
