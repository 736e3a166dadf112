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

typedef s16x4 __attribute__((address_space(3))) * lds_s16x4_ptr;
__device__ __forceinline__ bf16x4 tr_read_d(const char* p) {
    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(p));
    return __builtin_bit_cast(bf16x4, v);
}
__device__ __forceinline__ bf16x8 cat4_d(bf16x4 a, bf16x4 b) { return __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7); }
__device__ __forceinline__ bf16x8 pack8_d(float a0, float a1, float a2, float a3, float b0, float b1, float b2, float b3) {
    uint4 u = make_uint4(pack_bf16(a0, a1), pack_bf16(a2, a3), pack_bf16(b0, b1), pack_bf16(b2, b3));
    return __builtin_bit_cast(bf16x8, u);
}
// 8 packed bf16 + 8 packed bf16 -> 8 packed bf16 (fp32 add, RNE)
__device__ __forceinline__ u32x4 add_bf16x8(u32x4 a, u32x4 b) {
    u32x4 r;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float lo = __uint_as_float(a[i] << 16) + __uint_as_float(b[i] << 16);
        const float hi = __uint_as_float(a[i] & 0xffff0000u) + __uint_as_float(b[i] & 0xffff0000u);
        r[i] = pack_bf16(lo, hi);
    }
    return r;
}

#define DC 256           // channels of the decoder
#define ROW_B 512        // bytes per 256-channel bf16 row
#define VT_STRIDE 544    // padded LDS row stride for tr reads: 8 rows x 32 B hit disjoint banks
__device__ __forceinline__ int kswz(int row, int chunk) { return row * ROW_B + ((chunk ^ (row & 15)) << 4); }

// ------------------------------------------------------------------------------------------------ fold
// out[p][8h+t][d] = scale * sum_j a[p][t][16h+j] * W(16h+j, d);  W row-major [128][256] (mode 0: k_proj / q_proj
// weight) or [256][128] indexed W[d][16h+j] (mode 1: out_proj weight).  cb[p][8h+t] = scale * sum_j a . bias[16h+j].
__global__ __launch_bounds__(256) void dec_fold_kernel(const float* __restrict__ a, const bf16_t* __restrict__ W, const float* __restrict__ bias,
                                                       int mode, float scale, bf16_t* __restrict__ out, float* __restrict__ cb) {
    __shared__ float as[8 * 128];
    const int p = blockIdx.x, d = threadIdx.x;
    for (int i = d; i < 8 * 128; i += 256) as[i] = a[(int64_t)p * 1024 + i];
    __syncthreads();
    for (int h = 0; h < 8; ++h) {
        float w[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) w[j] = mode == 0 ? bf2f(W[(16 * h + j) * 256 + d]) : bf2f(W[d * 128 + 16 * h + j]);
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            float acc = 0.f;
#pragma unroll
            for (int j = 0; j < 16; ++j) acc += as[t * 128 + 16 * h + j] * w[j];
            out[((int64_t)p * 64 + 8 * h + t) * 256 + d] = f2bf(acc * scale);
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
//  * score = Qt.(x + pe) is evaluated as [Qt | Qt].[x ; pe] (contraction 512): the positional encoding enters through the
//    MFMA instead of a VALU add, so BOTH operand tiles are plain copies and go global -> LDS directly
//    (global_load_lds_dwordx4, 2-stage ring, no staging registers, no ds_write).
//  * one LDS image of the X tile (XOR-swizzled 512-B rows, source-side swizzle) serves the K operand (ds_read_b128 rows)
//    and the V operand (ds_read_b64_tr_b16 with the same XOR).
//  * 8 waves: wave = (key half kh) * 4 + (q tile); each wave keeps its own online-softmax partial over its 32 keys of
//    every block; the two halves are merged through LDS at the end.  Two waves per SIMD hide each other's LDS latency.
// Writes un-normalised partial O [P][split][64][256] and (m, l) [P][split][64][2] (log2 domain).
#define T2I_KB 64
#define T2I_STAGE (2 * T2I_KB * ROW_B)        // X tile + PE tile
#define T2I_LDS (2 * T2I_STAGE)
typedef __attribute__((address_space(1))) const void* gptr_d;
typedef __attribute__((address_space(3))) void* lptr_d;

__global__ __launch_bounds__(512) void dec_t2i_kernel(const bf16_t* __restrict__ X, int64_t x_bs, const bf16_t* __restrict__ pe,
                                                      const bf16_t* __restrict__ Qt, float* __restrict__ Opart, float* __restrict__ ML,
                                                      int split) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int qt = wave & 3, kh = wave >> 2;
    const int fi = lane & 15, fg = lane >> 4;
    const int p = blockIdx.x / split, sp = blockIdx.x - p * split;
    const int nkeys = 4096 / split, key0 = sp * nkeys, nkb = nkeys / T2I_KB;
    const bf16_t* Xp = X + (int64_t)p * x_bs + (int64_t)key0 * DC;
    const bf16_t* Pp = pe + (int64_t)key0 * DC;

    bf16x8 qf[8];
    {
        const bf16_t* qrow = Qt + ((int64_t)p * 64 + qt * 16 + fi) * DC;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) qf[ks] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(qrow + 32 * ks + 8 * fg));
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
    auto issue = [&](int kb, int stage) {
        char* sx = smem + stage * T2I_STAGE;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int64_t off = (int64_t)(kb * T2I_KB + srow[i]) * DC + schunk[i] * 8;
            __builtin_amdgcn_global_load_lds((gptr_d)(Xp + off), (lptr_d)(sx + (wave * 4 + i) * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((gptr_d)(Pp + off), (lptr_d)(sx + T2I_KB * ROW_B + (wave * 4 + i) * 1024), 16, 0, 0);
        }
    };
    // per-lane LDS offsets that do not depend on the block
    int koff[8];                                        // K-operand fragment of key row (16kt + fi), k-step ks
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) koff[ks] = fi * ROW_B + (((4 * ks + fg) ^ fi) << 4);
    const int vrow = 4 * fg + (fi >> 2);                // tr-read row within the 32-key half (+16 for the second read)
    const int vsel = (fi & 3) >> 1, vlow = (fi & 1) * 8;

    issue(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    for (int kb = 0; kb < nkb; ++kb) {
        const int stage = kb & 1;
        if (kb + 1 < nkb) issue(kb + 1, stage ^ 1);
        const char* xs = smem + stage * T2I_STAGE + kh * 32 * ROW_B;     // this wave's 32 keys
        const char* ps = xs + T2I_KB * ROW_B;
        f32x4 s[2];
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
            s[kt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                const bf16x8 kx = *reinterpret_cast<const bf16x8*>(xs + kt * 16 * ROW_B + koff[ks]);
                const bf16x8 kp = *reinterpret_cast<const bf16x8*>(ps + kt * 16 * ROW_B + koff[ks]);
                s[kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kx, qf[ks], s[kt], 0, 0, 0);
                s[kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kp, qf[ks], s[kt], 0, 0, 0);
            }
        }
        float mx = fmaxf(fmaxf(fmaxf(s[0][0], s[0][1]), fmaxf(s[0][2], s[0][3])), fmaxf(fmaxf(s[1][0], s[1][1]), fmaxf(s[1][2], s[1][3])));
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        if (__any(mx > m)) {        // rescale only when some row's running maximum grows (wave-uniform branch)
            const float mn = fmaxf(m, mx);
            const float alpha = exp2f(m - mn);
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
                const float e = exp2f(s[kt][r] - m);
                s[kt][r] = e;
                sum += e;
            }
        sum += __shfl_xor(sum, 16, 64);
        sum += __shfl_xor(sum, 32, 64);
        l += sum;
        const bf16x8 pf = pack8_d(s[0][0], s[0][1], s[0][2], s[0][3], s[1][0], s[1][1], s[1][2], s[1][3]);
#pragma unroll
        for (int dt = 0; dt < 16; ++dt) {
            const int sw = ((2 * dt + vsel) ^ (vrow & 15)) << 4;       // rows vrow and vrow+16 share (row & 15)
            const char* base = xs + vrow * ROW_B + sw + vlow;
            const bf16x8 vf = cat4_d(tr_read_d(base), tr_read_d(base + 16 * ROW_B));
            o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf, o[dt], 0, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // next block landed (this wave's part)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
    // merge the two key halves of each q tile: waves 4..7 park (m, l, O) in LDS, waves 0..3 combine and store
    float* mo = reinterpret_cast<float*>(smem) + (size_t)qt * 16 * 260;      // [16 q][256 + 4] floats per q tile
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
        const int q = qt * 16 + fi;
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
        for (int d = 0; d < DC; ++d) acc += bf2f(w[d]) * z[t][d];
        out[(int64_t)p * 1024 + t * 128 + oo] = f2bf(acc);
    }
}

const char* launch_dec_t2i(const bf16_t* X, int64_t x_bs, const bf16_t* pe, const bf16_t* Qt, float* Opart, float* ML, int P, int split,
                           const bf16_t* Wv, const float* bv, bf16_t* out, hipStream_t s) {
    if (P <= 0) return nullptr;
    if (split != 1 && split != 2 && split != 4 && split != 8) return "dec_t2i: split must be 1, 2, 4 or 8";
    hipLaunchKernelGGL(dec_t2i_kernel, dim3(P * split), dim3(512), T2I_LDS, s, X, x_bs, pe, Qt, Opart, ML, split);
    hipLaunchKernelGGL(dec_t2i_finish_kernel, dim3(P * 8), dim3(256), 0, s, (const float*)Opart, (const float*)ML, split, Wv, bv, out);
    return nullptr;
}

// ------------------------------------------------------------------------------------------------ image -> tokens
#define I2T_ROWS 64
#define I2T_TILES 8   // row tiles per block (512 rows), folded operands stay in LDS across them
#define I2T_LDS (64 * ROW_B + 64 * VT_STRIDE + I2T_ROWS * ROW_B + I2T_ROWS * VT_STRIDE)

// grid = P * (4096 / (I2T_ROWS * I2T_TILES)).  X_out[p][n] = LN(x_n + softmax_heads((x_n + pe_n).Kt + cb).Vt + b_o)
__global__ __launch_bounds__(256) void dec_i2t_kernel(const bf16_t* __restrict__ X, int64_t x_bs, const bf16_t* __restrict__ pe,
                                                      const bf16_t* __restrict__ Kt, const float* __restrict__ cb,
                                                      const bf16_t* __restrict__ Vt, const float* __restrict__ bo,
                                                      const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                                      bf16_t* __restrict__ Xout) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* kt_s = smem;                                  // [64 c][256] swizzled   (A operand of GEMM1)
    char* vt_s = kt_s + 64 * ROW_B;                     // [64 c][256] stride 544 (tr-read source of GEMM2)
    char* xp_s = vt_s + 64 * VT_STRIDE;                 // [64 rows][256] swizzled: bf16(x + pe)  (B operand of GEMM1)
    char* x_s = xp_s + I2T_ROWS * ROW_B;                // [64 rows][256] x (residual), padded rows: the 16 rows a wave reads hit different banks
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fi = lane & 15, fg = lane >> 4;
    const int blocks_per_p = 4096 / (I2T_ROWS * I2T_TILES);
    const int p = blockIdx.x / blocks_per_p, seg = blockIdx.x - p * blocks_per_p;
    const bf16_t* Xp = X + (int64_t)p * x_bs;
    bf16_t* Xo = Xout + (int64_t)p * 4096 * DC;

    // folded operands of this prompt
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int idx = tid + 256 * j, row = idx >> 5, ch = idx & 31;
        *reinterpret_cast<uint4*>(kt_s + kswz(row, ch)) = *reinterpret_cast<const uint4*>(Kt + ((int64_t)p * 64 + row) * DC + ch * 8);
        *reinterpret_cast<uint4*>(vt_s + row * VT_STRIDE + ch * 16) = *reinterpret_cast<const uint4*>(Vt + ((int64_t)p * 64 + row) * DC + ch * 8);
    }
    // per-lane constants: cb for c = 16ct + 4g + r ; bo/gamma/beta for d = 16dt + 4g + r are re-read per tile from L1
    float cbv[4][4];
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) {
        const float4 v = *reinterpret_cast<const float4*>(cb + (int64_t)p * 64 + 16 * ct + 4 * fg);
        cbv[ct][0] = v.x; cbv[ct][1] = v.y; cbv[ct][2] = v.z; cbv[ct][3] = v.w;
    }

    u32x4 rx[8];
    auto gload = [&](int t) {
        const int row0 = (seg * I2T_TILES + t) * I2T_ROWS;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int idx = tid + 256 * j, row = idx >> 5, ch = idx & 31;
            rx[j] = *reinterpret_cast<const u32x4*>(Xp + (int64_t)(row0 + row) * DC + ch * 8);
        }
    };
    gload(0);
#pragma unroll 1
    for (int t = 0; t < I2T_TILES; ++t) {
        u32x4 rp[8];   // positional encoding of this tile: shared by every prompt, L2-resident
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int idx = tid + 256 * j, row = idx >> 5, ch = idx & 31;
            rp[j] = *reinterpret_cast<const u32x4*>(pe + (int64_t)((seg * I2T_TILES + t) * I2T_ROWS + row) * DC + ch * 8);
        }
        __syncthreads();  // previous tile consumed (first time: folded operands visible after the 2nd barrier)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int idx = tid + 256 * j, row = idx >> 5, ch = idx & 31;
            *reinterpret_cast<u32x4*>(xp_s + kswz(row, ch)) = add_bf16x8(rx[j], rp[j]);
            *reinterpret_cast<u32x4*>(x_s + row * VT_STRIDE + ch * 16) = rx[j];
        }
        __syncthreads();
        if (t + 1 < I2T_TILES) gload(t + 1);
        const int mrow = wave * 16 + fi;  // row of this lane within the tile

        // GEMM1 (swapped): S^T[c][m] = Kt[c] . xpos[m]
        f32x4 s[4];
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) s[ct] = (f32x4){cbv[ct][0], cbv[ct][1], cbv[ct][2], cbv[ct][3]};
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            const bf16x8 xf = *reinterpret_cast<const bf16x8*>(xp_s + kswz(mrow, 4 * ks + fg));
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) {
                const bf16x8 kf = *reinterpret_cast<const bf16x8*>(kt_s + kswz(16 * ct + fi, 4 * ks + fg));
                s[ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, xf, s[ct], 0, 0, 0);
            }
        }
        // softmax over the 8 tokens of a head: c = 16ct + 4g + r -> head = 2ct + (g >> 1); members: r = 0..3 and lane ^ 16
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) {
            float mx = fmaxf(fmaxf(s[ct][0], s[ct][1]), fmaxf(s[ct][2], s[ct][3]));
            mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
            float sum = 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) { s[ct][r] = exp2f(s[ct][r] - mx); sum += s[ct][r]; }
            sum += __shfl_xor(sum, 16, 64);
            const float inv = __builtin_amdgcn_rcpf(sum);
#pragma unroll
            for (int r = 0; r < 4; ++r) s[ct][r] *= inv;
        }
        bf16x8 pf[2];
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2)
            pf[k2] = pack8_d(s[2 * k2][0], s[2 * k2][1], s[2 * k2][2], s[2 * k2][3], s[2 * k2 + 1][0], s[2 * k2 + 1][1], s[2 * k2 + 1][2],
                             s[2 * k2 + 1][3]);
        // GEMM2: Y^T[d][m] = Vt^T[d][c] . P^T[c][m]
        f32x4 y[16];
#pragma unroll
        for (int dt = 0; dt < 16; ++dt) {
            y[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int k2 = 0; k2 < 2; ++k2) {
                const char* base = vt_s + (32 * k2 + 4 * fg + (fi >> 2)) * VT_STRIDE + (16 * dt + 4 * (fi & 3)) * 2;
                const bf16x8 vf = cat4_d(tr_read_d(base), tr_read_d(base + 16 * VT_STRIDE));
                y[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf[k2], y[dt], 0, 0, 0);
            }
        }
        // residual + LayerNorm over the 256 channels of row m (lane holds d = 16dt + 4g + r; partners: lane ^ 16, ^ 32)
        float sum = 0.f;
#pragma unroll
        for (int dt = 0; dt < 16; ++dt) {
            const uint2 xr = *reinterpret_cast<const uint2*>(x_s + mrow * VT_STRIDE + (16 * dt + 4 * fg) * 2);
            const float4 b4 = *reinterpret_cast<const float4*>(bo + 16 * dt + 4 * fg);
            y[dt][0] += __uint_as_float(xr.x << 16) + b4.x;
            y[dt][1] += __uint_as_float(xr.x & 0xffff0000u) + b4.y;
            y[dt][2] += __uint_as_float(xr.y << 16) + b4.z;
            y[dt][3] += __uint_as_float(xr.y & 0xffff0000u) + b4.w;
            sum += (y[dt][0] + y[dt][1]) + (y[dt][2] + y[dt][3]);
        }
        sum += __shfl_xor(sum, 16, 64);
        sum += __shfl_xor(sum, 32, 64);
        const float mean = sum * (1.0f / DC);
        float var = 0.f;
#pragma unroll
        for (int dt = 0; dt < 16; ++dt)
#pragma unroll
            for (int r = 0; r < 4; ++r) { const float dlt = y[dt][r] - mean; var += dlt * dlt; }
        var += __shfl_xor(var, 16, 64);
        var += __shfl_xor(var, 32, 64);
        const float rstd = __builtin_amdgcn_rsqf(var * (1.0f / DC) + eps);
        bf16_t* orow = Xo + (int64_t)((seg * I2T_TILES + t) * I2T_ROWS + mrow) * DC;
#pragma unroll
        for (int dt = 0; dt < 16; ++dt) {
            const float4 g4 = *reinterpret_cast<const float4*>(gamma + 16 * dt + 4 * fg);
            const float4 e4 = *reinterpret_cast<const float4*>(beta + 16 * dt + 4 * fg);
            const float v0 = (y[dt][0] - mean) * rstd * g4.x + e4.x, v1 = (y[dt][1] - mean) * rstd * g4.y + e4.y;
            const float v2 = (y[dt][2] - mean) * rstd * g4.z + e4.z, v3 = (y[dt][3] - mean) * rstd * g4.w + e4.w;
            *reinterpret_cast<uint2*>(orow + 16 * dt + 4 * fg) = make_uint2(pack_bf16(v0, v1), pack_bf16(v2, v3));
        }
    }
}

const char* launch_dec_i2t(const bf16_t* X, int64_t x_bs, const bf16_t* pe, const bf16_t* Kt, const float* cb, const bf16_t* Vt,
                           const float* bo, const float* gamma, const float* beta, float eps, bf16_t* Xout, int P, hipStream_t s) {
    if (P <= 0) return nullptr;
    hipLaunchKernelGGL(dec_i2t_kernel, dim3(P * (4096 / (I2T_ROWS * I2T_TILES))), dim3(256), I2T_LDS, s, X, x_bs, pe, Kt, cb, Vt, bo, gamma,
                       beta, eps, Xout);
    return nullptr;
}


// ------------------------------------------------------------------------------------------------ upscaling head
// Fused output_upscaling + hypernetwork product:
//   u1 = GELU(LN64(ConvT1(x) + feat_s1));  u2 = GELU(ConvT2(u1) + feat_s0);  masks[k] = hyper[k] . u2
// Persistent blocks: one block owns a tile of 32 consecutive tokens (2x16 tokens = 8x64 output pixels) and walks over
// the prompts.  Both ConvTranspose weights stay resident in LDS (128 KB + 16 KB), the tile's feat_s1 / feat_s0 rows and
// the LayerNorm parameters stay in registers, and the next prompt's 16-KB X tile is prefetched while the current one is
// computed - per prompt the block touches HBM only for X (read) and the 8-KB logit tile (write).
// Phase A: [32 tok] x [256 = pos*64+ch] over K=256; wave w owns pos = w, so its LayerNorm groups are wave-local and its
// GELU outputs feed phase B straight from registers (k-slot permutation, W2 pre-permuted on the host).
// Phase B: [(pos w, 32 tok)] x [128 = pos2*32+ch2] over K=64, epilogue = +feat_s0, GELU, 4 dot products with hyper.
// The 32-channel 256x256 upscaled embedding (8 MB fp32 per prompt) never exists in memory.
#define UP_TOK 32
#define UP_W1S (256 * ROW_B)          // W1 [256 n][256 k] bf16, kswz
#define UP_W2S (128 * 128)            // W2p [128 n2][64] bf16 (k-slots pre-permuted), swz128
#define UP_XS (UP_TOK * ROW_B)        // X tile [32][256] bf16, kswz; reused as the output tile [4][8][64] fp32
#define UP_LDS (UP_W1S + UP_W2S + UP_XS)
__device__ __forceinline__ int swz128(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }

__global__ __launch_bounds__(512) void dec_upscale_kernel(const bf16_t* __restrict__ X, const bf16_t* __restrict__ W1, const float* __restrict__ b1,
                                                          const float* __restrict__ ln_g, const float* __restrict__ ln_b,
                                                          const bf16_t* __restrict__ W2p, const float* __restrict__ b2,
                                                          const float* __restrict__ fs1, const float* __restrict__ fs0,
                                                          const float* __restrict__ hyper, float* __restrict__ masks4, int P, int groups) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* w1s = smem;
    char* w2s = w1s + UP_W1S;
    char* xs = w2s + UP_W2S;
    float* outs = reinterpret_cast<float*>(xs);
    // 8 waves (two per SIMD: the epilogues are VALU-bound, a lone wave would issue at half rate):
    // wave = (token half mw) * 4 + (pos = ConvT1 output position)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int pos = wave & 3, mw = wave >> 2;
    const int fi = lane & 15, fg = lane >> 4;
    const int tile = blockIdx.x % 128, grp = blockIdx.x / 128;

    // resident weights
    for (int idx = tid; idx < 256 * 32; idx += 512) {
        const int row = idx >> 5, ch = idx & 31;
        *reinterpret_cast<u32x4*>(w1s + kswz(row, ch)) = *reinterpret_cast<const u32x4*>(W1 + row * 256 + ch * 8);
    }
    for (int idx = tid; idx < 128 * 8; idx += 512) {
        const int row = idx >> 3, ch = idx & 7;
        *reinterpret_cast<u32x4*>(w2s + swz128(row, ch)) = *reinterpret_cast<const u32x4*>(W2p + row * 64 + ch * 8);
    }
    // per-lane constants of this tile: feat_s1 + b1 (fp32), feat_s0 + b2 (packed bf16), LN parameters
    const int tl = mw * 16 + fi;                        // token within the tile: bits [X3 X2 X1][Y0][X0]
    const int tok = tile * UP_TOK + tl;
    float4 f1[4], gg[4], be[4];
    uint2 f0[4][2];
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) {
        gg[ni] = *reinterpret_cast<const float4*>(ln_g + ni * 16 + 4 * fg);
        be[ni] = *reinterpret_cast<const float4*>(ln_b + ni * 16 + 4 * fg);
        const float4 r = *reinterpret_cast<const float4*>(fs1 + ((int64_t)tok * 4 + pos) * 64 + ni * 16 + 4 * fg);
        const float4 b = *reinterpret_cast<const float4*>(b1 + pos * 64 + ni * 16 + 4 * fg);
        f1[ni] = make_float4(r.x + b.x, r.y + b.y, r.z + b.z, r.w + b.w);
    }
#pragma unroll
    for (int pos2 = 0; pos2 < 4; ++pos2)
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
            const float4 r = *reinterpret_cast<const float4*>(fs0 + (((int64_t)tok * 4 + pos) * 4 + pos2) * 32 + hh * 16 + 4 * fg);
            const float4 b = *reinterpret_cast<const float4*>(b2 + pos2 * 32 + hh * 16 + 4 * fg);
            f0[pos2][hh] = make_uint2(pack_bf16(r.x + b.x, r.y + b.y), pack_bf16(r.z + b.z, r.w + b.w));
        }
    const int dy1 = pos >> 1, dx1 = pos & 1;
    const int ty = (tl >> 1) & 1, tx = ((tl >> 2) & 7) * 2 + (tl & 1);
    int ty0, tx0;
    perm_coords(tile * UP_TOK, 2, &ty0, &tx0);

    u32x4 rx[2];
    auto xload = [&](int p) {
        const bf16_t* Xt = X + ((int64_t)p * 4096 + tile * UP_TOK) * DC;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int idx = tid + 512 * j;
            rx[j] = *reinterpret_cast<const u32x4*>(Xt + (int64_t)(idx >> 5) * DC + (idx & 31) * 8);
        }
    };
    if (grp < P) xload(grp);
    for (int p = grp; p < P; p += groups) {
        __syncthreads();  // weights visible (first pass) / previous output tile fully stored
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int idx = tid + 512 * j;
            *reinterpret_cast<u32x4*>(xs + kswz(idx >> 5, idx & 31)) = rx[j];
        }
        __syncthreads();
        if (p + groups < P) xload(p + groups);
        float4 hy[2][4];
#pragma unroll
        for (int hh = 0; hh < 2; ++hh)
#pragma unroll
            for (int k = 0; k < 4; ++k) hy[hh][k] = *reinterpret_cast<const float4*>(hyper + (int64_t)p * 128 + k * 32 + hh * 16 + 4 * fg);
        // ---------------- phase A: [16 tok of this wave] x [64 outputs of pos] over K = 256
        f32x4 acc[4];
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) acc[ni] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            const bf16x8 xf = *reinterpret_cast<const bf16x8*>(xs + kswz(tl, ks * 4 + fg));
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) {
                const bf16x8 wf = *reinterpret_cast<const bf16x8*>(w1s + kswz(pos * 64 + ni * 16 + fi, ks * 4 + fg));
                acc[ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, xf, acc[ni], 0, 0, 0);
            }
        }
        // epilogue A: + (bias + feat_s1), LayerNorm over the 64 channels of (tok, pos), GELU, pack as phase-B operand
        bf16x8 uf[2];
        {
            float v[4][4], sum = 0.f;
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) {
                v[ni][0] = acc[ni][0] + f1[ni].x; v[ni][1] = acc[ni][1] + f1[ni].y;
                v[ni][2] = acc[ni][2] + f1[ni].z; v[ni][3] = acc[ni][3] + f1[ni].w;
                sum += (v[ni][0] + v[ni][1]) + (v[ni][2] + v[ni][3]);
            }
            sum += __shfl_xor(sum, 16, 64);
            sum += __shfl_xor(sum, 32, 64);
            const float mean = sum * (1.0f / 64.0f);
            float var = 0.f;
#pragma unroll
            for (int ni = 0; ni < 4; ++ni)
#pragma unroll
                for (int r = 0; r < 4; ++r) { const float d = v[ni][r] - mean; var += d * d; }
            var += __shfl_xor(var, 16, 64);
            var += __shfl_xor(var, 32, 64);
            const float rstd = __builtin_amdgcn_rsqf(var * (1.0f / 64.0f) + 1e-6f);
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) {
                v[ni][0] = gelu_erf((v[ni][0] - mean) * rstd * gg[ni].x + be[ni].x);
                v[ni][1] = gelu_erf((v[ni][1] - mean) * rstd * gg[ni].y + be[ni].y);
                v[ni][2] = gelu_erf((v[ni][2] - mean) * rstd * gg[ni].z + be[ni].z);
                v[ni][3] = gelu_erf((v[ni][3] - mean) * rstd * gg[ni].w + be[ni].w);
            }
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
                uf[ks] = pack8_d(v[2 * ks][0], v[2 * ks][1], v[2 * ks][2], v[2 * ks][3], v[2 * ks + 1][0], v[2 * ks + 1][1], v[2 * ks + 1][2],
                                 v[2 * ks + 1][3]);
        }
        __syncthreads();  // every wave is done reading the X tile: its LDS now holds the output tile
        // ---------------- phase B (two halves of the 128 outputs: pos2 in {0,1} then {2,3})
#pragma unroll
        for (int hb = 0; hb < 2; ++hb) {
            f32x4 c2[4];
#pragma unroll
            for (int nl = 0; nl < 4; ++nl) c2[nl] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int nl = 0; nl < 4; ++nl) {
                    const bf16x8 w2f = *reinterpret_cast<const bf16x8*>(w2s + swz128((hb * 4 + nl) * 16 + fi, ks * 4 + fg));
                    c2[nl] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w2f, uf[ks], c2[nl], 0, 0, 0);
                }
#pragma unroll
            for (int pp = 0; pp < 2; ++pp) {           // pos2 = 2*hb + pp
                const int pos2 = 2 * hb + pp;
                float part[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) {
                    const int nl = 2 * pp + hh;
                    const uint2 fb = f0[pos2][hh];
                    const float u0 = gelu_erf(c2[nl][0] + __uint_as_float(fb.x << 16)), u1 = gelu_erf(c2[nl][1] + __uint_as_float(fb.x & 0xffff0000u));
                    const float u2 = gelu_erf(c2[nl][2] + __uint_as_float(fb.y << 16)), u3 = gelu_erf(c2[nl][3] + __uint_as_float(fb.y & 0xffff0000u));
#pragma unroll
                    for (int k = 0; k < 4; ++k) part[k] += (u0 * hy[hh][k].x + u1 * hy[hh][k].y) + (u2 * hy[hh][k].z + u3 * hy[hh][k].w);
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    part[k] += __shfl_xor(part[k], 16, 64);
                    part[k] += __shfl_xor(part[k], 32, 64);
                }
                const int py = ty * 4 + dy1 * 2 + (pos2 >> 1), px = tx * 4 + dx1 * 2 + (pos2 & 1);
                const float mine = fg == 0 ? part[0] : fg == 1 ? part[1] : fg == 2 ? part[2] : part[3];
                outs[(fg * 8 + py) * 64 + px] = mine;
            }
        }
        __syncthreads();
        {
            const int k = tid >> 7, row = (tid >> 4) & 7, c4 = tid & 15;   // 512 float4: [k][row 0..7][col/4]
            const float4 v = *reinterpret_cast<const float4*>(outs + (k * 8 + row) * 64 + c4 * 4);
            *reinterpret_cast<float4*>(masks4 + (((int64_t)p * 4 + k) * 256 + (ty0 * 4 + row)) * 256 + tx0 * 4 + c4 * 4) = v;
        }
    }
}

const char* launch_dec_upscale(const bf16_t* X, const bf16_t* W1, const float* b1, const float* ln_g, const float* ln_b, const bf16_t* W2p,
                               const float* b2, const float* fs1, const float* fs0, const float* hyper, float* masks4, int P, hipStream_t s) {
    if (P <= 0) return nullptr;
    const int groups = P >= 2 ? 2 : 1;   // 128 tiles x 2 groups = one resident block per CU
    hipLaunchKernelGGL(dec_upscale_kernel, dim3(128 * groups), dim3(512), UP_LDS, s, X, W1, b1, ln_g, ln_b, W2p, b2, fs1, fs0, hyper, masks4, P, groups);
    return nullptr;
}

const char* decoder_fused_init_device() {
    hipError_t st = hipFuncSetAttribute(reinterpret_cast<const void*>(dec_t2i_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, T2I_LDS);
    if (st == hipSuccess) st = hipFuncSetAttribute(reinterpret_cast<const void*>(dec_i2t_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, I2T_LDS);
    if (st == hipSuccess) st = hipFuncSetAttribute(reinterpret_cast<const void*>(dec_upscale_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, UP_LDS);
    return st == hipSuccess ? nullptr : hipGetErrorString(st);
}
