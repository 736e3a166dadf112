// Token side of the two-way transformer as ONE kernel per segment (VERDICT r02 "token-side layer as one kernel").
//
// Between two image-side kernels (dec_t2i / dec_i2t) the 8 tokens of every prompt go through a chain of small dependent steps: output
// projection + residual + LayerNorm, the 256 -> 2048 -> 256 MLP, the k / v projections and folds that prepare the next image-side kernel,
// the 8 x 8 self attention of the next layer, the q projection and fold of the next tokens -> image attention.  As separate launches that
// was ~58 kernels of 5-25 us per decoder batch (M = 8 192 rows at most: launch-to-drain latency, not work): 7-8 ms per slice.  Here a
// workgroup (8 waves) owns FOUR prompts (32 token rows) and walks the whole chain with the rows resident in LDS; weights stream from L2 straight
// into MFMA operand registers (every workgroup reads every weight once: ~3 MB per workgroup and segment).
//
// Arithmetic as in the separate kernels: bf16 MFMA operands (activations rounded where they were rounded before), fp32 accumulation, fp32
// residual stream / LayerNorm / softmax; the folds multiply the fp32 projections as a bf16 hi + lo pair (two MFMAs), which is closer to
// dec_fold_kernel's fp32 products than a single bf16 rounding would be.
#include "common.h"
#include "kernels.h"

#define TK_G 4                 // prompts per workgroup
#define TK_T 512               // threads per workgroup: 8 waves, two output tiles of 16 columns each per 256-wide projection
#define TK_R (8 * TK_G)        // token rows per workgroup
#define TK_AS 528              // bytes per row of a bf16 operand buffer (256 + 8 elements: 16 rows x 16 B hit disjoint banks)
#define TK_FS 132              // floats per row of an fp32 scratch buffer (128 + 4)

struct TokCtx { int tid, lane, wave, fi, fg; char* Q; char* B0; char* B1; char* F0; char* F1; char* F2; char* H; };

// acc[i][m][r] = sum_k W[n0 + 16 (wave NTW + i) + 4 fg + r][k] * A[row 16 m + fi][k]  (W rows beyond nrows are clamped: their results are not used)
// Wpk (optional) = launch_pack_w_kstep's copy [K / 32][Npk][4 chunks, XOR-permuted by row][8] of the matrix whose row `row_off` is W's row 0: a
// 16 x 32 fragment is then one contiguous KB (eight full lines) instead of sixteen half lines of sixteen rows.
__device__ __forceinline__ int tk_perm(int row) { return (0x78 >> (((row >> 2) & 3) * 2)) & 3; }          // gemm_rowln.hip rl_perm
template <int NTW, int MT>
__device__ __forceinline__ void wgemm(const TokCtx& c, const char* A, int K, const bf16_t* W, int ldw, int n0, int nrows, f32x4 (&acc)[NTW][MT],
                                      const bf16_t* Wpk = nullptr, int Npk = 0, int row_off = 0) {
#pragma unroll
    for (int i = 0; i < NTW; ++i)
#pragma unroll
        for (int m = 0; m < MT; ++m) acc[i][m] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const bf16_t* wr[NTW];
    int64_t kstride = 32;                   // elements from one K-step's fragment to the next
#pragma unroll
    for (int i = 0; i < NTW; ++i) {
        const int n = min(n0 + 16 * (c.wave * NTW + i) + c.fi, nrows - 1);
        if (Wpk) wr[i] = Wpk + (int64_t)(row_off + n) * 32 + ((c.fg ^ tk_perm(row_off + n)) << 3);
        else wr[i] = W + (int64_t)n * ldw + 8 * c.fg;
    }
    if (Wpk) kstride = (int64_t)Npk * 32;
#pragma unroll 8
    for (int ks = 0; ks < K / 32; ++ks) {
        op16x8 b[MT];
#pragma unroll
        for (int m = 0; m < MT; ++m) b[m] = *reinterpret_cast<const op16x8*>(A + (16 * m + c.fi) * TK_AS + (32 * ks + 8 * c.fg) * 2);
#pragma unroll
        for (int i = 0; i < NTW; ++i) {
            const op16x8 a = __builtin_bit_cast(op16x8, *reinterpret_cast<const uint4*>(wr[i] + ks * kstride));
#pragma unroll
            for (int m = 0; m < MT; ++m) acc[i][m] = MFMA_16x16x32(a, b[m], acc[i][m], 0, 0, 0);
        }
    }
}

// The same product with the weight fragments of the whole call (K = 256: 8 k-steps x 2 column tiles = 64 registers) REQUESTED AHEAD: wfrag_load
// is issued one call early (the MLP walks 16 dependent 128-KB weight panels per segment, each behind a workgroup barrier: fetched inside the
// call, every panel paid an L2 round trip with nothing to overlap it - round 5).  Same MFMA order as wgemm: bit-identical results.
__device__ __forceinline__ void wfrag_load(const TokCtx& c, const bf16_t* Wpk, int N, int n0, int ks0, uint4 (&w)[2][8]) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int n = n0 + 16 * (c.wave * 2 + i) + c.fi;
        const bf16_t* wr = Wpk + ((int64_t)ks0 * N + n) * 32 + ((c.fg ^ tk_perm(n)) << 3);
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) w[i][ks] = *reinterpret_cast<const uint4*>(wr + (int64_t)ks * N * 32);
    }
    __builtin_amdgcn_sched_barrier(0);          // the requests stay HERE (ahead of the previous panel's MFMAs), not next to their use
}
__device__ __forceinline__ void wgemm_pre(const TokCtx& c, const char* A, const uint4 (&w)[2][8], f32x4 (&acc)[2][2]) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int m = 0; m < 2; ++m) acc[i][m] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
        op16x8 b[2];
#pragma unroll
        for (int m = 0; m < 2; ++m) b[m] = *reinterpret_cast<const op16x8*>(A + (16 * m + c.fi) * TK_AS + (32 * ks + 8 * c.fg) * 2);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const op16x8 a = __builtin_bit_cast(op16x8, w[i][ks]);
#pragma unroll
            for (int m = 0; m < 2; ++m) acc[i][m] = MFMA_16x16x32(a, b[m], acc[i][m], 0, 0, 0);
        }
    }
}

// B = bf16(Q + (pe ? tok_pe : 0)) for the workgroup's 32 rows
__device__ __forceinline__ void to_operand(const TokCtx& c, char* B, const float* pe_rows /* global, this workgroup's first row, or null */, int rows_valid) {
    for (int idx = c.tid; idx < TK_R * 64; idx += TK_T) {
        const int r = idx >> 6, c4 = (idx & 63) * 4;
        float4 v = *reinterpret_cast<const float4*>(c.Q + (r * 256 + c4) * 4);
        if (pe_rows && r < rows_valid) {
            const float4 p = *reinterpret_cast<const float4*>(pe_rows + r * 256 + c4);
            v.x += p.x; v.y += p.y; v.z += p.z; v.w += p.w;
        }
        *reinterpret_cast<uint2*>(B + r * TK_AS + c4 * 2) = make_uint2(pack_op16(v.x, v.y), pack_op16(v.z, v.w));
    }
}
// Q[row] = LN(Q[row]) for the workgroup's 32 rows (wave w: rows 4 w .. 4 w + 3; a lane holds 4 channels)
__device__ __forceinline__ void ln_rows(const TokCtx& c, TokLn ln, float eps) {
    const float4 g = *reinterpret_cast<const float4*>(ln.g + 4 * c.lane), b = *reinterpret_cast<const float4*>(ln.b + 4 * c.lane);
    for (int r = 4 * c.wave; r < 4 * c.wave + 4; ++r) {
        float4* q = reinterpret_cast<float4*>(c.Q + (r * 256 + 4 * c.lane) * 4);
        const float4 v = *q;
        const float mean = wave_sum((v.x + v.y) + (v.z + v.w)) * (1.0f / 256.0f);
        const float a0 = v.x - mean, a1 = v.y - mean, a2 = v.z - mean, a3 = v.w - mean;
        const float rstd = 1.0f / sqrtf(wave_sum((a0 * a0 + a1 * a1) + (a2 * a2 + a3 * a3)) * (1.0f / 256.0f) + eps);
        *q = make_float4(a0 * rstd * g.x + b.x, a1 * rstd * g.y + b.y, a2 * rstd * g.z + b.z, a3 * rstd * g.w + b.w);
    }
}
// Q (+)= A . W^T + bias for N = 256 (residual: add to Q, else overwrite)
__device__ __forceinline__ void proj_to_q(const TokCtx& c, const char* A, int K, TokLin L, bool residual) {
    f32x4 acc[2][2];
    wgemm<2, 2>(c, A, K, L.w, L.ldw, 0, 256, acc, L.wpk, L.npk, 0);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int n = 16 * (c.wave * 2 + i) + 4 * c.fg;
        const float4 b = *reinterpret_cast<const float4*>(L.b + n);
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            float4* q = reinterpret_cast<float4*>(c.Q + ((16 * m + c.fi) * 256 + n) * 4);
            float4 v = make_float4(acc[i][m][0] + b.x, acc[i][m][1] + b.y, acc[i][m][2] + b.z, acc[i][m][3] + b.w);
            if (residual) { const float4 o = *q; v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w; }
            *q = v;
        }
    }
}
// F[row][0..127] = A . W[n0 .. n0+127]^T + bias (fp32 scratch, 128 columns)
__device__ __forceinline__ void proj_to_f(const TokCtx& c, const char* A, int K, TokLin L, int n0, char* F) {
    f32x4 acc[1][2];
    wgemm<1, 2>(c, A, K, L.w, L.ldw, n0, L.n, acc, L.wpk, L.npk, 0);
#pragma unroll
    for (int i = 0; i < 1; ++i) {
        const int n = 16 * (c.wave + i) + 4 * c.fg;
        const float4 b = *reinterpret_cast<const float4*>(L.b + n0 + n);
#pragma unroll
        for (int m = 0; m < 2; ++m)
            *reinterpret_cast<float4*>(F + ((16 * m + c.fi) * TK_FS + n) * 4) =
                make_float4(acc[i][m][0] + b.x, acc[i][m][1] + b.y, acc[i][m][2] + b.z, acc[i][m][3] + b.w);
    }
}
__device__ __forceinline__ op16x8 pack8(const float (&v)[8]) {
    const uint4 u = make_uint4(pack_op16(v[0], v[1]), pack_op16(v[2], v[3]), pack_op16(v[4], v[5]), pack_op16(v[6], v[7]));
    return __builtin_bit_cast(op16x8, u);
}
// block-diagonal bf16 hi / lo operand of a fold: row (hsel, t) = fi, k = 8 fg .. + 7 of [head 2 hp | head 2 hp + 1]
__device__ __forceinline__ void fold_operand(const TokCtx& c, const char* F, int pi, int hp, float scale, op16x8* hi, op16x8* lo) {
    const int hsel = c.fi >> 3, t = c.fi & 7, hk = c.fg >> 1;
    const float* a = reinterpret_cast<const float*>(F) + (8 * pi + t) * TK_FS + 16 * (2 * hp + hsel) + 8 * (c.fg & 1);
    const float z = hk == hsel ? scale : 0.f;
    float v[8], l[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        v[j] = a[j] * z;
        l[j] = v[j] - op2f(f2op(v[j]));
    }
    *hi = pack8(v); *lo = pack8(l);
}
// out[p][8 h + t][d] = scale sum_j a[t][16 h + j] W(16 h + j, d), W given TRANSPOSED as WT bf16 [256][128]  (dec_fold_kernel mode 0)
// (the wave's 8 weight fragments - 4 head pairs x 2 channel tiles - are loaded once, ahead of the loop over the prompts: fetched inside it
// they cost one L2 round trip per (prompt, head pair), 16 in a row)
__device__ __forceinline__ void fold_rows(const TokCtx& c, const char* F, const bf16_t* WT, float scale, bf16_t* out, int p0, int P) {
    op16x8 w[4][2];
#pragma unroll
    for (int hp = 0; hp < 4; ++hp)
#pragma unroll
        for (int i = 0; i < 2; ++i)
            w[hp][i] = __builtin_bit_cast(op16x8, *reinterpret_cast<const uint4*>(WT + (16 * (c.wave * 2 + i) + c.fi) * 128 + 32 * hp + 8 * c.fg));
    for (int pi = 0; pi < TK_G; ++pi) {
        if (p0 + pi >= P) break;
#pragma unroll
        for (int hp = 0; hp < 4; ++hp) {
            op16x8 hi, lo;
            fold_operand(c, F, pi, hp, scale, &hi, &lo);
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int dt = c.wave * 2 + i;
                f32x4 acc = MFMA_16x16x32(w[hp][i], hi, (f32x4){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                acc = MFMA_16x16x32(w[hp][i], lo, acc, 0, 0, 0);
                // D[d = 16 dt + 4 fg + r][(hsel, t) = fi]
                bf16_t* o = out + ((int64_t)(p0 + pi) * 64 + 8 * (2 * hp + (c.fi >> 3)) + (c.fi & 7)) * 256 + 16 * dt + 4 * c.fg;
                *reinterpret_cast<uint2*>(o) = make_uint2(pack_op16(acc[0], acc[1]), pack_op16(acc[2], acc[3]));
            }
        }
    }
}
// out[p][d][8 h + t] = sum_j a[t][16 h + j] W[d][16 h + j], W bf16 [256][128]  (dec_fold_kernel mode 1: transposed output)
__device__ __forceinline__ void fold_cols(const TokCtx& c, const char* F, const bf16_t* W, int ldw, bf16_t* out, int p0, int P) {
    op16x8 w[4][2];
#pragma unroll
    for (int hp = 0; hp < 4; ++hp)
#pragma unroll
        for (int i = 0; i < 2; ++i)
            w[hp][i] = __builtin_bit_cast(op16x8, *reinterpret_cast<const uint4*>(W + (int64_t)(16 * (c.wave * 2 + i) + c.fi) * ldw + 32 * hp + 8 * c.fg));
    for (int pi = 0; pi < TK_G; ++pi) {
        if (p0 + pi >= P) break;
#pragma unroll
        for (int hp = 0; hp < 4; ++hp) {
            op16x8 hi, lo;
            fold_operand(c, F, pi, hp, 1.0f, &hi, &lo);
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int dt = c.wave * 2 + i;
                f32x4 acc = MFMA_16x16x32(hi, w[hp][i], (f32x4){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                acc = MFMA_16x16x32(lo, w[hp][i], acc, 0, 0, 0);
                // D[(hsel, t) = 4 fg + r][d = 16 dt + fi]  ->  columns 16 hp + 4 fg + r of row d
                bf16_t* o = out + ((int64_t)(p0 + pi) * 256 + 16 * dt + c.fi) * 64 + 16 * hp + 4 * c.fg;
                *reinterpret_cast<uint2*>(o) = make_uint2(pack_op16(acc[0], acc[1]), pack_op16(acc[2], acc[3]));
            }
        }
    }
}

__global__ __launch_bounds__(TK_T) void dec_tokens_kernel(TokSeg s, unsigned long long* __restrict__ stamps) {
    // development: s_memtime at the phase boundaries (tools/tok_stamps.py), wave 0 of every workgroup; nullptr in production
    unsigned long long tst[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tprev = stamps ? __builtin_amdgcn_s_memtime() : 0;
#define TK_STAMP(k) do { if (stamps) { const unsigned long long _n = __builtin_amdgcn_s_memtime(); tst[k] += _n - tprev; tprev = _n; } } while (0)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    TokCtx c;
    c.tid = threadIdx.x; c.lane = c.tid & 63; c.wave = __builtin_amdgcn_readfirstlane(c.tid >> 6); c.fi = c.lane & 15; c.fg = c.lane >> 4;
    c.Q = smem; c.B0 = c.Q + TK_R * 256 * 4; c.B1 = c.B0 + TK_R * TK_AS; c.H = c.B1 + TK_R * TK_AS;
    c.F0 = c.H + TK_R * TK_AS; c.F1 = c.F0 + TK_R * TK_FS * 4; c.F2 = c.F1 + TK_R * TK_FS * 4;
    const int p0 = blockIdx.x * TK_G;
    const int np = min(TK_G, s.P - p0), rows = 8 * np;
    const int64_t row0 = (int64_t)p0 * 8;
    const float* pe = s.tok_pe + row0 * 256;
    // residual stream of the workgroup's tokens
    for (int idx = c.tid; idx < TK_R * 64; idx += TK_T) {
        const int r = idx >> 6, c4 = (idx & 63) * 4;
        *reinterpret_cast<float4*>(c.Q + (r * 256 + c4) * 4) = r < rows ? *reinterpret_cast<const float4*>(s.queries + (row0 + r) * 256 + c4) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    TK_STAMP(0);
    // ---------------- (1) output projection of the tokens -> image attention that has just run
    if (s.t_att) {
        for (int idx = c.tid; idx < TK_R * 16; idx += TK_T) {
            const int r = idx >> 4, c8 = (idx & 15) * 8;
            *reinterpret_cast<uint4*>(c.B1 + r * TK_AS + c8 * 2) = r < rows ? *reinterpret_cast<const uint4*>(s.t_att + (row0 + r) * 128 + c8) : make_uint4(0u, 0u, 0u, 0u);
        }
        __syncthreads();
        proj_to_q(c, c.B1, 128, s.att_o, true);
        __syncthreads();
        ln_rows(c, s.att_ln, s.att_eps);
    }
    __syncthreads();
    TK_STAMP(1);
    // ---------------- (2) MLP, LN3, operands of the image -> tokens attention
    if (s.do_mlp) {
        to_operand(c, c.B0, nullptr, rows);
        __syncthreads();
        f32x4 acc2[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int m = 0; m < 2; ++m) acc2[i][m] = (f32x4){0.f, 0.f, 0.f, 0.f};
        uint4 w1f[2][8], w2f[2][8];
        wfrag_load(c, s.mlp1_pk, 2048, 0, 0, w1f);
#pragma unroll 1
        for (int ch = 0; ch < 8; ++ch) {            // hidden columns 256 ch .. 256 ch + 255
            f32x4 acc[2][2];
            wfrag_load(c, s.mlp2_pk, 256, 0, 8 * ch, w2f);       // this chunk's second panel, under the first one's MFMAs
            wgemm_pre(c, c.B0, w1f, acc);
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int n = 16 * (c.wave * 2 + i) + 4 * c.fg;
                const float4 b = *reinterpret_cast<const float4*>(s.mlp1.b + 256 * ch + n);
#pragma unroll
                for (int m = 0; m < 2; ++m)
                    *reinterpret_cast<uint2*>(c.H + (16 * m + c.fi) * TK_AS + n * 2) =
                        make_uint2(pack_op16(fmaxf(acc[i][m][0] + b.x, 0.f), fmaxf(acc[i][m][1] + b.y, 0.f)), pack_op16(fmaxf(acc[i][m][2] + b.z, 0.f), fmaxf(acc[i][m][3] + b.w, 0.f)));
            }
            if (ch + 1 < 8) wfrag_load(c, s.mlp1_pk, 2048, 256 * (ch + 1), 0, w1f);      // the next chunk's first panel
            __syncthreads();
            f32x4 part[2][2];
            wgemm_pre(c, c.H, w2f, part);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int m = 0; m < 2; ++m) acc2[i][m] += part[i][m];
            __syncthreads();
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int n = 16 * (c.wave * 2 + i) + 4 * c.fg;
            const float4 b = *reinterpret_cast<const float4*>(s.mlp2.b + n);
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                float4* q = reinterpret_cast<float4*>(c.Q + ((16 * m + c.fi) * 256 + n) * 4);
                const float4 o = *q;
                *q = make_float4(o.x + acc2[i][m][0] + b.x, o.y + acc2[i][m][1] + b.y, o.z + acc2[i][m][2] + b.z, o.w + acc2[i][m][3] + b.w);
            }
        }
        __syncthreads();
        ln_rows(c, s.ln3, 1e-5f);
        __syncthreads();
        TK_STAMP(2);
        // image -> tokens: k = k_proj(queries + pe), v = v_proj(queries); folded with the image side's q_proj / out_proj
        to_operand(c, c.B0, pe, rows);
        to_operand(c, c.B1, nullptr, rows);
        __syncthreads();
        proj_to_f(c, c.B0, 256, s.i2t_k, 0, c.F0);
        proj_to_f(c, c.B1, 256, s.i2t_v, 0, c.F1);
        __syncthreads();
        for (int idx = c.tid; idx < rows * 32; idx += TK_T) {
            const int r = idx >> 5, c4 = (idx & 31) * 4;
            *reinterpret_cast<float4*>(s.tk_out + (row0 + r) * 128 + c4) = *reinterpret_cast<const float4*>(c.F0 + (r * TK_FS + c4) * 4);
        }
        if (c.tid < 64 * np) {      // cb[p][8 h + t] = scale sum_j k[t][16 h + j] b_q[16 h + j]
            const int pi = c.tid >> 6, h = (c.tid >> 3) & 7, t = c.tid & 7;
            const float* a = reinterpret_cast<const float*>(c.F0) + (8 * pi + t) * TK_FS + 16 * h;
            float acc = 0.f;
            for (int j = 0; j < 16; ++j) acc += a[j] * s.i2t_qb[16 * h + j];
            s.fold_cb[(int64_t)(p0 + pi) * 64 + 8 * h + t] = acc * s.kscale;
        }
        TK_STAMP(3);
        fold_rows(c, c.F0, s.i2t_qT, s.kscale, s.fold_k, p0, s.P);
        fold_cols(c, c.F1, s.i2t_o, 128, s.fold_v, p0, s.P);
        __syncthreads();
    }
    TK_STAMP(4);
    // ---------------- (3) self attention of the tokens
    if (s.do_self) {
        if (!s.do_mlp) {      // (after (2) B0 = bf16(queries + pe) and B1 = bf16(queries) are already in place)
            to_operand(c, c.B0, s.self_first ? nullptr : pe, rows);
            if (!s.self_first) to_operand(c, c.B1, nullptr, rows);
            __syncthreads();
        }
        const char* xv = s.self_first ? c.B0 : c.B1;
        for (int hc = 0; hc < 2; ++hc) {             // heads 4 hc .. 4 hc + 3 (128 columns)
            proj_to_f(c, c.B0, 256, s.sa_q, 128 * hc, c.F0);
            proj_to_f(c, c.B0, 256, s.sa_k, 128 * hc, c.F1);
            proj_to_f(c, xv, 256, s.sa_v, 128 * hc, c.F2);
            __syncthreads();
            if (c.tid < 32 * TK_G) {
                const int pi = c.tid >> 5, hh = (c.tid >> 3) & 3, qi = c.tid & 7;
                const float* qp = reinterpret_cast<const float*>(c.F0) + (8 * pi + qi) * TK_FS + 32 * hh;
                float qv[32];
#pragma unroll
                for (int d = 0; d < 32; ++d) qv[d] = qp[d];
                const float scale = rsqrtf(32.0f);
                float sc[8], mx = -3.0e38f;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float* kp = reinterpret_cast<const float*>(c.F1) + (8 * pi + j) * TK_FS + 32 * hh;
                    float a = 0.f;
#pragma unroll
                    for (int d = 0; d < 32; ++d) a += qv[d] * kp[d];
                    sc[j] = a * scale;
                    mx = fmaxf(mx, sc[j]);
                }
                float sum = 0.f;
#pragma unroll
                for (int j = 0; j < 8; ++j) { sc[j] = expf(sc[j] - mx); sum += sc[j]; }
                const float inv = 1.0f / sum;
                float o[32];
#pragma unroll
                for (int d = 0; d < 32; ++d) o[d] = 0.f;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float* vp = reinterpret_cast<const float*>(c.F2) + (8 * pi + j) * TK_FS + 32 * hh;
                    const float pj = sc[j] * inv;
#pragma unroll
                    for (int d = 0; d < 32; ++d) o[d] += pj * vp[d];
                }
                char* op = c.H + (8 * pi + qi) * TK_AS + (128 * hc + 32 * hh) * 2;
#pragma unroll
                for (int d = 0; d < 32; d += 4) *reinterpret_cast<uint2*>(op + d * 2) = make_uint2(pack_op16(o[d], o[d + 1]), pack_op16(o[d + 2], o[d + 3]));
            }
            __syncthreads();
        }
        proj_to_q(c, c.H, 256, s.sa_o, !s.self_first);
        __syncthreads();
        ln_rows(c, s.ln1, 1e-5f);
        __syncthreads();
    }
    TK_STAMP(5);
    // ---------------- (4) operands of the next tokens -> image attention: q = q_proj(queries + pe), folded with the image side's k_proj
    if (s.do_t2i) {
        to_operand(c, c.B0, pe, rows);
        __syncthreads();
        proj_to_f(c, c.B0, 256, s.t2i_q, 0, c.F0);
        __syncthreads();
        for (int idx = c.tid; idx < rows * 32; idx += TK_T) {
            const int r = idx >> 5, c4 = (idx & 31) * 4;
            *reinterpret_cast<float4*>(s.tq_out + (row0 + r) * 128 + c4) = *reinterpret_cast<const float4*>(c.F0 + (r * TK_FS + c4) * 4);
        }
        fold_rows(c, c.F0, s.t2i_kT, s.kscale, s.fold_q, p0, s.P);
    }
    TK_STAMP(6);
    // the residual stream goes back (the next segment, or saber_get_decoder_tokens, reads it)
    __syncthreads();
    for (int idx = c.tid; idx < rows * 64; idx += TK_T) {
        const int r = idx >> 6, c4 = (idx & 63) * 4;
        *reinterpret_cast<float4*>(s.queries + (row0 + r) * 256 + c4) = *reinterpret_cast<const float4*>(c.Q + (r * 256 + c4) * 4);
    }
    // ---------------- (5) heads on the final tokens: [obj, iou, mask 0..3, point, pad]
    if (s.do_heads) {
        // three-layer MLP on ONE token per prompt: rows 0..3 of a 16-row operand tile (the other rows are zero)
        auto mlp3 = [&](const TokLin* L, int64_t w_off0, int64_t w_off2, int r_off0, int r_off2, int b_off0, int b_off2, int token, int n_out, int sigmoid, float* out, int ldo, int o_off) {
            for (int idx = c.tid; idx < 16 * 64; idx += TK_T) {
                const int r = idx >> 6, c4 = (idx & 63) * 4;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (r < np) v = *reinterpret_cast<const float4*>(c.Q + ((8 * r + token) * 256 + c4) * 4);
                *reinterpret_cast<uint2*>(c.B0 + r * TK_AS + c4 * 2) = make_uint2(pack_op16(v.x, v.y), pack_op16(v.z, v.w));
            }
            __syncthreads();
            for (int l = 0; l < 2; ++l) {
                const char* in = l == 0 ? c.B0 : c.B1;
                char* outb = l == 0 ? c.B1 : c.H;
                f32x4 acc[2][1];
                wgemm<2, 1>(c, in, 256, L[l].w + w_off0, L[l].ldw, 0, 256, acc, L[l].wpk, L[l].npk, r_off0);
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int n = 16 * (c.wave * 2 + i) + 4 * c.fg;
                    const float4 b = *reinterpret_cast<const float4*>(L[l].b + b_off0 + n);
                    *reinterpret_cast<uint2*>(outb + c.fi * TK_AS + n * 2) =
                        make_uint2(pack_op16(fmaxf(acc[i][0][0] + b.x, 0.f), fmaxf(acc[i][0][1] + b.y, 0.f)), pack_op16(fmaxf(acc[i][0][2] + b.z, 0.f), fmaxf(acc[i][0][3] + b.w, 0.f)));
                }
                __syncthreads();
            }
            if (c.wave * 16 < n_out) {       // n_out <= 32: waves 0 (and 1)
                f32x4 acc[1][1];
                wgemm<1, 1>(c, c.H, 256, L[2].w + w_off2, L[2].ldw, 0, n_out, acc, L[2].wpk, L[2].npk, r_off2);
                if (c.fi < np) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int n = 16 * c.wave + 4 * c.fg + r;
                        if (n < n_out) {
                            float v = acc[0][0][r] + L[2].b[b_off2 + n];
                            if (sigmoid) v = 1.0f / (1.0f + __expf(-v));
                            out[(int64_t)(p0 + c.fi) * ldo + o_off + n] = v;
                        }
                    }
                }
            }
            __syncthreads();
        };
        mlp3(s.iou, 0, 0, 0, 0, 0, 0, 1, 4, 1, s.iou4, 4, 0);
        if (s.obj_out) mlp3(s.obj, 0, 0, 0, 0, 0, 0, 0, 1, 0, s.obj_out, 1, 0);
        for (int k = 0; k < 4; ++k)
            mlp3(s.hyper, (int64_t)k * 256 * s.hyper[0].ldw, (int64_t)k * 32 * s.hyper[2].ldw, 256 * k, 32 * k, 256 * k, 32 * k, 2 + k, 32, 0, s.hyper_out, 128, 32 * k);
    }
    TK_STAMP(7);
    if (stamps && c.tid == 0)
        for (int k = 0; k < 8; ++k) stamps[(int64_t)blockIdx.x * 8 + k] = tst[k];
#undef TK_STAMP
}

#define TK_LDS (TK_R * 256 * 4 + 3 * TK_R * TK_AS + 3 * TK_R * TK_FS * 4)
const char* launch_dec_tokens(const TokSeg& s, hipStream_t st) {
    if (s.P <= 0) return nullptr;
    if (s.do_mlp && (!s.mlp1_pk || !s.mlp2_pk)) return "dec_tokens: the MLP needs the K-step-packed copies of its two weights (launch_pack_w_kstep)";
    // development (tools/tok_stamps.py): with a stamp buffer set, the segment whose index in the decode (call count mod 4) equals
    // SABER_AMD_TOK_STAMP_SEG writes its stamps at element 500 000 of the buffer (dec_t2i / dec_i2t write theirs from element 0)
    unsigned long long* stamps = nullptr;
    if (g_saber_stamp_buf) {
        static int calls = 0;
        const char* e = getenv("SABER_AMD_TOK_STAMP_SEG");
        if (e && (calls & 3) == atoi(e)) stamps = g_saber_stamp_buf + 500000;
        ++calls;
    }
    hipLaunchKernelGGL(dec_tokens_kernel, dim3((s.P + TK_G - 1) / TK_G), dim3(TK_T), TK_LDS, st, s, stamps);
    return nullptr;
}
const char* decoder_tokens_init_device() {
    const hipError_t st = hipFuncSetAttribute(reinterpret_cast<const void*>(dec_tokens_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, TK_LDS);
    return st == hipSuccess ? nullptr : hipGetErrorString(st);
}
