// Exact-precision mode of the engine (saber_engine_set_precision(e, SABER_PRECISION_EXACT)).
//
// The reference runs the SAM2 image model in fp32 (saber/utils/io.py:127-132: TF32 allowed, autocast commented out) and the north star
// asks for logits within 1e-3 rel of it.  The production path rounds every GEMM / attention operand to bf16, which costs 3-8e-3 end to
// end.  This file is the mode that meets the tolerance: the same engine, the same token order, slots, weights and
// C-ABI, but every operand, every stored activation and every statistic in fp32:
//   * GEMMs on v_mfma_f32_16x16x4_f32 (fp32 in, fp32 accumulate: bit for bit an fmaf chain, MI355X_MICROARCH.md "Matrix cores"),
//   * attention products on the same fp32 MFMA (round 4), LayerNorm, GELU (libm erff, not the fitted form), softmax (expf) on the vector ALU,
//   * the mask decoder as the UNFOLDED composition upstream executes (k/v/q projections of the image tokens per prompt, 8-head
//     attention, out projection, residual, LayerNorm; two ConvTranspose2d as GEMMs, LayerNorm2d, GELU, hypernetwork product), so it
//     also checks the production kernels' folded t2i / i2t algebra and their fused upscaling against an independent formulation.
// A verification mode: a default-grid AMG slice takes 1.2 s (round 3: 3.7 s) against 0.14 s of the production arithmetic.
#include "engine.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>

#include "common.h"

#define TRY(x) do { int _r = (x); if (_r != SABER_OK) return _r; } while (0)

// ------------------------------------------------------------------------------------------------ fp32 GEMM
// C[m][n] = act(sum_k A[m][k] W[n][k] + bias[n]) (+ res) ; 128 x 64 tile per 256-thread workgroup, 16-deep K steps through LDS,
// wave w owns rows 32w..32w+31 of the tile as 2 x 4 MFMA tiles of 16 x 16.
struct XGemm {
    const float* A = nullptr; int64_t lda = 0; int64_t sA = 0;
    const float* W = nullptr; int64_t ldw = 0; int64_t sW = 0;
    const float* bias = nullptr; int64_t sBias = 0;
    const float* res = nullptr; int64_t ldres = 0; int res_shift = 0; int64_t res_mod = 0;
    // res_rows_per > 0: per-prompt rows against per-slot tables - the residual of row r is res[((r / res_rows_per + res_off) / res_div) * res_stride +
    // (r % res_rows_per) * ldres + n] (the xg_add_slot mapping folded into the epilogue)
    int64_t res_rows_per = 0, res_stride = 0; int res_div = 1, res_off = 0;
    // A2: the operand is A[m][k] + A2[(m % a2_mod)][k], summed in fp32 before the product exactly as a stored sum would be (keys + dense_pe)
    const float* A2 = nullptr; int64_t lda2 = 0; int64_t a2_mod = 1;
    int64_t row0 = 0;      // first row of this launch within the whole GEMM (row slabs): the A2 and slot-residual mappings count from there
    float* C = nullptr; int64_t ldc = 0; int64_t sC = 0;
    int M = 0, N = 0, K = 0, act = ACT_NONE, act_last = 0, pool4 = 0, batch = 1;
};

__device__ __attribute__((noinline)) float x_gelu(float x) {   // noinline: 64 epilogue values x libm erff would be 50 k instructions per GEMM kernel
    return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
__device__ __forceinline__ float x_act(float v, int act) {
    if (act == ACT_GELU) return x_gelu(v);
    if (act == ACT_RELU) return fmaxf(v, 0.0f);
    if (act == ACT_SIGMOID) return 1.0f / (1.0f + expf(-v));
    return v;
}

#define XG_BM 128
#define XG_BN 64
#define XG_BK 16
__global__ __launch_bounds__(256) void xg_gemm_kernel(XGemm p) {
    __shared__ float As[XG_BK][XG_BM + 16];
    __shared__ float Bs[XG_BK][XG_BN + 16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t m0 = (int64_t)blockIdx.y * XG_BM;
    const int n0 = blockIdx.x * XG_BN;
    const int b = blockIdx.z;
    const float* A = p.A + b * p.sA;
    const float* W = p.W + b * p.sW;
    f32x4 acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int arow = tid & 127, akh = tid >> 7;        // A: row of the tile, 8-float half of the K step
    const int wrow = tid & 63, wkq = tid >> 6;         // W: row (n) of the tile, 4-float quarter of the K step
    const bool a_ok = m0 + arow < p.M, w_ok = n0 + wrow < p.N;
    const float* ap = A + (m0 + arow) * p.lda + akh * 8;
    const float* wp = W + (int64_t)(n0 + wrow) * p.ldw + wkq * 4;
    // 16-byte loads where the rows allow it (every GEMM of the model: K and the leading dimensions are multiples of 4), and the NEXT K step's
    // operands are in flight while the current one is multiplied (round 4: the loop used to wait for its scalar loads at the top of every
    // step).  The products and their order are unchanged: results are bit-identical to the unpipelined loop.
    const bool vec = ((p.lda | p.ldw) & 3) == 0 && (p.K & 3) == 0 && ((reinterpret_cast<uintptr_t>(A) | reinterpret_cast<uintptr_t>(W)) & 15) == 0;
    float av[8], wv[4];
    auto gload = [&](int k0) {
        if (vec) {
            const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
            const float4 a0 = (a_ok && k0 + akh * 8 < p.K) ? *reinterpret_cast<const float4*>(ap + k0) : z;
            const float4 a1 = (a_ok && k0 + akh * 8 + 4 < p.K) ? *reinterpret_cast<const float4*>(ap + k0 + 4) : z;
            const float4 w0 = (w_ok && k0 + wkq * 4 < p.K) ? *reinterpret_cast<const float4*>(wp + k0) : z;
            av[0] = a0.x; av[1] = a0.y; av[2] = a0.z; av[3] = a0.w; av[4] = a1.x; av[5] = a1.y; av[6] = a1.z; av[7] = a1.w;
            wv[0] = w0.x; wv[1] = w0.y; wv[2] = w0.z; wv[3] = w0.w;
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) av[j] = (a_ok && k0 + akh * 8 + j < p.K) ? ap[k0 + j] : 0.0f;
#pragma unroll
            for (int j = 0; j < 4; ++j) wv[j] = (w_ok && k0 + wkq * 4 + j < p.K) ? wp[k0 + j] : 0.0f;
        }
    };
    gload(0);
    for (int k0 = 0; k0 < p.K; k0 += XG_BK) {
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 8; ++j) As[akh * 8 + j][arow] = av[j];
#pragma unroll
        for (int j = 0; j < 4; ++j) Bs[wkq * 4 + j][wrow] = wv[j];
        __syncthreads();
        if (k0 + XG_BK < p.K) gload(k0 + XG_BK);
#pragma unroll
        for (int kk = 0; kk < XG_BK; kk += 4) {
            float a[2], bb[4];
#pragma unroll
            for (int i = 0; i < 2; ++i) a[i] = As[kk + (lane >> 4)][wave * 32 + i * 16 + (lane & 15)];
#pragma unroll
            for (int j = 0; j < 4; ++j) bb[j] = Bs[kk + (lane >> 4)][j * 16 + (lane & 15)];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], bb[j], acc[i][j], 0, 0, 0);
        }
    }
    // epilogue: C/D layout col = lane & 15, row = (lane >> 4) * 4 + reg
    float* C = p.C + b * p.sC;
    const float* bias = p.bias ? p.bias + b * p.sBias : nullptr;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = n0 + j * 16 + (lane & 15);
            if (n >= p.N) continue;
            const float bv = bias ? bias[n] : 0.0f;
            const int64_t r0 = m0 + wave * 32 + i * 16 + (lane >> 4) * 4;
            if (p.pool4) {   // rows 4q..4q+3 (the lane's four registers) -> output row q
                if (r0 < p.M) {
                    float v = fmaxf(fmaxf(acc[i][j][0], acc[i][j][1]), fmaxf(acc[i][j][2], acc[i][j][3])) + bv;
                    C[(r0 >> 2) * p.ldc + n] = x_act(v, p.act);
                }
                continue;
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int64_t row = r0 + r;
                if (row >= p.M) continue;
                float v = acc[i][j][r] + bv;
                if (!p.act_last) v = x_act(v, p.act);
                if (p.res) {
                    int64_t rr = row >> p.res_shift;
                    if (p.res_mod > 0) rr %= p.res_mod;
                    v += p.res[rr * p.ldres + n];
                }
                if (p.act_last) v = x_act(v, p.act);
                C[row * p.ldc + n] = v;
            }
        }
}
// The pipelined tile (round 4): 128 x BN outputs per workgroup, 32-deep K steps, operands row-major in LDS ([row][k], rows padded to 36
// floats: the 16-byte fragment reads of 16 rows x 4 k-quads and the staging writes are conflict-free), each lane reading the four k values
// of its quad with one ds_read_b128 and feeding MFMA e with element e - the k index of the MFMA is a permutation of the tile's k, identical
// for A and W.  Global loads are whole 128-byte row pieces (8 lanes per row) and the NEXT K step's loads are in flight during the MFMAs.
// Every output is still an fp32 fmaf chain over all k; only the order within a 16-k group differs from xg_gemm_kernel (e-major instead
// of ascending).  Needs K, lda, ldw multiples of 4 and 16-byte aligned bases (every GEMM of the model); others take xg_gemm_kernel.
template <int BN>
__global__ __launch_bounds__(256, 3) void xg_gemm2_kernel(XGemm p) {
    constexpr int LDK = 36, WM = BN == 128 ? 64 : 32, TI = WM / 16, TJ = 4, NB = BN / 32;
    __shared__ __attribute__((aligned(16))) float As[128 * LDK];
    __shared__ __attribute__((aligned(16))) float Bs[BN * LDK];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = BN == 128 ? (wave >> 1) * 64 : wave * 32;
    const int wn = BN == 128 ? (wave & 1) * 64 : 0;
    const int n0 = blockIdx.x * BN;
    const int b = blockIdx.z;
    const float* A = p.A + b * p.sA;
    const float* W = p.W + b * p.sW;
    const int lr = tid >> 3, kq = (tid & 7) * 4;
    f32x4 pa[4], pb[NB];
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    // Persistent over the row tiles blockIdx.y, blockIdx.y + gridDim.y, ...: the first K step of the NEXT tile is loaded during the last K step of
    // this one, so its latency and the address set-up hide behind the MFMAs and the epilogue's stores (the decoder's K = 64 ... 256 GEMMs over
    // millions of rows spend a third of their time at tile boundaries otherwise).
    const int tiles_m = (int)(((int64_t)p.M + 127) / 128);
    // row pointers of a tile (the A2 row index is a modulo: not in the K loop)
    const float* arow[4];
    const float* a2row[4];
    const float* wrow[NB];
#pragma unroll
    for (int r = 0; r < NB; ++r) {
        const int n = n0 + lr + 32 * r;
        wrow[r] = n < p.N ? W + (int64_t)n * p.ldw + kq : nullptr;
    }
    auto setup = [&](int tile) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int64_t row = (int64_t)tile * 128 + lr + 32 * r;
            arow[r] = row < p.M ? A + row * p.lda + kq : nullptr;
            a2row[r] = (p.A2 && row < p.M) ? p.A2 + (int64_t)((unsigned)(row + p.row0) % (unsigned)p.a2_mod) * p.lda2 + kq : nullptr;      // (rows < 2^32: 32-bit modulo)
        }
    };
    auto gload = [&](int k0) {
        const bool kin = k0 + kq < p.K;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            f32x4 v = z;
            if (kin && arow[r]) {
                v = *reinterpret_cast<const f32x4*>(arow[r] + k0);
                if (a2row[r]) v += *reinterpret_cast<const f32x4*>(a2row[r] + k0);
            }
            pa[r] = v;
        }
#pragma unroll
        for (int r = 0; r < NB; ++r) {
            f32x4 v = z;
            if (kin && wrow[r]) v = *reinterpret_cast<const f32x4*>(wrow[r] + k0);
            pb[r] = v;
        }
    };
    float* C = p.C + b * p.sC;
    const float* bias = p.bias ? p.bias + b * p.sBias : nullptr;
    float bv[TJ];
#pragma unroll
    for (int j = 0; j < TJ; ++j) {
        const int n = n0 + wn + j * 16 + (lane & 15);
        bv[j] = (bias && n < p.N) ? bias[n] : 0.0f;
    }
    const int act = p.act, act_last = p.act_last;
    int tile = blockIdx.y;
    if (tile >= tiles_m) return;
    setup(tile);
    gload(0);
    for (; tile < tiles_m; tile += gridDim.y) {
        const int64_t m0 = (int64_t)tile * 128;
        f32x4 acc[TI][TJ];
#pragma unroll
        for (int i = 0; i < TI; ++i)
#pragma unroll
            for (int j = 0; j < TJ; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int k0 = 0; k0 < p.K; k0 += 32) {
            __syncthreads();
#pragma unroll
            for (int r = 0; r < 4; ++r) *reinterpret_cast<f32x4*>(&As[(lr + 32 * r) * LDK + kq]) = pa[r];
#pragma unroll
            for (int r = 0; r < NB; ++r) *reinterpret_cast<f32x4*>(&Bs[(lr + 32 * r) * LDK + kq]) = pb[r];
            __syncthreads();
            if (k0 + 32 < p.K) gload(k0 + 32);
            else if (tile + (int)gridDim.y < tiles_m) { setup(tile + gridDim.y); gload(0); }
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                f32x4 af[TI], bf[TJ];
#pragma unroll
                for (int i = 0; i < TI; ++i) af[i] = *reinterpret_cast<const f32x4*>(&As[(wm + i * 16 + (lane & 15)) * LDK + h * 16 + (lane >> 4) * 4]);
#pragma unroll
                for (int j = 0; j < TJ; ++j) bf[j] = *reinterpret_cast<const f32x4*>(&Bs[(wn + j * 16 + (lane & 15)) * LDK + h * 16 + (lane >> 4) * 4]);
#pragma unroll
                for (int i = 0; i < TI; ++i)
#pragma unroll
                    for (int j = 0; j < TJ; ++j) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i][0], bf[j][0], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i][1], bf[j][1], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i][2], bf[j][2], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i][3], bf[j][3], acc[i][j], 0, 0, 0);
                    }
            }
        }
        // epilogue: C/D layout col = lane & 15, row = (lane >> 4) * 4 + reg.  Row-dependent addressing (the residual's row mapping has integer
        // divisions) once per row, not per value; rows fit 32 bits (M is an int).
#pragma unroll
        for (int i = 0; i < TI; ++i) {
            const int64_t r0 = m0 + wm + i * 16 + (lane >> 4) * 4;
            if (p.pool4) {   // rows 4q..4q+3 (the lane's four registers) -> output row q
#pragma unroll
                for (int j = 0; j < TJ; ++j) {
                    const int n = n0 + wn + j * 16 + (lane & 15);
                    if (n < p.N && r0 < p.M) {
                        const float v = fmaxf(fmaxf(acc[i][j][0], acc[i][j][1]), fmaxf(acc[i][j][2], acc[i][j][3])) + bv[j];
                        C[(r0 >> 2) * p.ldc + n] = x_act(v, act);
                    }
                }
                continue;
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int64_t row = r0 + r;
                if (row >= p.M) continue;
                const float* resrow = nullptr;
                if (p.res) {
                    if (p.res_rows_per > 0) {
                        const unsigned grow = (unsigned)(row + p.row0);      // < 2^32
                        const unsigned pr = grow / (unsigned)p.res_rows_per;
                        resrow = p.res + (int64_t)((pr + (unsigned)p.res_off) / (unsigned)p.res_div) * p.res_stride + (int64_t)(grow - pr * (unsigned)p.res_rows_per) * p.ldres;
                    } else {
                        unsigned rr = (unsigned)row >> p.res_shift;
                        if (p.res_mod > 0) rr %= (unsigned)p.res_mod;
                        resrow = p.res + (int64_t)rr * p.ldres;
                    }
                }
                float* crow = C + row * p.ldc;
#pragma unroll
                for (int j = 0; j < TJ; ++j) {
                    const int n = n0 + wn + j * 16 + (lane & 15);
                    if (n >= p.N) continue;
                    float v = acc[i][j][r] + bv[j];
                    if (!act_last) v = x_act(v, act);
                    if (resrow) v += resrow[n];
                    if (act_last) v = x_act(v, act);
                    crow[n] = v;
                }
            }
        }
    }
}
static const char* xg_gemm(const XGemm& p, hipStream_t s) {
    if (p.M <= 0 || p.N <= 0) return nullptr;
    if (!p.A || !p.W || !p.C || p.K <= 0) return "exact gemm: bad argument";
    if (p.pool4 && (p.M % 4)) return "exact gemm: pool4 needs M % 4 == 0";
    const bool vec = ((p.lda | p.ldw | p.sA | p.sW | (int64_t)p.K) & 3) == 0 && ((reinterpret_cast<uintptr_t>(p.A) | reinterpret_cast<uintptr_t>(p.W)) & 15) == 0 &&
                     (!p.A2 || (((p.lda2 & 3) == 0) && (reinterpret_cast<uintptr_t>(p.A2) & 15) == 0));
    if (!vec && (p.A2 || p.res_rows_per > 0)) return "exact gemm: the fused operand sum / slot residual need 16-byte rows";
    const int64_t gy = ((int64_t)p.M + XG_BM - 1) / XG_BM;
    if (vec) {
        // short-K GEMMs (K <= 128: the 128 -> 256 out-projection of the image -> token attention, the K = 64 up-convolution) take 128 x 64 tiles:
        // 110 registers = four workgroups per CU to hide their tile-start and store latency
        const int bn = (p.N > 64 && !(p.K <= 128 && p.N <= 256)) ? 128 : 64;
        const int n_cu = saber_cu_count();
        // persistent row tiles: as many workgroups as stay resident (3 or 4 per CU), each walking blockIdx.y, + gridDim.y, ... ; the column tiles of
        // one row tile stay neighbours in dispatch order (they share the A tile in L2)
        const int gx = (p.N + bn - 1) / bn;
        const int64_t resident = (int64_t)n_cu * (bn == 128 ? 3 : 4);
        int64_t ny = std::max<int64_t>(1, resident / ((int64_t)gx * p.batch));
        if (gy < 8 * ny && gy <= 65535) ny = gy;      // few tiles per resident workgroup (the encoder's 672-tile GEMMs: 4.4): one tile per workgroup, the dispatcher balances
        ny = std::min<int64_t>(std::min<int64_t>(ny, gy), 65535);
        const dim3 grid(gx, (unsigned)ny, p.batch);
        if (bn == 128) hipLaunchKernelGGL(xg_gemm2_kernel<128>, grid, dim3(256), 0, s, p);
        else hipLaunchKernelGGL(xg_gemm2_kernel<64>, grid, dim3(256), 0, s, p);
        return nullptr;
    }
    if (gy > 65535 * 32) return "exact gemm: M too large";
    // gridDim.y is limited to 65535: large M is split into row slabs
    const int64_t slab = 65535;
    for (int64_t y0 = 0; y0 < gy; y0 += slab) {
        XGemm q = p;
        const int64_t rows0 = y0 * XG_BM;
        const int64_t ny = std::min(slab, gy - y0);
        q.A = p.A + rows0 * p.lda;
        q.C = p.C + (p.pool4 ? rows0 / 4 : rows0) * p.ldc;
        q.M = (int)std::min<int64_t>((int64_t)p.M - rows0, ny * XG_BM);
        q.row0 = rows0;
        if (p.res) {
            if (p.res_mod > 0 || p.res_shift) { if (y0 > 0) return "exact gemm: residual mapping with M beyond one slab"; }
            else if (p.res_rows_per <= 0) q.res = p.res + rows0 * p.ldres;
        }
        hipLaunchKernelGGL(xg_gemm_kernel, dim3((p.N + XG_BN - 1) / XG_BN, (unsigned)ny, p.batch), dim3(256), 0, s, q);
    }
    return nullptr;
}

// ------------------------------------------------------------------------------------------------ LayerNorm (fp32 in / out)
// one wave per row; two-pass statistics; rows flagged invalid (window padding of the 14 x 14 trunks) are written as zeros
__global__ __launch_bounds__(256) void xg_layernorm_kernel(const float* __restrict__ x, const float* __restrict__ g, const float* __restrict__ be,
                                                          float eps, float* __restrict__ out, int64_t rows, int C, int act,
                                                          const uint8_t* __restrict__ row_valid, int valid_mod) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* xr = x + row * C;
    float* o = out + row * C;
    if (row_valid && !row_valid[row % valid_mod]) {
        for (int c = lane; c < C; c += 64) o[c] = 0.0f;
        return;
    }
    float sum = 0.f;
    for (int c = lane; c < C; c += 64) sum += xr[c];
    const float mu = wave_sum(sum) / (float)C;
    float sq = 0.f;
    for (int c = lane; c < C; c += 64) { const float d = xr[c] - mu; sq += d * d; }
    const float var = wave_sum(sq) / (float)C;
    const float rstd = 1.0f / sqrtf(var + eps);
    for (int c = lane; c < C; c += 64) o[c] = x_act((xr[c] - mu) * rstd * g[c] + be[c], act);
}
// Register-resident rows (round 4): LPR lanes share a row (64, or 16 for the 64-channel LayerNorm2d of the upscaling: 4 rows per wave), every
// lane keeps its float4 pieces, so a row is read once (the kernel above reads it three times) with 16-byte accesses.  Two-pass statistics as
// above; only the order of the sums differs.  C % 4 == 0, C <= 1280.
template <int LPR, int NV>
__global__ __launch_bounds__(256) void xg_layernorm4_kernel(const float* __restrict__ x, const float* __restrict__ g, const float* __restrict__ be,
                                                           float eps, float* __restrict__ out, int64_t rows, int C, int act,
                                                           const uint8_t* __restrict__ row_valid, int valid_mod) {
    constexpr int RPW = 64 / LPR;      // NV = float4 pieces per lane: C <= 4 * LPR * NV
    const int lane = threadIdx.x & 63, sub = lane % LPR, rw = lane / LPR;
    const int64_t row = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * RPW + rw;
    const bool live = row < rows;
    const int C4 = C >> 2;
    const f32x4* xr = reinterpret_cast<const f32x4*>(x + row * C);
    f32x4* o = reinterpret_cast<f32x4*>(out + row * C);
    if (LPR == 64 && row_valid && live && !row_valid[row % valid_mod]) {      // wave-uniform: one row per wave
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
        for (int c = sub; c < C4; c += 64) o[c] = z;
        return;
    }
    f32x4 v[NV];
    float sum = 0.f;
#pragma unroll
    for (int t = 0; t < NV; ++t) {
        const int c = sub + LPR * t;
        v[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (live && c < C4) v[t] = xr[c];
        sum += (v[t][0] + v[t][1]) + (v[t][2] + v[t][3]);
    }
    const float mu = (LPR == 64 ? wave_sum(sum) : row16_sum(sum)) / (float)C;
    float sq = 0.f;
#pragma unroll
    for (int t = 0; t < NV; ++t) {
        const int c = sub + LPR * t;
        if (c < C4) {
            const f32x4 d = v[t] - mu;
            sq += (d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3]);
        }
    }
    const float var = (LPR == 64 ? wave_sum(sq) : row16_sum(sq)) / (float)C;
    const float rstd = 1.0f / sqrtf(var + eps);
    if (!live) return;
#pragma unroll
    for (int t = 0; t < NV; ++t) {
        const int c = sub + LPR * t;
        if (c >= C4) continue;
        const f32x4 gg = reinterpret_cast<const f32x4*>(g)[c], bb = reinterpret_cast<const f32x4*>(be)[c];
        f32x4 r;
#pragma unroll
        for (int q = 0; q < 4; ++q) r[q] = x_act((v[t][q] - mu) * rstd * gg[q] + bb[q], act);
        o[c] = r;
    }
}
static const char* xg_layernorm(const float* x, const LnW& w, float eps, float* out, int64_t rows, int C, int act, hipStream_t s,
                                const uint8_t* row_valid = nullptr, int valid_mod = 0) {
    if (rows <= 0) return nullptr;
    const bool al = (C & 3) == 0 && ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(w.g) | reinterpret_cast<uintptr_t>(w.b)) & 15) == 0;
    if (al && C == 64 && !row_valid)
        hipLaunchKernelGGL((xg_layernorm4_kernel<16, 1>), dim3((unsigned)((rows + 15) / 16)), dim3(256), 0, s, x, w.g, w.b, eps, out, rows, C, act, row_valid, valid_mod);
    else if (al && C <= 256)       // the decoder's 256-channel rows: one float4 per lane, a fifth of the registers of the general form
        hipLaunchKernelGGL((xg_layernorm4_kernel<64, 1>), dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, x, w.g, w.b, eps, out, rows, C, act, row_valid, valid_mod);
    else if (al && C <= 1280)
        hipLaunchKernelGGL((xg_layernorm4_kernel<64, 5>), dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, x, w.g, w.b, eps, out, rows, C, act, row_valid, valid_mod);
    else
        hipLaunchKernelGGL(xg_layernorm_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, x, w.g, w.b, eps, out, rows, C, act, row_valid, valid_mod);
    return nullptr;
}

// ------------------------------------------------------------------------------------------------ elementwise
// out[r][c] = x[r][c] + y[(ymod ? r % ymod : r)][c]
__global__ __launch_bounds__(256) void xg_add_kernel(const float* __restrict__ x, const float* __restrict__ y, int64_t ymod, float* __restrict__ out,
                                                    int64_t rows, int C) {
    const int64_t total = rows * C;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / C;
        const int c = (int)(i - r * C);
        out[i] = x[i] + y[(ymod ? r % ymod : r) * C + c];
    }
}
static void xg_add(const float* x, const float* y, int64_t ymod, float* out, int64_t rows, int C, hipStream_t s) {
    if (rows <= 0) return;
    const int64_t total = rows * C;
    hipLaunchKernelGGL(xg_add_kernel, dim3((unsigned)std::min<int64_t>((total + 255) / 256, 65536 * 8)), dim3(256), 0, s, x, y, ymod, out, rows, C);
}
// per-prompt tensors against per-slot tables: out[p][r][c] = act((in ? in[p][r][c] : 0) + tab[((p + off) / div) * stride + r * C + c] + (vec ? vec[c] : 0))
__global__ __launch_bounds__(256) void xg_add_slot_kernel(const float* __restrict__ in, const float* __restrict__ tab, XMap m, const float* __restrict__ vec,
                                                         float* __restrict__ out, int64_t rows_per, int C, int P, int act) {
    const int64_t per = rows_per * C, total = per * P;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int p = (int)(i / per);
        const int64_t j = i - (int64_t)p * per;
        float v = tab[(int64_t)((p + m.off) / m.div) * m.stride + j];
        if (in) v += in[i];
        if (vec) v += vec[j % C];
        out[i] = x_act(v, act);
    }
}
// the same on float4 pieces (C % 4 == 0, 16-byte aligned operands, slot stride a multiple of 4): one 64-bit division per 16 bytes instead of two per float
__global__ __launch_bounds__(256) void xg_add_slot4_kernel(const f32x4* __restrict__ in, const f32x4* __restrict__ tab, XMap m, const f32x4* __restrict__ vec,
                                                          f32x4* __restrict__ out, int64_t per4, int C4, int P, int act) {
    const int64_t total = per4 * P;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int p = (int)(i / per4);
        const int64_t j = i - (int64_t)p * per4;
        f32x4 v = tab[(int64_t)((p + m.off) / m.div) * (m.stride >> 2) + j];
        if (in) v += in[i];
        if (vec) v += vec[(int)(j % C4)];
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = x_act(v[q], act);
        out[i] = v;
    }
}
static void xg_add_slot(const float* in, const float* tab, XMap m, const float* vec, float* out, int64_t rows_per, int C, int P, int act, hipStream_t s) {
    if (P <= 0) return;
    const int64_t total = rows_per * C * P;
    if ((C & 3) == 0 && (m.stride & 3) == 0 &&
        ((reinterpret_cast<uintptr_t>(in) | reinterpret_cast<uintptr_t>(tab) | reinterpret_cast<uintptr_t>(vec) | reinterpret_cast<uintptr_t>(out)) & 15) == 0) {
        hipLaunchKernelGGL(xg_add_slot4_kernel, dim3((unsigned)std::min<int64_t>((total / 4 + 255) / 256, 65536 * 8)), dim3(256), 0, s, reinterpret_cast<const f32x4*>(in),
                           reinterpret_cast<const f32x4*>(tab), m, reinterpret_cast<const f32x4*>(vec), reinterpret_cast<f32x4*>(out), rows_per * C / 4, C / 4, P, act);
        return;
    }
    hipLaunchKernelGGL(xg_add_slot_kernel, dim3((unsigned)std::min<int64_t>((total + 255) / 256, 65536 * 8)), dim3(256), 0, s, in, tab, m, vec, out, rows_per, C, P, act);
}

// ------------------------------------------------------------------------------------------------ attention (fp32)
// out[b][i][h*HD + d] = softmax_j(scale q_i . k_j) v_j for one (batch, head); one thread per query, K / V tiles of 32 keys through LDS.
// qpool: query i is the element-wise maximum of q rows 4i .. 4i+3 (Hiera's 2 x 2 max-pooled queries: a pooling group is 4 consecutive
// rows in the engine's token order).  kmask: one byte per key (shared by every batch), 0 = the key takes no part.
#define XA_TK 32
template <int HD>
__global__ __launch_bounds__(128) void xg_attn_kernel(const float* __restrict__ q, int64_t q_bs, int ldq, const float* __restrict__ k, int64_t k_bs, int ldk,
                                                     const float* __restrict__ v, int64_t v_bs, int ldv, float* __restrict__ o, int64_t o_bs, int ldo,
                                                     int nq, int nk, int qpool, const uint8_t* __restrict__ kmask, float scale) {
    __shared__ __attribute__((aligned(16))) float Ks[XA_TK][HD];
    __shared__ __attribute__((aligned(16))) float Vs[XA_TK][HD];
    const int tid = threadIdx.x;
    // heads vary fastest over the grid: the 8 heads' 64-byte pieces of a q / k / v row share its cache lines while they are in L2
    const int h = blockIdx.x, b = blockIdx.z;
    const int i = blockIdx.y * 128 + tid;
    const bool live = i < nq;
    float qv[HD], acc[HD];
#pragma unroll
    for (int d = 0; d < HD; ++d) { qv[d] = 0.f; acc[d] = 0.f; }
    if (live) {
        const float* qp = q + b * q_bs + h * HD;
        if (qpool) {
#pragma unroll
            for (int d = 0; d < HD; ++d) {
                const float a0 = qp[(int64_t)(4 * i) * ldq + d], a1 = qp[(int64_t)(4 * i + 1) * ldq + d];
                const float a2 = qp[(int64_t)(4 * i + 2) * ldq + d], a3 = qp[(int64_t)(4 * i + 3) * ldq + d];
                qv[d] = fmaxf(fmaxf(a0, a1), fmaxf(a2, a3));
            }
        } else {
#pragma unroll
            for (int d = 0; d < HD; ++d) qv[d] = qp[(int64_t)i * ldq + d];
        }
    }
    float mrun = -INFINITY, lrun = 0.f;
    const float* kb = k + b * k_bs + h * HD;
    const float* vb = v + b * v_bs + h * HD;
    for (int j0 = 0; j0 < nk; j0 += XA_TK) {
        const int tk = min(XA_TK, nk - j0);
        __syncthreads();
        for (int idx = tid; idx < XA_TK * HD; idx += 128) {
            const int j = idx / HD, d = idx - j * HD;
            const bool in = j < tk;
            Ks[j][d] = in ? kb[(int64_t)(j0 + j) * ldk + d] : 0.0f;
            Vs[j][d] = in ? vb[(int64_t)(j0 + j) * ldv + d] : 0.0f;
        }
        __syncthreads();
        float sc[XA_TK];
        float mt = -INFINITY;
#pragma unroll
        for (int j = 0; j < XA_TK; ++j) {
            float a = 0.f;
#pragma unroll
            for (int d = 0; d < HD; ++d) a = fmaf(qv[d], Ks[j][d], a);
            a *= scale;
            const bool ok = j < tk && (!kmask || kmask[j0 + j]);
            sc[j] = ok ? a : -INFINITY;
            mt = fmaxf(mt, sc[j]);
        }
        if (mt == -INFINITY) continue;              // every key of the tile is masked (uniform across the block's live threads or harmless)
        const float mnew = fmaxf(mrun, mt);
        const float corr = expf(mrun - mnew);       // exp(-inf) = 0 on the first tile
        lrun *= corr;
#pragma unroll
        for (int d = 0; d < HD; ++d) acc[d] *= corr;
#pragma unroll
        for (int j = 0; j < XA_TK; ++j) {
            const float pj = expf(sc[j] - mnew);    // masked: exp(-inf) = 0
            lrun += pj;
#pragma unroll
            for (int d = 0; d < HD; ++d) acc[d] = fmaf(pj, Vs[j][d], acc[d]);
        }
        mrun = mnew;
    }
    if (live) {
        const float inv = 1.0f / lrun;
        float* op = o + b * o_bs + (int64_t)i * ldo + h * HD;
#pragma unroll
        for (int d = 0; d < HD; ++d) op[d] = acc[d] * inv;
    }
}
// Few queries against many keys (the decoder's token -> image attention: 7 + n_pts <= 16 queries, 4 096 keys per prompt and head).  The
// one-thread-per-query kernel above puts 8 192 threads on the chip for it; here a workgroup of 256 threads owns one (prompt, head), every thread
// takes the keys j = tid, tid + 256, ... against ALL the queries with its own online softmax (running maximum, sum, HD accumulators per
// query), and the partial results are merged at the end (rescaled to the common maximum; wave reduction, then the 4 waves through LDS).
// Same products and fp32 accumulation as the kernel above; only the order of the softmax sums over keys differs (~1e-7 relative).
template <int HD, int TQ>
__global__ __launch_bounds__(256) void xg_attn_fewq_kernel(const float* __restrict__ q, int64_t q_bs, int ldq, const float* __restrict__ k, int64_t k_bs, int ldk,
                                                          const float* __restrict__ v, int64_t v_bs, int ldv, float* __restrict__ o, int64_t o_bs, int ldo,
                                                          int nq, int nk, float scale) {
    __shared__ float qs[TQ][HD];
    __shared__ float part[4][TQ][HD + 2];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = blockIdx.x, b = blockIdx.y;           // heads fastest (see xg_attn_kernel)
    const float* kb = k + b * k_bs + h * HD;
    const float* vb = v + b * v_bs + h * HD;
    for (int q0 = 0; q0 < nq; q0 += TQ) {
        const int tq = min(TQ, nq - q0);
        __syncthreads();
        for (int idx = tid; idx < TQ * HD; idx += 256) {
            const int t = idx / HD, d = idx - t * HD;
            qs[t][d] = t < tq ? q[b * q_bs + (int64_t)(q0 + t) * ldq + h * HD + d] : 0.0f;
        }
        __syncthreads();
        float m[TQ], l[TQ], acc[TQ][HD];
#pragma unroll
        for (int t = 0; t < TQ; ++t) {
            m[t] = -INFINITY; l[t] = 0.f;
#pragma unroll
            for (int d = 0; d < HD; ++d) acc[t][d] = 0.f;
        }
        for (int j0 = tid; j0 < nk; j0 += 256 * 4) {
            float sc[4][TQ];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int j = j0 + 256 * c;
                if (j < nk) {
                    float kr[HD];
                    const float4* kp = reinterpret_cast<const float4*>(kb + (int64_t)j * ldk);
#pragma unroll
                    for (int d4 = 0; d4 < HD / 4; ++d4) { const float4 x = kp[d4]; kr[4 * d4] = x.x; kr[4 * d4 + 1] = x.y; kr[4 * d4 + 2] = x.z; kr[4 * d4 + 3] = x.w; }
#pragma unroll
                    for (int t = 0; t < TQ; ++t) {
                        float a = 0.f;
#pragma unroll
                        for (int d = 0; d < HD; ++d) a = fmaf(qs[t][d], kr[d], a);
                        sc[c][t] = a * scale;
                    }
                } else {
#pragma unroll
                    for (int t = 0; t < TQ; ++t) sc[c][t] = -INFINITY;
                }
            }
#pragma unroll
            for (int t = 0; t < TQ; ++t) {
                const float mt = fmaxf(fmaxf(sc[0][t], sc[1][t]), fmaxf(sc[2][t], sc[3][t]));      // finite: key j0 exists
                const float mnew = fmaxf(m[t], mt);
                const float corr = expf(m[t] - mnew);                                              // exp(-inf) = 0 on the first chunk
                l[t] *= corr;
#pragma unroll
                for (int d = 0; d < HD; ++d) acc[t][d] *= corr;
                m[t] = mnew;
            }
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int j = j0 + 256 * c;
                if (j >= nk) continue;
                float vr[HD];
                const float4* vp = reinterpret_cast<const float4*>(vb + (int64_t)j * ldv);
#pragma unroll
                for (int d4 = 0; d4 < HD / 4; ++d4) { const float4 x = vp[d4]; vr[4 * d4] = x.x; vr[4 * d4 + 1] = x.y; vr[4 * d4 + 2] = x.z; vr[4 * d4 + 3] = x.w; }
#pragma unroll
                for (int t = 0; t < TQ; ++t) {
                    const float pj = expf(sc[c][t] - m[t]);
                    l[t] += pj;
#pragma unroll
                    for (int d = 0; d < HD; ++d) acc[t][d] = fmaf(pj, vr[d], acc[t][d]);
                }
            }
        }
        // merge: the wave's threads to their common maximum, then the four waves
#pragma unroll
        for (int t = 0; t < TQ; ++t) {
            const float M = wave_max(m[t]);
            const float f = (m[t] == -INFINITY) ? 0.0f : expf(m[t] - M);
            const float L = wave_sum(l[t] * f);
            if (lane == 0) { part[wave][t][HD] = M; part[wave][t][HD + 1] = L; }
#pragma unroll
            for (int d = 0; d < HD; ++d) {
                const float a = wave_sum(acc[t][d] * f);
                if (lane == 0) part[wave][t][d] = a;
            }
        }
        __syncthreads();
        if (tid < TQ * HD) {
            const int t = tid / HD, d = tid - t * HD;
            if (t < tq) {
                float M = -INFINITY;
#pragma unroll
                for (int w = 0; w < 4; ++w) M = fmaxf(M, part[w][t][HD]);
                float L = 0.f, A = 0.f;
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                    const float mw = part[w][t][HD];
                    const float f = (mw == -INFINITY) ? 0.0f : expf(mw - M);
                    L += part[w][t][HD + 1] * f;
                    A += part[w][t][d] * f;
                }
                const float inv = 1.0f / L;
                o[b * o_bs + (int64_t)(q0 + t) * ldo + h * HD + d] = A * inv;
            }
        }
    }
}
// fp32 attention on the matrix cores (round 4) for the encoder's head dimensions (56 / 72 / 96): v_mfma_f32_16x16x4_f32 for both products, so
// every product and sum is still fp32 (the one-thread-per-query kernel above ran the same arithmetic on the vector ALU at a third of the rate).
// A wave owns 16 queries, a workgroup of NW waves shares 32-key K / V tiles through LDS.  Transposed formulation (as the bf16 kernels):
//   S^T[key][q] = sum_d K[key][d] Q[q][d]   A = K fragment (row = key, k = 4 step + g), B = Q fragment (k, col = q), 16-key tiles;
//   the C layout leaves lane (q = lane & 15, g = lane >> 4) with keys 4 g + r, r = 0..3, of the tile: softmax per q over its 4 registers and the
//   4 lanes q + 16 g; the same registers ARE the B operand of O^T[d][q] += sum_key V[key][d] P[key][q] when MFMA r takes key 4 g + r as its
//   k index g, i.e. the V fragment of lane (d, g) is V[4 g + r][d]: no transposition of P anywhere.
// qpool / kmask / scale as in xg_attn_kernel; the softmax sums differ from it only in their order.
template <int HD>
__global__ __launch_bounds__(512) void xg_attn_mfma_kernel(const float* __restrict__ q, int64_t q_bs, int ldq, const float* __restrict__ k, int64_t k_bs, int ldk,
                                                          const float* __restrict__ v, int64_t v_bs, int ldv, float* __restrict__ o, int64_t o_bs, int ldo,
                                                          int nq, int nk, int qpool, const uint8_t* __restrict__ kmask, float scale) {
    constexpr int KT = 32, LDK = HD + 4, NS = HD / 4, NDT = (HD + 15) / 16;
    __shared__ __attribute__((aligned(16))) float Ks[KT * LDK];
    __shared__ __attribute__((aligned(16))) float Vs[KT * LDK];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nthr = blockDim.x;
    const int fi = lane & 15, fg = lane >> 4;
    const int h = blockIdx.x, b = blockIdx.z;
    const int i = blockIdx.y * (16 * (nthr >> 6)) + wave * 16 + fi;          // this lane's query
    const bool live = i < nq;
    // Q fragments: Q[i][4 step + fg]
    float qf[NS];
    {
        const float* qp = q + b * q_bs + h * HD;
#pragma unroll
        for (int st = 0; st < NS; ++st) {
            const int d = 4 * st + fg;
            float x = 0.f;
            if (live) {
                if (qpool) {
                    const float a0 = qp[(int64_t)(4 * i) * ldq + d], a1 = qp[(int64_t)(4 * i + 1) * ldq + d];
                    const float a2 = qp[(int64_t)(4 * i + 2) * ldq + d], a3 = qp[(int64_t)(4 * i + 3) * ldq + d];
                    x = fmaxf(fmaxf(a0, a1), fmaxf(a2, a3));
                } else x = qp[(int64_t)i * ldq + d];
            }
            qf[st] = x;
        }
    }
    f32x4 oacc[NDT];
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt) oacc[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float mrun = -INFINITY, lrun = 0.f;      // lrun: this lane's share of the row sum (its 4 g-lanes are added at the end)
    const float* kb = k + b * k_bs + h * HD;
    const float* vb = v + b * v_bs + h * HD;
    for (int j0 = 0; j0 < nk; j0 += KT) {
        __syncthreads();
        for (int idx = tid; idx < KT * NS; idx += nthr) {
            const int j = idx / NS, c = idx - j * NS;
            f32x4 kv = {0.f, 0.f, 0.f, 0.f}, vv = {0.f, 0.f, 0.f, 0.f};
            if (j0 + j < nk) {
                kv = *reinterpret_cast<const f32x4*>(kb + (int64_t)(j0 + j) * ldk + 4 * c);
                vv = *reinterpret_cast<const f32x4*>(vb + (int64_t)(j0 + j) * ldv + 4 * c);
            }
            *reinterpret_cast<f32x4*>(&Ks[j * LDK + 4 * c]) = kv;
            *reinterpret_cast<f32x4*>(&Vs[j * LDK + 4 * c]) = vv;
        }
        __syncthreads();
        // scores of the two 16-key sub-tiles
        f32x4 sc[2];
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
            f32x4 a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int st = 0; st < NS; ++st) a = __builtin_amdgcn_mfma_f32_16x16x4f32(Ks[(16 * sub + fi) * LDK + 4 * st + fg], qf[st], a, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int j = j0 + 16 * sub + 4 * fg + r;
                const bool ok = j < nk && (!kmask || kmask[j]);
                a[r] = ok ? a[r] * scale : -INFINITY;
            }
            sc[sub] = a;
        }
        float mt = fmaxf(fmaxf(fmaxf(sc[0][0], sc[0][1]), fmaxf(sc[0][2], sc[0][3])), fmaxf(fmaxf(sc[1][0], sc[1][1]), fmaxf(sc[1][2], sc[1][3])));
        mt = xor32_max(xor16_max(mt));                      // over the 4 lanes of this query
        if (__builtin_amdgcn_ballot_w64(mt != -INFINITY) == 0) continue;      // wave-uniform: every key of the tile is masked (window padding)
        // no divergence around the MFMAs: a query that has not met a live key yet keeps the reference 0 (exp(-inf) = 0 everywhere)
        const float mnew = fmaxf(mrun, mt);
        const float mref = mnew == -INFINITY ? 0.f : mnew;
        const float corr = expf(mrun - mref);
        lrun *= corr;
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt) oacc[dt] *= corr;
        mrun = mnew;
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
            float pr[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) { pr[r] = expf(sc[sub][r] - mref); lrun += pr[r]; }
#pragma unroll
            for (int dt = 0; dt < NDT; ++dt) {
                const int d = dt * 16 + fi;
                const int dc = d < HD ? d : 0;             // the half tile of HD = 56 / 72: rows d >= HD multiply zeros
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float vf = Vs[(16 * sub + 4 * fg + r) * LDK + dc];
                    if (d >= HD) vf = 0.f;
                    oacc[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(vf, pr[r], oacc[dt], 0, 0, 0);
                }
            }
        }
    }
    const float ltot = xor32_sum(xor16_sum(lrun));
    if (live) {
        const float inv = 1.0f / ltot;
        float* op = o + b * o_bs + (int64_t)i * ldo + h * HD;
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt) {
            const int d0 = dt * 16 + 4 * fg;              // this lane's 4 consecutive output channels of tile dt
            if (d0 < HD) *reinterpret_cast<f32x4*>(op + d0) = oacc[dt] * inv;
        }
    }
}
// The same formulation for the decoder's token -> image attention (head dimension 16, <= 16 queries, thousands of keys), one WAVE per
// (prompt, head, key segment): S^T = K Q^T needs 4 MFMAs per 16 keys (the lane's float4 of a key row feeds MFMA e with channel 4 g + e - any
// permutation of the contraction index serves as long as Q uses it too), the second product one MFMA per 4 keys
// (transposed: O^T = V^T P^T, k index g <-> key 4 g + r: P's registers are the B operand as they stand, V rows are read coalesced as the A operand).  K and V go straight from global memory into operand
// registers; NSEG waves of a workgroup split the keys and merge their (max, sum, O) through LDS.
template <int NSEG>
__global__ __launch_bounds__(64 * NSEG) void xg_attn_fewq_mfma_kernel(const float* __restrict__ q, int64_t q_bs, int ldq, const float* __restrict__ k, int64_t k_bs, int ldk,
                                                                     const float* __restrict__ v, int64_t v_bs, int ldv, float* __restrict__ o, int64_t o_bs, int ldo,
                                                                     int nq, int nk, float scale) {
    constexpr int HD = 16;
    __shared__ float part[NSEG][16][HD + 2];       // per wave: O[q][d], max, sum
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fi = lane & 15, fg = lane >> 4;
    const int h = blockIdx.x, b = blockIdx.y;
    // B operand of S^T: Q[q = fi][4 fg + e]
    f32x4 qv = {0.f, 0.f, 0.f, 0.f};
    if (fi < nq) qv = *reinterpret_cast<const f32x4*>(q + b * q_bs + (int64_t)fi * ldq + h * HD + 4 * fg);
    const float* kb = k + b * k_bs + h * HD;
    const float* vb = v + b * v_bs + h * HD;
    f32x4 oacc = {0.f, 0.f, 0.f, 0.f};             // O^T[d = 4 fg + r][q = fi]
    float mrun = -INFINITY, lrun = 0.f;            // of query fi (this lane's share of the sum)
    const int per = ((nk + 16 * NSEG - 1) / (16 * NSEG)) * 16;
    const int j_lo = wave * per, j_hi = min(nk, j_lo + per);
    for (int j0 = j_lo; j0 < j_hi; j0 += 16) {
        const int jk = j0 + fi;
        f32x4 kv = {0.f, 0.f, 0.f, 0.f};
        if (jk < j_hi) kv = *reinterpret_cast<const f32x4*>(kb + (int64_t)jk * ldk + 4 * fg);
        float vv[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int jv = j0 + 4 * fg + r;
            vv[r] = jv < j_hi ? vb[(int64_t)jv * ldv + fi] : 0.f;
        }
        f32x4 a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int e = 0; e < 4; ++e) a = __builtin_amdgcn_mfma_f32_16x16x4f32(kv[e], qv[e], a, 0, 0, 0);      // S^T[key 4 fg + r][q fi]
        float mt = -INFINITY;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            a[r] = (j0 + 4 * fg + r < j_hi) ? a[r] * scale : -INFINITY;
            mt = fmaxf(mt, a[r]);
        }
        mt = xor32_max(xor16_max(mt));
        const float mnew = fmaxf(mrun, mt);        // finite: key j0 exists
        const float corr = expf(mrun - mnew);
        lrun *= corr;
        oacc *= corr;
        mrun = mnew;
        float pr[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) { pr[r] = expf(a[r] - mnew); lrun += pr[r]; }
#pragma unroll
        for (int r = 0; r < 4; ++r) oacc = __builtin_amdgcn_mfma_f32_16x16x4f32(vv[r], pr[r], oacc, 0, 0, 0);      // O^T: A = V^T[d fi][key 4 fg + r], B = P^T[key 4 fg + r][q fi]
    }
    const float ltot = xor32_sum(xor16_sum(lrun));
    // merge the NSEG key segments: every wave leaves O^T (channels 4 fg + r of query fi), and per query its max / sum
#pragma unroll
    for (int r = 0; r < 4; ++r) part[wave][fi][4 * fg + r] = oacc[r];
    if (fg == 0) { part[wave][fi][HD] = mrun; part[wave][fi][HD + 1] = ltot; }
    __syncthreads();
    if (tid < 16 * HD) {
        const int t = tid / HD, d = tid - t * HD;
        if (t < nq) {
            float M = -INFINITY;
#pragma unroll
            for (int w = 0; w < NSEG; ++w) M = fmaxf(M, part[w][t][HD]);
            float L = 0.f, A = 0.f;
#pragma unroll
            for (int w = 0; w < NSEG; ++w) {
                const float mw = part[w][t][HD];
                const float f = (mw == -INFINITY) ? 0.0f : expf(mw - M);
                L += part[w][t][HD + 1] * f;
                A += part[w][t][d] * f;
            }
            o[b * o_bs + (int64_t)t * ldo + h * HD + d] = A * (1.0f / L);
        }
    }
}
// Many queries against a handful of keys (the decoder's image -> token attention: 4 096 queries, 7 + n_pts <= 16 keys, 8 heads of 16): the
// eight heads of a query sit in eight neighbouring lanes, so a wave reads and writes whole 512-byte rows of q / out (the general kernel runs
// one head per workgroup: 64-byte pieces of every row, each cache line fetched twice).  Same arithmetic, in the same order, as xg_attn_kernel
// with its single key tile.
__global__ __launch_bounds__(256) void xg_attn_fewk_kernel(const float* __restrict__ q, int64_t q_bs, int ldq, const float* __restrict__ k, int64_t k_bs, int ldk,
                                                          const float* __restrict__ v, int64_t v_bs, int ldv, float* __restrict__ o, int64_t o_bs, int ldo,
                                                          int nq, int nk, float scale) {
    constexpr int HD = 16, NH = 8, MAXK = 16;
    __shared__ __attribute__((aligned(16))) float Ks[MAXK][NH * HD];
    __shared__ __attribute__((aligned(16))) float Vs[MAXK][NH * HD];
    const int tid = threadIdx.x, b = blockIdx.y;
    for (int idx = tid; idx < nk * NH * HD; idx += 256) {
        const int j = idx / (NH * HD), c = idx - j * (NH * HD);
        Ks[j][c] = k[b * k_bs + (int64_t)j * ldk + c];
        Vs[j][c] = v[b * v_bs + (int64_t)j * ldv + c];
    }
    __syncthreads();
    const int h = tid & 7, i = blockIdx.x * 32 + (tid >> 3);
    if (i >= nq) return;
    float qv[HD], acc[HD];
    {
        const f32x4* qp = reinterpret_cast<const f32x4*>(q + b * q_bs + (int64_t)i * ldq + h * HD);
#pragma unroll
        for (int c = 0; c < 4; ++c) { const f32x4 t = qp[c]; qv[4 * c] = t[0]; qv[4 * c + 1] = t[1]; qv[4 * c + 2] = t[2]; qv[4 * c + 3] = t[3]; }
    }
    float sc[MAXK];
    float mt = -INFINITY;
#pragma unroll
    for (int j = 0; j < MAXK; ++j) {
        float a = 0.f;
        if (j < nk) {
#pragma unroll
            for (int d = 0; d < HD; ++d) a = fmaf(qv[d], Ks[j][h * HD + d], a);
            a *= scale;
        }
        sc[j] = j < nk ? a : -INFINITY;
        mt = fmaxf(mt, sc[j]);
    }
    float l = 0.f;
#pragma unroll
    for (int d = 0; d < HD; ++d) acc[d] = 0.f;
#pragma unroll
    for (int j = 0; j < MAXK; ++j) {
        if (j < nk) {
            const float pj = expf(sc[j] - mt);
            l += pj;
#pragma unroll
            for (int d = 0; d < HD; ++d) acc[d] = fmaf(pj, Vs[j][h * HD + d], acc[d]);
        }
    }
    const float inv = 1.0f / l;
    f32x4* op = reinterpret_cast<f32x4*>(o + b * o_bs + (int64_t)i * ldo + h * HD);
#pragma unroll
    for (int c = 0; c < 4; ++c) op[c] = (f32x4){acc[4 * c] * inv, acc[4 * c + 1] * inv, acc[4 * c + 2] * inv, acc[4 * c + 3] * inv};
}
static const char* xg_attn(int hd, const float* q, int64_t q_bs, int ldq, const float* k, int64_t k_bs, int ldk, const float* v, int64_t v_bs, int ldv,
                           float* o, int64_t o_bs, int ldo, int nq, int nk, int batch, int heads, int qpool, const uint8_t* kmask, float scale, hipStream_t s) {
    if (batch <= 0 || nq <= 0) return nullptr;
    if (hd == 16 && nq <= 16 && nk >= 1024 && !qpool && !kmask && batch <= 65535 && ((ldk | ldv) & 3) == 0 && (((k_bs | v_bs) & 3) == 0) &&
        ((reinterpret_cast<uintptr_t>(k) | reinterpret_cast<uintptr_t>(v)) & 15) == 0) {
        static const bool fq_valu = getenv("SABER_AMD_XG_FEWQ_VALU") != nullptr;      // development A/B: the vector-ALU form
        if (fq_valu || ((ldq | q_bs) & 3) || (reinterpret_cast<uintptr_t>(q) & 15))
            hipLaunchKernelGGL((xg_attn_fewq_kernel<16, 8>), dim3(heads, batch), dim3(256), 0, s, q, q_bs, ldq, k, k_bs, ldk, v, v_bs, ldv, o, o_bs, ldo, nq, nk, scale);
        else
            hipLaunchKernelGGL((xg_attn_fewq_mfma_kernel<4>), dim3(heads, batch), dim3(256), 0, s, q, q_bs, ldq, k, k_bs, ldk, v, v_bs, ldv, o, o_bs, ldo, nq, nk, scale);
        return nullptr;
    }
    if (hd == 16 && heads == 8 && nk <= 16 && nq >= 256 && !qpool && !kmask && batch <= 65535 && ((ldq | ldo) & 3) == 0 && ((q_bs | o_bs) & 3) == 0 &&
        ((reinterpret_cast<uintptr_t>(q) | reinterpret_cast<uintptr_t>(o)) & 15) == 0) {
        hipLaunchKernelGGL(xg_attn_fewk_kernel, dim3((nq + 31) / 32, batch), dim3(256), 0, s, q, q_bs, ldq, k, k_bs, ldk, v, v_bs, ldv, o, o_bs, ldo, nq, nk, scale);
        return nullptr;
    }
    if ((hd == 56 || hd == 72 || hd == 96) && ((ldk | ldv | ldo) & 3) == 0 && ((k_bs | v_bs | o_bs) & 3) == 0 && batch <= 65535 &&
        ((reinterpret_cast<uintptr_t>(k) | reinterpret_cast<uintptr_t>(v) | reinterpret_cast<uintptr_t>(o)) & 15) == 0) {
        const int nw = std::min(8, (nq + 15) / 16);
        const dim3 g3(heads, (nq + 16 * nw - 1) / (16 * nw), batch), blk(64 * nw);
        if (hd == 56) hipLaunchKernelGGL(xg_attn_mfma_kernel<56>, g3, blk, 0, s, q, q_bs, ldq, k, k_bs, ldk, v, v_bs, ldv, o, o_bs, ldo, nq, nk, qpool, kmask, scale);
        else if (hd == 72) hipLaunchKernelGGL(xg_attn_mfma_kernel<72>, g3, blk, 0, s, q, q_bs, ldq, k, k_bs, ldk, v, v_bs, ldv, o, o_bs, ldo, nq, nk, qpool, kmask, scale);
        else hipLaunchKernelGGL(xg_attn_mfma_kernel<96>, g3, blk, 0, s, q, q_bs, ldq, k, k_bs, ldk, v, v_bs, ldv, o, o_bs, ldo, nq, nk, qpool, kmask, scale);
        return nullptr;
    }
    const dim3 block(128);
    if ((nq + 127) / 128 > 65535) return "exact attention: too many queries";
    // gridDim.z <= 65535
    for (int b0 = 0; b0 < batch; b0 += 65535) {
        const int nb = std::min(65535, batch - b0);
        dim3 g2(heads, (nq + 127) / 128, nb);
#define XA_LAUNCH(HDIM) hipLaunchKernelGGL(xg_attn_kernel<HDIM>, g2, block, 0, s, q + b0 * q_bs, q_bs, ldq, k + b0 * k_bs, k_bs, ldk, v + b0 * v_bs, v_bs, ldv, \
                                          o + b0 * o_bs, o_bs, ldo, nq, nk, qpool, kmask, scale)
        switch (hd) {
            case 16: XA_LAUNCH(16); break;
            case 32: XA_LAUNCH(32); break;
            case 56: XA_LAUNCH(56); break;
            case 72: XA_LAUNCH(72); break;
            case 96: XA_LAUNCH(96); break;
            default: return "exact attention: unsupported head dimension";
        }
#undef XA_LAUNCH
    }
    return nullptr;
}

// ------------------------------------------------------------------------------------------------ mask prompt embedding (first two stages)
// h2[p][tok][16] = GELU(LN2d(conv k2s2 4->16 (GELU(LN2d(conv k2s2 1->4 (mask))))))  for the 4 x 4 logit patch of token tok; the 1 x 1 conv to
// 256 channels that follows is a GEMM.  One thread per (prompt, token).
__global__ __launch_bounds__(256) void xg_mask_hidden_kernel(const float* __restrict__ mask_in, int P, MaskEmbedWeights w, float clamp_abs, int raw4_q0,
                                                            float* __restrict__ h2out) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (int64_t)P * 4096) return;
    const int p = (int)(idx >> 12), tok = (int)(idx & 4095);
    int ty, tx;
    perm_coords(tok, 2, &ty, &tx);
    const int64_t plane = raw4_q0 >= 0 ? (int64_t)(raw4_q0 + p) + (raw4_q0 + p) / 3 + 1 : (int64_t)p;
    const float* mp = mask_in + plane * 65536 + (int64_t)(ty * 4) * 256 + tx * 4;
    float h1[4][4];       // [position ky*2+kx of the second conv][channel]
#pragma unroll
    for (int pos = 0; pos < 4; ++pos) {
        const int py = pos >> 1, px = pos & 1;
        float in[4];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            float t = mp[(py * 2 + (kk >> 1)) * 256 + px * 2 + (kk & 1)];
            if (clamp_abs > 0.0f) t = fminf(fmaxf(t, -clamp_abs), clamp_abs);
            in[kk] = t;
        }
        float vv[4], mu = 0.f;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float a = w.b1[c];
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) a += w.w1[c * 4 + kk] * in[kk];
            vv[c] = a; mu += a;
        }
        mu *= 0.25f;
        float var = 0.f;
#pragma unroll
        for (int c = 0; c < 4; ++c) var += (vv[c] - mu) * (vv[c] - mu);
        const float rstd = 1.0f / sqrtf(var * 0.25f + 1e-6f);
#pragma unroll
        for (int c = 0; c < 4; ++c) h1[pos][c] = x_gelu((vv[c] - mu) * rstd * w.g1[c] + w.be1[c]);
    }
    float h2[16], mu = 0.f;
#pragma unroll
    for (int c = 0; c < 16; ++c) {
        float a = w.b2[c];
#pragma unroll
        for (int ci = 0; ci < 4; ++ci)
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) a += w.w2[(c * 4 + ci) * 4 + kk] * h1[kk][ci];
        h2[c] = a; mu += a;
    }
    mu *= (1.0f / 16.0f);
    float var = 0.f;
#pragma unroll
    for (int c = 0; c < 16; ++c) var += (h2[c] - mu) * (h2[c] - mu);
    const float rstd = 1.0f / sqrtf(var * (1.0f / 16.0f) + 1e-6f);
    float* o = h2out + idx * 16;
#pragma unroll
    for (int c = 0; c < 16; ++c) o[c] = x_gelu((h2[c] - mu) * rstd * w.g2[c] + w.be2[c]);
}

// masks[p][k][y][x] = sum_c hyper[p][k][c] * up[p][perm(y,x)][c]   (up: fp32 [P][65536][32], engine token order; hyper: [P][128])
__global__ __launch_bounds__(256) void xg_mask_dot_kernel(const float* __restrict__ up, const float* __restrict__ hyper, int P, float* __restrict__ masks4) {
    // A workgroup takes 256 consecutive tokens = four horizontally adjacent 8 x 8 blocks of the engine's token order = an 8 x 32 pixel patch: their
    // 32 KB of `up` are read as one contiguous stream into LDS (rows padded to 33 floats), then thread (ry, rx) of the patch reads its token's row
    // and writes 128-byte rows of the four mask planes.  (Round 3: one thread per pixel reading its 128-byte row straight from memory.)
    __shared__ float us[256 * 33];
    const int64_t t0 = (int64_t)blockIdx.x * 256;              // first token of the patch, over all prompts
    if (t0 >= (int64_t)P * 65536) return;
    const int tid = threadIdx.x;
    const f32x4* src = reinterpret_cast<const f32x4*>(up + t0 * 32);
#pragma unroll
    for (int it = 0; it < 8; ++it) {
        const int idx = it * 256 + tid;                        // float4 index in the 32-KB stream: token idx >> 3, channels 4 (idx & 7) ..
        const f32x4 vq = src[idx];
        float* d = &us[(idx >> 3) * 33 + 4 * (idx & 7)];
        d[0] = vq[0]; d[1] = vq[1]; d[2] = vq[2]; d[3] = vq[3];
    }
    __syncthreads();
    const int p = (int)(t0 >> 16), tl0 = (int)(t0 & 65535);
    int y0, x0;
    perm_coords256(tl0, &y0, &x0);                             // top-left pixel of the patch (tl0 is a multiple of 256)
    const int ry = tid >> 5, rx = tid & 31;
    const int tl = perm_index256(y0 + ry, x0 + rx) - tl0;      // this pixel's token within the patch
    const float* uv = &us[tl * 33];
    const float* hp = hyper + (int64_t)p * 128;
    const int pix = (y0 + ry) * 256 + x0 + rx;
#pragma unroll
    for (int kq = 0; kq < 4; ++kq) {
        float a = 0.f;
#pragma unroll
        for (int c = 0; c < 32; ++c) a = fmaf(hp[kq * 32 + c], uv[c], a);
        masks4[((int64_t)p * 4 + kq) * 65536 + pix] = a;
    }
}

// ------------------------------------------------------------------------------------------------ workspaces
struct ExactWs {
    // encoder (the n <= max_images crops of a pass as ONE batch of n x tokens rows)
    float *xn = nullptr, *qkv = nullptr, *att = nullptr, *hid = nullptr, *sb[4] = {nullptr, nullptr, nullptr, nullptr}, *lat3 = nullptr;
    // decoder (chunks of pc prompts)
    int pc = 0;
    float *keys = nullptr, *kpe = nullptr, *p0 = nullptr, *p1 = nullptr, *patt = nullptr, *up1 = nullptr, *up2 = nullptr, *h2 = nullptr;
    float *t0 = nullptr, *t1 = nullptr, *tq = nullptr, *tk = nullptr, *tv = nullptr, *ta = nullptr, *thid = nullptr, *hd0 = nullptr, *hd1 = nullptr;
};
static ExactWs* ws_of(saber_engine* e) { return reinterpret_cast<ExactWs*>(e->exact_ws); }

// 512 prompts per pass (round 3: 128): ~14 GB of fp32 workspaces on a handle with max_prompts >= 512, a quarter of the launches, and the
// token-side GEMMs (8 rows per prompt) get 4 096 rows instead of 1 024
int exact_chunk_prompts(const saber_engine* e) { return std::min(e->max_prompts, 512); }

static int ensure_ws_alloc(saber_engine* e, ExactWs* w);
//  the handle learns about the workspaces only when EVERY allocation has succeeded: these are ~14 GB next to a 58-GB production
// handle, so a hipMalloc failure half way is plausible, and a non-null exact_ws with null members would send the next exact call into
// kernels on nullptr buffers.  On failure what was allocated is released again.
static int ensure_ws(saber_engine* e) {
    if (e->exact_ws) return SABER_OK;
    ExactWs* w = new ExactWs();
    const int st = ensure_ws_alloc(e, w);
    if (st != SABER_OK) {
        void* bufs[] = {w->xn, w->qkv, w->att, w->hid, w->sb[0], w->sb[1], w->sb[2], w->sb[3], w->lat3, w->keys, w->kpe, w->p0, w->p1, w->patt, w->up1, w->up2, w->h2,
                        w->t0, w->t1, w->tq, w->tk, w->tv, w->ta, w->thid, w->hd0, w->hd1};
        for (void* b : bufs) if (b) eng_free(e, b);
        delete w;
        return st;
    }
    e->exact_ws = w;
    return SABER_OK;
}
static int ensure_ws_alloc(saber_engine* e, ExactWs* w) {
    const size_t C0 = (size_t)e->embed_dim, NI = (size_t)e->max_images;      // the whole pass of up to max_images crops at once (round 4)
    TRY(eng_alloc(e, &w->xn, NI * 65536 * C0));
    TRY(eng_alloc(e, &w->qkv, NI * 65536 * 6 * C0));
    TRY(eng_alloc(e, &w->att, NI * 65536 * C0));
    TRY(eng_alloc(e, &w->hid, NI * 65536 * 4 * C0));
    for (int s = 0; s < 4; ++s) TRY(eng_alloc(e, &w->sb[s], NI * (size_t)e->tok_rows[s] * (C0 << s)));
    TRY(eng_alloc(e, &w->lat3, NI * (size_t)e->tok_rows[3] * 256));
    const size_t P = w->pc = exact_chunk_prompts(e);
    TRY(eng_alloc(e, &w->keys, P * 4096 * 256));
    TRY(eng_alloc(e, &w->kpe, P * 4096 * 256));
    TRY(eng_alloc(e, &w->p0, P * 4096 * 128));
    TRY(eng_alloc(e, &w->p1, P * 4096 * 128));
    TRY(eng_alloc(e, &w->patt, P * 4096 * 128));
    TRY(eng_alloc(e, &w->up1, P * 16384 * 64));
    TRY(eng_alloc(e, &w->up2, P * 65536 * 32));
    TRY(eng_alloc(e, &w->h2, P * 4096 * 16));
    TRY(eng_alloc(e, &w->t0, P * 8 * 256));
    TRY(eng_alloc(e, &w->t1, P * 8 * 256));
    TRY(eng_alloc(e, &w->tq, P * 8 * 256));
    TRY(eng_alloc(e, &w->tk, P * 8 * 256));
    TRY(eng_alloc(e, &w->tv, P * 8 * 256));
    TRY(eng_alloc(e, &w->ta, P * 8 * 256));
    TRY(eng_alloc(e, &w->thid, P * 8 * 2048));
    TRY(eng_alloc(e, &w->hd0, 4 * P * 256));
    TRY(eng_alloc(e, &w->hd1, 4 * P * 256));
    return SABER_OK;
}
void exact_release(saber_engine* e) {
    delete ws_of(e);        // the device buffers are in e->allocs
    e->exact_ws = nullptr;
}

static XGemm mkx(const float* A, int64_t lda, int M, const LinW& w) {
    XGemm p;
    p.A = A; p.lda = lda; p.W = w.wf; p.ldw = w.in; p.bias = w.b; p.M = M; p.N = w.out; p.K = w.in;
    return p;
}
#define XK(call) do { const char* _m = (call); if (_m) return eng_fail(e, SABER_ERR_INVALID, _m); } while (0)

// ------------------------------------------------------------------------------------------------ encoder
// Continues eng_encode after the (fp32) resize + normalise and patch embedding: e->xa holds n images x 65536 tokens x C0.
int exact_encode_blocks(saber_engine* e, int n, int slot0, hipStream_t s) {
    TRY(ensure_ws(e));
    ExactWs* w = ws_of(e);
    for (const BlockW& bw : e->bw) if (!bw.qkv.wf) return eng_fail(e, SABER_ERR_STATE, "exact mode: the engine was finalized without fp32 weights (set the precision before finalize)");
    const int C0 = e->embed_dim;
    const size_t nblocks = e->blocks.size();
    const float hscale = 1.0f / sqrtf((float)e->head_dim);
    if (n > e->max_images) return eng_fail(e, SABER_ERR_INVALID, "exact encode: more images than max_images");
    // Every image of the pass goes through each block together: the rows of the GEMMs, LayerNorms and attention batches are n x (tokens of one
    // image).  Row r of image i computes exactly what it computed alone (no kernel mixes rows of different windows / images); round 3 looped
    // over the images, which left the small-M GEMMs and the global attention of stages 3 / 4 with one workgroup per CU.
    {
        float* x = e->xa;
        float* xalt = e->xb;
        int tokens = 65536, stage = 0;                        // tokens of ONE image; N below = rows of the batch
        XK(xg_layernorm(x, e->bw[0].n1, 1e-6f, w->xn, (int64_t)n * tokens, e->blocks[0].din, ACT_NONE, s, e->valid[0], tokens));
        for (size_t i = 0; i < nblocks; ++i) {
            const BlockSpec& bs = e->blocks[i];
            const BlockW& b = e->bw[i];
            const int N = n * tokens;
            float* xres = x;
            int Nq = N;
            if (bs.din != bs.dout) {     // shortcut = 2 x 2 max-pool of proj(norm1(x))
                XGemm g = mkx(w->xn, bs.din, N, b.sc);
                g.C = xalt; g.ldc = bs.dout; g.pool4 = 1;
                XK(xg_gemm(g, s));
                xres = xalt; Nq = N / 4;
            }
            {
                XGemm g = mkx(w->xn, bs.din, N, b.qkv);
                g.C = w->qkv; g.ldc = 3 * bs.dout;
                XK(xg_gemm(g, s));
            }
            const int nk = bs.window > 0 ? bs.window * bs.window : tokens;
            const uint8_t* kmask = (bs.window == 0 && e->valid[stage]) ? e->kmask2 : nullptr;
            const int C = bs.dout, qp = bs.q_stride > 1;
            XK(xg_attn(e->head_dim, w->qkv, (int64_t)nk * 3 * C, 3 * C, w->qkv + C, (int64_t)nk * 3 * C, 3 * C, w->qkv + 2 * C, (int64_t)nk * 3 * C, 3 * C,
                       w->att, (int64_t)(qp ? nk / 4 : nk) * C, C, qp ? nk / 4 : nk, nk, N / nk, bs.heads, qp, kmask, hscale, s));
            {
                XGemm g = mkx(w->att, bs.dout, Nq, b.proj);
                g.C = xres; g.ldc = bs.dout; g.res = xres; g.ldres = bs.dout;
                XK(xg_gemm(g, s));
            }
            if (bs.din != bs.dout) { std::swap(x, xalt); tokens /= 4; ++stage; }
            XK(xg_layernorm(x, b.n2, 1e-6f, w->xn, Nq, bs.dout, ACT_NONE, s));
            {
                XGemm g = mkx(w->xn, bs.dout, Nq, b.fc1);
                g.C = w->hid; g.ldc = 4 * bs.dout; g.act = ACT_GELU;
                XK(xg_gemm(g, s));
            }
            {
                XGemm g = mkx(w->hid, 4 * bs.dout, Nq, b.fc2);
                g.C = x; g.ldc = bs.dout; g.res = x; g.ldres = bs.dout;
                XK(xg_gemm(g, s));
            }
            if ((int)i == e->stage_ends[stage])
                ENG_HIP(e, hipMemcpyAsync(w->sb[stage], x, sizeof(float) * (size_t)Nq * bs.dout, hipMemcpyDeviceToDevice, s));
            if (e->padded && bs.din != bs.dout && stage == 2) {
                XK(launch_gather_rows(x, 4096, xalt, e->tok_rows[2], e->pack_idx, bs.dout, n, s));
                std::swap(x, xalt);
                tokens = e->tok_rows[2];
            }
            if (i + 1 < nblocks) XK(xg_layernorm(x, e->bw[i + 1].n1, 1e-6f, w->xn, (int64_t)n * tokens, bs.dout, ACT_NONE, s, e->valid[stage], tokens));
        }
        // neck + conv_s0 / conv_s1 (composed with their lateral convs at finalize, in double); slots slot0 .. slot0 + n - 1 are contiguous
        {
            XGemm g = mkx(w->sb[3], e->stage_dims[3], n * e->tok_rows[3], e->neck3);
            g.C = w->lat3; g.ldc = 256;
            XK(xg_gemm(g, s));
        }
        {
            float* emb_slot = e->emb + (size_t)slot0 * 4096 * 256;
            XGemm g = mkx(w->sb[2], e->stage_dims[2], n * e->tok_rows[2], e->neck2);
            g.C = e->padded ? w->hid : emb_slot; g.ldc = 256; g.res = w->lat3; g.ldres = 256; g.res_shift = 2;      // 4 x tok_rows[3] = tok_rows[2]: the shift holds across images
            XK(xg_gemm(g, s));
            if (e->padded) XK(launch_gather_rows(w->hid, e->tok_rows[2], emb_slot, 4096, e->unpack_idx, 256, n, s));
        }
        {
            XGemm g = mkx(w->sb[1], e->stage_dims[1], n * 16384, e->s1);
            g.C = e->fs1 + (size_t)slot0 * 16384 * 64; g.ldc = 64;
            XK(xg_gemm(g, s));
        }
        {
            XGemm g = mkx(w->sb[0], e->stage_dims[0], n * 65536, e->s0);
            g.C = e->fs0 + (size_t)slot0 * 65536 * 32; g.ldc = 32;
            XK(xg_gemm(g, s));
        }
    }
    ENG_HIP(e, hipGetLastError());
    return SABER_OK;
}

// ------------------------------------------------------------------------------------------------ decoder
// One chunk of P <= exact_chunk_prompts() prompts through the prompt encoder's dense branch, the two-way transformer, the heads and the
// upscaling.  Leaves: e->queries (the 8 output tokens per prompt), e->iou4 [P][4], out_obj [P] (optional), masks4 [P][4][256 x 256]
// (row-major pixels); the caller applies the same mask selection as the production path.
int exact_decode_core(saber_engine* e, int slot0, int per_slot, int p_base, const float* pts, const int* labels, int P, const float* mask_in,
                      float mask_clamp, int mask_in_q0, float* out_obj, float* masks4, hipStream_t s, int n_pts) {
    TRY(ensure_ws(e));
    ExactWs* w = ws_of(e);
    if (P > w->pc) return eng_fail(e, SABER_ERR_INVALID, "exact decode: chunk larger than the exact-mode workspace");
    if (!e->dl[0].t2i.q.wf) return eng_fail(e, SABER_ERR_STATE, "exact mode: the engine was finalized without fp32 weights (set the precision before finalize)");
    // n_pts points per prompt (several clicks, a box as its two corners): 6 output tokens + the points + upstream's padding point.  Every
    // kernel below takes the token count as a parameter; the workspaces are sized for 8 tokens x the chunk.
    const int T = 7 + n_pts, PT = P * T;
    if (n_pts < 1 || (size_t)PT > (size_t)w->pc * 8 || (size_t)PT > (size_t)e->max_prompts * 8)
        return eng_fail(e, SABER_ERR_INVALID, "exact decode: prompts x tokens exceed the workspace (8 tokens x the prompt chunk)");
    const XMap slots{(int64_t)4096 * 256, per_slot, p_base};
    const float* emb0 = e->emb + (size_t)slot0 * 4096 * 256;
    XK(launch_prompt_tokens_multi(pts, labels, P, n_pts, e->pw, e->tok_pe, s));
    ENG_HIP(e, hipMemcpyAsync(e->queries, e->tok_pe, sizeof(float) * PT * 256, hipMemcpyDeviceToDevice, s));
    // src = image_embed + dense prompt embedding
    if (!mask_in) {
        xg_add_slot(nullptr, emb0, slots, e->no_mask_embed, w->keys, 4096, 256, P, ACT_NONE, s);
    } else {
        hipLaunchKernelGGL(xg_mask_hidden_kernel, dim3((unsigned)(((int64_t)P * 4096 + 255) / 256)), dim3(256), 0, s, mask_in, P, e->mw, mask_clamp, mask_in_q0, w->h2);
        XGemm g;
        g.A = w->h2; g.lda = 16; g.W = e->mw.w3; g.ldw = 16; g.bias = e->mw.b3; g.M = P * 4096; g.N = 256; g.K = 16; g.C = w->keys; g.ldc = 256;
        XK(xg_gemm(g, s));
        xg_add_slot(w->keys, emb0, slots, nullptr, w->keys, 4096, 256, P, ACT_NONE, s);
    }
    float* queries = e->queries;
    const float* tokpe = e->tok_pe;
    const int64_t NI = (int64_t)P * 4096;

    auto lin = [&](const float* A, int64_t lda, int64_t M, const LinW& L, float* C, int64_t ldc, const float* res = nullptr, int act = ACT_NONE) -> const char* {
        XGemm g = mkx(A, lda, (int)M, L);
        g.C = C; g.ldc = ldc; g.act = act;
        if (res) { g.res = res; g.ldres = ldc; }
        return xg_gemm(g, s);
    };
    // tokens -> image cross attention: queries += out_proj(attn(q_proj(queries + pe), k_proj(keys + pos), v_proj(keys))), then LayerNorm
    auto t2i = [&](const AttnW& a, const LnW& ln) -> int {
        xg_add(queries, tokpe, 0, w->t0, PT, 256, s);
        XK(lin(w->t0, 256, PT, a.q, w->tq, 128));
        {   // k_proj(keys + pos): the sum is formed in the GEMM's operand load (fp32, rounded once, as the stored sum was)
            XGemm g = mkx(w->keys, 256, (int)NI, a.k);
            g.A2 = e->dense_pe; g.lda2 = 256; g.a2_mod = 4096; g.C = w->p0; g.ldc = 128;
            XK(xg_gemm(g, s));
        }
        XK(lin(w->keys, 256, NI, a.v, w->p1, 128));
        XK(xg_attn(16, w->tq, (int64_t)T * 128, 128, w->p0, (int64_t)4096 * 128, 128, w->p1, (int64_t)4096 * 128, 128, w->ta, (int64_t)T * 128, 128, T, 4096, P, 8, 0,
                   nullptr, 0.25f, s));
        XK(lin(w->ta, 128, PT, a.o, queries, 256, queries));
        XK(xg_layernorm(queries, ln, 1e-5f, queries, PT, 256, ACT_NONE, s));
        return SABER_OK;
    };
    for (int l = 0; l < 2; ++l) {
        const DecLayerW& d = e->dl[l];
        // (1) self attention of the tokens (layer 0: no positional term, no residual)
        const float* qk_in = queries;
        if (l > 0) { xg_add(queries, tokpe, 0, w->t0, PT, 256, s); qk_in = w->t0; }
        XK(lin(qk_in, 256, PT, d.self_attn.q, w->tq, 256));
        XK(lin(qk_in, 256, PT, d.self_attn.k, w->tk, 256));
        XK(lin(queries, 256, PT, d.self_attn.v, w->tv, 256));
        XK(xg_attn(32, w->tq, (int64_t)T * 256, 256, w->tk, (int64_t)T * 256, 256, w->tv, (int64_t)T * 256, 256, w->ta, (int64_t)T * 256, 256, T, T, P, 8, 0, nullptr,
                   1.0f / sqrtf(32.0f), s));
        XK(lin(w->ta, 256, PT, d.self_attn.o, queries, 256, l > 0 ? queries : nullptr));
        XK(xg_layernorm(queries, d.n1, 1e-5f, queries, PT, 256, ACT_NONE, s));
        // (2) tokens -> image
        TRY(t2i(d.t2i, d.n2));
        // (3) MLP
        XK(lin(queries, 256, PT, d.mlp1, w->thid, 2048, nullptr, ACT_RELU));
        XK(lin(w->thid, 2048, PT, d.mlp2, queries, 256, queries));
        XK(xg_layernorm(queries, d.n3, 1e-5f, queries, PT, 256, ACT_NONE, s));
        // (4) image -> tokens: keys = LN(keys + out_proj(attn(q_proj(keys + pos), k_proj(queries + pe), v_proj(queries))))
        xg_add(queries, tokpe, 0, w->t0, PT, 256, s);
        XK(lin(w->t0, 256, PT, d.i2t.k, w->tk, 128));
        XK(lin(queries, 256, PT, d.i2t.v, w->tv, 128));
        {
            XGemm g = mkx(w->keys, 256, (int)NI, d.i2t.q);
            g.A2 = e->dense_pe; g.lda2 = 256; g.a2_mod = 4096; g.C = w->p0; g.ldc = 128;
            XK(xg_gemm(g, s));
        }
        XK(xg_attn(16, w->p0, (int64_t)4096 * 128, 128, w->tk, (int64_t)T * 128, 128, w->tv, (int64_t)T * 128, 128, w->patt, (int64_t)4096 * 128, 128, 4096, T, P, 8, 0,
                   nullptr, 0.25f, s));
        XK(lin(w->patt, 128, NI, d.i2t.o, w->kpe, 256, w->keys));
        XK(xg_layernorm(w->kpe, d.n4, 1e-5f, w->keys, NI, 256, ACT_NONE, s));
    }
    TRY(t2i(e->final_attn, e->final_ln));

    // heads: token 0 = object score, 1 = IoU, 2..5 = mask tokens
    auto mlp3 = [&](const LinW* L, const float* A, int last_act, float* outf, int ldo) -> int {
        XK(lin(A, (int64_t)T * 256, P, L[0], w->hd0, 256, nullptr, ACT_RELU));
        XK(lin(w->hd0, 256, P, L[1], w->hd1, 256, nullptr, ACT_RELU));
        XK(lin(w->hd1, 256, P, L[2], outf, ldo, nullptr, last_act));
        return SABER_OK;
    };
    TRY(mlp3(e->iou_head, queries + 256, ACT_SIGMOID, e->iou4, 4));
    if (out_obj) TRY(mlp3(e->obj_head, queries, ACT_NONE, out_obj, 1));
    {   // the 4 hypernetwork MLPs, batched over the mask token
        XGemm g = mkx(queries + 2 * 256, (int64_t)T * 256, P, e->hyper[0]);
        g.batch = 4; g.sA = 256; g.sW = 256 * 256; g.sBias = 256; g.C = w->hd0; g.ldc = 256; g.sC = (int64_t)P * 256; g.act = ACT_RELU;
        XK(xg_gemm(g, s));
        g = mkx(w->hd0, 256, P, e->hyper[1]);
        g.batch = 4; g.sA = (int64_t)P * 256; g.sW = 256 * 256; g.sBias = 256; g.C = w->hd1; g.ldc = 256; g.sC = (int64_t)P * 256; g.act = ACT_RELU;
        XK(xg_gemm(g, s));
        g = mkx(w->hd1, 256, P, e->hyper[2]);
        g.batch = 4; g.sA = (int64_t)P * 256; g.sW = 32 * 256; g.sBias = 32; g.C = e->hyper_out; g.ldc = 128; g.sC = 32;
        XK(xg_gemm(g, s));
    }
    // upscaling: ConvTranspose2d(256 -> 64, k2 s2) as a GEMM whose N index is (ky*2+kx)*64 + co: row t of the 64^2 grid becomes rows
    // 4t .. 4t+3 of the 128^2 grid in the engine's token order; + feat_s1; LayerNorm2d; GELU; the same for 64 -> 32 with feat_s0; GELU
    // (+ feat_s1 / feat_s0 of the prompt's slot in the GEMM epilogues: output row t, column n of the first GEMM is element t * 256 + n of the
    // slot's 128^2 x 64 map in the engine's token order, likewise t * 128 + n of the 256^2 x 32 map for the second)
    {
        XGemm g = mkx(w->keys, 256, (int)NI, e->dc1);
        g.C = w->up1; g.ldc = 256;
        g.res = e->fs1 + (size_t)slot0 * 16384 * 64; g.ldres = 256; g.res_rows_per = 4096; g.res_stride = (int64_t)16384 * 64; g.res_div = per_slot; g.res_off = p_base;
        XK(xg_gemm(g, s));
    }
    XK(xg_layernorm(w->up1, e->up_ln, 1e-6f, w->up1, (int64_t)P * 16384, 64, ACT_GELU, s));
    {
        XGemm g = mkx(w->up1, 64, (int)((int64_t)P * 16384), e->dc2);
        g.C = w->up2; g.ldc = 128; g.act = ACT_GELU; g.act_last = 1;
        g.res = e->fs0 + (size_t)slot0 * 65536 * 32; g.ldres = 128; g.res_rows_per = 16384; g.res_stride = (int64_t)65536 * 32; g.res_div = per_slot; g.res_off = p_base;
        XK(xg_gemm(g, s));
    }
    hipLaunchKernelGGL(xg_mask_dot_kernel, dim3((unsigned)(((int64_t)P * 65536 + 255) / 256)), dim3(256), 0, s, w->up2, e->hyper_out, P, masks4);
    ENG_HIP(e, hipGetLastError());
    return SABER_OK;
}
