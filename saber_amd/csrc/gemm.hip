// bf16 MFMA GEMM with fused epilogues:  C[M,N] = epi( A[M,K] . W[N,K]^T + bias )
//
// Replaces the torch ops nn.Linear / 1x1 Conv2d / ConvTranspose2d(k2s2) executed by the
// third-party sam2 package under the reference call site saber/adapters/sam2/predictor.py:70
// (SURVEY.md 8a rows b4-b7, b9, b10).  gfx950 only: v_mfma_f32_16x16x32_bf16, wave64.
//
// Tile 128x128x64, 4 waves (2x2), each wave 64x64 = 4x4 MFMA tiles.  Both operands are
// K-contiguous, staged global->registers->LDS (double-buffered, XOR-swizzled 16-B chunks so
// ds_read_b128 fragment reads are bank-conflict-free).  The MFMA is issued "swapped"
// (A-operand = W fragment, B-operand = A fragment) so each lane ends up owning 4 consecutive
// output columns of one row: epilogue loads/stores are 16-B (fp32) / 8-B (bf16) vectors.
#include "common.h"
#include "kernels.h"

#define BM 128
#define BN 128
#define BK 64
#define TILE_BYTES (BM * BK * 2)  // 16 KiB per operand per buffer

// XCD-aware block -> tile map: blocks b and b+8 share an XCD (and its L2).  All N-tiles of one M-tile are given the same
// b % 8 and consecutive dispatch slots, so the A panel is fetched from HBM once instead of once per XCD that touches it.
// Grid must be ceil(tiles_m / 8) * 8 * tiles_n blocks; returns false for the padding blocks.
__device__ __forceinline__ bool tile_map(int b, int tiles_m, int tiles_n, int* tm, int* tn) {
    const int xcd = b & 7, q = b >> 3;
    *tn = q % tiles_n;
    *tm = (q / tiles_n) * 8 + xcd;
    return *tm < tiles_m;
}

__device__ __forceinline__ int swz(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }

template <int T>
__device__ __forceinline__ void gemm_epilogue(const GemmParams& p, f32x4 (&acc)[T][T], int m0, int n0, int wm, int wn, int fi, int fg, int64_t z) {
    // ---------------- epilogue: lane owns C[m][n..n+3]
    const float* bias = p.bias ? p.bias + z * p.strideBias : nullptr;
    const float* res = p.res ? p.res + z * p.strideRes : nullptr;
    float* Cf = p.Cf ? p.Cf + z * p.strideCf : nullptr;
    bf16_t* Cb = p.Cb ? p.Cb + z * p.strideCb : nullptr;
    const bool vec_ok = (p.N & 3) == 0;
    auto apply_act = [&](float (&v)[4]) {
        if (p.act == ACT_GELU) {
            { const f32x2 g0_ = gelu_erf2((f32x2){v[0], v[1]}), g1_ = gelu_erf2((f32x2){v[2], v[3]}); v[0] = g0_.x; v[1] = g0_.y; v[2] = g1_.x; v[3] = g1_.y; }
        } else if (p.act == ACT_RELU) {
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
        } else if (p.act == ACT_SIGMOID) {
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = 1.0f / (1.0f + expf(-v[r]));
        }
    };
#pragma unroll
    for (int i = 0; i < T; ++i) {
        const int m = m0 + wm * (16 * T) + i * 16 + fi;
#pragma unroll
        for (int j = 0; j < T; ++j) {
            const int n = n0 + wn * (16 * T) + j * 16 + fg * 4;
            float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
            const bool nvec = vec_ok && (n + 3 < p.N);
            if (bias) {
                if (nvec) {
                    const float4 b = *reinterpret_cast<const float4*>(bias + n);
                    v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w;
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) if (n + r < p.N) v[r] += bias[n + r];
                }
            }
            if (!p.act_last) apply_act(v);
            int mo = m;
            bool writer = m < p.M;
            if (p.pool4) {  // max over 4 consecutive rows (lanes 4q..4q+3 of the 16-lane group)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float x = (m < p.M) ? v[r] : -3.0e38f;
                    x = fmaxf(x, dpp_mov<DPP_XOR1>(x));
                    x = fmaxf(x, dpp_mov<DPP_XOR2>(x));
                    v[r] = x;
                }
                writer = writer && ((fi & 3) == 0);
                mo = m >> 2;
            }
            if (!writer) continue;
            if (res) {
                int rr = mo >> p.res_shift;
                if (p.res_mod > 0) rr %= p.res_mod;
                const float* rp = res + (int64_t)rr * p.ldres + n;
                if (nvec) {
                    const float4 b = *reinterpret_cast<const float4*>(rp);
                    v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w;
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) if (n + r < p.N) v[r] += rp[r];
                }
            }
            if (p.act_last) apply_act(v);
            if (Cf) {
                float* cp = Cf + (int64_t)mo * p.ldcf + n;
                if (nvec) *reinterpret_cast<float4*>(cp) = make_float4(v[0], v[1], v[2], v[3]);
                else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) if (n + r < p.N) cp[r] = v[r];
                }
            }
            if (Cb) {
                bf16_t* cp = Cb + (int64_t)mo * p.ldcb + n;
                if (nvec) *reinterpret_cast<uint2*>(cp) = make_uint2(pack_op16(v[0], v[1]), pack_op16(v[2], v[3]));
                else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) if (n + r < p.N) cp[r] = f2op(v[r]);
                }
            }
        }
    }
}

// T = MFMA tiles per wave and dimension: T = 4 -> 128x128 block tile, T = 2 -> 64x64 (small problems: 4x the blocks)
template <int T>
__global__ __launch_bounds__(256) void gemm_bf16_kernel(GemmParams p) {
    constexpr int TB = 32 * T;            // block tile edge
    constexpr int TBYTES = TB * BK * 2;   // bytes per operand tile
    constexpr int NL = T;                 // 16-B chunks per thread and operand
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int tiles_n = (p.N + TB - 1) / TB;
    int tm, tn;
    if (!tile_map(blockIdx.x, (p.M + TB - 1) / TB, tiles_n, &tm, &tn)) return;
    const int m0 = tm * TB, n0 = tn * TB;
    const int64_t z = blockIdx.z;
    const bf16_t* __restrict__ A = p.A + z * p.strideA;
    const bf16_t* __restrict__ W = p.W + z * p.strideW;

    const int c = tid & 7, r0 = tid >> 3;
    u32x4 ra[NL], rw[NL];
    // Loads are unconditional from clamped in-bounds addresses and zeroed by value selects afterwards: a
    // "load or zero" written as a branch makes hipcc wait vmcnt(0) per load and serialises the prefetch.
    int64_t aoff[NL], woff[NL];
    bool aok[NL], wok[NL];
#pragma unroll
    for (int i = 0; i < NL; ++i) {
        const int row = m0 + r0 + 32 * i, n = n0 + r0 + 32 * i;
        aok[i] = row < p.M;
        wok[i] = n < p.N;
        aoff[i] = (int64_t)(aok[i] ? row : p.M - 1) * p.lda;
        woff[i] = (int64_t)(wok[i] ? n : p.N - 1) * p.ldw;
    }
    bool kok_cur = true;
    auto gload = [&](int kt) {   // raw loads only: nothing consumes the registers until lstore()
        const int k = kt * BK + c * 8;
        kok_cur = k < p.K;
        const int kc = kok_cur ? k : 0;
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            ra[i] = *reinterpret_cast<const u32x4*>(A + aoff[i] + kc);
            rw[i] = *reinterpret_cast<const u32x4*>(W + woff[i] + kc);
        }
    };
    auto lstore = [&](int buf) {
        char* sa = smem + buf * 2 * TBYTES;
        char* sw = sa + TBYTES;
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            const int row = r0 + 32 * i;
            const int off = swz(row, c);
            const u32x4 z4 = {0u, 0u, 0u, 0u};
            *reinterpret_cast<u32x4*>(sa + off) = (kok_cur && aok[i]) ? ra[i] : z4;
            *reinterpret_cast<u32x4*>(sw + off) = (kok_cur && wok[i]) ? rw[i] : z4;
        }
    };

    f32x4 acc[T][T];
#pragma unroll
    for (int i = 0; i < T; ++i)
#pragma unroll
        for (int j = 0; j < T; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int nk = (p.K + BK - 1) / BK;
    gload(0);
    lstore(0);
    __syncthreads();
    const int fi = lane & 15, fg = lane >> 4;
    for (int kt = 0; kt < nk; ++kt) {
        if (kt + 1 < nk) gload(kt + 1);
        const char* sa = smem + (kt & 1) * 2 * TBYTES;
        const char* sw = sa + TBYTES;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            op16x8 af[T], wf[T];
#pragma unroll
            for (int i = 0; i < T; ++i) {
                af[i] = *reinterpret_cast<const op16x8*>(sa + swz(wm * (16 * T) + i * 16 + fi, ks * 4 + fg));
                wf[i] = *reinterpret_cast<const op16x8*>(sw + swz(wn * (16 * T) + i * 16 + fi, ks * 4 + fg));
            }
#pragma unroll
            for (int i = 0; i < T; ++i)
#pragma unroll
                for (int j = 0; j < T; ++j)
                    acc[i][j] = MFMA_16x16x32(wf[j], af[i], acc[i][j], 0, 0, 0);
        }
        if (kt + 1 < nk) lstore((kt + 1) & 1);
        __syncthreads();
    }

    gemm_epilogue<T>(p, acc, m0, n0, wm, wn, fi, fg, z);
}

// ------------------------------------------------------------------------------------------------
// Same tile, but operands go global -> LDS directly (global_load_lds_dwordx4, no staging VGPRs, no ds_write) through a
// 3-stage ring with counted vmcnt and raw s_barrier, so two K-tiles of loads stay in flight across the barrier
// (cdna_hip_programming.md "Pipelining across barriers").  The XOR swizzle is applied on the SOURCE address (the LDS
// image of one wave-instruction is lane-linear).  Out-of-range rows are clamped instead of zeroed: rows >= M / >= N
// only feed outputs that are never stored.  Requires W rows zero-padded to a multiple of 64 in K (p.w_kpad).
typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// WM wave-rows x 2 wave-columns, each wave 64x64: WM = 2 -> 128x128 tile (4 waves), WM = 4 -> 256x128 tile (8 waves,
// two per SIMD, so one wave's ds_read latency hides under the other's MFMAs).
// PERSISTENT: the block walks over its tiles (XCD-aware order, stride = gridDim.x) with ONE continuous K-tile stream, so the
// loads of the next tile's first K-tiles are in flight while the current tile's epilogue stores drain.
template <int WM>
__global__ __launch_bounds__(WM * 128) void gemm_bf16_glds_kernel(GemmParams p) {
    constexpr int TBM = WM * 64;
    constexpr int A_BYTES = TBM * BK * 2, W_BYTES = BN * BK * 2, STAGE = A_BYTES + W_BYTES;
    constexpr int NW = WM * 2;                 // waves
    constexpr int WI = 16 / NW;                // W wave-instructions per wave per K-tile (A: always 4)
    constexpr int NLOADS = 4 + WI;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* epi_lds = smem + 3 * STAGE;          // NW x 2 KB: per-wave transposition scratch of the bf16 epilogue
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int tiles_m = (p.M + TBM - 1) / TBM, tiles_n = (p.N + BN - 1) / BN;
    const int padded = ((tiles_m + 7) / 8) * 8 * tiles_n;
    const int64_t z = blockIdx.z;
    const bf16_t* __restrict__ A = p.A + z * p.strideA;
    const bf16_t* __restrict__ W = p.W + z * p.strideW;
    const int fi = lane & 15, fg = lane >> 4;
    const int nk = (p.K + BK - 1) / BK;

    int achunk[4], wchunk[WI], arow[4], wrow[WI];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        arow[i] = (wave * 4 + i) * 8 + (lane >> 3);
        achunk[i] = (lane & 7) ^ ((arow[i] >> 1) & 7);      // logical 16-B chunk that lands at physical slot lane&7
    }
#pragma unroll
    for (int i = 0; i < WI; ++i) {
        wrow[i] = (wave * WI + i) * 8 + (lane >> 3);
        wchunk[i] = (lane & 7) ^ ((wrow[i] >> 1) & 7);
    }
    auto next_tile = [&](int L, int* tm, int* tn) {   // first valid tile at or after linear slot L (stride gridDim.x)
        while (L < padded && !tile_map(L, tiles_m, tiles_n, tm, tn)) L += gridDim.x;
        return L;
    };

    // ---- issue cursor
    int Li, tmi = 0, tni = 0, kti = 0, si = 0;
    const bf16_t* asrc[4];
    const bf16_t* wsrc[WI];
    auto set_issue_tile = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i) asrc[i] = A + (int64_t)min(tmi * TBM + arow[i], p.M - 1) * p.lda;
#pragma unroll
        for (int i = 0; i < WI; ++i) wsrc[i] = W + (int64_t)min(tni * BN + wrow[i], p.N - 1) * p.ldw;
    };
    auto issue = [&]() {   // one K-tile of the issue cursor into ring stage si, then advance the cursor
        char* sa = smem + si * STAGE;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int k = kti * BK + achunk[i] * 8;
            const int ka = k < p.K ? k : 0;              // K tail of A: any finite data, W supplies the zeros
            if (!(p.dbg & 4)) __builtin_amdgcn_global_load_lds((gptr_t)(asrc[i] + ka), (lptr_t)(sa + (wave * 4 + i) * 1024), 16, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < WI; ++i)
            if (!(p.dbg & 8)) __builtin_amdgcn_global_load_lds((gptr_t)(wsrc[i] + kti * BK + wchunk[i] * 8), (lptr_t)(sa + A_BYTES + (wave * WI + i) * 1024), 16, 0, 0);
        si = si == 2 ? 0 : si + 1;
        if (++kti == nk) {
            kti = 0;
            Li = next_tile(Li + gridDim.x, &tmi, &tni);
            if (Li < padded) set_issue_tile();
        }
    };
    Li = next_tile(blockIdx.x, &tmi, &tni);
    if (Li >= padded) return;                      // block-uniform
    set_issue_tile();
    // ---- compute cursor
    int Lc = Li, tmc = tmi, tnc = tni, ktc = 0, sc = 0;

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // bias of the bf16 fast path is fetched when a tile STARTS: a global load inside the epilogue would have to wait (vmcnt is
    // in-order) for the K-tile prefetch and for the stores of the previous rows.
    const bool fast_bf16 = p.Cb && !p.Cf && !p.res && !p.pool4 && (p.N & 7) == 0 && (p.ldcb & 7) == 0;
    // fp32 output (+ fp32 residual): proj / fc2 of every block.  The residual tile is fetched into registers right before the
    // LAST K-tile of the output tile is computed (older in the vmcnt queue than that iteration's prefetch, so the counted wait
    // that precedes the epilogue covers it) instead of 16 dependent load -> wait -> add -> store round trips.
    const bool fast_f32 = p.Cf && !p.Cb && !p.pool4 && p.act == ACT_NONE && (p.N & 3) == 0 && (p.ldcf & 3) == 0 &&
                          (!p.res || (p.res_shift == 0 && p.res_mod == 0 && (p.ldres & 3) == 0));
    float4 bias4[4];
    auto load_bias = [&](int tn) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = tn * BN + wn * 64 + j * 16 + fg * 4;
            bias4[j] = ((fast_bf16 || fast_f32) && p.bias && n + 3 < p.N) ? *reinterpret_cast<const float4*>(p.bias + z * p.strideBias + n) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    load_bias(tnc);
    float4 rres[4][4];
    auto load_res = [&](int tm, int tn) {
        const float* res = p.res + z * p.strideRes;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = min(tm * TBM + wm * 64 + i * 16 + fi, p.M - 1);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int n = min(tn * BN + wn * 64 + j * 16 + fg * 4, p.N - 4);     // clamped lanes are never stored
                rres[i][j] = *reinterpret_cast<const float4*>(res + (int64_t)m * p.ldres + n);
            }
        }
    };
    int ahead = 0;                                  // K-tiles issued but not yet computed
    issue(); ++ahead;
    if (Li < padded) { issue(); ++ahead; }
    if (ahead == 2) { if (NLOADS == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); }
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    while (Lc < padded) {
        const bool more = Li < padded;              // a third K-tile can be put in flight
        if (fast_f32 && p.res && ktc == nk - 1) load_res(tmc, tnc);
        if (more) { issue(); ++ahead; }
        const char* sa = smem + sc * STAGE;
        const char* sw = sa + A_BYTES;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            op16x8 af[4], wf[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                af[i] = *reinterpret_cast<const op16x8*>(sa + swz(wm * 64 + i * 16 + fi, ks * 4 + fg));
                wf[i] = *reinterpret_cast<const op16x8*>(sw + swz(wn * 64 + i * 16 + fi, ks * 4 + fg));
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = MFMA_16x16x32(wf[j], af[i], acc[i][j], 0, 0, 0);
        }
        --ahead;
        // the K-tile computed next must have landed (this wave's part) before the barrier; the youngest one may stay in flight.
        // This wait sits BEFORE the epilogue so that the epilogue's stores (younger in the vmcnt queue) are not waited for.
        if (ahead == 2) { if (NLOADS == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); }
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        sc = sc == 2 ? 0 : sc + 1;
        if (++ktc == nk) {
            const int m0 = tmc * TBM, n0 = tnc * BN;
            if (p.dbg & 2) {
            } else if (fast_bf16) {
                // bf16-only output (qkv, fc1: the widest matrices of the encoder): bias + activation in registers, then 16 rows at a
                // time are transposed through the wave's 2 KB of LDS so that every store instruction writes 8 full 128-B rows.
                bf16_t* Cb = p.Cb + z * p.strideCb;
                const uint32_t tb_a = (uint32_t)(uintptr_t)(lptr_t)(epi_lds + wave * 2048);
                const uint32_t tb_r0 = tb_a + (lane >> 3) * 128 + (((lane & 7) ^ ((lane >> 3) & 7)) << 4);   // rows 0-7; rows 8-15 are +1024 (same swizzle)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int n = n0 + wn * 64 + j * 16 + fg * 4;
                        float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
                        v[0] += bias4[j].x; v[1] += bias4[j].y; v[2] += bias4[j].z; v[3] += bias4[j].w;
                        if (p.act == ACT_GELU) {
                            { const f32x2 g0_ = gelu_erf2((f32x2){v[0], v[1]}), g1_ = gelu_erf2((f32x2){v[2], v[3]}); v[0] = g0_.x; v[1] = g0_.y; v[2] = g1_.x; v[3] = g1_.y; }
                        } else if (p.act == ACT_RELU) {
#pragma unroll
                            for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
                        } else if (p.act == ACT_SIGMOID) {
#pragma unroll
                            for (int r = 0; r < 4; ++r) v[r] = 1.0f / (1.0f + expf(-v[r]));
                        }
                        const int chunk = j * 2 + (fg >> 1);
                        // LDS traffic of the epilogue is inline asm: hipcc orders every VISIBLE ds access behind the direct-to-LDS
                        // loads in flight with s_waitcnt vmcnt(0), which would also drain the stores of the previous 16 rows.
                        const uint64_t pk = ((uint64_t)pack_op16(v[2], v[3]) << 32) | pack_op16(v[0], v[1]);
                        asm volatile("ds_write_b64 %0, %1" ::"v"(tb_a + fi * 128 + ((chunk ^ (fi & 7)) << 4) + (fg & 1) * 8), "v"(pk) : "memory");
                    }
                    u32x4 val0, val1;
                    asm volatile("s_waitcnt lgkmcnt(0)\n\tds_read_b128 %0, %2\n\tds_read_b128 %1, %3\n\ts_waitcnt lgkmcnt(0)"
                                 : "=&v"(val0), "=&v"(val1) : "v"(tb_r0), "v"(tb_r0 + 1024) : "memory");
#pragma unroll
                    for (int it = 0; it < 2; ++it) {
                        const int row = it * 8 + (lane >> 3), chunk = lane & 7;
                        const int m = m0 + wm * 64 + i * 16 + row, n = n0 + wn * 64 + chunk * 8;
                        if (m < p.M && n < p.N && !(p.dbg & 1)) *reinterpret_cast<u32x4*>(Cb + (int64_t)m * p.ldcb + n) = it ? val1 : val0;
                    }
                }
            } else if (fast_f32) {
                float* Cf = p.Cf + z * p.strideCf;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int m = m0 + wm * 64 + i * 16 + fi;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int n = n0 + wn * 64 + j * 16 + fg * 4;
                        float4 v = make_float4(acc[i][j][0] + bias4[j].x, acc[i][j][1] + bias4[j].y, acc[i][j][2] + bias4[j].z, acc[i][j][3] + bias4[j].w);
                        if (p.res) { v.x += rres[i][j].x; v.y += rres[i][j].y; v.z += rres[i][j].z; v.w += rres[i][j].w; }
                        if (m < p.M && n + 3 < p.N) *reinterpret_cast<float4*>(Cf + (int64_t)m * p.ldcf + n) = v;
                    }
                }
            } else {
                gemm_epilogue<4>(p, acc, m0, n0, wm, wn, fi, fg, z);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
            ktc = 0;
            Lc = next_tile(Lc + gridDim.x, &tmc, &tnc);
            if (Lc < padded) load_bias(tnc);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
}
// ------------------------------------------------------------------------------------------------
// Two workgroups per CU: 256x128 tile, K-tiles of 32, 3-stage direct-to-LDS ring (72 KB), 8 waves of 64x64, <= 128 VGPRs.
// While one workgroup runs its epilogue (bias / activation / residual, LDS transposition, stores) the other one keeps the
// matrix cores busy; with one workgroup per CU the epilogue is serial time (25-60 % of the shapes of a Hiera block).
// LDS rows are 64 B (4 chunks of 16 B); physical chunk = logical chunk ^ G[(row >> 2) & 3], G = {0, 2, 3, 1}, which makes the
// four 16-lane service groups of ds_read_b128 hit 16 distinct (row mod 4, chunk) pairs = all 64 banks.
#define BK2 32
#define G2_STAGE ((256 + BN) * BK2 * 2)
#define G2_LDS (3 * G2_STAGE)
__device__ __forceinline__ int g2perm(int row) { return (0x78 >> (((row >> 2) & 3) * 2)) & 3; }   // {0, 2, 3, 1}
__device__ __forceinline__ int swz2(int row, int chunk) { return row * 64 + ((chunk ^ g2perm(row)) << 4); }

__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4, 4))) void gemm_bf16_glds2_kernel(GemmParams p) {
    constexpr int A_BYTES = 256 * BK2 * 2;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // waves 0-3 own the left 64 columns, waves 4-7 the right 64: one wave of each column half per SIMD, so that in an N-edge tile
    // whose right half lies beyond N (N = 576, 288, 144: a tenth to a quarter of all tiles' columns) EVERY SIMD loses half of this
    // workgroup's MFMA + fragment-read work to the co-resident workgroup, not two SIMDs all of it
    const int wm = wave & 3, wn = wave >> 2;
    const int tiles_m = (p.M + 255) / 256, tiles_n = (p.N + BN - 1) / BN;
    int tm, tn;
    if (!tile_map(blockIdx.x, tiles_m, tiles_n, &tm, &tn)) return;
    if (p.rev) tm = tiles_m - 1 - tm;
    const int m0 = tm * 256, n0 = tn * BN;
    const int64_t z = blockIdx.z;
    const bf16_t* __restrict__ A = p.A + z * p.strideA;
    const bf16_t* __restrict__ W = p.W + z * p.strideW;
    const int fi = lane & 15, fg = lane >> 4;
    const int nk = (p.K + BK2 - 1) / BK2;
    const bool cols_live = (n0 + wn * 64 < p.N) || (p.dbg & 2);   // wave-uniform: this wave's 64 columns hold at least one real one

    // one wave-instruction = 16 rows x 64 B; per K-tile: A 16 instructions (2 per wave), W 8 (1 per wave)
    const int lrow = lane >> 2, lslot = lane & 3;
    const bf16_t* asrc[2];
    int achunk[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = (wave * 2 + i) * 16 + lrow;
        asrc[i] = A + (int64_t)min(m0 + row, p.M - 1) * p.lda;
        achunk[i] = lslot ^ g2perm(row);
    }
    const int wrow_l = wave * 16 + lrow;
    const bf16_t* wsrc = W + (int64_t)min(n0 + wrow_l, p.N - 1) * p.ldw;
    const int wchunk = lslot ^ g2perm(wrow_l);
    auto issue = [&](int kt) {
        char* sa = smem + (kt % 3) * G2_STAGE;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int k = kt * BK2 + achunk[i] * 8;
            __builtin_amdgcn_global_load_lds((gptr_t)(asrc[i] + (k < p.K ? k : 0)), (lptr_t)(sa + (wave * 2 + i) * 1024), 16, 0, 0);
        }
        __builtin_amdgcn_global_load_lds((gptr_t)(wsrc + kt * BK2 + wchunk * 8), (lptr_t)(sa + A_BYTES + wave * 1024), 16, 0, 0);
    };

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    issue(0);
    if (nk > 1) issue(1);
    if (nk > 1) asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    for (int kt = 0; kt < nk; ++kt) {
        const char* sa = smem + (kt % 3) * G2_STAGE;
        const char* sw = sa + A_BYTES;
        if (cols_live) {
            op16x8 af[4], wf[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                af[i] = *reinterpret_cast<const op16x8*>(sa + swz2(wm * 64 + i * 16 + fi, fg));
                wf[i] = *reinterpret_cast<const op16x8*>(sw + swz2(wn * 64 + i * 16 + fi, fg));
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = MFMA_16x16x32(wf[j], af[i], acc[i][j], 0, 0, 0);
        }
        // the K-tile two ahead is issued BEHIND this one's MFMAs (an LDS-DMA issue stalls the issuing wave 60-185 cycles; at the top of
        // the K-tile, right behind the barrier, nothing of this wave is in flight to hide it): -0.3 ms per slice
        if (kt + 2 < nk) issue(kt + 2);
        // K-tile kt+1 must have landed (this wave's pieces); the youngest one may stay in flight
        if (kt + 2 < nk) asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }

    const bool fast_bf16 = p.Cb && !p.Cf && !p.res && !p.pool4 && (p.N & 7) == 0 && (p.ldcb & 7) == 0;
    const bool fast_f32 = p.Cf && !p.Cb && !p.pool4 && p.act == ACT_NONE && (p.N & 3) == 0 && (p.ldcf & 3) == 0 &&
                          (!p.res || (p.res_shift == 0 && p.res_mod == 0 && (p.ldres & 3) == 0));
    if (fast_bf16) {
        // bias + activation in registers, 16 rows at a time transposed through 2 KB of the (now idle) ring so that every store
        // instruction writes 8 full 128-B rows; LDS traffic in inline asm (see gemm_bf16_glds_kernel)
        bf16_t* Cb = p.Cb + z * p.strideCb;
        const uint32_t tb_a = (uint32_t)(uintptr_t)(lptr_t)(smem + wave * 2048);
        const uint32_t tb_r0 = tb_a + (lane >> 3) * 128 + (((lane & 7) ^ ((lane >> 3) & 7)) << 4);
        // the wave's 4 bias vectors: unconditional loads from clamped addresses, ONE wait (a load inside a branch is waited for
        // with vmcnt(0) on its own: that was 16 dependent L2 round trips per tile)
        float4 bias4[4];
        {
            const float* bp = p.bias ? p.bias + z * p.strideBias : reinterpret_cast<const float*>(p.W);
#pragma unroll
            for (int j = 0; j < 4; ++j) bias4[j] = *reinterpret_cast<const float4*>(bp + min(n0 + wn * 64 + j * 16 + fg * 4, p.N - 4));
#pragma unroll
            for (int j = 0; j < 4; ++j) if (!p.bias) bias4[j] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float v[4] = {acc[i][j][0] + bias4[j].x, acc[i][j][1] + bias4[j].y, acc[i][j][2] + bias4[j].z, acc[i][j][3] + bias4[j].w};
                if (p.act == ACT_GELU) {
                    { const f32x2 g0_ = gelu_erf2((f32x2){v[0], v[1]}), g1_ = gelu_erf2((f32x2){v[2], v[3]}); v[0] = g0_.x; v[1] = g0_.y; v[2] = g1_.x; v[3] = g1_.y; }
                } else if (p.act == ACT_RELU) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
                } else if (p.act == ACT_SIGMOID) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = __builtin_amdgcn_rcpf(1.0f + __expf(-v[r]));
                }
                const int chunk = j * 2 + (fg >> 1);
                const uint64_t pk = ((uint64_t)pack_op16(v[2], v[3]) << 32) | pack_op16(v[0], v[1]);
                asm volatile("ds_write_b64 %0, %1" ::"v"(tb_a + fi * 128 + ((chunk ^ (fi & 7)) << 4) + (fg & 1) * 8), "v"(pk) : "memory");
            }
            u32x4 val0, val1;
            asm volatile("s_waitcnt lgkmcnt(0)\n\tds_read_b128 %0, %2\n\tds_read_b128 %1, %3\n\ts_waitcnt lgkmcnt(0)"
                         : "=&v"(val0), "=&v"(val1) : "v"(tb_r0), "v"(tb_r0 + 1024) : "memory");
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                const int row = it * 8 + (lane >> 3), chunk = lane & 7;
                const int m = m0 + wm * 64 + i * 16 + row, n = n0 + wn * 64 + chunk * 8;
                if (m < p.M && n < p.N) *reinterpret_cast<u32x4*>(Cb + (int64_t)m * p.ldcb + n) = it ? val1 : val0;
            }
        }
    } else if (fast_f32) {
        float* Cf = p.Cf + z * p.strideCf;
        // bias and residual: unconditional loads from clamped addresses (clamped lanes are never stored).  The bias goes into the
        // accumulators first (its registers die), then the residual rows of group i + 1 are in flight while group i is added and
        // stored: one exposed HBM latency per tile instead of four (the kernel lives on 128 VGPRs: 64 accumulators + 2 x 16 residual)
        // residual through a buffer descriptor over this tile's rows: ONE 32-bit per-lane offset for all 16 loads (row group in the
        // scalar offset, column group in the immediate), rows beyond M read as zero (never stored), no clamps
        const float* rp = p.res ? p.res + z * p.strideRes : p.Cf + z * p.strideCf;     // no residual: any readable fp32 (discarded)
        const int64_t ldr = p.res ? p.ldres : p.ldcf;
        const float* rtile = rp + (int64_t)m0 * ldr;
        const uint32_t rbytes = (uint32_t)(min(256, p.M - m0) * ldr * 4);
        // (every input is a kernel argument or blockIdx-derived: the descriptor is built in SGPRs without readfirstlane)
        const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)rtile, 0, (int)rbytes, 0x00020000);
        int lane_e = lane;                            // opaque copy: keeps the epilogue's address arithmetic from being hoisted above the
        asm volatile("" : "+v"(lane_e));              // main loop, where every live register costs a spill (128-register budget)
        const uint32_t voff = (uint32_t)(((wm * 64 + (lane_e & 15)) * ldr + n0 + wn * 64 + (lane_e >> 4) * 4) * 4);
        const uint32_t sstep = (uint32_t)(16 * ldr * 4);
        u32x4 ra[4], rb[4];
        auto load_group = [&](int i, u32x4 (&rr)[4]) {
#pragma unroll
            for (int j = 0; j < 4; ++j) rr[j] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff + j * 64, i * sstep, 0);
        };
        load_group(0, ra);
        {   // unconditional (a branch around the accumulator update doubles their live ranges): no bias = any readable fp32, zeroed
            const float* bp = p.bias ? p.bias + z * p.strideBias : reinterpret_cast<const float*>(p.W);
            const bool has_bias = p.bias != nullptr;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float4 b4 = *reinterpret_cast<const float4*>(bp + min(n0 + wn * 64 + j * 16 + fg * 4, p.N - 4));
                if (!has_bias) b4 = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int i = 0; i < 4; ++i) { acc[i][j][0] += b4.x; acc[i][j][1] += b4.y; acc[i][j][2] += b4.z; acc[i][j][3] += b4.w; }
            }
        }
        // output through a descriptor over the same rows of C (rows beyond M are dropped by the range check; columns by the predicate)
        float* ctile = Cf + (int64_t)m0 * p.ldcf;
        const __amdgpu_buffer_rsrc_t csrc = __builtin_amdgcn_make_buffer_rsrc((void*)ctile, 0, (int)(uint32_t)(min(256, p.M - m0) * p.ldcf * 4), 0x00020000);
        const int ncol0 = n0 + wn * 64 + (lane_e >> 4) * 4;
        const uint32_t coff = (uint32_t)(((wm * 64 + (lane_e & 15)) * p.ldcf + ncol0) * 4);
        const uint32_t cstep = (uint32_t)(16 * p.ldcf * 4);
        auto store_group = [&](int i, const u32x4 (&rr)[4]) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                u32x4 v;
                if (p.res) {
                    v[0] = __float_as_uint(acc[i][j][0] + __uint_as_float(rr[j][0])); v[1] = __float_as_uint(acc[i][j][1] + __uint_as_float(rr[j][1]));
                    v[2] = __float_as_uint(acc[i][j][2] + __uint_as_float(rr[j][2])); v[3] = __float_as_uint(acc[i][j][3] + __uint_as_float(rr[j][3]));
                } else {
                    v[0] = __float_as_uint(acc[i][j][0]); v[1] = __float_as_uint(acc[i][j][1]); v[2] = __float_as_uint(acc[i][j][2]); v[3] = __float_as_uint(acc[i][j][3]);
                }
                if (ncol0 + j * 16 + 3 < p.N) __builtin_amdgcn_raw_buffer_store_b128(v, csrc, coff + j * 64, i * cstep, 0);
            }
        };
        // sched_barrier: without it the scheduler hoists all four load groups to the top (64 more registers: spills)
        __builtin_amdgcn_sched_barrier(0);
        load_group(1, rb);
        __builtin_amdgcn_sched_barrier(0);
        store_group(0, ra);
        __builtin_amdgcn_sched_barrier(0);
        load_group(2, ra);
        __builtin_amdgcn_sched_barrier(0);
        store_group(1, rb);
        __builtin_amdgcn_sched_barrier(0);
        load_group(3, rb);
        __builtin_amdgcn_sched_barrier(0);
        store_group(2, ra);
        __builtin_amdgcn_sched_barrier(0);
        store_group(3, rb);
    } else {
        gemm_epilogue<4>(p, acc, m0, n0, wm, wn, fi, fg, z);
    }
}

// ------------------------------------------------------------------------------------------------
// 256x256 tiles, 8 waves of 128x64, K-tiles of 32 through a 4-stage direct-to-LDS ring (128 KB, three K-tiles in flight), PERSISTENT
// with one continuous K-tile stream across tiles (gemm_bf16_p256s_kernel; bf16 output only: qkv / fc1 of the Hiera blocks.  An fp32 +
// residual epilogue in the same kernel cost 19 SGPR spills and 12 % of the main loop's speed, and was slower than the 256x128 kernel
// on every fp32 shape anyway: those stay on gemm_bf16_glds2_kernel).
// Why: in the 256x128 kernels every K-tile costs a wave 3 LDS-DMA issues (60-185 cycles each, the vector-memory path is shared
// with the epilogue's stores) for 16 MFMAs (256 cycles); here it is 4 issues for 32 MFMAs, and half the L2 -> LDS bytes per FLOP.
#define P2_STAGE ((256 + 256) * BK2 * 2)
#define P2_NST 4
#define P2_LDS (P2_NST * P2_STAGE + 8 * 2048)

template <bool STAMPS>
__global__ __launch_bounds__(512) void gemm_bf16_p256s_kernel(GemmParams p) {
    constexpr int A_BYTES = 256 * BK2 * 2;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* epi_lds = smem + P2_NST * P2_STAGE;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;           // 2 x 4 waves: rows 128 wm .., columns 64 wn ..
    const int tiles_m = (p.M + 255) / 256, tiles_n = (p.N + 255) / 256;
    const int padded = ((tiles_m + 7) / 8) * 8 * tiles_n;
    const int64_t z = blockIdx.z;
    const bf16_t* __restrict__ A = p.A + z * p.strideA;
    const bf16_t* __restrict__ W = p.W + z * p.strideW;
    const int fi = lane & 15, fg = lane >> 4;
    const int nk = (p.K + BK2 - 1) / BK2;

    // one wave-instruction = 16 rows x 64 B; per K-tile 16 A pieces + 16 W pieces, 2 + 2 per wave
    const int lrow = lane >> 2, lslot = lane & 3;
    int prow[2], pchunk[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        prow[i] = (wave * 2 + i) * 16 + lrow;
        pchunk[i] = lslot ^ g2perm(prow[i]);
    }
    auto next_tile = [&](int L, int* tm, int* tn) {
        while (L < padded && !tile_map(L, tiles_m, tiles_n, tm, tn)) L += gridDim.x;
        if (p.rev && L < padded) *tm = tiles_m - 1 - *tm;
        return L;
    };
    int Li, tmi = 0, tni = 0, kti = 0, si = 0;
    const bool wpk = p.Wpk != nullptr && p.batch <= 1 && !(p.dbg & 16384);
    const int64_t wpk_kstride = (int64_t)p.N * 32;
    const bf16_t* asrc[2];
    const bf16_t* wsrc[2];
    auto set_issue_tile = [&]() {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            asrc[i] = A + (int64_t)min(tmi * 256 + prow[i], p.M - 1) * p.lda;
            // W pre-packed per K-step (launch_pack_w_kstep: Wpk[ks][n][physical chunk][8]): the piece of 16 rows is one contiguous KB
            if (wpk) wsrc[i] = p.Wpk + ((int64_t)min(tni * 256 + prow[i], p.N - 1) * 4 + lslot) * 8;
            else wsrc[i] = W + (int64_t)min(tni * 256 + prow[i], p.N - 1) * p.ldw;
        }
    };
    // one K-tile = 4 pieces per wave (A0, W0, A1, W1); piece q of the issue cursor's K-tile, then advance() moves the cursor
    auto issue_piece = [&](int q) {
        if (STAMPS && (p.dbg & 4)) return;          // development (stamps build only): the loop without its transfers (results are garbage)
        char* sa = smem + si * P2_STAGE;
        const int i = q >> 1;
        const int k = kti * BK2 + pchunk[i] * 8;
        if (q & 1) __builtin_amdgcn_global_load_lds((gptr_t)(wpk ? wsrc[i] + (int64_t)kti * wpk_kstride : wsrc[i] + k), (lptr_t)(sa + A_BYTES + (wave * 2 + i) * 1024), 16, 0, 0);
        else __builtin_amdgcn_global_load_lds((gptr_t)(asrc[i] + (k < p.K ? k : 0)), (lptr_t)(sa + (wave * 2 + i) * 1024), 16, 0, 0);   // K tail of A: any finite data, W supplies the zeros
    };
    auto advance = [&]() {
        si = (si + 1) & (P2_NST - 1);
        if (++kti == nk) {
            kti = 0;
            Li = next_tile(Li + gridDim.x, &tmi, &tni);
            if (Li < padded) set_issue_tile();
        }
    };
    auto issue = [&]() {
#pragma unroll
        for (int q = 0; q < 4; ++q) issue_piece(q);
        advance();
    };
    Li = next_tile(blockIdx.x, &tmi, &tni);
    if (Li >= padded) return;                      // block-uniform
    if (p.dbg >> 21) {      // development (tools/gemm_bench.py DBGS; GEMM micro-benchmarks only: the decoder reads these bits too): late start of some workgroups, as in gemm_rowln_kernel
        const int units = (p.dbg >> 21) & 127, mode = (p.dbg >> 28) & 3;
        int cnt = 0, a_, b_;
        for (int L = blockIdx.x; L < padded; L += gridDim.x) cnt += tile_map(L, tiles_m, tiles_n, &a_, &b_) ? 1 : 0;
        const int cmax = (tiles_m * tiles_n + (int)gridDim.x - 1) / (int)gridDim.x;
        const bool late = mode == 0 ? cnt < cmax : mode == 1 ? ((blockIdx.x >> 3) & 1) : (cnt < cmax || ((blockIdx.x >> 3) & 1));
        if (late) {
            const unsigned long long t0 = __builtin_amdgcn_s_memtime();
            while (__builtin_amdgcn_s_memtime() - t0 < (unsigned long long)units * 4096ull) __builtin_amdgcn_s_sleep(32);
        }
    }
    set_issue_tile();
    int Lc = Li, tmc = tmi, tnc = tni, ktc = 0, sc = 0;

    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float4 bias4[4];
    auto load_bias = [&](int tn) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = tn * 256 + wn * 64 + j * 16 + fg * 4;
            // unconditional load from a clamped address + select: a load inside a branch is waited for on its own with vmcnt(0), which
            // here also drains the LDS-DMA pieces in flight
            const float4 bv = *reinterpret_cast<const float4*>((p.bias ? p.bias + z * p.strideBias : reinterpret_cast<const float*>(p.W)) + min(n, p.N - 4));
            bias4[j] = (p.bias && n + 3 < p.N) ? bv : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    load_bias(tnc);
    auto wait_next = [&](int ahead) {               // all K-tiles older than the (ahead - 1) youngest ones have landed (4 DMAs per tile and wave)
        if (ahead >= 3) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if (ahead == 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    };
    // STAGGERED: the two wave groups (rows 0-127: waves 0-3, rows 128-255: waves 4-7; wave w and w + 4 share a SIMD) run half an
    // iteration apart.  In every half-step one group of each SIMD feeds the matrix core with the 32 MFMAs of its K-tile while the
    // other one does the memory work - 12 fragment reads of the next K-tile and 4 LDS-DMA issues - behind them:
    //     half-step 2t     : group 0  MFMA(t)                    | group 1  read(t), issue(t + 3)
    //     half-step 2t + 1 : group 0  read(t + 1), issue(t + 3)  | group 1  MFMA(t)
    // The stage of K-tile t - 1 is free from half-step 2t - 1 on (group 1 read it in 2t - 2), which is when tile t + 3 goes into it;
    // tile t + 1 was issued in half-steps 2t - 4 / 2t - 3 and must have landed (own pieces, counted vmcnt) before the barrier that
    // ends half-step 2t.
    const int grp = wave >> 2;
    unsigned long long ts[6] = {0, 0, 0, 0, 0, 0}, tprev = 0;
#define P2S_STAMP(k) do { if (STAMPS) { const unsigned long long _n = __builtin_amdgcn_s_memtime(); ts[k] += _n - tprev; tprev = _n; } } while (0)
    int issued = 0;                                 // K-tiles of the stream this wave has issued
    int computed = 0;                               // K-tiles of the stream consumed so far (index of the current one)
    for (int i = 0; i < 3 && Li < padded; ++i) { issue(); ++issued; }
    // tile 0 landed (own pieces) -> barrier -> group 0 pre-reads its fragments of tile 0
    if (issued >= 3) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (issued == 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    op16x8 af[8], wf[4];
    auto read_frags = [&](int stage) {
        const char* sa = smem + stage * P2_STAGE;
        const char* sw = sa + A_BYTES;
#pragma unroll
        for (int j = 0; j < 4; ++j) wf[j] = *reinterpret_cast<const op16x8*>(sw + swz2(wn * 64 + j * 16 + fi, fg));
#pragma unroll
        for (int i = 0; i < 8; ++i) af[i] = *reinterpret_cast<const op16x8*>(sa + swz2(wm * 128 + i * 16 + fi, fg));
    };
    // The MFMA cluster must stay inside its half-step: MFMAs touch no memory, so hipcc is free to move them across the raw barriers and
    // the inline-asm waits (it did, depending on unrelated edits: 305 us <-> 335 us on the fc1 shape).  s_setprio around the cluster
    // keeps it together (cdna_hip_programming.md T5) and sched_barrier(0) pins its place.
    auto mfma_tile = [&]() {
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[i][j] = MFMA_16x16x32(wf[j], af[i], acc[i][j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
    };
    auto wait_tile = [&](int t) {                   // own pieces of stream tile t have landed: everything but the tiles issued after it
        const int younger = issued - (t + 1);
        if (younger >= 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if (younger == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    };
    auto epilogue = [&]() {
        const int m0 = tmc * 256, n0 = tnc * 256;
        bf16_t* Cb = p.Cb + z * p.strideCb;
        const uint32_t tb_a = (uint32_t)(uintptr_t)(lptr_t)(epi_lds + wave * 2048);
        const uint32_t tb_r0 = tb_a + (lane >> 3) * 128 + (((lane & 7) ^ ((lane >> 3) & 7)) << 4);   // rows 0-7; rows 8-15 are +1024 (same swizzle)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float v[4] = {acc[i][j][0] + bias4[j].x, acc[i][j][1] + bias4[j].y, acc[i][j][2] + bias4[j].z, acc[i][j][3] + bias4[j].w};
                if (p.act == ACT_GELU) {                       // packed form: the same arithmetic as gelu_erf, two values per VALU issue slot
                    const f32x2 g0 = gelu_erf2((f32x2){v[0], v[1]}), g1 = gelu_erf2((f32x2){v[2], v[3]});
                    v[0] = g0.x; v[1] = g0.y; v[2] = g1.x; v[3] = g1.y;
                }                                              // (only ACT_NONE / ACT_GELU are routed to this kernel)
                const int chunk = j * 2 + (fg >> 1);
                // LDS traffic of the epilogue is inline asm: hipcc orders every VISIBLE ds access behind the direct-to-LDS loads in
                // flight with s_waitcnt vmcnt(0), which would also drain the stores of the previous rows
                const uint64_t pk = ((uint64_t)pack_op16(v[2], v[3]) << 32) | pack_op16(v[0], v[1]);
                asm volatile("ds_write_b64 %0, %1" ::"v"(tb_a + fi * 128 + ((chunk ^ (fi & 7)) << 4) + (fg & 1) * 8), "v"(pk) : "memory");
                acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
            }
            u32x4 val0, val1;
            asm volatile("s_waitcnt lgkmcnt(0)\n\tds_read_b128 %0, %2\n\tds_read_b128 %1, %3\n\ts_waitcnt lgkmcnt(0)"
                         : "=&v"(val0), "=&v"(val1) : "v"(tb_r0), "v"(tb_r0 + 1024) : "memory");
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                const int row = it * 8 + (lane >> 3), chunk = lane & 7;
                const int m = m0 + wm * 128 + i * 16 + row, n = n0 + wn * 64 + chunk * 8;
                if (m < p.M && n < p.N) {
                    // nontemporal: the qkv / hidden tensors of a 21-crop pass (297 / 396 MB in stage 3) are larger than the Infinity Cache;
                    // written normally they push the fp32 residual stream out of it before the row-owner GEMMs read it back
                    // (same-box A/B: GEMM class 66.85 -> 63.87 ms, Hiera attention 11.04 -> 10.67 ms per slice; fc1 alone: no change)
                    __builtin_nontemporal_store(it ? val1 : val0, reinterpret_cast<u32x4*>(Cb + (int64_t)m * p.ldcb + n));
                }
            }
        }
        ktc = 0;
        Lc = next_tile(Lc + gridDim.x, &tmc, &tnc);
        if (Lc < padded) load_bias(tnc);
    };
    if (STAMPS) tprev = __builtin_amdgcn_s_memtime();
    // two straight-line loops (one per group) with the same barrier sequence: a single loop that branches on the group inside
    // every half-step made hipcc spill 215 VGPRs
    // (All four LDS-DMA issues stay in the memory half-step: moving two of them between the MFMAs made the pieces land later and
    // the MFMA half-step longer - 330 us instead of 303 us on the fc1 shape.)
    if (grp == 0) {
        read_frags(0);
        while (Lc < padded) {
            const int t = computed;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            mfma_tile();                                    // half-step 2t
            P2S_STAMP(0);
            wait_tile(t + 1);
            P2S_STAMP(1);
            __builtin_amdgcn_s_barrier();
            P2S_STAMP(2);
            sc = (sc + 1) & (P2_NST - 1);
            read_frags(sc);                                 // half-step 2t + 1: fragments of the next stream tile (a surplus read of a stale stage at the stream's end is harmless)
            if (Li < padded) { issue(); ++issued; }
            P2S_STAMP(3);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            ++computed;
            if (++ktc == nk) { epilogue(); P2S_STAMP(4); }
            P2S_STAMP(5);
        }
    } else {
        while (Lc < padded) {
            const int t = computed;
            read_frags(sc);                                 // half-step 2t
            if (Li < padded) { issue(); ++issued; }
            P2S_STAMP(0);
            wait_tile(t + 1);
            P2S_STAMP(1);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            P2S_STAMP(2);
            mfma_tile();                                    // half-step 2t + 1
            P2S_STAMP(3);
            __builtin_amdgcn_s_barrier();
            sc = (sc + 1) & (P2_NST - 1);
            ++computed;
            if (++ktc == nk) { epilogue(); P2S_STAMP(4); }
            P2S_STAMP(5);
        }
    }
    if (STAMPS && lane == 0)
        for (int k = 0; k < 6; ++k) p.stamps[((int64_t)blockIdx.x * 8 + wave) * 6 + k] = ts[k];
}


// ------------------------------------------------------------------------------------------------
#define GS_LDS_128 (3 * (128 * BK * 2 + BN * BK * 2) + 4 * 2048)
#define GS_LDS_256 (3 * (256 * BK * 2 + BN * BK * 2) + 8 * 2048)

const char* gemm_init_device() {
    hipError_t st = hipSuccess;
    if (st == hipSuccess) st = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_bf16_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, 4 * TILE_BYTES);
    if (st == hipSuccess) st = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_bf16_glds_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, GS_LDS_128);
    if (st == hipSuccess) st = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_bf16_glds_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, GS_LDS_256);
    if (st == hipSuccess) st = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_bf16_glds2_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, G2_LDS);
    if (st == hipSuccess) st = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_bf16_p256s_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, P2_LDS);
    if (st == hipSuccess) st = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_bf16_p256s_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, P2_LDS);

    return st == hipSuccess ? nullptr : hipGetErrorString(st);
}

const char* launch_gemm(const GemmParams& p_in, hipStream_t stream) {
    GemmParams p = p_in;
    p.dbg = g_saber_debug_flags;
    p.stamps = g_saber_stamp_buf;
    if (p.M <= 0 || p.N <= 0 || p.K <= 0) return "gemm: empty problem";
    if ((p.K & 7) || (p.lda & 7) || (p.ldw & 7)) return "gemm: K, lda, ldw must be multiples of 8";
    if (((uintptr_t)p.A & 15) || ((uintptr_t)p.W & 15)) return "gemm: A/W must be 16-byte aligned";
    if ((p.strideA & 7) || (p.strideW & 7)) return "gemm: batch strides must be multiples of 8";
    if ((p.N & 3) == 0) {
        if ((p.Cf && ((p.ldcf & 3) || ((uintptr_t)p.Cf & 15))) || (p.Cb && ((p.ldcb & 3) || ((uintptr_t)p.Cb & 7))) ||
            (p.res && ((p.ldres & 3) || ((uintptr_t)p.res & 15))) || (p.bias && ((uintptr_t)p.bias & 15)))
            return "gemm: output/residual/bias alignment";
    }
    if (p.pool4 && (p.M & 3)) return "gemm: pool4 needs M % 4 == 0";
    const int tiles = ((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN);
    auto padded = [](int tiles_m, int tiles_n) { return ((tiles_m + 7) / 8) * 8 * tiles_n; };
    dim3 grid(padded((p.M + BM - 1) / BM, (p.N + BN - 1) / BN), 1, p.batch > 0 ? p.batch : 1);
    const bool direct_ok = p.w_kpad || (p.K % BK) == 0;
    const int tiles256 = ((p.M + 255) / 256) * ((p.N + BN - 1) / BN);
    const bool bf16_only = p.Cb && !p.Cf && !p.res && !p.pool4 && (p.N & 7) == 0 && (p.ldcb & 7) == 0 && grid.z == 1 && (p.act == ACT_NONE || p.act == ACT_GELU);
    const int tiles_p2 = ((p.M + 255) / 256) * ((p.N + 255) / 256);
    // (the one-wave-per-SIMD experiment of round 3, DESIGN.md section 4, is parked in tools/experiments/gemm_w1.hip: it lost its A/B)
    // (round 4's one-wave-per-SIMD direct-to-LDS kernel, tools/experiments/gemm_w1d.hip, matched this kernel's K-loop rate and lost on its epilogue: DESIGN.md section 8)
    if (direct_ok && bf16_only && ((tiles_p2 >= 1024 && (p.N >= 1024 || (p.act == ACT_NONE && p.N >= 384)) && !(p.dbg & 64)) || (p.dbg & 128))) {
        // widest bf16-output GEMMs (qkv, fc1 of stages 2-3): persistent 256x256 tiles, one workgroup per CU
        // (round 5's one-wave-per-SIMD kernel whose epilogue runs under the NEXT tile's K loop, tools/experiments/gemm_w1e.hip: parity-green and
        // bit-identical to this kernel, 5-27 % ahead of it in tools/gemm_bench.py, 5-10 % BEHIND it inside the engine - DESIGN.md section 8)
        const int slots = padded((p.M + 255) / 256, (p.N + 255) / 256);
        // (the 32x32x16 form of this kernel without staggered groups, round 3's opt-in SABER_AMD_P256X, lost its A/B inside the slice and breaks the
        // batch-size invariance of the features: parked in tools/experiments/gemm_p256x.hip)
        if (p.stamps) hipLaunchKernelGGL(gemm_bf16_p256s_kernel<true>, dim3(slots < 256 ? slots : 256), dim3(512), P2_LDS, stream, p);   // development build with cycle stamps
        else hipLaunchKernelGGL(gemm_bf16_p256s_kernel<false>, dim3(slots < 256 ? slots : 256), dim3(512), P2_LDS, stream, p);
    } else if (direct_ok && tiles256 >= 512 && !(p.dbg & 16)) {
        // two co-resident workgroups per CU: one's epilogue overlaps the other's main loop
        hipLaunchKernelGGL(gemm_bf16_glds2_kernel, dim3(padded((p.M + 255) / 256, (p.N + BN - 1) / BN), 1, grid.z), dim3(512), G2_LDS, stream, p);
    } else if (direct_ok && tiles256 >= 256) {
        // big problems: 256x128 tiles, 8 waves, operands straight into a 3-stage LDS ring
        {
            const int slots = padded((p.M + 255) / 256, (p.N + BN - 1) / BN);
            hipLaunchKernelGGL(gemm_bf16_glds_kernel<4>, dim3(slots < 256 ? slots : 256, 1, grid.z), dim3(512), GS_LDS_256, stream, p);   // persistent: one block per CU
        }
    } else if (tiles >= 384) {
        hipLaunchKernelGGL(gemm_bf16_kernel<4>, grid, dim3(256), 4 * TILE_BYTES, stream, p);
    } else {
        // small problems (token-side GEMMs of the decoder): 64x64 tiles give 4x the blocks
        const int tiles64 = ((p.M + 63) / 64) * ((p.N + 63) / 64);
        hipLaunchKernelGGL(gemm_bf16_kernel<2>, dim3(padded((p.M + 63) / 64, (p.N + 63) / 64), 1, grid.z), dim3(256), 2 * TILE_BYTES, stream, p);
    }
    return nullptr;
}
