// Parked in round 5 (VERDICT r04 housekeeping): cut out of saber_amd/csrc/gemm.hip, where it was an opt-in (SABER_AMD_P256X=1) that lost its A/B
// inside the slice (145.8 vs 145.5 ms) and whose fp32 summation order differs from the 16x16x32 kernels smaller batches run on.  It uses
// gemm.hip's helpers (tile_map, g2perm, the P2 ring constants): to revive, paste it back above `#define GS_LDS_128`, restore the two
// hipFuncSetAttribute lines in gemm_init_device and the SABER_AMD_P256X branch of launch_gemm's 256 x 256 path (git show 0b050aa:saber_amd/csrc/gemm.hip).
// The same tile, ring and persistent stream on v_mfma_f32_32x32x16_bf16, WITHOUT the staggered wave groups (gemm_bf16_p256x_kernel).
// Measured first (tools/probes/mfma_issue_probe.py, tools/gemm_stamps.py): a 32x32x16 accumulation stream runs at 97-98 % of the MFMA peak
// from one wave with any number of accumulators, and the staggered kernel spends 2 290 cycles per K-tile against 1 024 of MFMA - its
// fragment-read half-steps (330-790 cycles) and barriers, not its transfers (2 044 without them), are what is left.  Here a wave keeps the
// matrix core busy by itself: per K = 16 sub-step 8 MFMAs of 32 cycles, and in their issue shadow the 6 ds_read_b128 of the NEXT sub-step's
// fragments (second register set) and 2 of the K-tile's 4 LDS-DMA issues; the two waves of a SIMD just share the pipe.  One barrier per
// K-tile, in its middle: by then the next K-tile's stage must have landed (counted vmcnt), because the second sub-step pre-reads it.
// Wave = 128 x 64 = 4 x 2 tiles of 32 x 32, swapped operands (W rows = MFMA A operand): D[row n][col m], a lane owns row m = lane & 31 of a
// tile and 16 of its 32 columns n = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5).
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define PX_LDS (P2_NST * P2_STAGE + 8 * 4096)

template <bool STAMPS>
__global__ __launch_bounds__(512) void gemm_bf16_p256x_kernel(GemmParams p) {
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
    const int r32 = lane & 31, h = lane >> 5;
    const int nk = (p.K + BK2 - 1) / BK2;

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
            if (wpk) wsrc[i] = p.Wpk + ((int64_t)min(tni * 256 + prow[i], p.N - 1) * 4 + lslot) * 8;
            else wsrc[i] = W + (int64_t)min(tni * 256 + prow[i], p.N - 1) * p.ldw;
        }
    };
    // one K-tile = 4 pieces per wave (A0, W0, A1, W1).  Past the end of the stream the pieces are still issued (from the last addresses, into
    // a stage nobody reads any more): a branch around them would split the loop body and hipcc then sinks the MFMAs below it.
    auto issue_piece = [&](int q) {
        if (STAMPS && (p.dbg & 4)) return;          // development (stamps build only): the loop without its transfers
        char* sa = smem + si * P2_STAGE;
        const int i = q >> 1;
        const int k = kti * BK2 + pchunk[i] * 8;
        if (q & 1) __builtin_amdgcn_global_load_lds((gptr_t)(wpk ? wsrc[i] + (int64_t)kti * wpk_kstride : wsrc[i] + k), (lptr_t)(sa + A_BYTES + (wave * 2 + i) * 1024), 16, 0, 0);
        else __builtin_amdgcn_global_load_lds((gptr_t)(asrc[i] + (k < p.K ? k : 0)), (lptr_t)(sa + (wave * 2 + i) * 1024), 16, 0, 0);
    };
    auto advance = [&]() {
        si = (si + 1) & (P2_NST - 1);
        if (Li >= padded) return;
        if (++kti == nk) {
            kti = 0;
            Li = next_tile(Li + gridDim.x, &tmi, &tni);
            if (Li < padded) set_issue_tile();
        }
    };
    Li = next_tile(blockIdx.x, &tmi, &tni);
    if (Li >= padded) return;                      // block-uniform
    set_issue_tile();
    int Lc = Li, tmc = tmi, tnc = tni, ktc = 0, sc = 0;

    // The accumulators START as the bias (a lane's 16 columns of a tile: n = 32 j + 8 g + 4 h + 0..3): no bias registers across the K loop,
    // no adds in the epilogue.  The next tile's bias is fetched at the START of an epilogue, ahead of its stores in the vmcnt queue.
    f32x16 acc[4][2];
    const float* bias_z = p.bias ? p.bias + z * p.strideBias : nullptr;
    auto bias_of = [&](int tn, float4 (&b)[2][4]) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int n = tn * 256 + wn * 64 + 32 * j + 8 * g + 4 * h;
                const float4 bv = *reinterpret_cast<const float4*>((bias_z ? bias_z : reinterpret_cast<const float*>(p.W)) + min(n, p.N - 4));     // unconditional load + select
                b[j][g] = (bias_z && n + 3 < p.N) ? bv : make_float4(0.f, 0.f, 0.f, 0.f);
            }
    };
    auto acc_init = [&](const float4 (&b)[2][4]) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int g = 0; g < 4; ++g) { acc[i][j][4 * g] = b[j][g].x; acc[i][j][4 * g + 1] = b[j][g].y; acc[i][j][4 * g + 2] = b[j][g].z; acc[i][j][4 * g + 3] = b[j][g].w; }
    };
    {
        float4 b0[2][4];
        bias_of(tnc, b0);
        acc_init(b0);
    }

    // fragment addresses: row of the operand panel, logical chunk 2 s + h of its 64-byte row (K-tile of 32 = two K = 16 sub-steps)
    const uint32_t lds0 = (uint32_t)(uintptr_t)(lptr_t)smem;
    uint32_t a_ad[2][4], w_ad[2][2];               // [sub-step][tile]
#pragma unroll
    for (int s = 0; s < 2; ++s) {
#pragma unroll
        for (int i = 0; i < 4; ++i) a_ad[s][i] = lds0 + swz2(wm * 128 + 32 * i + r32, 2 * s + h);
#pragma unroll
        for (int j = 0; j < 2; ++j) w_ad[s][j] = lds0 + A_BYTES + swz2(wn * 64 + 32 * j + r32, 2 * s + h);
    }
    u32x4 fa[2][4], fw[2][2];                      // two fragment sets
#define PX_RD(dst, addr) asm volatile("ds_read_b128 %0, %1" : "=v"(dst) : "v"(addr))
#define PX_MFMA(set, i, j) acc[i][j] = MFMA_32x32x16(__builtin_bit_cast(op16x8, fw[set][j]), __builtin_bit_cast(op16x8, fa[set][i]), acc[i][j], 0, 0, 0)
#define PX_FENCE() __builtin_amdgcn_sched_barrier(0)
    unsigned long long ts[6] = {0, 0, 0, 0, 0, 0}, tprev = 0;

    // prologue: three K-tiles in flight, the first one landed, fragments of its first sub-step in set 0
#pragma unroll
    for (int t = 0; t < 3; ++t) {
#pragma unroll
        for (int q = 0; q < 4; ++q) issue_piece(q);
        advance();
    }
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int i = 0; i < 4; ++i) PX_RD(fa[0][i], a_ad[0][i]);
#pragma unroll
    for (int j = 0; j < 2; ++j) PX_RD(fw[0][j], w_ad[0][j]);
    if (STAMPS) tprev = __builtin_amdgcn_s_memtime();
    while (Lc < padded) {
        const uint32_t so = sc * P2_STAGE;                              // stage of the K-tile being computed
        const uint32_t sn = ((sc + 1) & (P2_NST - 1)) * P2_STAGE;       // stage of the next one
        // ---- sub-step 0: products of (t, 0) from set 0; in their shadow the fragments of (t, 1) into set 1 and two pieces of K-tile t + 3
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+v"(fa[0][0]), "+v"(fa[0][1]), "+v"(fa[0][2]), "+v"(fa[0][3]), "+v"(fw[0][0]), "+v"(fw[0][1]));
        PX_FENCE();
        PX_MFMA(0, 0, 0); PX_FENCE(); PX_RD(fw[1][0], w_ad[1][0] + so); PX_FENCE();
        PX_MFMA(0, 0, 1); PX_FENCE(); PX_RD(fw[1][1], w_ad[1][1] + so); PX_FENCE();
        PX_MFMA(0, 1, 0); PX_FENCE(); PX_RD(fa[1][0], a_ad[1][0] + so); PX_FENCE();
        PX_MFMA(0, 1, 1); PX_FENCE(); PX_RD(fa[1][1], a_ad[1][1] + so); PX_FENCE();
        PX_MFMA(0, 2, 0); PX_FENCE(); PX_RD(fa[1][2], a_ad[1][2] + so); PX_FENCE();
        PX_MFMA(0, 2, 1); PX_FENCE(); PX_RD(fa[1][3], a_ad[1][3] + so); PX_FENCE();
        PX_MFMA(0, 3, 0); PX_FENCE(); issue_piece(0); PX_FENCE();
        PX_MFMA(0, 3, 1); PX_FENCE(); issue_piece(1); PX_FENCE();
        if (STAMPS) { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); ts[0] += n_ - tprev; tprev = n_; }
        // the next K-tile's stage has landed (own pieces: all but the 4 of K-tile t + 2 and the 2 just issued) -> visible to everybody
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        if (STAMPS) { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); ts[1] += n_ - tprev; tprev = n_; }
        __builtin_amdgcn_s_barrier();
        PX_FENCE();
        if (STAMPS) { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); ts[2] += n_ - tprev; tprev = n_; }
        // ---- sub-step 1: products of (t, 1) from set 1; fragments of (t + 1, 0) into set 0, the other two pieces
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+v"(fa[1][0]), "+v"(fa[1][1]), "+v"(fa[1][2]), "+v"(fa[1][3]), "+v"(fw[1][0]), "+v"(fw[1][1]));
        PX_FENCE();
        PX_MFMA(1, 0, 0); PX_FENCE(); PX_RD(fw[0][0], w_ad[0][0] + sn); PX_FENCE();
        PX_MFMA(1, 0, 1); PX_FENCE(); PX_RD(fw[0][1], w_ad[0][1] + sn); PX_FENCE();
        PX_MFMA(1, 1, 0); PX_FENCE(); PX_RD(fa[0][0], a_ad[0][0] + sn); PX_FENCE();
        PX_MFMA(1, 1, 1); PX_FENCE(); PX_RD(fa[0][1], a_ad[0][1] + sn); PX_FENCE();
        PX_MFMA(1, 2, 0); PX_FENCE(); PX_RD(fa[0][2], a_ad[0][2] + sn); PX_FENCE();
        PX_MFMA(1, 2, 1); PX_FENCE(); PX_RD(fa[0][3], a_ad[0][3] + sn); PX_FENCE();
        PX_MFMA(1, 3, 0); PX_FENCE(); issue_piece(2); PX_FENCE();
        PX_MFMA(1, 3, 1); PX_FENCE(); issue_piece(3); PX_FENCE();
        if (STAMPS) { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); ts[3] += n_ - tprev; tprev = n_; }
        advance();
        sc = (sc + 1) & (P2_NST - 1);
        if (++ktc < nk) continue;
        // ---------------- epilogue (bf16 out): activation, 32 rows x 64 columns of the wave at a time through its 4 KB of LDS
        {
            const int m0 = tmc * 256 + wm * 128, n0 = tnc * 256 + wn * 64;
            bf16_t* Cb = p.Cb + z * p.strideCb;
            const uint32_t tb = (uint32_t)(uintptr_t)(lptr_t)(epi_lds + wave * 4096);
            ktc = 0;
            Lc = next_tile(Lc + gridDim.x, &tmc, &tnc);
            float4 nb[2][4];
            bias_of(Lc < padded ? tnc : 0, nb);                   // the next tile's bias: ahead of this tile's stores in the vmcnt queue
#pragma unroll
            for (int i = 0; i < 4; ++i) {
#pragma unroll
                for (int j = 0; j < 2; ++j) {
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        float v0 = acc[i][j][4 * g + 0], v1 = acc[i][j][4 * g + 1], v2 = acc[i][j][4 * g + 2], v3 = acc[i][j][4 * g + 3];
                        if (p.act == ACT_GELU) { const f32x2 g0_ = gelu_erf2((f32x2){v0, v1}), g1_ = gelu_erf2((f32x2){v2, v3}); v0 = g0_.x; v1 = g0_.y; v2 = g1_.x; v3 = g1_.y; }
                        const uint64_t pk = ((uint64_t)pack_op16(v2, v3) << 32) | pack_op16(v0, v1);
                        const int chunk = 4 * j + g;              // 16-byte chunk of the 128-byte row; its 8-byte half is h
                        asm volatile("ds_write_b64 %0, %1" ::"v"(tb + r32 * 128 + ((chunk ^ (r32 & 7)) << 4) + h * 8), "v"(pk) : "memory");
                    }
                }
                u32x4 val[4];
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    const int row = it * 8 + (lane >> 3);
                    asm volatile("ds_read_b128 %0, %1" : "=v"(val[it]) : "v"(tb + row * 128 + (((lane & 7) ^ (row & 7)) << 4)) : "memory");
                }
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(val[0]), "+v"(val[1]), "+v"(val[2]), "+v"(val[3]));
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    const int row = it * 8 + (lane >> 3);
                    const int m = m0 + 32 * i + row, n = n0 + (lane & 7) * 8;
                    if (m < p.M && n < p.N) __builtin_nontemporal_store(val[it], reinterpret_cast<u32x4*>(Cb + (int64_t)m * p.ldcb + n));
                }
            }
            acc_init(nb);
        }
        if (STAMPS) { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); ts[4] += n_ - tprev; tprev = n_; }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // trailing (dummy) transfers must land before the LDS is released
    if (STAMPS && lane == 0)
        for (int k = 0; k < 6; ++k) p.stamps[((int64_t)blockIdx.x * 8 + wave) * 6 + k] = ts[k];
}


