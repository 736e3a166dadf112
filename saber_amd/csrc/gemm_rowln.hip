// Row-owner bf16 MFMA GEMM whose epilogue finishes the Hiera residual step AND the LayerNorm that follows it:
//
//     y  = A[M,K] . W[N,K]^T + bias + res          (fp32, written to Cf: the residual stream)
//     xn = LayerNorm(y) * gamma + beta             (bf16, written to ln_out: the A operand of the next GEMM)
//     [Cb = bf16(y)]                                (optional: the stage output that feeds the FPN neck)
//
// Replaces, per MultiScaleBlock of the Hiera trunk (third-party sam2 `MultiScaleBlock.forward`, reached from the reference
// through saber/adapters/sam2/predictor.py:70; HF restatement modeling_sam2.py:457-546), the pairs
//     attn.proj GEMM (+ shortcut)  ->  norm2         and        mlp.layers.1 GEMM (+ residual)  ->  norm1 of the NEXT block
// which round 1 ran as a GEMM with an fp32 + residual epilogue followed by a separate LayerNorm pass that re-read the fp32 rows
// it had just written (7.8 ms and 34 GB of HBM traffic per slice).  A LayerNorm needs whole rows, so the workgroup owns ALL N
// columns of its rows: tile = R rows x N columns with N = 144 WN, R = 64 WM, 8 waves as WM x WN, every wave a 64 x 144
// sub-tile (4 x 9 MFMA tiles of 16x16x32, 144 accumulator registers).  That covers the residual widths of Hiera-L stages 0-2
// (144, 288, 576) and of tiny/small stage 1-2 (192?  no: only multiples of 144, other trunks keep the unfused pair).
//
//   * A is streamed ONCE (no N tiling), W is re-read per tile from L2: (R + N) x 64 B per 32-deep K-step.
//   * W comes PRE-PACKED per K-step (pack_w_kstep_kernel, once per weight at engine finalize): Wpk[ks][n][32] with the LDS chunk
//     permutation already applied, so the W tile of a K-step is ONE contiguous 64 N-byte block and every direct-to-LDS instruction
//     copies 1 KB = 8 whole 128-B lines.  Row-major W costs 16 half-used lines per instruction (16 rows x 64 B), and the CU's
//     texture-address path works per line: the operand feed of the round-1 GEMMs sits at ~13 B/clk/CU for that reason.
//   * operands go global -> LDS directly (global_load_lds_dwordx4) into a 3-stage ring, K-steps of 32, 64-B LDS rows with the
//     chunk permutation of gemm_bf16_glds2_kernel (conflict-free ds_read_b128), counted vmcnt, one raw barrier per K-step;
//     13 fragment reads per 36 MFMAs per wave.
//   * persistent: min(tiles, CUs) workgroups walk the row tiles; the first two K-steps of the next tile are put in flight before
//     the epilogue of the current one.
//   * epilogue: bias into the accumulators, residual rows added, fp32 rows stored, two-pass LayerNorm statistics (row sums
//     reduced over the lanes of a row with permlane swaps, over the WN waves of a row through 2 x 2 KB of LDS), normalised
//     rows stored as bf16.
#include <type_traits>
#include "common.h"
#include "kernels.h"

typedef __attribute__((address_space(1))) const void* gptr_r;
typedef __attribute__((address_space(3))) void* lptr_r;
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

#define RL_BK 32
#define RL_PITCH 304                 // bytes per row of the bf16 transposition scratch (288 + 16)
#define RL_SCR (16 * RL_PITCH)       // per wave
// pieces of the K-step two ahead issued at the top of a K-step | behind its first MFMA group | behind the second (measured on the
// stage-2 shapes against all of them at the top: fc2 401 -> 376 us, proj 219 -> 211 us)
#define RL_SPLIT 1
#define RL_Q1 1
#define RL_Q2 3
// The timing-only switches of tools/rowln_bench.py (DBG=1024|2048|4096|8192|32768, the late-start sweep) exist in development builds only
// (make EXTRA=-DRL_DEV=1 BUILD=build_dev LIB=...): as run-time tests inside the K loop they cost a live register and a branch around every
// MFMA group, which pushed the fp16 build of the <2,4> configuration into spilling inside the loop.
#ifndef RL_DEV
#define RL_DEV 0
#endif
#define RL_DBG(bit) (RL_DEV && (p.dbg & (bit)))
#ifndef RL_ANTIPHASE
#define RL_ANTIPHASE 1               // the two waves of a SIMD issue their transfers at different points of the K-step (0: the lockstep loop of round 2)
#endif
__device__ __forceinline__ int rl_perm(int row) { return (0x78 >> (((row >> 2) & 3) * 2)) & 3; }   // {0, 2, 3, 1}: gemm.hip g2perm
__device__ __forceinline__ int rl_swz(int row, int chunk) { return row * 64 + ((chunk ^ rl_perm(row)) << 4); }

template <int WM, int WN> struct RowLnCfg {
    static constexpr int R = 64 * WM, N = 144 * WN;
    static constexpr int RA = R / 16, RW = N / 16, PT = RA + RW;          // 1-KB pieces (16 rows x 64 B) of one K-step
    static constexpr int STAGE = (R + N) * 64;
    static constexpr int STAT = R * WN * 4;                                // one statistics buffer
    // transposition scratch of the bf16 rows (RL_SCR per wave): ring stage 2 when an eighth of it is enough, else behind the statistics
    static constexpr bool SCR_IN_RING = STAGE / 8 >= RL_SCR;
    static constexpr int SCR_OFF = SCR_IN_RING ? 2 * STAGE : 3 * STAGE + 2 * STAT;
    static constexpr int LDS = 3 * STAGE + 2 * STAT + (SCR_IN_RING ? 0 : 8 * RL_SCR);
    static constexpr int NQ = (PT + 7) / 8;                                // pieces per wave (upper bound)
};

template <int WM, int WN>
__global__ __launch_bounds__(512) void gemm_rowln_kernel(GemmParams p) {
    using CF = RowLnCfg<WM, WN>;
    constexpr int R = CF::R, RA = CF::RA, RW = CF::RW, STAGE = CF::STAGE, NQ = CF::NQ;
    static_assert(WM * WN == 8, "8 waves");
    static_assert(RA % 8 == 0, "A pieces must split evenly over the waves");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* stat0 = reinterpret_cast<float*>(smem + 3 * STAGE);      // two statistics buffers of R x WN floats
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int fi = lane & 15, fg = lane >> 4;
    const int tiles = (p.M + R - 1) / R;
    const int nk = (p.K + RL_BK - 1) / RL_BK;

    // ---- direct-to-LDS pieces (16 rows x 64 B = 1 KB per wave-instruction): piece = wave + 8 q; pieces < RA are A rows, the rest W
    // rows.  Both operands go through buffer descriptors: ONE 32-bit offset register per piece, the K-step in the scalar offset, rows
    // beyond M (and the K tail of A beyond the buffer) read as zero by the range check.  Every wave issues exactly NQ pieces per
    // K-step (the surplus ones repeat the last W piece: same bytes, same place) so that the counted vmcnt is the same for all waves.
    // Per-lane offsets of the pieces are NOT kept in registers across the K loop: issue() re-derives them from the lane id (v_mbcnt through
    // volatile asm, so that the compiler can neither hoist the arithmetic out of the loop nor keep its results alive) - ~12 VALU
    // instructions per call against 36 MFMAs per K-step.  With the offsets live (first one register per piece, then two in all) the fp16
    // build of the <2,4> / <4,2> configurations (144 accumulator registers of 256) spilled them and reloaded them in front of every issue,
    // behind an s_waitcnt vmcnt(0) that drained the LDS-DMA ring: 584 us against 349 us on the fc2 shape (round 4; the bf16 build happened
    // to keep them).  The W pieces' index is wave-uniform and goes into the scalar offset.
    int ldsoff[NQ], woff_s[NQ];                               // wave-uniform (scalar registers)
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        if (q * 8 < RA) {
            ldsoff[q] = (wave + 8 * q) * 1024;
            woff_s[q] = 0;
        } else {
            const int wp = min(wave + 8 * q - RA, RW - 1);      // wave-uniform
            woff_s[q] = wp * 1024;
            ldsoff[q] = (RA + wp) * 1024;
        }
    }
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.Wpk, 0, (int)((uint32_t)nk * (uint32_t)CF::N * 64u), 0x00020000);
    // (the voffset argument is cast explicitly: voff[] has a template-dependent bound, which makes voff[q] type-dependent, and a
    // type-dependent argument of this builtin makes the HOST instantiation of the kernel template fail silently - no host stub)
    // q0 .. q1: which of the wave's NQ pieces (the K-step two ahead is issued in three parts BETWEEN the MFMA groups of the current one:
    // an LDS-DMA issue stalls the issuing wave 60-185 cycles, and as a burst at the K-step's start - all eight waves at once, right
    // behind the barrier - the matrix cores idle through every one of them)
    auto issue = [&](const __amdgpu_buffer_rsrc_t& wrs, int tile, int kt, int q0 = 0, int q1 = 100) {
        const int arows = min(R, p.M - tile * R);
        const __amdgpu_buffer_rsrc_t arsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(p.A + (int64_t)tile * R * p.lda), 0, (int)((uint32_t)arows * (uint32_t)p.lda * 2u), 0x00020000);
        char* st = smem + (kt % 3) * STAGE;
        uint32_t ln;
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(ln));      // lane id, opaque (see above)
        const int lr = (int)(ln >> 2);
        const uint32_t voff_a = (uint32_t)((wave * 16 + lr) * (int)p.lda * 2) + (((ln & 3u) ^ (uint32_t)rl_perm(lr)) << 4);   // logical 16-B chunk that lands at physical slot lane & 3
        const uint32_t voff_w = ln * 16u;                         // packed W: the K-step's tile is a straight copy
        const int qstride = 128 * (int)p.lda * 2;                 // bytes between the A pieces of one wave (8 pieces x 16 rows)
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            if (q < q0 || q >= q1) continue;
            // (the row part of an A piece stays in the VECTOR offset: the descriptor's range check - rows beyond M read as zero - looks at it)
            if (q * 8 < RA) __builtin_amdgcn_raw_ptr_buffer_load_lds(arsrc, (lptr_r)(st + ldsoff[q]), 16, (int)(voff_a + (uint32_t)(q * qstride)), kt * (RL_BK * 2), 0, 0);
            else __builtin_amdgcn_raw_ptr_buffer_load_lds(wrs, (lptr_r)(st + ldsoff[q]), 16, (int)voff_w, (int)(kt * (CF::N * 64) + woff_s[q]), 0, 0);      // (explicit casts: a type-dependent argument - woff_s has a template-dependent bound - silently drops the kernel's HOST stub)
        }
    };
    static_assert(NQ == 6 || NQ == 5, "piece counts handled: 5 or 6 per wave");

    int tl = blockIdx.x;                                   // position in this workgroup's walk; t = the tile (reversed walk: p.rev)
    if (tl >= tiles) return;
    auto tile_of = [&](int l) { return p.rev ? tiles - 1 - l : l; };
    // Workgroups whose walk is one tile shorter than the longest (tiles is rarely a multiple of the grid: 2.6 / 5.25 / 10.5 tiles per
    // workgroup on the 21-crop pass) start ~0.4 tile times late.  Every workgroup spends a tile time the same way - K loop (no HBM
    // traffic to speak of), then an epilogue that moves 737 KB - and all start together, so the epilogues of all 256 CUs meet in HBM
    // (5.7 TB/s while they last, nothing in between).  The late ones have a whole tile time of slack; their epilogues now fall into the
    // others' K loops and the others' into theirs: proj 183 -> 177 us, fc2 341 -> 316 us, stage 2 311 -> 297 / 468 -> 444 us,
    // stage 1 556 -> 535 / 807 -> 776 us (tools/rowln_bench.py; delaying workgroups WITH a full walk loses, as it must).
    {
        const int my_tiles = (tiles - 1 - (int)blockIdx.x) / (int)gridDim.x + 1, max_tiles = (tiles - 1) / (int)gridDim.x + 1;
        if (my_tiles < max_tiles && !RL_DBG(32768)) {
            unsigned long long late = 1080ull * (unsigned)nk + 40000ull;                // shader cycles: 0.4 x (2 700 per K-step + ~100 000 of epilogue)
            if (RL_DEV && ((p.dbg >> 21) & 127)) late = (unsigned long long)((p.dbg >> 21) & 127) * 4096ull;      // development: tools/rowln_bench.py DBGS sweep (micro-benchmark only: the decoder reads these bits too)
            const unsigned long long t0 = __builtin_amdgcn_s_memtime();
            while (__builtin_amdgcn_s_memtime() - t0 < late) __builtin_amdgcn_s_sleep(32);
        }
    }
    issue(wrsrc, tile_of(tl), 0);
    if (nk > 1) issue(wrsrc, tile_of(tl), 1);
    for (; tl < tiles; tl += gridDim.x) {
        const int t = tile_of(tl);
        const int m0 = t * R;
        f32x4 acc[4][9];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 9; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        // K-steps 0 (and 1) of this tile were issued before the previous tile's epilogue: everything has landed after vmcnt(0)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
#if RL_ANTIPHASE
        // The two waves of a SIMD (w and w + 4) take their transfer issues at different points of the K-step: waves 0-3 issue ahead of
        // each MFMA group, waves 4-7 behind the first and second.  An LDS-DMA issue holds its wave for as long as the L2 -> LDS path is
        // backed up (60-185 cycles); in lockstep both waves of the SIMD sat in their issues at the same time and the matrix core idled.
        // Same-box A/B against the lockstep loop (tools/rowln_bench.py, two builds interleaved): fc2 of stage 2 358 -> 340 us, of stage 1
        // 453 -> 439, stage 0 531 -> 484 / 764 -> 719 us; the K = 576 / K = 288 proj shapes level.  (Waves 0-3 issuing 3 + 3 pieces ahead
        // of the first two groups instead of 2 + 2 + 2: no difference.)
        // Two straight-line K loops (one per half) with the same barrier sequence: a branch inside one loop makes hipcc spill.
        auto kstep = [&](auto ph_tag, int kt) {
            constexpr bool PH = decltype(ph_tag)::value;
            const bool more = kt + 2 < nk;
            const char* sa = smem + (kt % 3) * STAGE;
            const char* sw = sa + RA * 1024;
            if (!PH && more) issue(wrsrc, t, kt + 2, 0, 2);
            // fragment addresses from a fresh (opaque) lane id: like the pieces' offsets they must not live across the K loop (see issue())
            uint32_t lk;
            asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lk));
            const int kfi = (int)(lk & 15u), kfg = (int)(lk >> 4);
            op16x8 af[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) af[i] = *reinterpret_cast<const op16x8*>(sa + rl_swz(wm * 64 + i * 16 + kfi, kfg));
#pragma unroll
            for (int jg = 0; jg < 3; ++jg) {
                op16x8 wf[3];
#pragma unroll
                for (int jj = 0; jj < 3; ++jj) wf[jj] = *reinterpret_cast<const op16x8*>(sw + rl_swz(wn * 144 + (jg * 3 + jj) * 16 + kfi, kfg));
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int jj = 0; jj < 3; ++jj)
                        if (!RL_DBG(8192)) acc[i][jg * 3 + jj] = MFMA_16x16x32(wf[jj], af[i], acc[i][jg * 3 + jj], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (more) {
                    if (!PH) { if (jg == 0) issue(wrsrc, t, kt + 2, 2, 4); else if (jg == 1) issue(wrsrc, t, kt + 2, 4, NQ); }
                    else { if (jg == 0) issue(wrsrc, t, kt + 2, 0, 3); else if (jg == 1) issue(wrsrc, t, kt + 2, 3, NQ); }
                }
            }
            if (more) { if (NQ == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); }
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        };
        if (wave < 4) { for (int kt = 0; kt < nk; ++kt) kstep(std::false_type{}, kt); }
        else { for (int kt = 0; kt < nk; ++kt) kstep(std::true_type{}, kt); }
#else
        for (int kt = 0; kt < nk; ++kt) {
#if RL_SPLIT
            if (kt + 2 < nk) issue(wrsrc, t, kt + 2, 0, RL_Q1);
#else
            if (kt + 2 < nk) issue(wrsrc, t, kt + 2);
#endif
            const char* sa = smem + (kt % 3) * STAGE;
            const char* sw = sa + RA * 1024;
            // fragments in three column groups of 3 tiles: 16 + 12 live operand registers instead of 52 (the kernel lives on 256 VGPRs
            // with 144 of them accumulators)
            uint32_t lk;
            asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lk));      // (see the anti-phase loop)
            const int kfi = (int)(lk & 15u), kfg = (int)(lk >> 4);
            op16x8 af[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) af[i] = *reinterpret_cast<const op16x8*>(sa + rl_swz(wm * 64 + i * 16 + kfi, kfg));
#pragma unroll
            for (int jg = 0; jg < 3; ++jg) {
                op16x8 wf[3];
#pragma unroll
                for (int jj = 0; jj < 3; ++jj) wf[jj] = *reinterpret_cast<const op16x8*>(sw + rl_swz(wn * 144 + (jg * 3 + jj) * 16 + kfi, kfg));
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int jj = 0; jj < 3; ++jj)
                        if (!RL_DBG(8192)) acc[i][jg * 3 + jj] = MFMA_16x16x32(wf[jj], af[i], acc[i][jg * 3 + jj], 0, 0, 0);
#if RL_SPLIT
                if (kt + 2 < nk) { if (jg == 0) issue(wrsrc, t, kt + 2, RL_Q1, RL_Q2); else if (jg == 1) issue(wrsrc, t, kt + 2, RL_Q2, NQ); }
#endif
            }
            // K-step kt + 1 has landed (this wave's pieces); the youngest (kt + 2) may stay in flight
            if (kt + 2 < nk) { if (NQ == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); }
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
#endif
        // ---------------- epilogue: lane owns rows m0 + wm*64 + i*16 + fi, columns wn*144 + j*16 + fg*4 .. +3
        // (opaque lane copy: keeps the epilogue's address arithmetic from being hoisted above the main loop, where every live register
        // costs a spill).  Residual, fp32 rows and bf16 rows go through buffer descriptors over this tile's rows: one 32-bit offset per
        // lane and tensor, the row group in the scalar offset, the column group in the immediate; rows beyond M are dropped by the range check.
        int lane_e = lane;
        asm volatile("" : "+v"(lane_e));
        const int efi = lane_e & 15, efg = lane_e >> 4;
        const int ncol = wn * 144 + efg * 4;
        const int trow = wm * 64 + efi;                       // row within the tile (+ 16 i)
        const int rows = min(R, p.M - m0);
        const float* rbase = p.res ? p.res : p.Cf;            // no residual: any readable fp32 (discarded)
        const int64_t ldr = p.res ? p.ldres : p.ldcf;
        const __amdgpu_buffer_rsrc_t rrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(rbase + (int64_t)m0 * ldr), 0, (int)((uint32_t)rows * (uint32_t)ldr * 4u), 0x00020000);
        const __amdgpu_buffer_rsrc_t crsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(p.Cf + (int64_t)m0 * p.ldcf), 0, (int)((uint32_t)rows * (uint32_t)p.ldcf * 4u), 0x00020000);
        const uint32_t roff = (uint32_t)((trow * (int)ldr + ncol) * 4), coff = (uint32_t)((trow * (int)p.ldcf + ncol) * 4);
        const uint32_t rstep = (uint32_t)(16 * (int)ldr * 4), cstep = (uint32_t)(16 * (int)p.ldcf * 4);
        const bool has_res = p.res != nullptr;
        const bool dbg_nores = RL_DBG(1024), dbg_nof32 = RL_DBG(2048), dbg_nobf = RL_DBG(4096);   // development: timing-only switches (tools/rowln_bench.py)
        // No direct-to-LDS load is in flight here (the last K-step ended with vmcnt(0)), so hipcc counts these loads instead of draining
        // the queue at every use; the residual rows of group i + 1 are requested before group i is added and stored.
        u32x4 ra[9], rb[9];
        auto load_group = [&](const __amdgpu_buffer_rsrc_t& rsr, int i, u32x4 (&rr)[9]) {
#pragma unroll
            for (int j = 0; j < 9; ++j) rr[j] = dbg_nores ? (u32x4){0u, 0u, 0u, 0u} : __builtin_amdgcn_raw_buffer_load_b128(rsr, roff + j * 64, i * rstep, 0);
        };
        float rs[4];
        auto finish_group = [&](const __amdgpu_buffer_rsrc_t& csr, int i, const u32x4 (&rr)[9]) {
            float s = 0.f;
#pragma unroll
            for (int j = 0; j < 9; ++j) {
                if (has_res) {
                    acc[i][j][0] += __uint_as_float(rr[j][0]); acc[i][j][1] += __uint_as_float(rr[j][1]);
                    acc[i][j][2] += __uint_as_float(rr[j][2]); acc[i][j][3] += __uint_as_float(rr[j][3]);
                }
                s += (acc[i][j][0] + acc[i][j][1]) + (acc[i][j][2] + acc[i][j][3]);
            }
#pragma unroll
            for (int j = 0; j < 9; ++j) {
                u32x4 v;
                v[0] = __float_as_uint(acc[i][j][0]); v[1] = __float_as_uint(acc[i][j][1]); v[2] = __float_as_uint(acc[i][j][2]); v[3] = __float_as_uint(acc[i][j][3]);
                if (!dbg_nof32) __builtin_amdgcn_raw_buffer_store_b128(v, csr, coff + j * 64, i * cstep, 0);
            }
            rs[i] = xor32_sum(xor16_sum(s));
        };
        load_group(rrsrc, 0, ra);
#pragma unroll
        for (int j = 0; j < 9; ++j) {
            const float4 b4 = p.bias ? *reinterpret_cast<const float4*>(p.bias + ncol + j * 16) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int i = 0; i < 4; ++i) { acc[i][j][0] += b4.x; acc[i][j][1] += b4.y; acc[i][j][2] += b4.z; acc[i][j][3] += b4.w; }
        }
        __builtin_amdgcn_sched_barrier(0);
        load_group(rrsrc, 1, rb);
        __builtin_amdgcn_sched_barrier(0);
        finish_group(crsrc, 0, ra);
        __builtin_amdgcn_sched_barrier(0);
        load_group(rrsrc, 2, ra);
        __builtin_amdgcn_sched_barrier(0);
        finish_group(crsrc, 1, rb);
        __builtin_amdgcn_sched_barrier(0);
        load_group(rrsrc, 3, rb);
        __builtin_amdgcn_sched_barrier(0);
        finish_group(crsrc, 2, ra);
        __builtin_amdgcn_sched_barrier(0);
        finish_group(crsrc, 3, rb);
        __builtin_amdgcn_sched_barrier(0);
        if (p.Cb) {
            const __amdgpu_buffer_rsrc_t brsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(p.Cb + (int64_t)m0 * p.ldcb), 0, (int)((uint32_t)rows * (uint32_t)p.ldcb * 2u), 0x00020000);
            const uint32_t boff = (uint32_t)((trow * (int)p.ldcb + ncol) * 2), bstep = (uint32_t)(16 * (int)p.ldcb * 2);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 9; ++j) {
                    u32x2 v;
                    v[0] = pack_op16(acc[i][j][0], acc[i][j][1]); v[1] = pack_op16(acc[i][j][2], acc[i][j][3]);
                    __builtin_amdgcn_raw_buffer_store_b64(v, brsrc, boff + j * 32, i * bstep, 0);
                }
        }
        // every wave is past its last fragment read (the K loop's final barrier): the ring is free.  The next tile's first K-steps go in
        // flight now, under the LayerNorm statistics, the bf16 stores and the drain of this tile's stores.
        if (tl + (int)gridDim.x < tiles) {
            issue(wrsrc, tile_of(tl + gridDim.x), 0);
            if (nk > 1) issue(wrsrc, tile_of(tl + gridDim.x), 1);
        }
        // mean / variance over the N columns of each row: the partial sums of the WN waves of a row meet in LDS.  LDS traffic and the
        // barrier are inline asm / raw: a visible ds access or __syncthreads() would drain the direct-to-LDS loads and all stores (vmcnt(0)).
        float mean[4], rstd[4];
        const uint32_t st_w = (uint32_t)(uintptr_t)(lptr_r)stat0 + (uint32_t)((trow * WN + wn) * 4);   // this wave's slot of row trow (+ 16 i rows)
        const uint32_t st_r = (uint32_t)(uintptr_t)(lptr_r)stat0 + (uint32_t)(trow * WN * 4);
        auto exchange = [&](int buf, float (&v)[4]) {        // v[i] <- sum over the WN waves of the row
            if (WN == 1) return;
            const uint32_t bo = (uint32_t)(buf * R * WN * 4);
            if (efg == 0) {
#pragma unroll
                for (int i = 0; i < 4; ++i) asm volatile("ds_write_b32 %0, %1" ::"v"(st_w + bo + i * 16 * WN * 4), "v"(v[i]) : "memory");
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (WN == 4) {
                    f32x4 q4;
                    asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(q4) : "v"(st_r + bo + i * 16 * WN * 4) : "memory");
                    v[i] = (q4[0] + q4[1]) + (q4[2] + q4[3]);
                } else {
                    f32x2 q2;
                    asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(q2) : "v"(st_r + bo + i * 16 * WN * 4) : "memory");
                    v[i] = q2[0] + q2[1];
                }
            }
        };
        exchange(0, rs);
#pragma unroll
        for (int i = 0; i < 4; ++i) mean[i] = rs[i] * (1.0f / CF::N);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float q = 0.f;
#pragma unroll
            for (int j = 0; j < 9; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) { acc[i][j][r] -= mean[i]; q = fmaf(acc[i][j][r], acc[i][j][r], q); }
            rs[i] = xor32_sum(xor16_sum(q));
        }
        exchange(1, rs);
#pragma unroll
        for (int i = 0; i < 4; ++i) rstd[i] = __builtin_amdgcn_rsqf(rs[i] * (1.0f / CF::N) + p.ln_eps);
        const __amdgpu_buffer_rsrc_t lrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(p.ln_out + (int64_t)m0 * p.ldln), 0, (int)((uint32_t)rows * (uint32_t)p.ldln * 2u), 0x00020000);
#pragma unroll
        for (int j = 0; j < 9; ++j) {
            const float4 g4 = *reinterpret_cast<const float4*>(p.ln_gamma + ncol + j * 16);
            const float4 be4 = *reinterpret_cast<const float4*>(p.ln_beta + ncol + j * 16);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                acc[i][j][0] = fmaf(acc[i][j][0] * rstd[i], g4.x, be4.x); acc[i][j][1] = fmaf(acc[i][j][1] * rstd[i], g4.y, be4.y);
                acc[i][j][2] = fmaf(acc[i][j][2] * rstd[i], g4.z, be4.z); acc[i][j][3] = fmaf(acc[i][j][3] * rstd[i], g4.w, be4.w);
            }
        }
        // bf16 rows leave through LDS: in the accumulator layout a store instruction covers 16 rows x 32 B (the same bytes cost twice
        // what the fp32 rows cost: tools/rowln_bench.py, DBG=4096 against DBG=2048); transposed, a lane stores 16 B of a 288-B row
        // segment, 3 rows per instruction.  The scratch is wave-private (16 rows x 304 B: the pitch spreads the 16 rows of a
        // ds_write_b64 over the banks), in ring stage 2 - the next tile's K-steps 0 and 1 are landing in stages 0 and 1, K-step 2 is
        // issued behind the barrier at the top of the next tile - or, where an eighth of a stage is too small, behind the statistics.
        const uint32_t scr = (uint32_t)(uintptr_t)(lptr_r)(smem + CF::SCR_OFF) + (uint32_t)(wave * RL_SCR);
        const uint32_t scr_w = scr + (uint32_t)(efi * RL_PITCH + efg * 8);
        const int lrow = lane_e / 18, lch = lane_e - 18 * lrow;                  // lanes 0..53: 3 rows x 18 chunks of 16 B
        const uint32_t scr_r = scr + (uint32_t)(lrow * RL_PITCH + lch * 16);
        const uint32_t loff = (uint32_t)(((wm * 64 + lrow) * (int)p.ldln + wn * 144) * 2 + lch * 16);
        const uint32_t lrowb = (uint32_t)((int)p.ldln * 2);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
            for (int j = 0; j < 9; ++j) {
                const uint64_t pk = ((uint64_t)pack_op16(acc[i][j][2], acc[i][j][3]) << 32) | pack_op16(acc[i][j][0], acc[i][j][1]);
                asm volatile("ds_write_b64 %0, %1" ::"v"(scr_w + j * 32), "v"(pk) : "memory");
            }
            u32x4 val[6];
#pragma unroll
            for (int r = 0; r < 6; ++r) asm volatile("ds_read_b128 %0, %1" : "=v"(val[r]) : "v"(scr_r + r * 3 * RL_PITCH) : "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(val[0]), "+v"(val[1]), "+v"(val[2]), "+v"(val[3]), "+v"(val[4]), "+v"(val[5]));
            if (!dbg_nobf && lane_e < 54) {
#pragma unroll
                for (int r = 0; r < 5; ++r) __builtin_amdgcn_raw_buffer_store_b128(val[r], lrsrc, loff, (i * 16 + r * 3) * lrowb, 0);
                if (lane_e < 18) __builtin_amdgcn_raw_buffer_store_b128(val[5], lrsrc, loff, (i * 16 + 15) * lrowb, 0);
            }
        }
    }
}

// Wpk[ks][n][physical chunk][8] <- W[n][32 ks + 8 chunk + e], physical chunk = chunk ^ rl_perm(n); k beyond ldw reads as zero
__global__ __launch_bounds__(256) void pack_w_kstep_kernel(const bf16_t* __restrict__ W, int ldw, int N, int nk, bf16_t* __restrict__ out) {
    const int64_t total = (int64_t)nk * N * 4;           // 16-B chunks
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        const int c = (int)(idx & 3);
        const int64_t rn = idx >> 2;
        const int n = (int)(rn % N), ks = (int)(rn / N);
        const int k = ks * 32 + c * 8;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (k + 8 <= ldw) v = *reinterpret_cast<const uint4*>(W + (int64_t)n * ldw + k);
        *reinterpret_cast<uint4*>(out + ((rn * 4) + (c ^ rl_perm(n))) * 8) = v;
    }
}
size_t gemm_rowln_packed_elems(int N, int K) { return (size_t)((K + 31) / 32) * N * 32; }
const char* launch_pack_w_kstep(const bf16_t* W, int ldw, int N, int K, bf16_t* out, hipStream_t s) {
    if ((ldw & 7) || ldw < K) return "pack_w_kstep: ldw must be a multiple of 8 and >= K";
    const int nk = (K + 31) / 32;
    const int64_t total = (int64_t)nk * N * 4;
    hipLaunchKernelGGL(pack_w_kstep_kernel, dim3((unsigned)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048)), dim3(256), 0, s, W, ldw, N, nk, out);
    return nullptr;
}

const char* gemm_rowln_init_device() {
    hipError_t st = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_rowln_kernel<2, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, RowLnCfg<2, 4>::LDS);
    if (st == hipSuccess) st = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_rowln_kernel<4, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, RowLnCfg<4, 2>::LDS);
    if (st == hipSuccess) st = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_rowln_kernel<8, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, RowLnCfg<8, 1>::LDS);
    return st == hipSuccess ? nullptr : hipGetErrorString(st);
}

bool gemm_rowln_supported(const GemmParams& p) {
    return (p.N == 144 || p.N == 288 || p.N == 576) && p.Wpk && p.Cf && p.ln_out && p.ln_gamma && p.ln_beta && p.batch <= 1 && !p.pool4 &&
           p.act == ACT_NONE && p.res_shift == 0 && p.res_mod == 0;
}

const char* launch_gemm_rowln(const GemmParams& p_in, hipStream_t stream) {
    GemmParams p = p_in;
    p.dbg = g_saber_debug_flags;
    if (!gemm_rowln_supported(p)) return "gemm_rowln: unsupported problem (N must be 144, 288 or 576; fp32 + LayerNorm outputs required)";
    if (p.M <= 0 || p.K <= 0) return "gemm_rowln: empty problem";
    if ((p.K & 7) || (p.lda & 7)) return "gemm_rowln: K and lda must be multiples of 8";
    if (((uintptr_t)p.A & 15) || ((uintptr_t)p.Wpk & 15) || ((uintptr_t)p.Cf & 15) || (p.ldcf & 3) || ((uintptr_t)p.ln_out & 7) || (p.ldln & 3) ||
        (p.res && (((uintptr_t)p.res & 15) || (p.ldres & 3))) || (p.bias && ((uintptr_t)p.bias & 15)) || (p.Cb && (((uintptr_t)p.Cb & 7) || (p.ldcb & 3))) ||
        ((uintptr_t)p.ln_gamma & 15) || ((uintptr_t)p.ln_beta & 15))
        return "gemm_rowln: operand alignment";
    constexpr int lds24 = RowLnCfg<2, 4>::LDS, lds42 = RowLnCfg<4, 2>::LDS, lds81 = RowLnCfg<8, 1>::LDS;
    const int n_cu = saber_cu_count();
    if (p.N == 576) {
        const int tiles = (p.M + 127) / 128;
        hipLaunchKernelGGL((gemm_rowln_kernel<2, 4>), dim3(tiles < n_cu ? tiles : n_cu), dim3(512), lds24, stream, p);
    } else if (p.N == 288) {
        const int tiles = (p.M + 255) / 256;
        hipLaunchKernelGGL((gemm_rowln_kernel<4, 2>), dim3(tiles < n_cu ? tiles : n_cu), dim3(512), lds42, stream, p);
    } else {
        const int tiles = (p.M + 511) / 512;
        hipLaunchKernelGGL((gemm_rowln_kernel<8, 1>), dim3(tiles < n_cu ? tiles : n_cu), dim3(512), lds81, stream, p);
    }
    return nullptr;
}
