// 16-bit-output GEMM, one wave per SIMD, whose EPILOGUE RUNS UNDER THE NEXT TILE'S K LOOP (round 5).
//
// Replaces gemm_bf16_p256s_kernel for the widest 16-bit-output GEMMs of the Hiera trunk (attn.qkv and mlp.layers.0 of the third-party sam2
// MultiScaleBlock in stages 2-3, reached through saber/adapters/sam2/predictor.py:70).  Shape and K loop are round 4's experiment
// (tools/experiments/gemm_w1d.hip, which reproduced the vendor library's MT256x256x64 / four-wave / direct-to-LDS structure): a 256 x 256
// tile on FOUR waves of 128 x 128 (256 fp32 accumulators in the AccVGPR half of the register file, pinned there by inline-asm MFMAs), both
// operands global -> LDS directly through a 4-stage ring of 32-deep K-steps with three K-steps in flight, K-step-packed W, inline-asm
// fragment reads, one counted vmcnt + one barrier per K-step.  That loop ran at the staggered 8-wave kernel's rate (0.74 us per K-step of
// a 256 x 256 tile) and lost on what sits OUTSIDE it: at K = 576 a tile is 13.6 us of K loop and ~9.5 us of epilogue that overlaps with
// nothing ("the waves are held at the issue of the stores").
//
// What this kernel does with the ~100 VGPRs the one-wave shape leaves free:
//   * K-step 0 of tile T + 1 is the only step that overwrites the accumulators (srcC = 0), group by group (16 groups of 4 MFMAs).  The
//     results of tile T are taken out of the AccVGPRs just ahead of that: the conversion of group g + 1 (AccVGPR reads, + bias, GELU for
//     mlp.layers.0, pack to the 16-bit type) is interleaved, one quarter per MFMA, with the MFMAs of group g.
//   * a lane owns 4 consecutive columns of 16 rows per accumulator; v_permlane16_swap between the lane pairs (fg, fg ^ 1) of two column
//     fragments turns four 8-byte pieces into two 16-byte ones, so a group leaves as TWO buffer_store_dwordx4 covering 16 rows x 128 B.
//   * groups 0-7 are stored at once (during K-step 0), groups 8-15 are PARKED in 64 VGPRs and stored one group per K-step under K-steps
//     1 .. 8: the store path of the CU (~13 B/clk, 128 KB per tile) works while the matrix cores do.
//   * the bias vector lives in LDS (20 KB next to the 128-KB ring): no global load in the stream of LDS-DMA transfers, whose counted waits
//     would otherwise have to drain it.
// The first tile of a workgroup converts garbage (its stores are suppressed), the last one is drained after the loop.
#include "common.h"
#include "kernels.h"

#include <type_traits>

#define WE_BK 32
#define WE_STAGE ((256 + 256) * WE_BK * 2)       // 32 KB: A rows then W rows, 64 B per row
#define WE_NST 4
#define WE_RING (WE_NST * WE_STAGE)
#define WE_BIAS_MAX 5120                          // floats (N rounded up to a multiple of 256)
#define WE_LDS (WE_RING + WE_BIAS_MAX * 4)
// groups parked in registers (the other 16 - WE_PARK are stored during K-step 0)
#ifndef WE_PARK
#define WE_PARK 8
#endif

// timing-only ablations (results are garbage): -DWE_ABL=1 no stores, 2 no conversion (AccVGPR reads / bias / GELU / pack), 4 no bias reads,
// 8 no AccVGPR reads, 16 AccVGPR reads but no arithmetic
#ifndef WE_ABL
#define WE_ABL 0
#endif
// experiments: 1 the parked stores leave at a different group of the K-step in every wave, 2 plain instead of nontemporal stores,
// 4 (timing only) every second store suppressed, 8 odd workgroups start ~7 us late, 16 (timing only) every tile is stored over the
// workgroup's FIRST tile (the stores stay, their HBM traffic goes: the region lives in L2)
#ifndef WE_EXP
#define WE_EXP 0
#endif
#ifndef WE_ST_AUX
#define WE_ST_AUX ((WE_EXP & 2) ? 0 : 2)
#endif
typedef __attribute__((address_space(3))) void* we_lptr;
__device__ __forceinline__ int we_perm(int row) { return (0x78 >> (((row >> 2) & 3) * 2)) & 3; }   // {0, 2, 3, 1}: gemm.hip's swz2
__device__ __forceinline__ int we_swz(int row, int chunk) { return row * 64 + ((chunk ^ we_perm(row)) << 4); }
__device__ __forceinline__ bool we_tile_map(int b, int tiles_m, int tiles_n, int* tm, int* tn) {       // XCD-aware (gemm.hip tile_map)
    const int xcd = b & 7, q = b >> 3;
    *tn = q % tiles_n;
    *tm = (q / tiles_n) * 8 + xcd;
    return *tm < tiles_m;
}
#ifdef SABER_OP_F16
#define WE_MFMA_OP "v_mfma_f32_16x16x32_f16"
#else
#define WE_MFMA_OP "v_mfma_f32_16x16x32_bf16"
#endif

template <bool GELU, bool STAMPS = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void gemm_w1e_kernel(GemmParams p) {
    // STAMPS (development, tools/gemm_w1e_stamps.py): s_memtime per phase of a tile - K-step 0 (the conversion + direct stores), the K-steps that
    // carry parked stores, the plain rest - summed over the workgroup's tiles, per wave
    unsigned long long ts[4] = {0, 0, 0, 0}, tprev = 0;
#define WE_STAMP(k) do { if (STAMPS) { const unsigned long long _n = __builtin_amdgcn_s_memtime(); ts[k] += _n - tprev; tprev = _n; } } while (0)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int fi = lane & 15, fg = lane >> 4;
    const int tiles_m = (p.M + 255) / 256, tiles_n = (p.N + 255) / 256;
    const int padded = ((tiles_m + 7) / 8) * 8 * tiles_n;
    const int nk = p.K / WE_BK;
    auto next_tile = [&](int L, int* tm, int* tn) {
        while (L < padded && !we_tile_map(L, tiles_m, tiles_n, tm, tn)) L += gridDim.x;
        if (p.rev && L < padded) *tm = tiles_m - 1 - *tm;
        return L;
    };

    // ---- bias -> LDS (zero beyond N and when there is none), before any LDS-DMA is in flight
    {
        float* bias_s = reinterpret_cast<float*>(smem + WE_RING);
        for (int idx = tid; idx < tiles_n * 256; idx += 256) bias_s[idx] = (p.bias && idx < p.N) ? p.bias[idx] : 0.f;
        __syncthreads();
    }

    // ---- LDS-DMA pieces (gemm_w1d.hip): one wave-instruction = 16 rows x 64 B = 1 KB; a K-step is 16 A pieces + 16 W pieces, 4 + 4 per wave
    // Piece q of a wave covers rows (4 wave + q) * 16 .. + 15 of its operand panel, so its per-lane offset is piece 0's plus q * 16 rows: ONE
    // VGPR per operand (aoff0 / woff0: the lane's row of piece 0 and its swizzled 16-byte chunk) and the q-dependent part in the scalar
    // offset.  M and N are multiples of 16, so a piece lies inside its matrix or outside it as a whole: pieces beyond the last row re-read
    // the last valid 16 rows (their products land in rows / columns nobody stores).
    uint32_t aoff0 = 0, woff0 = 0;
    int a_sq[4] = {0, 0, 0, 0}, w_sq[4] = {0, 0, 0, 0};                       // scalar byte offsets of the four pieces relative to piece 0's row
    int Li, tmi = 0, tni = 0, kti = 0, si = 0;                     // issue cursor (tile, K-step, ring stage)
    auto set_issue_tile = [&]() {
        int ln;                                            // the lane id, re-read from the execution mask (volatile: not a loop invariant the compiler can keep - and spill)
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(ln));
        const int lr = ln >> 2, ls = ln & 3;
        const int chunk = ls ^ we_perm(lr);                // (the permutation depends on bits 2-3 of the row: the same for every piece)
        const int a_row0 = min(tmi * 256 + wave * 64, p.M - 16), w_row0 = min(tni * 256 + wave * 64, p.N - 16);
        aoff0 = (uint32_t)(((int64_t)(a_row0 + lr) * p.lda + chunk * 8) * 2);
        woff0 = (uint32_t)(((int64_t)(w_row0 + lr) * 4 + ls) * 16);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            a_sq[q] = min(q * 16, p.M - 16 - a_row0) * p.lda * 2;
            w_sq[q] = min(q * 16, p.N - 16 - w_row0) * 64;
        }
    };
    const __amdgpu_buffer_rsrc_t arsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, (int)(uint32_t)((int64_t)p.M * p.lda * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.Wpk, 0, (int)((uint32_t)nk * (uint32_t)p.N * 64u), 0x00020000);
    const __amdgpu_buffer_rsrc_t crsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.Cb, 0, (int)(uint32_t)((int64_t)p.M * p.ldcb * 2), 0x00020000);
    auto issue_piece = [&](int q) {                                 // q = 0..7: A pieces 0..3, then W pieces 0..3 of the issue cursor's K-step
        char* st = smem + si * WE_STAGE;
        if (q < 4) __builtin_amdgcn_raw_ptr_buffer_load_lds(arsrc, (we_lptr)(st + (wave * 4 + q) * 1024), 16, (int)aoff0, kti * (WE_BK * 2) + a_sq[q], 0, 0);
        else __builtin_amdgcn_raw_ptr_buffer_load_lds(wrsrc, (we_lptr)(st + 256 * 64 + (wave * 4 + (q - 4)) * 1024), 16, (int)woff0, (int)(kti * (p.N * 64)) + w_sq[q - 4], 0, 0);
    };
    auto advance_issue = [&]() {        // past the end of the stream the cursor keeps cycling over its last tile (pieces nobody reads): every step stays identical
        si = (si + 1) & (WE_NST - 1);
        if (++kti == nk) {
            kti = 0;
            if (Li < padded) {
                Li = next_tile(Li + gridDim.x, &tmi, &tni);
                if (Li < padded) set_issue_tile();
            }
        }
    };
    Li = next_tile(blockIdx.x, &tmi, &tni);
    if (Li >= padded) return;                              // block-uniform
    if ((WE_EXP & 8) && (blockIdx.x & 8)) { __builtin_amdgcn_s_sleep(127); __builtin_amdgcn_s_sleep(127); }
    set_issue_tile();
    int Lc = Li, tmc = tmi, tnc = tni;                     // compute cursor

    // The 256 accumulators are PHYSICAL registers named in the instruction strings: accumulator (i, j) = rows i * 16 + fi, columns j * 16 +
    // 4 fg .. + 3 of the wave's 128 x 128 block lives in a[WE_A(i, j) .. + 3].  As compiler-allocated values ("+a" operands) they were moved
    // through VGPRs wholesale around K-step 0 (live-range splitting: 250 v_accvgpr_read / write in a row, 5 000 cycles per tile - stamped);
    // the compiler now never sees them.  This statement tells it that the AccVGPR file is taken (tests/test_abi.py checks that it does not
    // put anything of its own there).
    asm volatile("" ::: "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15", "a16", "a17", "a18", "a19", "a20", "a21", "a22", "a23", "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31", "a32", "a33", "a34", "a35", "a36", "a37", "a38", "a39", "a40", "a41", "a42", "a43", "a44", "a45", "a46", "a47", "a48", "a49", "a50", "a51", "a52", "a53", "a54", "a55", "a56", "a57", "a58", "a59", "a60", "a61", "a62", "a63", "a64", "a65", "a66", "a67", "a68", "a69", "a70", "a71", "a72", "a73", "a74", "a75", "a76", "a77", "a78", "a79", "a80", "a81", "a82", "a83", "a84", "a85", "a86", "a87", "a88", "a89", "a90", "a91", "a92", "a93", "a94", "a95", "a96", "a97", "a98", "a99", "a100", "a101", "a102", "a103", "a104", "a105", "a106", "a107", "a108", "a109", "a110", "a111", "a112", "a113", "a114", "a115", "a116", "a117", "a118", "a119", "a120", "a121", "a122", "a123", "a124", "a125", "a126", "a127", "a128", "a129", "a130", "a131", "a132", "a133", "a134", "a135", "a136", "a137", "a138", "a139", "a140", "a141", "a142", "a143", "a144", "a145", "a146", "a147", "a148", "a149", "a150", "a151", "a152", "a153", "a154", "a155", "a156", "a157", "a158", "a159", "a160", "a161", "a162", "a163", "a164", "a165", "a166", "a167", "a168", "a169", "a170", "a171", "a172", "a173", "a174", "a175", "a176", "a177", "a178", "a179", "a180", "a181", "a182", "a183", "a184", "a185", "a186", "a187", "a188", "a189", "a190", "a191", "a192", "a193", "a194", "a195", "a196", "a197", "a198", "a199", "a200", "a201", "a202", "a203", "a204", "a205", "a206", "a207", "a208", "a209", "a210", "a211", "a212", "a213", "a214", "a215", "a216", "a217", "a218", "a219", "a220", "a221", "a222", "a223", "a224", "a225", "a226", "a227", "a228", "a229", "a230", "a231", "a232", "a233", "a234", "a235", "a236", "a237", "a238", "a239", "a240", "a241", "a242", "a243", "a244", "a245", "a246", "a247", "a248", "a249", "a250", "a251", "a252", "a253", "a254", "a255");
    op16x8 fa0[8], fw0[8], fa1[8], fw1[8];
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)(smem);
    const uint32_t fa_base = lds0 + we_swz(wm * 128 + fi, fg);                 // + i * 1024 (16 rows of 64 B; the swizzle term repeats every 16 rows)
    const uint32_t fw_base = lds0 + 256 * 64 + we_swz(wn * 128 + fi, fg);
#define WE_DSR(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))
#define WE_A(i, j) (((i) * 8 + (j)) * 4)
#define WE_MFMA(i, j, fa, fw) asm volatile(WE_MFMA_OP " a[%0:%1], %2, %3, a[%0:%1]" : : "n"(WE_A(i, j)), "n"(WE_A(i, j) + 3), "v"(fw[j]), "v"(fa[i]))
#define WE_MFMA0(i, j, fa, fw) asm volatile(WE_MFMA_OP " a[%0:%1], %2, %3, 0" : : "n"(WE_A(i, j)), "n"(WE_A(i, j) + 3), "v"(fw[j]), "v"(fa[i]))
// one quarter of a group's conversion: accumulator (i, jj) + bias -> (GELU) -> two packed registers pk[j][0..1]
#define WE_CONVQ(i, jj, j)                                                                                                                   \
    do {                                                                                                                                     \
        if (WE_ABL & 2) { pk[j][0] = (uint32_t)((i) + (jj)); pk[j][1] = (uint32_t)(j); break; }                                              \
        float x0_, x1_, x2_, x3_;                                                                                                            \
        if (WE_ABL & 8) { x0_ = bq[jj][1]; x1_ = bq[jj][2]; x2_ = bq[jj][3]; x3_ = bq[jj][0]; } else                                             \
        asm volatile("v_accvgpr_read_b32 %0, a[%4]\n\tv_accvgpr_read_b32 %1, a[%5]\n\tv_accvgpr_read_b32 %2, a[%6]\n\tv_accvgpr_read_b32 %3, a[%7]"   \
                     : "=v"(x0_), "=v"(x1_), "=v"(x2_), "=v"(x3_)                                                                           \
                     : "n"(WE_A(i, jj)), "n"(WE_A(i, jj) + 1), "n"(WE_A(i, jj) + 2), "n"(WE_A(i, jj) + 3));                                  \
        if (WE_ABL & 16) { pk[j][0] = __float_as_uint(x0_) ^ __float_as_uint(x1_); pk[j][1] = __float_as_uint(x2_) ^ __float_as_uint(x3_); break; } \
        f32x2 lo_ = {x0_ + bq[jj][0], x1_ + bq[jj][1]}, hi_ = {x2_ + bq[jj][2], x3_ + bq[jj][3]};                                                \
        if (GELU) { lo_ = gelu_erf2(lo_); hi_ = gelu_erf2(hi_); }                                                                            \
        pk[j][0] = pack_op16(lo_.x, lo_.y);                                                                                                  \
        pk[j][1] = pack_op16(hi_.x, hi_.y);                                                                                                  \
    } while (0)
    auto frag_read = [&](int q, uint32_t ab, uint32_t wb, op16x8 (&na)[8], op16x8 (&nw)[8]) {     // q-th of the 16 fragment reads of a K-step
        switch (q) {
            case 0: WE_DSR(nw[0], wb, 0); break;      case 1: WE_DSR(nw[1], wb, 1024); break;
            case 2: WE_DSR(nw[2], wb, 2048); break;   case 3: WE_DSR(nw[3], wb, 3072); break;
            case 4: WE_DSR(na[0], ab, 0); break;      case 5: WE_DSR(nw[4], wb, 4096); break;
            case 6: WE_DSR(nw[5], wb, 5120); break;   case 7: WE_DSR(nw[6], wb, 6144); break;
            case 8: WE_DSR(nw[7], wb, 7168); break;   case 9: WE_DSR(na[1], ab, 1024); break;
            case 10: WE_DSR(na[2], ab, 2048); break;  case 11: WE_DSR(na[3], ab, 3072); break;
            case 12: WE_DSR(na[4], ab, 4096); break;  case 13: WE_DSR(na[5], ab, 5120); break;
            case 14: WE_DSR(na[6], ab, 6144); break;  default: WE_DSR(na[7], ab, 7168); break;
        }
    };
    auto frags_landed = [&](op16x8 (&fa)[8], op16x8 (&fw)[8]) {
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+v"(fa[0]), "+v"(fa[1]), "+v"(fa[2]), "+v"(fa[3]), "+v"(fa[4]), "+v"(fa[5]), "+v"(fa[6]), "+v"(fa[7]),
                       "+v"(fw[0]), "+v"(fw[1]), "+v"(fw[2]), "+v"(fw[3]), "+v"(fw[4]), "+v"(fw[5]), "+v"(fw[6]), "+v"(fw[7]));
    };

    // ---- epilogue of the FINISHED tile (coordinates tmp / tnp), in 16 groups: group g = accumulators acc[g >> 1][4 (g & 1) .. + 3]
    // = rows i * 16 + fi, columns (j0 + j) * 16 + 4 fg .. + 3 of the wave's 128 x 128 block
    int tmp = 0, tnp = 0;
    f32x4 bq[8];                                           // bias of the finished tile's eight column fragments (columns jj * 16 + 4 fg .. + 3), read once per tile
    u32x4 park[WE_PARK][2];
    uint32_t bias_addr = lds0 + WE_RING + (uint32_t)((wn * 128 + fg * 4) * 4);     // + tnp * 1024 per tile
    // per-lane part of a store's byte offset into C: row fi, the 16-byte piece this lane ends up with after the swap (see below)
    const int lane_col = (fg & 1) * 16 + (fg & 2) * 4;     // column of the piece inside a 32-column pair of fragments
    const uint32_t lane_off = (uint32_t)((fi * p.ldcb + lane_col) * 2);
    uint32_t tile_off = 0;                                 // ((tmp * 256 + wm * 128) * ldcb + tnp * 256 + wn * 128) * 2
    int ncol_left = -(1 << 20);                            // N - (tnp * 256 + wn * 128): columns of C that exist from this wave's first one on
    auto set_prev_tile = [&]() {
        tmp = tmc; tnp = tnc;
        int ln;                                            // (re-derived per tile for the same reason as in set_issue_tile)
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(ln));
        bias_addr = lds0 + WE_RING + (uint32_t)((wn * 128 + (ln >> 4) * 4) * 4) + (uint32_t)tnp * 1024u;
        tile_off = (uint32_t)((((int64_t)tmp * 256 + wm * 128) * p.ldcb + tnp * 256 + wn * 128) * 2);
        if (WE_EXP & 16) { int t0m, t0n; (void)next_tile(blockIdx.x, &t0m, &t0n); tile_off = (uint32_t)((((int64_t)t0m * 256 + wm * 128) * p.ldcb + t0n * 256 + wn * 128) * 2); }
        ncol_left = p.N - (tnp * 256 + wn * 128);
    };
    // all eight ds_read_b128 at once, issued right BEFORE a wait on the LDS counter that the K loop needs anyway (the fragments of K-step 0):
    // per-group bias reads cost a ~150-cycle LDS round trip per group behind the fragment reads issued ahead of them (stamped: 2 500 cycles per tile)
    auto bias_issue = [&]() {
        if (WE_ABL & 6) return;
        WE_DSR(bq[0], bias_addr, 0); WE_DSR(bq[1], bias_addr, 64); WE_DSR(bq[2], bias_addr, 128); WE_DSR(bq[3], bias_addr, 192);
        WE_DSR(bq[4], bias_addr, 256); WE_DSR(bq[5], bias_addr, 320); WE_DSR(bq[6], bias_addr, 384); WE_DSR(bq[7], bias_addr, 448);
    };
    auto bias_landed = [&]() {
        if (WE_ABL & 6) { for (int j = 0; j < 8; ++j) bq[j] = (f32x4){0.f, 0.f, 0.f, 0.f}; return; }
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(bq[0]), "+v"(bq[1]), "+v"(bq[2]), "+v"(bq[3]), "+v"(bq[4]), "+v"(bq[5]), "+v"(bq[6]), "+v"(bq[7]));
    };
    // v_permlane16_swap(V, S): lanes 16-31 of V <-> lanes 0-15 of S (and 48-63 <-> 32-47).  With V = fragment j0's register and S = fragment
    // j0 + 1's: an even-fg lane ends with {own j0, partner's j0} = columns 4 fg .. 4 fg + 7 of fragment j0, an odd-fg lane with {partner's
    // j0 + 1, own j0 + 1} = columns 4 (fg - 1) .. + 7 of fragment j0 + 1: 16 contiguous bytes per lane, 128 B per row and pair of stores
    auto swap_pair = [&](const uint32_t (&pk)[4][2], int j) {
        const auto r0 = __builtin_amdgcn_permlane16_swap(pk[j][0], pk[j + 1][0], false, false);
        const auto r1 = __builtin_amdgcn_permlane16_swap(pk[j][1], pk[j + 1][1], false, false);
        return (u32x4){r0[0], r1[0], r0[1], r1[1]};
    };
    auto store_group = [&](int g, const u32x4& v0, const u32x4& v1) {
        if (WE_ABL & 1) { if (v0[0] + v1[0] == 0x12345679u) __builtin_amdgcn_raw_buffer_store_b128(v0, crsrc, lane_off, 0, 2); return; }
        const int i = g >> 1, j0 = (g & 1) * 4;
        // everything in the per-lane offset, so that the descriptor's range check drops rows beyond M.  N is a multiple of 32, so a store
        // (one 32-column pair of fragments) is valid or not as a whole: a SCALAR select pushes columns beyond N - and everything while there
        // is nothing to store yet (first tile of the workgroup: ncol_left hugely negative) - out of the descriptor's range
        const uint32_t sb = tile_off + (uint32_t)(i * 16 * p.ldcb * 2 + j0 * 32);
        const uint32_t sb0 = j0 * 16 + 32 <= ncol_left ? sb : 0x80000000u, sb1 = j0 * 16 + 64 <= ncol_left ? sb + 64u : 0x80000000u;
        __builtin_amdgcn_raw_buffer_store_b128(v0, crsrc, lane_off + sb0, 0, WE_ST_AUX);            // aux 2: nontemporal (the qkv / hidden tensors are larger than the Infinity Cache)
        if (!(WE_EXP & 4)) __builtin_amdgcn_raw_buffer_store_b128(v1, crsrc, lane_off + sb1, 0, WE_ST_AUX);
    };

    // one K-step: 16 groups of 4 MFMAs (i = g / 2, j = 4 (g % 2) ..); behind each of the first eight groups two fragment reads of the NEXT K-step,
    // behind every second group one LDS-DMA piece of the K-step FOUR ahead.  MODE 0: plain.  MODE 1: K-step 0 of a tile - the MFMAs write the
    // accumulators (srcC = 0) and the finished tile's results leave them just ahead (see the file header).  MODE 2 + q: plain, plus the
    // stores of parked group q.
    int sstage = 0;                                        // stage of the compute cursor's K-step
    auto step = [&](op16x8 (&fa)[8], op16x8 (&fw)[8], op16x8 (&na)[8], op16x8 (&nw)[8], auto mode_c) {
        constexpr int MODE = decltype(mode_c)::value;
        constexpr bool FIRST = MODE == 1;
        uint32_t pk[4][2];
        if constexpr (FIRST) {
            bias_issue();
        }
        frags_landed(fa, fw);
        if constexpr (FIRST) {
            bias_landed();
            // group 0's conversion has nothing to hide under
            WE_CONVQ(0, 0, 0); WE_CONVQ(0, 1, 1); WE_CONVQ(0, 2, 2); WE_CONVQ(0, 3, 3);
        }
        const int s1 = (sstage + 1) & (WE_NST - 1);
        const uint32_t ab = fa_base + s1 * WE_STAGE, wb = fw_base + s1 * WE_STAGE;
        // (the groups are written out by the preprocessor: the accumulators' register numbers are literals of the instruction strings)
#define WE_NI(g) ((((g) + 1) >> 1) & 7)
#define WE_NJ(g, j) ((((g) + 1) & 1) * 4 + (j))
#define WE_GROUP(g)                                                                                                                          \
        {                                                                                                                                    \
            if constexpr (FIRST) {                                                                                                           \
                /* this group's packed results (converted during the previous group) leave: swap, then store or park */                      \
                const u32x4 v0 = swap_pair(pk, 0), v1 = swap_pair(pk, 2);                                                                    \
                if ((g) < 16 - WE_PARK) store_group((g), v0, v1);                                                                            \
                else { park[(g) >= 16 - WE_PARK ? (g) - (16 - WE_PARK) : 0][0] = v0; park[(g) >= 16 - WE_PARK ? (g) - (16 - WE_PARK) : 0][1] = v1; } \
                __builtin_amdgcn_sched_barrier(0);                                                                                           \
                /* the MFMAs of group g overwrite its accumulators; group g + 1's conversion rides between them, one quarter per MFMA */     \
                WE_MFMA0((g) >> 1, ((g) & 1) * 4 + 0, fa, fw);                                                                               \
                if ((g) < 15) WE_CONVQ(WE_NI(g), WE_NJ(g, 0), 0);                                                                             \
                __builtin_amdgcn_sched_barrier(0);                                                                                           \
                WE_MFMA0((g) >> 1, ((g) & 1) * 4 + 1, fa, fw);                                                                               \
                if ((g) < 15) WE_CONVQ(WE_NI(g), WE_NJ(g, 1), 1);                                                                             \
                __builtin_amdgcn_sched_barrier(0);                                                                                           \
                WE_MFMA0((g) >> 1, ((g) & 1) * 4 + 2, fa, fw);                                                                               \
                if ((g) < 15) WE_CONVQ(WE_NI(g), WE_NJ(g, 2), 2);                                                                             \
                __builtin_amdgcn_sched_barrier(0);                                                                                           \
                WE_MFMA0((g) >> 1, ((g) & 1) * 4 + 3, fa, fw);                                                                               \
                if ((g) < 15) WE_CONVQ(WE_NI(g), WE_NJ(g, 3), 3);                                                                             \
                __builtin_amdgcn_sched_barrier(0);                                                                                           \
            } else {                                                                                                                         \
                WE_MFMA((g) >> 1, ((g) & 1) * 4 + 0, fa, fw); WE_MFMA((g) >> 1, ((g) & 1) * 4 + 1, fa, fw);                                  \
                WE_MFMA((g) >> 1, ((g) & 1) * 4 + 2, fa, fw); WE_MFMA((g) >> 1, ((g) & 1) * 4 + 3, fa, fw);                                  \
            }                                                                                                                                \
            if ((g) < 8) { frag_read(2 * (g), ab, wb, na, nw); frag_read(2 * (g) + 1, ab, wb, na, nw); }     /* all 16 reads in the first half: 8 groups of MFMAs cover the last one's latency */ \
            if ((g) & 1) issue_piece((g) >> 1);                                                                                              \
            if constexpr (MODE >= 2) if ((WE_EXP & 1) ? ((g) == 1 + 4 * wave) : ((g) == 5)) store_group(16 - WE_PARK + (MODE - 2), park[MODE - 2][0], park[MODE - 2][1]); \
        }
        WE_GROUP(0) WE_GROUP(1) WE_GROUP(2) WE_GROUP(3) WE_GROUP(4) WE_GROUP(5) WE_GROUP(6) WE_GROUP(7)
        WE_GROUP(8) WE_GROUP(9) WE_GROUP(10) WE_GROUP(11) WE_GROUP(12) WE_GROUP(13) WE_GROUP(14) WE_GROUP(15)
#undef WE_GROUP
        advance_issue();
        sstage = s1;
        // the next K-step's pieces have landed (this wave's), AND this wave's fragment reads of the stage that the next step's transfers
        // overwrite are complete - before the barrier, so that no wave's transfer can race another wave's read.  Loads return in order, so
        // "at most 24 outstanding" still implies the next K-step's eight pieces whatever the stores in between have done.
        asm volatile("s_waitcnt vmcnt(24) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    };

    // stream prologue: K-steps 0 .. 3 on their way, the first one landed and published, its fragments on their way
#pragma unroll
    for (int t = 0; t < 4; ++t) {
#pragma unroll
        for (int q = 0; q < 8; ++q) issue_piece(q);
        advance_issue();
    }
#pragma unroll
    for (int q = 0; q < WE_PARK; ++q) { park[q][0] = (u32x4){0, 0, 0, 0}; park[q][1] = (u32x4){0, 0, 0, 0}; }
    asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int q = 0; q < 16; ++q) frag_read(q, fa_base, fw_base, fa0, fw0);
    // a tile is an even number of K-steps (K % 64 == 0), so every tile starts with its fragments in buffer 0; K-steps 1 .. WE_PARK carry the parked stores
    constexpr int HEAD = (1 + WE_PARK + 1) & ~1;            // K-step 0, WE_PARK store-carrying steps, padded to an even count
    const int pairs = (nk - HEAD) / 2;
#define WE_HEAD_STEP(s)                                                                                                             \
    if constexpr ((s) == 1) WE_STAMP(0);                                                                                            \
    if constexpr ((s) < HEAD) {                                                                                                     \
        if constexpr (((s) & 1) == 0) step(fa0, fw0, fa1, fw1, std::integral_constant<int, (s) == 0 ? 1 : (s) <= WE_PARK ? (s) + 1 : 0>{}); \
        else step(fa1, fw1, fa0, fw0, std::integral_constant<int, (s) <= WE_PARK ? (s) + 1 : 0>{});                                 \
    }
    if (STAMPS) tprev = __builtin_amdgcn_s_memtime();
    while (Lc < padded) {
        // K-step 0 writes the accumulators and takes the finished tile's results out; K-steps 1 .. WE_PARK carry the parked stores
        WE_HEAD_STEP(0) WE_HEAD_STEP(1) WE_HEAD_STEP(2) WE_HEAD_STEP(3) WE_HEAD_STEP(4) WE_HEAD_STEP(5) WE_HEAD_STEP(6) WE_HEAD_STEP(7) WE_HEAD_STEP(8) WE_HEAD_STEP(9)
        WE_STAMP(1);
        for (int q = 0; q < pairs; ++q) {
            step(fa0, fw0, fa1, fw1, std::integral_constant<int, 0>{});
            step(fa1, fw1, fa0, fw0, std::integral_constant<int, 0>{});
        }
        WE_STAMP(2);
        set_prev_tile();
        Lc = next_tile(Lc + gridDim.x, &tmc, &tnc);
    }
    // the last tile's results: the same conversion without a K loop to hide under
    {
        uint32_t pk[4][2];
        bias_issue();
        bias_landed();
#define WE_DRAIN(g)                                                                                                                          \
        {                                                                                                                                    \
            WE_CONVQ((g) >> 1, ((g) & 1) * 4 + 0, 0); WE_CONVQ((g) >> 1, ((g) & 1) * 4 + 1, 1);                                              \
            WE_CONVQ((g) >> 1, ((g) & 1) * 4 + 2, 2); WE_CONVQ((g) >> 1, ((g) & 1) * 4 + 3, 3);                                              \
            store_group((g), swap_pair(pk, 0), swap_pair(pk, 2));                                                                            \
        }
        WE_DRAIN(0) WE_DRAIN(1) WE_DRAIN(2) WE_DRAIN(3) WE_DRAIN(4) WE_DRAIN(5) WE_DRAIN(6) WE_DRAIN(7)
        WE_DRAIN(8) WE_DRAIN(9) WE_DRAIN(10) WE_DRAIN(11) WE_DRAIN(12) WE_DRAIN(13) WE_DRAIN(14) WE_DRAIN(15)
#undef WE_DRAIN
    }
    WE_STAMP(3);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");      // the surplus transfers of the stream's end must land before the LDS is released
    if (STAMPS && p.stamps && lane == 0)
        for (int k = 0; k < 4; ++k) p.stamps[((int64_t)blockIdx.x * 4 + wave) * 4 + k] = ts[k];
#undef WE_STAMP
}

bool gemm_w1e_supported(const GemmParams& p) {
    return p.Cb && !p.Cf && !p.res && !p.pool4 && p.batch <= 1 && (p.act == ACT_NONE || p.act == ACT_GELU) && (p.K % (2 * WE_BK)) == 0 && p.K >= ((1 + WE_PARK + 1) & ~1) * WE_BK && p.Wpk &&
           (p.N & 31) == 0 && (p.M & 15) == 0 && p.M >= 16 && (p.ldcb & 7) == 0 && (p.lda & 7) == 0 && !p.ln_out && p.N >= 32 && ((p.N + 255) / 256) * 256 <= WE_BIAS_MAX &&
           (int64_t)p.M * p.lda * 2 < (1ll << 31) && (int64_t)(p.K / WE_BK) * p.N * 64 < (1ll << 31) && (int64_t)p.M * p.ldcb * 2 < (1ll << 31) &&
           (reinterpret_cast<uintptr_t>(p.Cb) & 15) == 0;
}

const char* launch_gemm_w1e(const GemmParams& p, hipStream_t stream) {
    if (!gemm_w1e_supported(p)) return "gemm_w1e: unsupported problem";
    const int tiles_m = (p.M + 255) / 256, tiles_n = (p.N + 255) / 256;
    const int slots = ((tiles_m + 7) / 8) * 8 * tiles_n;
    if (p.stamps) {          // development build with cycle stamps
        if (p.act == ACT_GELU) hipLaunchKernelGGL((gemm_w1e_kernel<true, true>), dim3(slots < 256 ? slots : 256), dim3(256), WE_LDS, stream, p);
        else hipLaunchKernelGGL((gemm_w1e_kernel<false, true>), dim3(slots < 256 ? slots : 256), dim3(256), WE_LDS, stream, p);
        return nullptr;
    }
    if (p.act == ACT_GELU) hipLaunchKernelGGL((gemm_w1e_kernel<true>), dim3(slots < 256 ? slots : 256), dim3(256), WE_LDS, stream, p);
    else hipLaunchKernelGGL((gemm_w1e_kernel<false>), dim3(slots < 256 ? slots : 256), dim3(256), WE_LDS, stream, p);
    return nullptr;
}

const char* gemm_w1e_init_device() {
    hipError_t st = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_w1e_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, WE_LDS);
    if (st == hipSuccess) st = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_w1e_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, WE_LDS);
    if (st == hipSuccess) st = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_w1e_kernel<true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, WE_LDS);
    if (st == hipSuccess) st = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_w1e_kernel<false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, WE_LDS);
    return st == hipSuccess ? nullptr : hipGetErrorString(st);
}
