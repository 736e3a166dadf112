// 16-bit-output GEMM with ONE wave per SIMD and direct-to-LDS operands: 256 x 256 tile, 4 waves of 128 x 128 (256 fp32 accumulators in the
// AccVGPR half of the wave's register file), K-steps of 32 through a 4-stage LDS-DMA ring with three K-steps in flight.
//
// Replaces (for the widest 16-bit-output GEMMs of the Hiera trunk: attn.qkv and mlp.layers.0 of the third-party sam2 MultiScaleBlock,
// reached through saber/adapters/sam2/predictor.py:70) gemm_bf16_p256s_kernel where it wins.  Why this shape:
// the vendor library's kernel for these shapes is MT256x256x64 on FOUR waves with direct-to-LDS operands (its name in hipblaslt's gfx950
// code object says so: MIWT8_8, WG32_8_1, DTLA1_DTLB1, PGR2); a 128 x 128 wave tile reads (128 + 128) x 64 B of fragments per K-step for 64
// MFMAs where two 128 x 64 waves read 2 x (128 + 64) x 64 B, and with one wave per SIMD nothing is shared, so there is no half-step
// choreography, only one wave's instruction stream: per K-step 64 MFMAs, and in their shadow the 16 fragment reads of the NEXT K-step and
// the 8 LDS-DMA pieces of the K-step four ahead.  Round 3's gemm_w1 (tools/experiments/) staged its operands through registers (plain loads
// + ds_write): 64 staging registers and the LDS writes in the MFMA issue stream; here the operands never touch a register.
//
// The accumulators are pinned to AccVGPRs by inline-asm MFMAs (left alone hipcc keeps them in VGPRs and copies every fragment); the
// fragment reads are inline asm too (invisible to the compiler, which would otherwise order every LDS access behind the LDS-DMA in
// flight with vmcnt(0)); one counted vmcnt + one barrier per K-step.
#include "common.h"
#include "kernels.h"

#include <type_traits>

#define WD_BK 32
#define WD_STAGE ((256 + 256) * WD_BK * 2)       // 32 KB: A rows then W rows, 64 B per row
#define WD_NST 4
#define WD_LDS (WD_NST * WD_STAGE)

typedef __attribute__((address_space(3))) void* wd_lptr;
__device__ __forceinline__ int wd_perm(int row) { return (0x78 >> (((row >> 2) & 3) * 2)) & 3; }   // {0, 2, 3, 1}: gemm.hip's swz2
__device__ __forceinline__ int wd_swz(int row, int chunk) { return row * 64 + ((chunk ^ wd_perm(row)) << 4); }
__device__ __forceinline__ bool wd_tile_map(int b, int tiles_m, int tiles_n, int* tm, int* tn) {       // XCD-aware (gemm.hip tile_map)
    const int xcd = b & 7, q = b >> 3;
    *tn = q % tiles_n;
    *tm = (q / tiles_n) * 8 + xcd;
    return *tm < tiles_m;
}
#ifdef SABER_OP_F16
#define WD_MFMA_OP "v_mfma_f32_16x16x32_f16"
#else
#define WD_MFMA_OP "v_mfma_f32_16x16x32_bf16"
#endif

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void gemm_w1d_kernel(GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int fi = lane & 15, fg = lane >> 4;
    const int tiles_m = (p.M + 255) / 256, tiles_n = (p.N + 255) / 256;
    const int padded = ((tiles_m + 7) / 8) * 8 * tiles_n;
    const int nk = p.K / WD_BK;
    auto next_tile = [&](int L, int* tm, int* tn) {
        while (L < padded && !wd_tile_map(L, tiles_m, tiles_n, tm, tn)) L += gridDim.x;
        if (p.rev && L < padded) *tm = tiles_m - 1 - *tm;
        return L;
    };

    // ---- LDS-DMA pieces: one wave-instruction = 16 rows x 64 B = 1 KB; a K-step is 16 A pieces + 16 W pieces, 4 + 4 per wave.  Piece q of
    // this wave covers rows (4 wave + q) * 16 .. + 15 of its operand panel; the 16-B chunk a lane fetches is permuted on the SOURCE side so that
    // the linear LDS image of the instruction is the swizzled layout the fragment reads expect.  Both operands go through buffer descriptors
    // (one 32-bit offset per lane and piece; the K-step in the scalar offset): A over the whole matrix, W over its K-step-packed copy
    // (launch_pack_w_kstep: Wpk[ks][n][physical chunk][8], the 16 rows of a piece are one contiguous KB).
    const int lrow = lane >> 2, lslot = lane & 3;
    uint32_t aoff[4], woff[4];
    int Li, tmi = 0, tni = 0, kti = 0, si = 0;                     // issue cursor (tile, K-step, ring stage)
    auto set_issue_tile = [&]() {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int row = (wave * 4 + q) * 16 + lrow;
            const int chunk = lslot ^ wd_perm(row);
            aoff[q] = (uint32_t)(((int64_t)min(tmi * 256 + row, p.M - 1) * p.lda + chunk * 8) * 2);
            woff[q] = (uint32_t)(((int64_t)min(tni * 256 + row, p.N - 1) * 4 + lslot) * 16);
        }
    };
    const __amdgpu_buffer_rsrc_t arsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, (int)(uint32_t)((int64_t)p.M * p.lda * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.Wpk, 0, (int)((uint32_t)nk * (uint32_t)p.N * 64u), 0x00020000);
    auto issue_piece = [&](int q) {                                 // q = 0..7: A pieces 0..3, then W pieces 0..3 of the issue cursor's K-step
        char* st = smem + si * WD_STAGE;
        if (q < 4) __builtin_amdgcn_raw_ptr_buffer_load_lds(arsrc, (wd_lptr)(st + (wave * 4 + q) * 1024), 16, (int)aoff[q], kti * (WD_BK * 2), 0, 0);
        else __builtin_amdgcn_raw_ptr_buffer_load_lds(wrsrc, (wd_lptr)(st + 256 * 64 + (wave * 4 + (q - 4)) * 1024), 16, (int)woff[q - 4], (int)(kti * (p.N * 64)), 0, 0);
    };
    auto advance_issue = [&]() {        // past the end of the stream the cursor keeps cycling over its last tile (pieces nobody reads): every step stays identical
        si = (si + 1) & (WD_NST - 1);
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
    set_issue_tile();
    int Lc = Li, tmc = tmi, tnc = tni;                     // compute cursor

    f32x4 acc[8][8];
    op16x8 fa0[8], fw0[8], fa1[8], fw1[8];
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)(smem);
    const uint32_t fa_base = lds0 + wd_swz(wm * 128 + fi, fg);                 // + i * 1024 (16 rows of 64 B; the swizzle term repeats every 16 rows)
    const uint32_t fw_base = lds0 + 256 * 64 + wd_swz(wn * 128 + fi, fg);
#define WD_DSR(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))
#define WD_MFMA(i, j, fa, fw) asm volatile(WD_MFMA_OP " %0, %1, %2, %0" : "+a"(acc[i][j]) : "v"(fw[j]), "v"(fa[i]))
#define WD_MFMA0(i, j, fa, fw) asm volatile(WD_MFMA_OP " %0, %1, %2, 0" : "=a"(acc[i][j]) : "v"(fw[j]), "v"(fa[i]))
    auto frag_read = [&](int q, uint32_t ab, uint32_t wb, op16x8 (&na)[8], op16x8 (&nw)[8]) {     // q-th of the 16 fragment reads of a K-step
        switch (q) {
            case 0: WD_DSR(nw[0], wb, 0); break;      case 1: WD_DSR(nw[1], wb, 1024); break;
            case 2: WD_DSR(nw[2], wb, 2048); break;   case 3: WD_DSR(nw[3], wb, 3072); break;
            case 4: WD_DSR(na[0], ab, 0); break;      case 5: WD_DSR(nw[4], wb, 4096); break;
            case 6: WD_DSR(nw[5], wb, 5120); break;   case 7: WD_DSR(nw[6], wb, 6144); break;
            case 8: WD_DSR(nw[7], wb, 7168); break;   case 9: WD_DSR(na[1], ab, 1024); break;
            case 10: WD_DSR(na[2], ab, 2048); break;  case 11: WD_DSR(na[3], ab, 3072); break;
            case 12: WD_DSR(na[4], ab, 4096); break;  case 13: WD_DSR(na[5], ab, 5120); break;
            case 14: WD_DSR(na[6], ab, 6144); break;  default: WD_DSR(na[7], ab, 7168); break;
        }
    };
    auto frags_landed = [&](op16x8 (&fa)[8], op16x8 (&fw)[8]) {
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+v"(fa[0]), "+v"(fa[1]), "+v"(fa[2]), "+v"(fa[3]), "+v"(fa[4]), "+v"(fa[5]), "+v"(fa[6]), "+v"(fa[7]),
                       "+v"(fw[0]), "+v"(fw[1]), "+v"(fw[2]), "+v"(fw[3]), "+v"(fw[4]), "+v"(fw[5]), "+v"(fw[6]), "+v"(fw[7]));
    };
    // epilogue of the compute cursor's tile: bias (+ GELU), 16-bit, 8-byte stores straight from the accumulator layout (lane = row fi,
    // columns 4 fg ..: four lanes cover 32 B of a row, the four fragments j .. j + 3 of a row complete its 128-B line back to back);
    // nontemporal (the qkv / hidden tensors are larger than the Infinity Cache: see gemm_bf16_p256s_kernel)
    auto epilogue = [&]() {
        const int m0 = tmc * 256 + wm * 128, n0 = tnc * 256 + wn * 128;
        char* Cb = reinterpret_cast<char*>(p.Cb);
#pragma unroll
        for (int jh = 0; jh < 2; ++jh) {
            float4 bias4[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int n = n0 + (jh * 4 + j) * 16 + fg * 4;
                const float4 bv = *reinterpret_cast<const float4*>((p.bias ? p.bias : reinterpret_cast<const float*>(p.W)) + min(n, p.N - 4));
                bias4[j] = (p.bias && n + 3 < p.N) ? bv : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int m = m0 + i * 16 + fi;
                const int64_t rowoff = (int64_t)min(m, p.M - 1) * p.ldcb * 2;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const f32x4 a = acc[i][jh * 4 + j];
                    float v[4] = {a[0] + bias4[j].x, a[1] + bias4[j].y, a[2] + bias4[j].z, a[3] + bias4[j].w};
                    if (p.act == ACT_GELU) {
                        const f32x2 g0 = gelu_erf2((f32x2){v[0], v[1]}), g1 = gelu_erf2((f32x2){v[2], v[3]});
                        v[0] = g0.x; v[1] = g0.y; v[2] = g1.x; v[3] = g1.y;
                    }
                    const int n = n0 + (jh * 4 + j) * 16 + fg * 4;
                    if (m < p.M && n < p.N) {
                        typedef unsigned int u32x2_w __attribute__((ext_vector_type(2)));
                        __builtin_nontemporal_store((u32x2_w){pack_op16(v[0], v[1]), pack_op16(v[2], v[3])}, reinterpret_cast<u32x2_w*>(Cb + rowoff + n * 2));
                    }
                }
            }
        }
        Lc = next_tile(Lc + gridDim.x, &tmc, &tnc);
    };
    // one K-step: 16 groups of 4 MFMAs (i = g / 2, j = 4 (g % 2) ..); behind each of the first eight groups two fragment reads of the NEXT K-step,
    // behind every second group one LDS-DMA piece of the K-step FOUR ahead (its ring stage held this step's operands: their fragments were read during the
    // previous step, and every wave has passed that step's barrier).  At the end the pieces of the next K-step must have landed: everything
    // but the 24 youngest (three K-steps in flight), then the barrier publishes them.
    int sstage = 0;                                        // stage of the compute cursor's K-step
    auto step = [&](op16x8 (&fa)[8], op16x8 (&fw)[8], op16x8 (&na)[8], op16x8 (&nw)[8], auto first_c) {
        constexpr bool FIRST = decltype(first_c)::value;
        frags_landed(fa, fw);
        const int s1 = (sstage + 1) & (WD_NST - 1);
        const uint32_t ab = fa_base + s1 * WD_STAGE, wb = fw_base + s1 * WD_STAGE;
#pragma unroll
        for (int g = 0; g < 16; ++g) {
            const int i = g >> 1, j0 = (g & 1) * 4;
            if (FIRST) { WD_MFMA0(i, j0, fa, fw); WD_MFMA0(i, j0 + 1, fa, fw); WD_MFMA0(i, j0 + 2, fa, fw); WD_MFMA0(i, j0 + 3, fa, fw); }
            else { WD_MFMA(i, j0, fa, fw); WD_MFMA(i, j0 + 1, fa, fw); WD_MFMA(i, j0 + 2, fa, fw); WD_MFMA(i, j0 + 3, fa, fw); }
            if (g < 8) { frag_read(2 * g, ab, wb, na, nw); frag_read(2 * g + 1, ab, wb, na, nw); }     // all 16 reads in the first half: 8 groups of MFMAs cover the last one's latency
            if (g & 1) issue_piece(g >> 1);
        }
        advance_issue();
        sstage = s1;
        // the next K-step's pieces have landed (this wave's), AND this wave's fragment reads of the stage that the next step's transfers
        // overwrite are complete - before the barrier, so that no wave's transfer can race another wave's read
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
    asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int q = 0; q < 16; ++q) frag_read(q, fa_base, fw_base, fa0, fw0);
    // a tile is an even number of K-steps (K % 64 == 0), so every tile starts with its fragments in buffer 0
    const int pairs = nk / 2 - 1;
    while (Lc < padded) {
        step(fa0, fw0, fa1, fw1, std::true_type{});       // K-step 0 writes the accumulators (srcC = 0)
        step(fa1, fw1, fa0, fw0, std::false_type{});
        for (int q = 0; q < pairs; ++q) {
            step(fa0, fw0, fa1, fw1, std::false_type{});
            step(fa1, fw1, fa0, fw0, std::false_type{});
        }
        epilogue();
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");      // the surplus transfers of the stream's end must land before the LDS is released
}

bool gemm_w1d_supported(const GemmParams& p) {
    return p.Cb && !p.Cf && !p.res && !p.pool4 && p.batch <= 1 && (p.act == ACT_NONE || p.act == ACT_GELU) && (p.K % (2 * WD_BK)) == 0 && p.Wpk &&
           (p.N & 7) == 0 && (p.ldcb & 7) == 0 && (p.lda & 7) == 0 && !p.ln_out && p.N >= 4 &&
           (int64_t)p.M * p.lda * 2 < (1ll << 31) && (int64_t)(p.K / WD_BK) * p.N * 64 < (1ll << 31);
}

const char* launch_gemm_w1d(const GemmParams& p, hipStream_t stream) {
    if (!gemm_w1d_supported(p)) return "gemm_w1d: unsupported problem";
    const int tiles_m = (p.M + 255) / 256, tiles_n = (p.N + 255) / 256;
    const int slots = ((tiles_m + 7) / 8) * 8 * tiles_n;
    hipLaunchKernelGGL(gemm_w1d_kernel, dim3(slots < 256 ? slots : 256), dim3(256), WD_LDS, stream, p);
    return nullptr;
}

const char* gemm_w1d_init_device() {
    hipError_t st = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_w1d_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, WD_LDS);
    return st == hipSuccess ? nullptr : hipGetErrorString(st);
}
