// bf16-output GEMM with ONE wave per SIMD: 256x256 tile, 4 waves of 128x128 (256 fp32 accumulator registers each, the unified 512-entry
// register file of a wave that has its SIMD to itself), K-steps of 32 through a 3-stage LDS ring.
//
// Why (vendor-library yardstick): with 8 waves of 128x64 a K-step costs the CU 96 KB of LDS fragment reads + 32 KB of
// operand writes = 1 024 cycles of LDS at 128 B/clk against 1 024 cycles of MFMA - the LDS is co-critical.  A 128x128 wave tile reads
// (128 + 128) x 64 B per K-step for 64 MFMAs instead of (128 + 64) x 64 B for 32: 64 + 32 = 96 KB per K-step, 25 % off the LDS.
// There is no partner wave on the SIMD to hide latency behind, so everything is pipelined inside the wave: operands travel
// global -> registers (issued two K-steps ahead, plain vector loads: an LDS-DMA issue blocks the issuing wave for 60-185 cycles, which
// a lone wave cannot afford) -> LDS (one K-step ahead) -> fragment registers (double-buffered, read during the previous step's MFMAs).
#include "common.h"
#include "kernels.h"

#include <hip/hip_runtime.h>

#include <type_traits>

#define W1_BK 32
#define W1_STAGE ((256 + 256) * W1_BK * 2)       // 32 KB: A rows then W rows, 64 B per row
#define W1_NST 3
#define W1_LDS (W1_NST * W1_STAGE)

__device__ __forceinline__ int w1_perm(int row) { return (0x78 >> (((row >> 2) & 3) * 2)) & 3; }   // {0, 2, 3, 1}: gemm.hip's swz2
__device__ __forceinline__ int w1_swz(int row, int chunk) { return row * 64 + ((chunk ^ w1_perm(row)) << 4); }
__device__ __forceinline__ bool w1_tile_map(int b, int tiles_m, int tiles_n, int* tm, int* tn) {       // XCD-aware (gemm.hip tile_map)
    const int xcd = b & 7, q = b >> 3;
    *tn = q % tiles_n;
    *tm = (q / tiles_n) * 8 + xcd;
    return *tm < tiles_m;
}

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void gemm_bf16_w1_kernel(GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int fi = lane & 15, fg = lane >> 4;
    const int tiles_m = (p.M + 255) / 256, tiles_n = (p.N + 255) / 256;
    const int padded = ((tiles_m + 7) / 8) * 8 * tiles_n;
    const int nk = p.K / W1_BK;
    auto next_tile = [&](int L, int* tm, int* tn) {
        while (L < padded && !w1_tile_map(L, tiles_m, tiles_n, tm, tn)) L += gridDim.x;
        return L;
    };

    // PERSISTENT, one continuous stream of K-steps across the tiles of this workgroup (stream index s): the issue cursor (global loads)
    // runs three K-steps ahead of the compute cursor, also across a tile boundary, so a tile's epilogue is followed by the next tile's
    // first MFMAs without a prologue.
    // operand pieces of a K-step: 256 rows x 4 chunks of 16 B per operand = 1 024 pieces, 4 per thread and operand (byte offsets from the
    // operand's base fit 32 bits: the launcher checks M x lda and N x ldw)
    uint32_t aoff[4], woff[4];
    const int prow = tid >> 2, pchunk = tid & 3;           // piece i: row prow + 64 i, chunk pchunk
    const int ldst0 = w1_swz(prow, pchunk);                // piece i lands 64 rows = 4 096 B further (the swizzle term repeats every 16 rows)
    int Li, tmi = 0, tni = 0, kti = 0;                     // issue cursor
    auto set_issue_tile = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            aoff[i] = (uint32_t)(((int64_t)min(tmi * 256 + prow + 64 * i, p.M - 1) * p.lda + pchunk * 8) * 2);
            woff[i] = (uint32_t)(((int64_t)min(tni * 256 + prow + 64 * i, p.N - 1) * p.ldw + pchunk * 8) * 2);
        }
    };
    Li = next_tile(blockIdx.x, &tmi, &tni);
    if (Li >= padded) return;                              // block-uniform
    set_issue_tile();
    int Lc = Li, tmc = tmi, tnc = tni;                     // compute cursor
    const char* Ab = reinterpret_cast<const char*>(p.A);
    const char* Wb = reinterpret_cast<const char*>(p.W);
    u32x4 ga0[4], gw0[4], ga1[4], gw1[4];                  // two sets of staging registers: a global load has two K-steps to land
    auto advance_issue = [&]() {                            // past the end of the stream the cursor keeps cycling over its last tile: up to
        if (++kti == nk) {                                 // three K-steps are loaded and never consumed, which keeps every step identical
            kti = 0;
            if (Li < padded) {
                Li = next_tile(Li + gridDim.x, &tmi, &tni);
                if (Li < padded) set_issue_tile();
            }
        }
    };
    auto gload_all = [&](u32x4 (&ga)[4], u32x4 (&gw)[4]) {  // the issue cursor's K-step into a set of staging registers
        const char* a = Ab + kti * (W1_BK * 2);
        const char* w = Wb + kti * (W1_BK * 2);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            ga[i] = *reinterpret_cast<const u32x4*>(a + aoff[i]);
            gw[i] = *reinterpret_cast<const u32x4*>(w + woff[i]);
        }
        advance_issue();
    };
    auto lstore_all = [&](int stage, u32x4 (&ga)[4], u32x4 (&gw)[4]) {
        char* sa = smem + stage * W1_STAGE;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            *reinterpret_cast<u32x4*>(sa + ldst0 + i * 4096) = ga[i];
            *reinterpret_cast<u32x4*>(sa + 256 * 64 + ldst0 + i * 4096) = gw[i];
        }
    };
    f32x4 acc[8][8];
    // Fragments: inline-asm LDS reads (the compiler cannot see them, so it neither waits for them one by one nor moves them), awaited
    // by one counted wait that names the registers.  MFMAs: inline asm with the accumulator constrained to an AccVGPR and the fragments
    // to VGPRs - left to itself hipcc put the 256 accumulators into VGPRs and shuttled every fragment through v_accvgpr_read.
    bf16x8 fa0[8], fw0[8], fa1[8], fw1[8];
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)(smem);
    const uint32_t fa_base = lds0 + w1_swz(wm * 128 + fi, fg);                 // + i * 1024 (16 rows of 64 B; the swizzle term repeats every 16 rows)
    const uint32_t fw_base = lds0 + 256 * 64 + w1_swz(wn * 128 + fi, fg);
#define W1_DSR(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))
#define W1_MFMA(i, j, fa, fw) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[i][j]) : "v"(fw[j]), "v"(fa[i]))
#define W1_MFMA0(i, j, fa, fw) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=a"(acc[i][j]) : "v"(fw[j]), "v"(fa[i]))
    auto frag_read = [&](int q, uint32_t ab, uint32_t wb, bf16x8 (&na)[8], bf16x8 (&nw)[8]) {     // q-th of the 16 fragment reads of a K-step
        switch (q) {
            case 0: W1_DSR(nw[0], wb, 0); break;      case 1: W1_DSR(nw[1], wb, 1024); break;
            case 2: W1_DSR(nw[2], wb, 2048); break;   case 3: W1_DSR(nw[3], wb, 3072); break;
            case 4: W1_DSR(na[0], ab, 0); break;      case 5: W1_DSR(nw[4], wb, 4096); break;
            case 6: W1_DSR(nw[5], wb, 5120); break;   case 7: W1_DSR(nw[6], wb, 6144); break;
            case 8: W1_DSR(nw[7], wb, 7168); break;   case 9: W1_DSR(na[1], ab, 1024); break;
            case 10: W1_DSR(na[2], ab, 2048); break;  case 11: W1_DSR(na[3], ab, 3072); break;
            case 12: W1_DSR(na[4], ab, 4096); break;  case 13: W1_DSR(na[5], ab, 5120); break;
            case 14: W1_DSR(na[6], ab, 6144); break;  default: W1_DSR(na[7], ab, 7168); break;
        }
    };
    auto frags_landed = [&](bf16x8 (&fa)[8], bf16x8 (&fw)[8]) {
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+v"(fa[0]), "+v"(fa[1]), "+v"(fa[2]), "+v"(fa[3]), "+v"(fa[4]), "+v"(fa[5]), "+v"(fa[6]), "+v"(fa[7]),
                       "+v"(fw[0]), "+v"(fw[1]), "+v"(fw[2]), "+v"(fw[3]), "+v"(fw[4]), "+v"(fw[5]), "+v"(fw[6]), "+v"(fw[7]));
    };
    // epilogue of the compute cursor's tile: bias (+ GELU), bf16, 8-byte stores straight from the accumulator layout (lane = row fi,
    // columns 4 fg ..: four lanes cover 32 B of a row, the four fragments j .. j + 3 of a row complete its 128-B line back to back)
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
                const uint32_t rowoff = (uint32_t)(((int64_t)min(m, p.M - 1) * p.ldcb) * 2);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const f32x4 a = acc[i][jh * 4 + j];
                    float v[4] = {a[0] + bias4[j].x, a[1] + bias4[j].y, a[2] + bias4[j].z, a[3] + bias4[j].w};
                    if (p.act == ACT_GELU) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] = gelu_erf(v[r]);
                    }
                    const int n = n0 + (jh * 4 + j) * 16 + fg * 4;
                    if (m < p.M && n < p.N) *reinterpret_cast<uint2*>(Cb + rowoff + n * 2) = make_uint2(pack_bf16(v[0], v[1]), pack_bf16(v[2], v[3]));
                }
            }
        }
        Lc = next_tile(Lc + gridDim.x, &tmc, &tnc);
    };
    // one K-step: 16 groups of 4 MFMAs (i = g / 2, j = 4 (g % 2) ..), each followed by one piece of the step's memory work.
    // Stream step s computes K-step s from the fragments read in step s - 1, reads the fragments of s + 1 (stage (s + 1) % 3), writes
    // K-step s + 2 from staging set s % 2 (loaded in step s - 2) into stage (s + 2) % 3 (held s - 1: its fragments were read in step
    // s - 2 and every wave has passed the barrier of step s - 1 since), and issues the global loads of K-step s + 4 into the same set.
    int sstage = 0;                                        // stage of the compute cursor's K-step
    auto step = [&](bf16x8 (&fa)[8], bf16x8 (&fw)[8], bf16x8 (&na)[8], bf16x8 (&nw)[8], u32x4 (&ga)[4], u32x4 (&gw)[4], auto first_c) {
        // straight-line; the tile loop below is a fixed sequence of these (variants selected at run time inside the loop made hipcc copy
        // fragments and accumulators between register sets at every join: 2 400-6 000 spills)
        constexpr bool FIRST = decltype(first_c)::value;
        frags_landed(fa, fw);
        const int s1 = sstage == W1_NST - 1 ? 0 : sstage + 1, s2 = s1 == W1_NST - 1 ? 0 : s1 + 1;
        const uint32_t ab = fa_base + s1 * W1_STAGE, wb = fw_base + s1 * W1_STAGE;
        char* sd = smem + s2 * W1_STAGE;
        const char* ga_src = Ab + kti * (W1_BK * 2);
        const char* gw_src = Wb + kti * (W1_BK * 2);
#pragma unroll
        for (int g = 0; g < 16; ++g) {
            const int i = g >> 1, j0 = (g & 1) * 4;
            if (FIRST) { W1_MFMA0(i, j0, fa, fw); W1_MFMA0(i, j0 + 1, fa, fw); W1_MFMA0(i, j0 + 2, fa, fw); W1_MFMA0(i, j0 + 3, fa, fw); }
            else { W1_MFMA(i, j0, fa, fw); W1_MFMA(i, j0 + 1, fa, fw); W1_MFMA(i, j0 + 2, fa, fw); W1_MFMA(i, j0 + 3, fa, fw); }
#ifndef W1_NO_FRAG
            frag_read(g, ab, wb, na, nw);
#endif
#ifndef W1_NO_MEM
            // the 8 LDS writes of K-step s + 2 go behind groups 0, 2, .., 14; each frees its staging register for the global load of
            // K-step s + 3, issued one group later
            if ((g & 1) == 0) {
                const int k = g >> 1;
                if (k < 4) *reinterpret_cast<u32x4*>(sd + ldst0 + k * 4096) = ga[k];
                else *reinterpret_cast<u32x4*>(sd + 256 * 64 + ldst0 + (k - 4) * 4096) = gw[k - 4];
            } else {
                const int k = g >> 1;
                if (k < 4) ga[k] = *reinterpret_cast<const u32x4*>(ga_src + aoff[k]);
                else gw[k - 4] = *reinterpret_cast<const u32x4*>(gw_src + woff[k - 4]);
            }
#endif
        }
        advance_issue();
        sstage = s1;
#ifndef W1_NO_BAR
        __syncthreads();
#endif
    };

    // stream prologue: K-steps 0 and 1 into the ring, K-steps 2 and 3 on their way in the two staging sets, fragments of K-step 0 on their way
    gload_all(ga0, gw0);
    lstore_all(0, ga0, gw0);
    gload_all(ga1, gw1);
    lstore_all(1, ga1, gw1);
    gload_all(ga0, gw0);
    gload_all(ga1, gw1);
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 16; ++q) frag_read(q, fa_base, fw_base, fa0, fw0);
    // a tile is an even number of K-steps (K % 64 == 0), so every tile starts with its fragments in buffer 0 and staging set 0
    const int pairs = nk / 2 - 1;
    while (Lc < padded) {
        step(fa0, fw0, fa1, fw1, ga0, gw0, std::true_type{});       // K-step 0 writes the accumulators (srcC = 0)
        step(fa1, fw1, fa0, fw0, ga1, gw1, std::false_type{});
        for (int q = 0; q < pairs; ++q) {
            step(fa0, fw0, fa1, fw1, ga0, gw0, std::false_type{});
            step(fa1, fw1, fa0, fw0, ga1, gw1, std::false_type{});
        }
        epilogue();
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");      // the surplus loads of the stream's end
}

bool gemm_w1_supported(const GemmParams& p) {
    return p.Cb && !p.Cf && !p.res && !p.pool4 && p.batch <= 1 && (p.act == ACT_NONE || p.act == ACT_GELU) && (p.K % (2 * W1_BK)) == 0 &&
           (p.N & 7) == 0 && (p.ldcb & 7) == 0 && (p.lda & 7) == 0 && (p.ldw & 7) == 0 && !p.ln_out &&
           (int64_t)p.M * p.lda * 2 < (1ll << 32) && (int64_t)p.N * p.ldw * 2 < (1ll << 32) && (int64_t)p.M * p.ldcb * 2 < (1ll << 32) && p.N >= 4;
}

const char* launch_gemm_w1(const GemmParams& p, hipStream_t stream) {
    if (!gemm_w1_supported(p)) return "gemm_w1: unsupported problem";
    const int tiles_m = (p.M + 255) / 256, tiles_n = (p.N + 255) / 256;
    const int slots = ((tiles_m + 7) / 8) * 8 * tiles_n;
    hipLaunchKernelGGL(gemm_bf16_w1_kernel, dim3(slots < 256 ? slots : 256), dim3(256), W1_LDS, stream, p);
    return nullptr;
}

const char* gemm_w1_init_device() {
    hipError_t st = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_bf16_w1_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, W1_LDS);
    return st == hipSuccess ? nullptr : hipGetErrorString(st);
}
