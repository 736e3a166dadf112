// Intake probe: how many bytes per clock can one CU pull from L2/HBM with the access pattern of the 256x256-tile GEMM's operand
// loads?  Variants: LDS-DMA (global_load_lds_dwordx4) or register loads (global_load_dwordx4); pieces of 16 rows x 64 B or 8 rows x 128 B;
// 2..8 issuing waves per workgroup.  No MFMA, no LDS reads: this is the ceiling of the feed alone.
//   hipcc --offload-arch=gfx950 -O3 -o dma_probe dma_probe.hip && ./dma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// A [M][K] bf16 row-major (lda = K).  Every workgroup walks tiles of 256 rows (tile = blockIdx.x + i * gridDim.x); per K-step of KS
// elements it fetches 256 rows x KS*2 bytes for "A" and the same again for "W" (a second matrix of 256*tiles_n rows).
template <int ROWB, bool TO_LDS, int INFLIGHT, bool PANEL = false, bool PANEL_W_ONLY = false>
__global__ __launch_bounds__(512) void probe(const unsigned short* __restrict__ A, const unsigned short* __restrict__ W, int M, int K, int n_issuers,
                                             unsigned long long* out_cycles, float* sink, int passes) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr int ROWS_PER_PIECE = 1024 / ROWB;            // 16 or 8
    constexpr int LANES_PER_ROW = ROWB / 16;                // 4 or 8
    constexpr int KS = ROWB / 2;                            // elements per K-step: 32 or 64
    const int lrow = lane / LANES_PER_ROW, lslot = lane % LANES_PER_ROW;
    const int pieces_per_operand = 256 / ROWS_PER_PIECE;    // 16 or 32
    const int tiles = M / 256, nk = K / KS;
    float acc = 0.f;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (wave < n_issuers) {
        int slot = 0;
        for (int pass = 0; pass < passes; ++pass)
        for (int tile = (blockIdx.x + pass * 37) % gridDim.x; tile < tiles; tile += gridDim.x) {
            for (int kt = 0; kt < nk; ++kt) {
                // this wave's share of the 2 * pieces_per_operand pieces of the K-step
                for (int pc = wave; pc < 2 * pieces_per_operand; pc += n_issuers) {
                    const bool isw = pc >= pieces_per_operand;
                    const int pr = (isw ? pc - pieces_per_operand : pc) * ROWS_PER_PIECE + lrow;
                    const unsigned short* src = (PANEL && (isw || !PANEL_W_ONLY)) ? (isw ? W + ((size_t)kt * 2048 + (tile % 8) * 256 + pr) * KS + lslot * 8 : A + ((size_t)kt * M + tile * 256 + pr) * KS + lslot * 8) : (isw ? W + (size_t)((tile % 8) * 256 + pr) * K : A + (size_t)(tile * 256 + pr) * K) + kt * KS + lslot * 8;
                    if (TO_LDS) {
                        __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(smem + ((slot & 63) * 8 + wave) * 1024 % (128 * 1024)), 16, 0, 0);
                    } else {
                        const uint4 v = *reinterpret_cast<const uint4*>(src);
                        acc += __uint_as_float(v.x & 0x3f800000u);
                    }
                    ++slot;
                }
                if (TO_LDS) {
                    if (INFLIGHT >= 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
                    else if (INFLIGHT >= 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                    else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                }
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (tid == 0) out_cycles[blockIdx.x] = t1 - t0;
    if (acc == 123.456f) sink[0] = acc;
}

template <int ROWB, bool TO_LDS, int INFLIGHT, bool PANEL = false, bool PANEL_W_ONLY = false>
void run(const char* name, const unsigned short* A, const unsigned short* W, int M, int K, int issuers, unsigned long long* cyc, float* sink, int passes) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(probe<ROWB, TO_LDS, INFLIGHT, PANEL, PANEL_W_ONLY>), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(a);
        hipLaunchKernelGGL((probe<ROWB, TO_LDS, INFLIGHT, PANEL, PANEL_W_ONLY>), dim3(256), dim3(512), 128 * 1024, 0, A, W, M, K, issuers, cyc, sink, passes);
        hipEventRecord(b); hipEventSynchronize(b);
    }
    float ms; hipEventElapsedTime(&ms, a, b);
    const double bytes = 2.0 * (double)M * K * 2 * passes;   // A and W-sized streams
    printf("%-34s issuers %d  %.3f ms  %.2f TB/s chip  %.1f GB/s per CU\n", name, issuers, ms, bytes / ms / 1e9, bytes / ms / 1e6 / 256);
}

int main() {
    const int M = 86016, K = 576 * 4;    // fc2-sized A: 396 MB
    unsigned short *A, *W; unsigned long long* cyc; float* sink;
    hipMalloc(&A, (size_t)M * K * 2); hipMalloc(&W, (size_t)2048 * K * 2); hipMalloc(&cyc, 256 * 8); hipMalloc(&sink, 4);
    hipMemset(A, 0, (size_t)M * K * 2); hipMemset(W, 0, (size_t)2048 * K * 2);
    struct Case { int M; int passes; const char* what; } cases[] = {{86016, 4, "A streamed from HBM (396 MB, 4 passes)"}, {16384, 24, "A 75 MB: MALL-resident, 24 passes"}, {65536 / 32, 192, "A 9.4 MB: L2-sized, 192 passes"}};
    for (const Case& c : cases) {
        printf("---- %s\n", c.what);
        for (int issuers : {8, 4}) {
            run<64, true, 8>("lds-dma 16 rows x 64 B  vm8", A, W, c.M, K, issuers, cyc, sink, c.passes);
            run<128, true, 8>("lds-dma  8 rows x 128 B vm8", A, W, c.M, K, issuers, cyc, sink, c.passes);
            run<128, true, 12>("lds-dma  8 rows x 128 B vm12", A, W, c.M, K, issuers, cyc, sink, c.passes);
            run<64, true, 8, true>("lds-dma K-panel layout (1 KB contiguous) vm8", A, W, c.M, K, issuers, cyc, sink, c.passes);
            run<64, true, 12, true>("lds-dma K-panel layout vm12", A, W, c.M, K, issuers, cyc, sink, c.passes);
            run<64, true, 8, true, true>("lds-dma A 16x64B rows, W panel vm8", A, W, c.M, K, issuers, cyc, sink, c.passes);
        }
    }
    return 0;
}
