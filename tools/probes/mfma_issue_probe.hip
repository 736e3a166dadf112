// Probe (development): how many independent accumulators / waves per SIMD it takes to keep the bf16 MFMA pipe full.
// hipcc --offload-arch=gfx950 -O2 -shared -fPIC tools/probes/mfma_issue_probe.hip -o /tmp/mfma_issue_probe.so ; driven by tools/probes/mfma_issue_probe.py
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC>
__global__ __launch_bounds__(256) void k16(int iters, float* sink) {
    f32x4 acc[NACC];
    for (int q = 0; q < NACC; ++q) acc[q] = (f32x4){0, 0, 0, 0};
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.001f * threadIdx.x); b[i] = (__bf16)(0.002f * i); }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 32 / NACC; ++k)
#pragma unroll
            for (int q = 0; q < NACC; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[q], 0, 0, 0);
    }
    float v = 0.f;
    for (int q = 0; q < NACC; ++q) v += acc[q][0] + acc[q][3];
    sink[blockIdx.x * 256 + threadIdx.x] = v;
}
template <int NACC>
__global__ __launch_bounds__(256) void k32(int iters, float* sink) {
    f32x16 acc[NACC];
    for (int q = 0; q < NACC; ++q) for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.001f * threadIdx.x); b[i] = (__bf16)(0.002f * i); }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 16 / NACC; ++k)
#pragma unroll
            for (int q = 0; q < NACC; ++q) acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[q], 0, 0, 0);
    }
    float v = 0.f;
    for (int q = 0; q < NACC; ++q) v += acc[q][0] + acc[q][15];
    sink[blockIdx.x * 256 + threadIdx.x] = v;
}
extern "C" void run(int kind, int nacc, int blocks, int iters, float* sink) {
#define L(K, N) hipLaunchKernelGGL((K<N>), dim3(blocks), dim3(256), 0, 0, iters, sink)
    if (kind == 16) { if (nacc == 1) L(k16, 1); else if (nacc == 2) L(k16, 2); else if (nacc == 4) L(k16, 4); else if (nacc == 8) L(k16, 8); else L(k16, 16); }
    else { if (nacc == 1) L(k32, 1); else if (nacc == 2) L(k32, 2); else if (nacc == 4) L(k32, 4); else L(k32, 8); }
}
