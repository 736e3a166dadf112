// Probe (development): shader clock under load.  s_memtime counts shader-engine cycles, s_memrealtime a constant 100 MHz reference: their
// ratio over a long loop is the clock the CU actually ran at.  Three loads: idle-ish (one wave, VALU only), MFMA bf16 on every SIMD, MFMA + LDS.
// hipcc --offload-arch=gfx950 -O2 -shared -fPIC tools/probes/clock_probe.hip -o /tmp/clock_probe.so ; driven by tools/probes/clock_probe.py
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void clock_kernel(int mode, int iters, unsigned long long* out, float* sink) {
    f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.001f * threadIdx.x); b[i] = (__bf16)(0.002f * i); }
    float v = threadIdx.x * 1e-3f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
        if (mode == 0) {
#pragma unroll
            for (int k = 0; k < 16; ++k) v = v * 1.0001f + 0.5f;
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[q], 0, 0, 0);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = t1 - t0; out[2 * blockIdx.x + 1] = r1 - r0; }
    if (sink) sink[blockIdx.x * 256 + threadIdx.x] = v + acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3];
}
extern "C" void clock_run(int mode, int blocks, int iters, unsigned long long* out, float* sink) {
    hipLaunchKernelGGL(clock_kernel, dim3(blocks), dim3(256), 0, 0, mode, iters, out, sink);
}
