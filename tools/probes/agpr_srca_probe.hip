// Probe (development): does v_mfma_f32_16x16x32_bf16 read its A operand from AccVGPRs, and do global loads land there?
// hipcc --offload-arch=gfx950 -o agpr_srca_probe agpr_srca_probe.hip && ./agpr_srca_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
__global__ void k(const uint16_t* A, const uint16_t* B, float* out) {
    const int lane = threadIdx.x;
    bf16x8 a = *reinterpret_cast<const bf16x8*>(A + lane * 8), b = *reinterpret_cast<const bf16x8*>(B + lane * 8);
    f32x4 c0 = {0, 0, 0, 0}, c1 = {0, 0, 0, 0}, c2 = {0, 0, 0, 0};
    c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c0, 0, 0, 0);
    // A through v_accvgpr_write
    asm volatile("v_accvgpr_write_b32 a0, %0\n\tv_accvgpr_write_b32 a1, %1\n\tv_accvgpr_write_b32 a2, %2\n\tv_accvgpr_write_b32 a3, %3\n\ts_nop 4"
                 : : "v"(((uint32_t*)&a)[0]), "v"(((uint32_t*)&a)[1]), "v"(((uint32_t*)&a)[2]), "v"(((uint32_t*)&a)[3]) : "a0", "a1", "a2", "a3");
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, a[0:3], %1, %0\n\ts_nop 15\n\ts_nop 7" : "+v"(c1) : "v"(b));
    // A through a global load into AccVGPRs
    const uint16_t* ap = A + lane * 8;
    asm volatile("global_load_dwordx4 a[4:7], %0, off\n\ts_waitcnt vmcnt(0)" : : "v"(ap) : "memory", "a4", "a5", "a6", "a7");
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, a[4:7], %1, %0\n\ts_nop 15\n\ts_nop 7" : "+v"(c2) : "v"(b));
    for (int r = 0; r < 4; ++r) { out[lane * 4 + r] = c0[r]; out[256 + lane * 4 + r] = c1[r]; out[512 + lane * 4 + r] = c2[r]; }
}
int main() {
    std::vector<uint16_t> A(512), B(512);
    for (int i = 0; i < 512; ++i) { A[i] = 0x3f80 + (i * 7) % 64; B[i] = 0x3f00 + (i * 13) % 128; }
    uint16_t *dA, *dB; float* dO;
    hipMalloc(&dA, 1024); hipMalloc(&dB, 1024); hipMalloc(&dO, 768 * 4);
    hipMemcpy(dA, A.data(), 1024, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), 1024, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dO);
    std::vector<float> o(768);
    hipMemcpy(o.data(), dO, 768 * 4, hipMemcpyDeviceToHost);
    double d1 = 0, d2 = 0;
    for (int i = 0; i < 256; ++i) { d1 += fabs(o[256 + i] - o[i]); d2 += fabs(o[512 + i] - o[i]); }
    printf("builtin c[0..3] = %g %g %g %g\nA from AccVGPR (v_accvgpr_write): sum |diff| = %g\nA from AccVGPR (global_load): sum |diff| = %g\n", o[0], o[1], o[2], o[3], d1, d2);
    return 0;
}
