// Probe (development): lane layout / scale semantics of v_mfma_scale_f32_16x16x128_f8f6f4 with e4m3 operands, and of v_cvt_pk_fp8_f32.
// hipcc --offload-arch=gfx950 -shared -fPIC tools/probes/mx_probe.hip -o /tmp/mx_probe.so ; driven by tools/probes/mx_probe.py
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
// A [16][128] bytes, B [16][128] bytes (row-major, e4m3), D [16][16] fp32: D[i][j] = sum_k A[i][k] B[j][k] * 2^(sa-127) * 2^(sb-127)
__global__ void mx_kernel(const uint8_t* A, const uint8_t* B, float* D, int sa, int sb) {
    const int lane = threadIdx.x, fi = lane & 15, fg = lane >> 4;
    v8i a = *reinterpret_cast<const v8i*>(A + fi * 128 + 32 * fg);
    v8i b = *reinterpret_cast<const v8i*>(B + fi * 128 + 32 * fg);
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, 0, sa, 0, sb);
    for (int r = 0; r < 4; ++r) D[(4 * fg + r) * 16 + fi] = c[r];     // row = 4 fg + r (A index), col = fi (B index)
}
__global__ void cvt_kernel(const float* x, uint8_t* out, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (2 * i + 1 < n) {
        const int v = __builtin_amdgcn_cvt_pk_fp8_f32(x[2 * i], x[2 * i + 1], 0, false);
        out[2 * i] = v & 0xff; out[2 * i + 1] = (v >> 8) & 0xff;
    }
}
extern "C" void mx_run(const uint8_t* A, const uint8_t* B, float* D, int sa, int sb) { hipLaunchKernelGGL(mx_kernel, dim3(1), dim3(64), 0, 0, A, B, D, sa, sb); }
extern "C" void cvt_run(const float* x, uint8_t* out, int n) { hipLaunchKernelGGL(cvt_kernel, dim3((n / 2 + 255) / 256), dim3(256), 0, 0, x, out, n); }
// per-lane block scales: SA [16][4] e8m0 bytes (row, 32-element K block), SB likewise; each lane passes ITS byte (row fi, block fg) in
// byte `sel` of the scale VGPR (the other bytes hold 0xff = NaN so that a wrong byte select shows)
template <int SEL>
__global__ void mx_scale_kernel(const uint8_t* A, const uint8_t* B, const uint8_t* SA, const uint8_t* SB, float* D) {
    const int lane = threadIdx.x, fi = lane & 15, fg = lane >> 4;
    // hardware K order (found with mx_scale_raw_kernel below): register bytes 0..15 of lane (fi, fg) are K = 16 fg .. 16 fg + 15, bytes 16..31 are
    // K = 64 + 16 fg .. 64 + 16 fg + 15; the scale of K block b (K = 32 b .. 32 b + 31) comes from lane fg = b
    typedef int v4i __attribute__((ext_vector_type(4)));
    const v4i a0 = *reinterpret_cast<const v4i*>(A + fi * 128 + 16 * fg), a1 = *reinterpret_cast<const v4i*>(A + fi * 128 + 64 + 16 * fg);
    const v4i b0 = *reinterpret_cast<const v4i*>(B + fi * 128 + 16 * fg), b1 = *reinterpret_cast<const v4i*>(B + fi * 128 + 64 + 16 * fg);
    v8i a = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
    v8i b = {b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
    const int sa = (int)(0xffffffffu & ~(0xffu << (8 * SEL))) | ((int)SA[fi * 4 + fg] << (8 * SEL));
    const int sb = (int)(0xffffffffu & ~(0xffu << (8 * SEL))) | ((int)SB[fi * 4 + fg] << (8 * SEL));
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, SEL, sa, SEL, sb);
    for (int r = 0; r < 4; ++r) D[(4 * fg + r) * 16 + fi] = c[r];
}
extern "C" void mx_scale_run(const uint8_t* A, const uint8_t* B, const uint8_t* SA, const uint8_t* SB, float* D, int sel) {
    if (sel == 0) hipLaunchKernelGGL(mx_scale_kernel<0>, dim3(1), dim3(64), 0, 0, A, B, SA, SB, D);
    else if (sel == 1) hipLaunchKernelGGL(mx_scale_kernel<1>, dim3(1), dim3(64), 0, 0, A, B, SA, SB, D);
    else if (sel == 2) hipLaunchKernelGGL(mx_scale_kernel<2>, dim3(1), dim3(64), 0, 0, A, B, SA, SB, D);
    else hipLaunchKernelGGL(mx_scale_kernel<3>, dim3(1), dim3(64), 0, 0, A, B, SA, SB, D);
}
// raw per-lane scale words: SAw / SBw [64] ints, one per lane
__global__ void mx_scale_raw_kernel(const uint8_t* A, const uint8_t* B, const int* SAw, const int* SBw, float* D) {
    const int lane = threadIdx.x, fi = lane & 15, fg = lane >> 4;
    v8i a = *reinterpret_cast<const v8i*>(A + fi * 128 + 32 * fg);
    v8i b = *reinterpret_cast<const v8i*>(B + fi * 128 + 32 * fg);
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, 0, SAw[lane], 0, SBw[lane]);
    for (int r = 0; r < 4; ++r) D[(4 * fg + r) * 16 + fi] = c[r];
}
extern "C" void mx_scale_raw_run(const uint8_t* A, const uint8_t* B, const int* SAw, const int* SBw, float* D) {
    hipLaunchKernelGGL(mx_scale_raw_kernel, dim3(1), dim3(64), 0, 0, A, B, SAw, SBw, D);
}
