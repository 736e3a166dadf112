// Shared device helpers for the gfx950 kernels (wave64, MFMA 16x16x32 on 16-bit operands).
//
// OPERAND TYPE.  Every MFMA kernel of the engine takes 16-bit operands and accumulates in fp32.  Which 16-bit type is a build-time
// parameter of the kernel sources: each of them is compiled twice (csrc/Makefile, op_wrap.hip), once with bf16 operands (namespace
// op_bf16: 8 exponent / 7 mantissa bits, the headline arithmetic of BASELINE configs[1]) and once with SABER_OP_F16 (namespace op_f16:
// IEEE half, 5 / 10 bits - the mantissa width of the TF32 arithmetic the reference enables on its GPUs, saber/utils/io.py:127-130; same
// MFMA rate: v_mfma_f32_16x16x32_f16).  Kernels only ever name the type through op16x8 / pack_op16 / f2op / op2f / op16_lo / op16_hi /
// MFMA_16x16x32 below; storage is uint16_t (`bf16_t`, the name predates the second type) in either build.  kernels.h dispatches every
// launcher on the calling thread's operand type (saber_op_is_f16()), which the engine sets from its precision mode.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>
#include <stdint.h>

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));  // staging registers: a first-class vector (HIP's uint4 is a struct and can end up in scratch)
typedef uint16_t bf16_t;  // storage type of a 16-bit operand in global memory (bf16 or fp16 bits, see above)

#define WAVE 64

// both conversion sets exist in every build (the classifier's two element-wise kernels pick one at run time)
typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float bf16_bits_to_f32(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
__device__ __forceinline__ float f16_bits_to_f32(bf16_t v) { return (float)__builtin_bit_cast(_Float16, v); }
// round-to-nearest-even f32 -> 16-bit pair (v_cvt_pk_bf16_f32 / v_cvt_pk_f16_f32)
// (a vector conversion: ONE v_cvt_pk_bf16_f32.  Through hip_bf16.h's __float22bfloat162_rn the same pair came out, inside the big kernels,
// as two single conversions + a shift + an SDWA or - four VALU instructions per packed register in every bf16 epilogue; found in round 5
// in the instruction stream of gemm_w1e.hip)
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pack_bf16_rn(float lo, float hi) {
    const f32x2 x = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(x, bf16x2_t));
}
__device__ __forceinline__ uint32_t pack_f16_rn(float lo, float hi) {
    const f16x2_t v = {(_Float16)lo, (_Float16)hi};
    return __builtin_bit_cast(uint32_t, v);
}

#ifdef SABER_OP_F16
typedef _Float16 op16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 op16x4 __attribute__((ext_vector_type(4)));
#define MFMA_16x16x32 __builtin_amdgcn_mfma_f32_16x16x32_f16
#define MFMA_32x32x16 __builtin_amdgcn_mfma_f32_32x32x16_f16
#define SABER_OP_NAME "f16"
__device__ __forceinline__ float op2f(bf16_t v) { return f16_bits_to_f32(v); }
__device__ __forceinline__ uint32_t pack_op16(float lo, float hi) { return pack_f16_rn(lo, hi); }
// the two halves of a packed pair as fp32 (v_cvt_f32_f16, the upper one with an SDWA word select)
__device__ __forceinline__ float op16_lo(uint32_t u) { return (float)__builtin_bit_cast(f16x2_t, u)[0]; }
__device__ __forceinline__ float op16_hi(uint32_t u) { return (float)__builtin_bit_cast(f16x2_t, u)[1]; }
#else
typedef __bf16 op16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 op16x4 __attribute__((ext_vector_type(4)));
#define MFMA_16x16x32 __builtin_amdgcn_mfma_f32_16x16x32_bf16
#define MFMA_32x32x16 __builtin_amdgcn_mfma_f32_32x32x16_bf16
#define SABER_OP_NAME "bf16"
__device__ __forceinline__ float op2f(bf16_t v) { return bf16_bits_to_f32(v); }
__device__ __forceinline__ uint32_t pack_op16(float lo, float hi) { return pack_bf16_rn(lo, hi); }
__device__ __forceinline__ float op16_lo(uint32_t u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float op16_hi(uint32_t u) { return __uint_as_float(u & 0xffff0000u); }
#endif
__device__ __forceinline__ bf16_t f2op(float x) { return (bf16_t)(pack_op16(x, 0.f) & 0xffffu); }

// erf by Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7 absolute, ~12 VALU ops instead of libm's ~60): the exact-erf GELU of
// SAM2 is evaluated ~5e10 times per slice, so libm erff alone would cost tens of ms of pure VALU time.
__device__ __forceinline__ float fast_erf(float x) {
    const float ax = fabsf(x);
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.0f));   // v_rcp_f32 (1 ulp); an IEEE divide costs ~10 VALU ops
    float p = 1.061405429f;
    p = fmaf(p, t, -1.453152027f);
    p = fmaf(p, t, 1.421413741f);
    p = fmaf(p, t, -0.284496736f);
    p = fmaf(p, t, 0.254829592f);
    const float y = 1.0f - p * t * __expf(-ax * ax);
    return copysignf(y, x);
}
// GELU(x) = x * Phi(x), written as x * sigmoid(x * q(x^2)) with q fitted (tools/fit_gelu.py, minimax over [-9, 9]) to the EXACT
// erf form the reference uses: max |error| 2.6e-5 absolute -- below the bf16 rounding of every consumer of these values --
// in 9 VALU ops instead of ~25 for the erf polynomial.  SAM2 evaluates GELU ~5e10 times per slice (MLP of every Hiera
// block, both stages of the mask upscaling), which made the erf form a VALU roofline of its own (~30 ms per slice).
// x^2 is clamped at 50: beyond |x| = 7.07 the sigmoid is saturated and the quartic term must not take over.
__device__ __forceinline__ float gelu_erf(float x) {
    const float x2 = fminf(x * x, 50.0f);
    float q = fmaf(x2, 1.01426305e-3f, -1.06775724e-1f);      // -log2(e) * (c2 x^4 + c1 x^2 + c0), c = {1.59501577, 7.4011292e-2, -7.03033575e-4}
    q = fmaf(q, x2, -2.30112134f);
    const float e = __builtin_amdgcn_exp2f(x * q);            // exp(-x q(x^2))
    return x * __builtin_amdgcn_rcpf(1.0f + e);
}

// Cross-lane reductions over lane ^ 16 and lane ^ 32 (the two steps that leave a 16-lane DPP row).  __shfl_xor compiles to
// ds_bpermute_b32, an LDS round trip of ~100 cycles on the critical path of every softmax / LayerNorm statistic; gfx950's
// v_permlane16_swap / v_permlane32_swap are plain VALU instructions.  swap(x, x) leaves {x[own half], x[other half]} in the pair.
__device__ __forceinline__ void xor16_pair(float x, float* a, float* b) {
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    *a = __uint_as_float(r[0]); *b = __uint_as_float(r[1]);
}
__device__ __forceinline__ void xor32_pair(float x, float* a, float* b) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    *a = __uint_as_float(r[0]); *b = __uint_as_float(r[1]);
}
// v_max_f32 without the canonicalising v_max(x, x) hipcc puts in front of fmaxf on values it cannot prove quiet
__device__ __forceinline__ float vmax(float a, float b) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float vmin(float a, float b) { float r; asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float xor16_max(float x) { float a, b; xor16_pair(x, &a, &b); return vmax(a, b); }
__device__ __forceinline__ float xor32_max(float x) { float a, b; xor32_pair(x, &a, &b); return vmax(a, b); }
__device__ __forceinline__ float xor16_min(float x) { float a, b; xor16_pair(x, &a, &b); return vmin(a, b); }
__device__ __forceinline__ float xor32_min(float x) { float a, b; xor32_pair(x, &a, &b); return vmin(a, b); }
__device__ __forceinline__ float xor16_sum(float x) { float a, b; xor16_pair(x, &a, &b); return a + b; }
__device__ __forceinline__ float xor32_sum(float x) { float a, b; xor32_pair(x, &a, &b); return a + b; }

// value of lane ^ 16 / lane ^ 32 (odd = this lane sits in the odd 16-lane row of its pair / in the upper half of the wave)
__device__ __forceinline__ float shfl_xor16(float x, bool odd) { float a, b; xor16_pair(x, &a, &b); return odd ? a : b; }
__device__ __forceinline__ float shfl_xor32(float x, bool upper) { float a, b; xor32_pair(x, &a, &b); return upper ? a : b; }

// two GELUs at once: the fma / mul / add steps as packed-f32 instructions (v_pk_mul_f32, v_pk_fma_f32, v_pk_add_f32: two lanes'
// worth of work per issue slot), min / exp2 / rcp per component.  Same arithmetic as gelu_erf.
__device__ __forceinline__ f32x2 gelu_erf2(f32x2 x) {
    f32x2 x2 = x * x;
    x2.x = fminf(x2.x, 50.0f); x2.y = fminf(x2.y, 50.0f);
    f32x2 q = __builtin_elementwise_fma(x2, (f32x2){1.01426305e-3f, 1.01426305e-3f}, (f32x2){-1.06775724e-1f, -1.06775724e-1f});
    q = __builtin_elementwise_fma(q, x2, (f32x2){-2.30112134f, -2.30112134f});
    const f32x2 t = x * q;
    f32x2 d;
    d.x = __builtin_amdgcn_exp2f(t.x); d.y = __builtin_amdgcn_exp2f(t.y);
    d = d + (f32x2){1.0f, 1.0f};
    f32x2 r;
    r.x = __builtin_amdgcn_rcpf(d.x); r.y = __builtin_amdgcn_rcpf(d.y);
    return x * r;
}

// Lane exchanges inside a 16-lane row as DPP modifiers of ordinary VALU instructions (no LDS round trip: ds_bpermute_b32, which
// __shfl / __shfl_xor compile to, costs ~100 cycles of latency per exchange and shares the LDS pipe with everything else).
//   quad_perm [1,0,3,2] = lane ^ 1, [2,3,0,1] = lane ^ 2, [q,q,q,q] = broadcast of lane q of the quad;
//   row_half_mirror (lane i <-> 7 - i of each 8) and row_mirror (i <-> 15 - i) pair the quads / the two halves of a row, which is
//   all a reduction needs once every lane of a quad (of an 8-group) already holds the same partial result.
#define DPP_XOR1 0xB1
#define DPP_XOR2 0x4E
#define DPP_HALF_MIRROR 0x141
#define DPP_MIRROR 0x140
template <int CTRL> __device__ __forceinline__ float dpp_mov(float x) {
    return __uint_as_float((uint32_t)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(x), CTRL, 0xf, 0xf, true));
}
template <int Q> __device__ __forceinline__ float quad_bcast(float x) { return dpp_mov<Q * 0x55>(x); }   // value of lane Q of this lane's quad
__device__ __forceinline__ float quad_sum(float v) { v += dpp_mov<DPP_XOR1>(v); v += dpp_mov<DPP_XOR2>(v); return v; }
__device__ __forceinline__ float row16_sum(float v) {
    v += dpp_mov<DPP_XOR1>(v); v += dpp_mov<DPP_XOR2>(v); v += dpp_mov<DPP_HALF_MIRROR>(v); v += dpp_mov<DPP_MIRROR>(v);
    return v;
}
__device__ __forceinline__ float row16_max(float v) {
    v = fmaxf(v, dpp_mov<DPP_XOR1>(v)); v = fmaxf(v, dpp_mov<DPP_XOR2>(v)); v = fmaxf(v, dpp_mov<DPP_HALF_MIRROR>(v)); v = fmaxf(v, dpp_mov<DPP_MIRROR>(v));
    return v;
}
__device__ __forceinline__ float row16_min(float v) {
    v = fminf(v, dpp_mov<DPP_XOR1>(v)); v = fminf(v, dpp_mov<DPP_XOR2>(v)); v = fminf(v, dpp_mov<DPP_HALF_MIRROR>(v)); v = fminf(v, dpp_mov<DPP_MIRROR>(v));
    return v;
}
// whole-wave all-reduce: 4 DPP steps inside the rows, then the two permlane swaps across rows
__device__ __forceinline__ float wave_sum(float v) { return xor32_sum(xor16_sum(row16_sum(v))); }
__device__ __forceinline__ float wave_max(float v) { return xor32_max(xor16_max(row16_max(v))); }
__device__ __forceinline__ float wave_min(float v) { return xor32_min(xor16_min(row16_min(v))); }

// Hiera token order used throughout the engine (DESIGN.md "token order"): for the
// 256x256 stage-0 grid, index bits are [y7 y6 x7 x6][y5 y4 y3 x5 x4 x3][y2 x2][y1 x1][y0 x0].
// Every attention window of every stage and every 2x2 pooling group is then a contiguous
// run of rows; stage s uses (idx >> 2s).
__host__ __device__ __forceinline__ int perm_index256(int y, int x) {
    return (((y >> 6) * 4 + (x >> 6)) << 12) | ((((y >> 3) & 7) * 8 + ((x >> 3) & 7)) << 6) |
           ((((y >> 2) & 1) * 2 + ((x >> 2) & 1)) << 4) | ((((y >> 1) & 1) * 2 + ((x >> 1) & 1)) << 2) |
           ((y & 1) * 2 + (x & 1));
}
__host__ __device__ __forceinline__ void perm_coords256(int idx, int* y, int* x) {
    int top = idx >> 12, mid = (idx >> 6) & 63;
    int yy = ((top >> 2) << 6) | ((mid >> 3) << 3) | (((idx >> 5) & 1) << 2) | (((idx >> 3) & 1) << 1) | ((idx >> 1) & 1);
    int xx = ((top & 3) << 6) | ((mid & 7) << 3) | (((idx >> 4) & 1) << 2) | (((idx >> 2) & 1) << 1) | (idx & 1);
    *y = yy; *x = xx;
}
// generic: grid of side g = 256 >> s
__host__ __device__ __forceinline__ int perm_index(int y, int x, int s) { return perm_index256(y << s, x << s) >> (2 * s); }
__host__ __device__ __forceinline__ void perm_coords(int idx, int s, int* y, int* x) {
    int yy, xx; perm_coords256(idx << (2 * s), &yy, &xx); *y = yy >> s; *x = xx >> s;
}
