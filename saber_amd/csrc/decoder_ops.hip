// Prompt-encoder / mask-decoder side kernels.
// These are the HBM / latency-bound pieces around the GEMMs of the two-way transformer:
// point-prompt tokens, the mask_downscaling dense prompt (m2m pass), the tiny fp32
// attentions over 8 tokens, the hypernetwork mask product, dynamic multimask selection,
// and K8: fused bilinear upsample + threshold + stability counts + bbox + bit-packing.
#include <algorithm>

#include "common.h"
#include "kernels.h"

#define DEC_C 256

// ------------------------------------------------------------------------------------------------
// tokens[p] = [obj, iou, mask0..3, point(p), pad]  (8 x 256 fp32)
__global__ __launch_bounds__(256) void prompt_tokens_kernel(const float* __restrict__ pts, const int* __restrict__ labels, int P,
                                                            PromptWeights w, float* __restrict__ tokens) {
    const int p = blockIdx.x, c = threadIdx.x;
    float* t = tokens + (int64_t)p * 8 * DEC_C;
#pragma unroll
    for (int k = 0; k < 6; ++k) t[k * DEC_C + c] = w.out_tokens[k * DEC_C + c];
    const int label = labels ? labels[p] : 1;
    float e = 0.f;
    if (label >= 0) {
        const float x = 2.0f * ((pts[2 * p] + 0.5f) / 1024.0f) - 1.0f;
        const float y = 2.0f * ((pts[2 * p + 1] + 0.5f) / 1024.0f) - 1.0f;
        const int f = c & 127;
        const float a = 6.283185307179586f * (x * w.gauss[f] + y * w.gauss[128 + f]);
        e = (c < 128 ? sinf(a) : cosf(a)) + w.point_embed[(label & 3) * DEC_C + c];
    } else {
        e = w.not_a_point[c];
    }
    t[6 * DEC_C + c] = e;
    t[7 * DEC_C + c] = w.not_a_point[c];
}

// tokens[p] = [obj, iou, mask0..3, point(p, 0) .. point(p, K - 1), pad]  ((7 + K) x 256 fp32): K points per prompt (clicks; a box is its two
// corners with labels 2 / 3, as upstream's SAM2VideoPredictor.add_new_points_or_box hands it to the prompt encoder), then the padding point
__global__ __launch_bounds__(256) void prompt_tokens_multi_kernel(const float* __restrict__ pts, const int* __restrict__ labels, int P, int K,
                                                                  PromptWeights w, float* __restrict__ tokens) {
    const int p = blockIdx.x, c = threadIdx.x;
    float* t = tokens + (int64_t)p * (7 + K) * DEC_C;
#pragma unroll
    for (int k = 0; k < 6; ++k) t[k * DEC_C + c] = w.out_tokens[k * DEC_C + c];
    for (int k = 0; k < K; ++k) {
        const int label = labels ? labels[p * K + k] : 1;
        float e;
        if (label >= 0) {
            const float x = 2.0f * ((pts[2 * (p * K + k)] + 0.5f) / 1024.0f) - 1.0f;
            const float y = 2.0f * ((pts[2 * (p * K + k) + 1] + 0.5f) / 1024.0f) - 1.0f;
            const int f = c & 127;
            const float a = 6.283185307179586f * (x * w.gauss[f] + y * w.gauss[128 + f]);
            e = (c < 128 ? sinf(a) : cosf(a)) + w.point_embed[(label & 3) * DEC_C + c];
        } else {
            e = w.not_a_point[c];
        }
        t[(6 + k) * DEC_C + c] = e;
    }
    t[(6 + K) * DEC_C + c] = w.not_a_point[c];
}
const char* launch_prompt_tokens_multi(const float* pts, const int* labels, int P, int K, PromptWeights w, float* tokens, hipStream_t s) {
    if (P <= 0) return nullptr;
    if (K < 1) return "prompt_tokens: at least one point per prompt";
    hipLaunchKernelGGL(prompt_tokens_multi_kernel, dim3(P), dim3(256), 0, s, pts, labels, P, K, w, tokens);
    return nullptr;
}

const char* launch_prompt_tokens(const float* pts, const int* labels, int P, PromptWeights w, float* tokens, hipStream_t s) {
    if (P <= 0) return nullptr;
    hipLaunchKernelGGL(prompt_tokens_kernel, dim3(P), dim3(256), 0, s, pts, labels, P, w, tokens);
    return nullptr;
}

// ------------------------------------------------------------------------------------------------
// mask_downscaling (conv k2s2 1->4, LN2d, GELU, conv k2s2 4->16, LN2d, GELU, conv 1x1 16->256) fused
// with "src = image_embed + dense".  One block = 64 consecutive tokens of one prompt: phase 1, one thread per
// token computes the 16-channel hidden vector from its 4x4 logit patch; phase 2, thread d produces channel d
// of every token (coalesced 512-B rows).
#define ME_NP 8
template <bool MFMA2>
__global__ __launch_bounds__(256) void mask_embed_src_kernel(const float* __restrict__ mask_in, int P,
                                                             const float* __restrict__ image_embed_base, XMap em, const float* __restrict__ pos,
                                                             MaskEmbedWeights w, float* __restrict__ src_f,
                                                             bf16_t* __restrict__ src_bf, bf16_t* __restrict__ srcpos_bf, float clamp_abs, int raw4_q0) {
    // one block = 64 tokens x ME_NP consecutive prompts: the image-embedding rows (fp32, shared by every prompt of the crop) are
    // read once per block instead of once per prompt
    __shared__ __attribute__((aligned(16))) float h2s[ME_NP][64][20];
    __shared__ __attribute__((aligned(16))) char oscr[MFMA2 ? 4 * 8192 : 16];   // MFMA2: per wave 16 token rows x 512 B for the full-line stores
    const int tid = threadIdx.x;
    const int pg = blockIdx.x >> 6, tok0 = (blockIdx.x & 63) * 64;
    const int p0 = pg * ME_NP, np = min(ME_NP, P - p0);
    for (int pi = 0; pi < np; ++pi) {
        const int p = p0 + pi;
        // phase 1: four adjacent lanes share a token; lane q computes position q of the first conv (2x2 patch -> 4 channels,
        // LN2d, GELU), the four h1 vectors are exchanged by shuffles, then lane q computes channels 4q..4q+3 of the second conv.
        const int tl = tid >> 2, q = tid & 3;
        const int py = q >> 1, px = q & 1;
        int ty, tx;
        perm_coords(tok0 + tl, 2, &ty, &tx);
        // raw4_q0 >= 0: mask_in is the 4-plane-per-prompt output of a multimask decode, prompt q = raw4_q0 + p refines plane 1 + q % 3 of prompt q / 3
        const int64_t plane = raw4_q0 >= 0 ? (int64_t)(raw4_q0 + p) + (raw4_q0 + p) / 3 + 1 : (int64_t)p;
        const float* mp = mask_in + plane * 65536 + (int64_t)(ty * 4 + py * 2) * 256 + tx * 4 + px * 2;
        const float2 r0 = *reinterpret_cast<const float2*>(mp), r1 = *reinterpret_cast<const float2*>(mp + 256);
        // clamp_abs: SAM2ImagePredictor._predict clamps the low-res logits it hands back to +-32 before they are re-used as a mask
        // prompt; the AMG driver passes the raw first-pass logits and has the clamp applied here instead of in a pass of its own
        const float in[4] = {fminf(fmaxf(r0.x, -clamp_abs), clamp_abs), fminf(fmaxf(r0.y, -clamp_abs), clamp_abs),
                             fminf(fmaxf(r1.x, -clamp_abs), clamp_abs), fminf(fmaxf(r1.y, -clamp_abs), clamp_abs)};
        float v[4], mu = 0.f;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float a = w.b1[c];
#pragma unroll
            for (int k = 0; k < 4; ++k) a += w.w1[c * 4 + k] * in[k];
            v[c] = a;
            mu += a;
        }
        mu *= 0.25f;
        float var = 0.f;
#pragma unroll
        for (int c = 0; c < 4; ++c) var += (v[c] - mu) * (v[c] - mu);
        float rstd = __builtin_amdgcn_rsqf(var * 0.25f + 1e-6f);
        float h1[4][4];     // [position ky*2+kx][channel]
        float mine[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) mine[c] = gelu_erf((v[c] - mu) * rstd * w.g1[c] + w.be1[c]);
#pragma unroll
        for (int c = 0; c < 4; ++c) {     // the quad's four h1 vectors: DPP quad broadcasts (were 16 ds_bpermute round trips per token)
            h1[0][c] = quad_bcast<0>(mine[c]); h1[1][c] = quad_bcast<1>(mine[c]); h1[2][c] = quad_bcast<2>(mine[c]); h1[3][c] = quad_bcast<3>(mine[c]);
        }
        float h2[4];
        mu = 0.f;
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) {
            const int c = 4 * q + cc;
            float a = w.b2[c];
#pragma unroll
            for (int ci = 0; ci < 4; ++ci)
#pragma unroll
                for (int k = 0; k < 4; ++k) a += w.w2[(c * 4 + ci) * 4 + k] * h1[k][ci];
            h2[cc] = a;
            mu += a;
        }
        mu = quad_sum(mu);
        mu *= (1.0f / 16.0f);
        var = 0.f;
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) var += (h2[cc] - mu) * (h2[cc] - mu);
        var = quad_sum(var);
        rstd = __builtin_amdgcn_rsqf(var * (1.0f / 16.0f) + 1e-6f);
        float4 o;
        o.x = gelu_erf((h2[0] - mu) * rstd * w.g2[4 * q + 0] + w.be2[4 * q + 0]);
        o.y = gelu_erf((h2[1] - mu) * rstd * w.g2[4 * q + 1] + w.be2[4 * q + 1]);
        o.z = gelu_erf((h2[2] - mu) * rstd * w.g2[4 * q + 2] + w.be2[4 * q + 2]);
        o.w = gelu_erf((h2[3] - mu) * rstd * w.g2[4 * q + 3] + w.be2[4 * q + 3]);
        *reinterpret_cast<float4*>(&h2s[pi][tl][4 * q]) = o;
    }
    __syncthreads();
    if constexpr (MFMA2) {
        // phase 2 on the matrix cores (only the bf16 src is wanted): the 1x1 conv 16 -> 256 is [16 tokens] x [256] over k = 16 (zero-
        // padded to 32) = 16 MFMAs per wave, token group and prompt, instead of 64 FMAs per 8 bytes of output on the VALU (which made
        // this kernel VALU-bound at half the HBM write rate).  Wave w owns tokens 16 w .. 16 w + 15; W3 (bf16) and image_embed + b3
        // stay in registers across the prompts; the tile leaves through LDS as full 512-B token rows (dwordx4).
        const int lane = tid & 63, wv = tid >> 6, fi = lane & 15, fg = lane >> 4;
        op16x8 w3f[16];
#pragma unroll
        for (int nt = 0; nt < 16; ++nt) {
            float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            if (fg < 2) {
                const float4 a = *reinterpret_cast<const float4*>(w.w3 + (16 * nt + fi) * 16 + 8 * fg), b = *reinterpret_cast<const float4*>(w.w3 + (16 * nt + fi) * 16 + 8 * fg + 4);
                v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
            }
            const uint4 u = make_uint4(pack_op16(v[0], v[1]), pack_op16(v[2], v[3]), pack_op16(v[4], v[5]), pack_op16(v[6], v[7]));
            w3f[nt] = __builtin_bit_cast(op16x8, u);
        }
        const int tok = tok0 + 16 * wv + fi;
        float4 eb[16];
        int cur_slot = -1;
        char* my = oscr + wv * 8192;
        for (int pi = 0; pi < np; ++pi) {
            const int p = p0 + pi;
            const int slot = (p + em.off) / em.div;
            if (slot != cur_slot) {        // wave-uniform: prompts of a block share the crop unless the block straddles two crops
                const float* ep = image_embed_base + (int64_t)slot * em.stride + (int64_t)tok * DEC_C;
#pragma unroll
                for (int nt = 0; nt < 16; ++nt) {
                    const float4 e = *reinterpret_cast<const float4*>(ep + 16 * nt + 4 * fg), b3 = *reinterpret_cast<const float4*>(w.b3 + 16 * nt + 4 * fg);
                    eb[nt] = make_float4(e.x + b3.x, e.y + b3.y, e.z + b3.z, e.w + b3.w);
                }
                cur_slot = slot;
            }
            float hv[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            if (fg < 2) {
                const float4 a = *reinterpret_cast<const float4*>(&h2s[pi][16 * wv + fi][8 * fg]), b = *reinterpret_cast<const float4*>(&h2s[pi][16 * wv + fi][8 * fg + 4]);
                hv[0] = a.x; hv[1] = a.y; hv[2] = a.z; hv[3] = a.w; hv[4] = b.x; hv[5] = b.y; hv[6] = b.z; hv[7] = b.w;
            }
            const uint4 hu = make_uint4(pack_op16(hv[0], hv[1]), pack_op16(hv[2], hv[3]), pack_op16(hv[4], hv[5]), pack_op16(hv[6], hv[7]));
            const op16x8 hf = __builtin_bit_cast(op16x8, hu);
#pragma unroll
            for (int nt = 0; nt < 16; ++nt) {
                const f32x4 z = {eb[nt].x, eb[nt].y, eb[nt].z, eb[nt].w};
                const f32x4 a = MFMA_16x16x32(w3f[nt], hf, z, 0, 0, 0);    // a[r] = src[token fi][channel 16 nt + 4 fg + r]
                *reinterpret_cast<uint2*>(my + fi * 512 + (((2 * nt + (fg >> 1)) ^ fi) << 4) + (fg & 1) * 8) = make_uint2(pack_op16(a[0], a[1]), pack_op16(a[2], a[3]));
            }
            __builtin_amdgcn_wave_barrier();
            bf16_t* dst = src_bf + ((int64_t)p * 4096 + tok0 + 16 * wv) * DEC_C;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int row = 2 * j + (lane >> 5), chunk = lane & 31;
                const u32x4 v = *reinterpret_cast<const u32x4*>(my + row * 512 + ((chunk ^ (row & 15)) << 4));
                __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(dst + row * DEC_C + chunk * 8));       // streamed: the 2 MB per prompt are read back by the next kernel long after they left the caches
            }
            __builtin_amdgcn_wave_barrier();
        }
        return;
    }
    // phase 2: thread = (4 consecutive channels, one of 4 token rows): one wave writes one whole 512-B token row per store
    const int c0 = (tid & 63) * 4, tr = tid >> 6;
    float w3[4][16];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int k4 = 0; k4 < 4; ++k4) {
            const float4 t = *reinterpret_cast<const float4*>(w.w3 + (c0 + c) * 16 + 4 * k4);
            w3[c][4 * k4] = t.x; w3[c][4 * k4 + 1] = t.y; w3[c][4 * k4 + 2] = t.z; w3[c][4 * k4 + 3] = t.w;
        }
    const float4 b3 = *reinterpret_cast<const float4*>(w.b3 + c0);
    // prompts of one block may straddle two crops (slots) only if ME_NP does not divide the prompts per crop: handled per prompt
    for (int t0 = tr; t0 < 64; t0 += 16) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int t = t0 + 4 * u, tok = tok0 + t;
            int cur_slot = -1;
            float4 e = make_float4(0.f, 0.f, 0.f, 0.f);
            float4 pp = make_float4(0.f, 0.f, 0.f, 0.f);
            if (srcpos_bf) pp = *reinterpret_cast<const float4*>(pos + (int64_t)tok * DEC_C + c0);
            for (int pi = 0; pi < np; ++pi) {
                const int p = p0 + pi;
                const int slot = (p + em.off) / em.div;
                if (slot != cur_slot) {
                    e = *reinterpret_cast<const float4*>(image_embed_base + (int64_t)slot * em.stride + (int64_t)tok * DEC_C + c0);
                    cur_slot = slot;
                }
                float a[4] = {b3.x + e.x, b3.y + e.y, b3.z + e.z, b3.w + e.w};
#pragma unroll
                for (int k4 = 0; k4 < 4; ++k4) {
                    const float4 h = *reinterpret_cast<const float4*>(&h2s[pi][t][4 * k4]);
#pragma unroll
                    for (int c = 0; c < 4; ++c) a[c] += w3[c][4 * k4] * h.x + w3[c][4 * k4 + 1] * h.y + w3[c][4 * k4 + 2] * h.z + w3[c][4 * k4 + 3] * h.w;
                }
                const int64_t off = ((int64_t)p * 4096 + tok) * DEC_C + c0;
                if (src_f) *reinterpret_cast<float4*>(src_f + off) = make_float4(a[0], a[1], a[2], a[3]);
                *reinterpret_cast<uint2*>(src_bf + off) = make_uint2(pack_op16(a[0], a[1]), pack_op16(a[2], a[3]));
                if (srcpos_bf) *reinterpret_cast<uint2*>(srcpos_bf + off) = make_uint2(pack_op16(a[0] + pp.x, a[1] + pp.y), pack_op16(a[2] + pp.z, a[3] + pp.w));
            }
        }
    }
}

// phase 1 of mask_embed_src_kernel alone: the hidden vectors go to HBM as bf16 (the operand precision of the 1 x 1 conv's MFMA)
__global__ __launch_bounds__(256) void mask_hidden_kernel(const float* __restrict__ mask_in, int P, MaskEmbedWeights w, bf16_t* __restrict__ h2out,
                                                         float clamp_abs, int raw4_q0) {
    const int tid = threadIdx.x;
    const int p = blockIdx.x >> 6, tok0 = (blockIdx.x & 63) * 64;
    const int tl = tid >> 2, q = tid & 3;
    const int py = q >> 1, px = q & 1;
    int ty, tx;
    perm_coords(tok0 + tl, 2, &ty, &tx);
    const int64_t plane = raw4_q0 >= 0 ? (int64_t)(raw4_q0 + p) + (raw4_q0 + p) / 3 + 1 : (int64_t)p;
    const float* mp = mask_in + plane * 65536 + (int64_t)(ty * 4 + py * 2) * 256 + tx * 4 + px * 2;
    const float2 r0 = *reinterpret_cast<const float2*>(mp), r1 = *reinterpret_cast<const float2*>(mp + 256);
    const float in[4] = {fminf(fmaxf(r0.x, -clamp_abs), clamp_abs), fminf(fmaxf(r0.y, -clamp_abs), clamp_abs),
                         fminf(fmaxf(r1.x, -clamp_abs), clamp_abs), fminf(fmaxf(r1.y, -clamp_abs), clamp_abs)};
    float v[4], mu = 0.f;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        float a = w.b1[c];
#pragma unroll
        for (int k = 0; k < 4; ++k) a += w.w1[c * 4 + k] * in[k];
        v[c] = a;
        mu += a;
    }
    mu *= 0.25f;
    float var = 0.f;
#pragma unroll
    for (int c = 0; c < 4; ++c) var += (v[c] - mu) * (v[c] - mu);
    float rstd = __builtin_amdgcn_rsqf(var * 0.25f + 1e-6f);
    float h1[4][4], mine[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) mine[c] = gelu_erf((v[c] - mu) * rstd * w.g1[c] + w.be1[c]);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        h1[0][c] = quad_bcast<0>(mine[c]); h1[1][c] = quad_bcast<1>(mine[c]); h1[2][c] = quad_bcast<2>(mine[c]); h1[3][c] = quad_bcast<3>(mine[c]);
    }
    float h2[4];
    mu = 0.f;
#pragma unroll
    for (int cc = 0; cc < 4; ++cc) {
        const int c = 4 * q + cc;
        float a = w.b2[c];
#pragma unroll
        for (int ci = 0; ci < 4; ++ci)
#pragma unroll
            for (int k = 0; k < 4; ++k) a += w.w2[(c * 4 + ci) * 4 + k] * h1[k][ci];
        h2[cc] = a;
        mu += a;
    }
    mu = quad_sum(mu);
    mu *= (1.0f / 16.0f);
    var = 0.f;
#pragma unroll
    for (int cc = 0; cc < 4; ++cc) var += (h2[cc] - mu) * (h2[cc] - mu);
    var = quad_sum(var);
    rstd = __builtin_amdgcn_rsqf(var * (1.0f / 16.0f) + 1e-6f);
    const float o0 = gelu_erf((h2[0] - mu) * rstd * w.g2[4 * q + 0] + w.be2[4 * q + 0]);
    const float o1 = gelu_erf((h2[1] - mu) * rstd * w.g2[4 * q + 1] + w.be2[4 * q + 1]);
    const float o2 = gelu_erf((h2[2] - mu) * rstd * w.g2[4 * q + 2] + w.be2[4 * q + 2]);
    const float o3 = gelu_erf((h2[3] - mu) * rstd * w.g2[4 * q + 3] + w.be2[4 * q + 3]);
    *reinterpret_cast<uint2*>(h2out + ((int64_t)p * 4096 + tok0 + tl) * 16 + 4 * q) = make_uint2(pack_op16(o0, o1), pack_op16(o2, o3));
}
const char* launch_mask_hidden(const float* mask_in, int P, MaskEmbedWeights w, bf16_t* h2, float clamp_abs, hipStream_t s, int raw4_q0) {
    if (P <= 0) return nullptr;
    if (!(clamp_abs > 0.f)) clamp_abs = 3.0e38f;
    hipLaunchKernelGGL(mask_hidden_kernel, dim3(P * 64), dim3(256), 0, s, mask_in, P, w, h2, clamp_abs, raw4_q0);
    return nullptr;
}

__global__ __launch_bounds__(256) void embb_tiles_kernel(const float* __restrict__ emb, const float* __restrict__ b3, float* __restrict__ out) {
    const int idx = blockIdx.x * 256 + threadIdx.x;        // one float4 of the output: (row tile, channel tile, lane)
    const int lane = idx & 63, ct = (idx >> 6) & 15, rt = idx >> 10;
    const int row = 16 * rt + (lane & 15), ch = 16 * ct + 4 * (lane >> 4);
    const float4 e = *reinterpret_cast<const float4*>(emb + (int64_t)row * 256 + ch), b = *reinterpret_cast<const float4*>(b3 + ch);
    *reinterpret_cast<float4*>(out + (int64_t)idx * 4) = make_float4(e.x + b.x, e.y + b.y, e.z + b.z, e.w + b.w);
}
const char* launch_embb_tiles(const float* emb, const float* b3, float* out, hipStream_t s) {
    hipLaunchKernelGGL(embb_tiles_kernel, dim3(4096 * 256 / 4 / 256), dim3(256), 0, s, emb, b3, out);
    return nullptr;
}

const char* launch_mask_embed_src(const float* mask_in, int P, const float* image_embed, XMap em, const float* pos, MaskEmbedWeights w,
                                  float* src_f, bf16_t* src_bf, bf16_t* srcpos_bf, float clamp_abs, hipStream_t s, int raw4_q0) {
    if (P <= 0) return nullptr;
    if (em.div <= 0) return "mask_embed_src: XMap.div must be positive";
    if (!(clamp_abs > 0.f)) clamp_abs = 3.0e38f;
    const dim3 grid(((P + ME_NP - 1) / ME_NP) * 64);
    if (!src_f && !srcpos_bf && src_bf) hipLaunchKernelGGL(mask_embed_src_kernel<true>, grid, dim3(256), 0, s, mask_in, P, image_embed, em, pos, w, src_f, src_bf, srcpos_bf, clamp_abs, raw4_q0);
    else hipLaunchKernelGGL(mask_embed_src_kernel<false>, grid, dim3(256), 0, s, mask_in, P, image_embed, em, pos, w, src_f, src_bf, srcpos_bf, clamp_abs, raw4_q0);
    return nullptr;
}

// ------------------------------------------------------------------------------------------------
// attention, few keys (nk <= 8): one thread per (b, q row, head)
template <int HDIM>
__global__ __launch_bounds__(256) void dec_attn_fewkeys_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                               const float* __restrict__ v, bf16_t* __restrict__ out, int B, int nq,
                                                               int nk, int heads, int64_t q_bs, int64_t k_bs, int64_t v_bs,
                                                               int64_t o_bs) {
    const int C = heads * HDIM;
    const int64_t total = (int64_t)B * nq * heads;
    const float scale = rsqrtf((float)HDIM);
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        const int h = (int)(idx % heads);
        const int64_t r = idx / heads;
        const int qi = (int)(r % nq);
        const int b = (int)(r / nq);
        const float* qp = q + b * q_bs + (int64_t)qi * C + h * HDIM;
        float qv[HDIM];
#pragma unroll
        for (int d = 0; d < HDIM; d += 4) {
            const float4 t = *reinterpret_cast<const float4*>(qp + d);
            qv[d] = t.x; qv[d + 1] = t.y; qv[d + 2] = t.z; qv[d + 3] = t.w;
        }
        float sc[8], mx = -3.0e38f;
        for (int j = 0; j < nk; ++j) {
            const float* kp = k + b * k_bs + (int64_t)j * C + h * HDIM;
            float a = 0.f;
#pragma unroll
            for (int d = 0; d < HDIM; ++d) a += qv[d] * kp[d];
            sc[j] = a * scale;
            mx = fmaxf(mx, sc[j]);
        }
        float sum = 0.f;
        for (int j = 0; j < nk; ++j) { sc[j] = expf(sc[j] - mx); sum += sc[j]; }
        const float inv = 1.0f / sum;
        float o[HDIM];
#pragma unroll
        for (int d = 0; d < HDIM; ++d) o[d] = 0.f;
        for (int j = 0; j < nk; ++j) {
            const float* vp = v + b * v_bs + (int64_t)j * C + h * HDIM;
            const float pj = sc[j] * inv;
#pragma unroll
            for (int d = 0; d < HDIM; ++d) o[d] += pj * vp[d];
        }
        bf16_t* op = out + b * o_bs + (int64_t)qi * C + h * HDIM;
#pragma unroll
        for (int d = 0; d < HDIM; d += 4) *reinterpret_cast<uint2*>(op + d) = make_uint2(pack_op16(o[d], o[d + 1]), pack_op16(o[d + 2], o[d + 3]));
    }
}

// attention, few queries (nq <= 8) over many keys: one block per (b, head); each thread streams keys
// with an online softmax for all nq queries, then the 256 partial states are merged through LDS.
template <int HDIM>
__global__ __launch_bounds__(256) void dec_attn_fewq_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                            const float* __restrict__ v, bf16_t* __restrict__ out, int nq, int nk,
                                                            int heads, int64_t q_bs, int64_t k_bs, int64_t v_bs, int64_t o_bs) {
    constexpr int NQ = 8;
    __shared__ float red[4][NQ][HDIM + 2];
    __shared__ float qs[NQ][HDIM];
    const int b = blockIdx.x / heads, h = blockIdx.x - b * heads;
    const int C = heads * HDIM;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float scale = rsqrtf((float)HDIM);
    for (int i = tid; i < NQ * HDIM; i += 256) {
        const int t = i / HDIM, d = i - t * HDIM;
        qs[t][d] = t < nq ? q[b * q_bs + (int64_t)t * C + h * HDIM + d] * scale : 0.f;
    }
    __syncthreads();
    float m[NQ], l[NQ], acc[NQ][HDIM];
#pragma unroll
    for (int t = 0; t < NQ; ++t) {
        m[t] = -3.0e38f; l[t] = 0.f;
#pragma unroll
        for (int d = 0; d < HDIM; ++d) acc[t][d] = 0.f;
    }
    for (int j = tid; j < nk; j += 256) {
        const float* kp = k + b * k_bs + (int64_t)j * C + h * HDIM;
        const float* vp = v + b * v_bs + (int64_t)j * C + h * HDIM;
        float kv[HDIM], vv[HDIM];
#pragma unroll
        for (int d = 0; d < HDIM; d += 4) {
            const float4 a = *reinterpret_cast<const float4*>(kp + d);
            const float4 c4 = *reinterpret_cast<const float4*>(vp + d);
            kv[d] = a.x; kv[d + 1] = a.y; kv[d + 2] = a.z; kv[d + 3] = a.w;
            vv[d] = c4.x; vv[d + 1] = c4.y; vv[d + 2] = c4.z; vv[d + 3] = c4.w;
        }
#pragma unroll
        for (int t = 0; t < NQ; ++t) {
            float s = 0.f;
#pragma unroll
            for (int d = 0; d < HDIM; ++d) s += qs[t][d] * kv[d];
            const float mn = fmaxf(m[t], s);
            const float alpha = expf(m[t] - mn), e = expf(s - mn);
            m[t] = mn;
            l[t] = l[t] * alpha + e;
#pragma unroll
            for (int d = 0; d < HDIM; ++d) acc[t][d] = acc[t][d] * alpha + e * vv[d];
        }
    }
    // merge across the wave, then across the 4 waves
#pragma unroll
    for (int t = 0; t < NQ; ++t) {
        const float mw = wave_max(m[t]);
        const float f = expf(m[t] - mw);
        l[t] = wave_sum(l[t] * f);
#pragma unroll
        for (int d = 0; d < HDIM; ++d) acc[t][d] = wave_sum(acc[t][d] * f);
        if (lane == 0) {
            red[wave][t][HDIM] = mw;
            red[wave][t][HDIM + 1] = l[t];
#pragma unroll
            for (int d = 0; d < HDIM; ++d) red[wave][t][d] = acc[t][d];
        }
    }
    __syncthreads();
    for (int i = tid; i < nq * HDIM; i += 256) {
        const int t = i / HDIM, d = i - t * HDIM;
        float mm = -3.0e38f;
        for (int w = 0; w < 4; ++w) mm = fmaxf(mm, red[w][t][HDIM]);
        float L = 0.f, A = 0.f;
        for (int w = 0; w < 4; ++w) {
            const float f = expf(red[w][t][HDIM] - mm);
            L += red[w][t][HDIM + 1] * f;
            A += red[w][t][d] * f;
        }
        out[b * o_bs + (int64_t)t * C + h * HDIM + d] = f2op(A / L);
    }
}

const char* launch_dec_attention(const float* q, const float* k, const float* v, bf16_t* out, int B, int nq, int nk, int heads,
                                 int hd, int64_t q_bs, int64_t k_bs, int64_t v_bs, int64_t o_bs, hipStream_t s) {
    if (B <= 0) return nullptr;
    if (hd != 16 && hd != 32) return "dec_attention: head_dim must be 16 or 32";
    if (nk <= 8) {
        const int64_t total = (int64_t)B * nq * heads;
        int blocks = (int)((total + 255) / 256);
        if (blocks > 16384) blocks = 16384;
        if (hd == 16) hipLaunchKernelGGL(dec_attn_fewkeys_kernel<16>, dim3(blocks), dim3(256), 0, s, q, k, v, out, B, nq, nk, heads, q_bs, k_bs, v_bs, o_bs);
        else hipLaunchKernelGGL(dec_attn_fewkeys_kernel<32>, dim3(blocks), dim3(256), 0, s, q, k, v, out, B, nq, nk, heads, q_bs, k_bs, v_bs, o_bs);
        return nullptr;
    }
    if (nq > 8) return "dec_attention: need nq <= 8 or nk <= 8";
    if (hd != 16) return "dec_attention: few-query path is instantiated for head_dim 16";
    hipLaunchKernelGGL(dec_attn_fewq_kernel<16>, dim3(B * heads), dim3(256), 0, s, q, k, v, out, nq, nk, heads, q_bs, k_bs, v_bs, o_bs);
    return nullptr;
}

// ------------------------------------------------------------------------------------------------
// hypernetwork product: masks[p][k][y][x] = sum_c hyper[p][k][c] * up[p][perm(y,x)][c]
__global__ __launch_bounds__(256) void mask_dot_kernel(const bf16_t* __restrict__ up, const float* __restrict__ hyper, int P,
                                                       float* __restrict__ masks4) {
    __shared__ float hs[4][32];
    const int p = blockIdx.y;
    if (threadIdx.x < 128) hs[threadIdx.x >> 5][threadIdx.x & 31] = hyper[(int64_t)p * 128 + threadIdx.x];
    __syncthreads();
    const int pix = blockIdx.x * 256 + threadIdx.x;  // row-major output pixel
    const int y = pix >> 8, x = pix & 255;
    const int tok = perm_index256(y, x);
    const uint4* src = reinterpret_cast<const uint4*>(up + ((int64_t)p * 65536 + tok) * 32);
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const uint4 u = src[q];
        const uint32_t wds[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float lo = op16_lo(wds[j]), hi = op16_hi(wds[j]);
            const int c = q * 8 + j * 2;
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) acc[kk] += hs[kk][c] * lo + hs[kk][c + 1] * hi;
        }
    }
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) masks4[((int64_t)p * 4 + kk) * 65536 + pix] = acc[kk];
}

const char* launch_mask_dot(const bf16_t* up, const float* hyper, int P, float* masks4, hipStream_t s) {
    if (P <= 0) return nullptr;
    hipLaunchKernelGGL(mask_dot_kernel, dim3(256, P), dim3(256), 0, s, up, hyper, P, masks4);
    return nullptr;
}

// dynamic multimask: counts of mask0 > +delta, > -delta
__global__ __launch_bounds__(256) void mask_select_kernel(const float* __restrict__ masks4, const float* __restrict__ iou4, int multimask,
                                                          float* __restrict__ out_masks, float* __restrict__ out_iou,
                                                          const int* __restrict__ counts, float thresh) {
    const int p = blockIdx.y;
    if (multimask) {
        for (int i = blockIdx.x * 256 + threadIdx.x; i < 3 * 65536 / 4; i += gridDim.x * 256)
            reinterpret_cast<float4*>(out_masks + (int64_t)p * 3 * 65536)[i] =
                reinterpret_cast<const float4*>(masks4 + ((int64_t)p * 4 + 1) * 65536)[i];
        if (blockIdx.x == 0 && threadIdx.x < 3) out_iou[p * 3 + threadIdx.x] = iou4[p * 4 + 1 + threadIdx.x];
        return;
    }
    const float ai = (float)counts[2 * p], au = (float)counts[2 * p + 1];
    const float stab = au > 0.f ? ai / au : 1.0f;
    int sel = 0;
    if (!(stab >= thresh)) {
        sel = 1;
        float best = iou4[p * 4 + 1];
        if (iou4[p * 4 + 2] > best) { best = iou4[p * 4 + 2]; sel = 2; }
        if (iou4[p * 4 + 3] > best) { best = iou4[p * 4 + 3]; sel = 3; }
    }
    for (int i = blockIdx.x * 256 + threadIdx.x; i < 65536 / 4; i += gridDim.x * 256)
        reinterpret_cast<float4*>(out_masks + (int64_t)p * 65536)[i] =
            reinterpret_cast<const float4*>(masks4 + ((int64_t)p * 4 + sel) * 65536)[i];
    if (blockIdx.x == 0 && threadIdx.x == 0) out_iou[p] = iou4[p * 4 + sel];
}

// multimask_output=False (the m2m pass): stability of the single-mask output decides between it and the best of the three
// multimask outputs (_dynamic_multimask_via_stability, delta 0.05 / threshold 0.98).  One block per prompt: count, decide, copy
// the chosen 256x256 plane - one launch and one pass over mask 0 (L2-hot for the copy) instead of memset + count + select.
__global__ __launch_bounds__(1024) void mask_select_dynamic_kernel(const float* __restrict__ masks4, const float* __restrict__ iou4,
                                                                   float* __restrict__ out_masks, float* __restrict__ out_iou, float delta, float thresh) {
    __shared__ int red[2][16];
    __shared__ int sel_s;
    const int p = blockIdx.x, tid = threadIdx.x;
    const float4* m0 = reinterpret_cast<const float4*>(masks4 + (int64_t)p * 4 * 65536);
    int a = 0, u = 0;
#pragma unroll 4
    for (int i = tid; i < 16384; i += 1024) {
        const float4 v = m0[i];
        a += (v.x > delta) + (v.y > delta) + (v.z > delta) + (v.w > delta);
        u += (v.x > -delta) + (v.y > -delta) + (v.z > -delta) + (v.w > -delta);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { a += __shfl_xor(a, o, 64); u += __shfl_xor(u, o, 64); }
    if ((tid & 63) == 0) { red[0][tid >> 6] = a; red[1][tid >> 6] = u; }
    __syncthreads();
    if (tid == 0) {
        int ai = 0, au = 0;
        for (int w = 0; w < 16; ++w) { ai += red[0][w]; au += red[1][w]; }
        const float stab = au > 0 ? (float)ai / (float)au : 1.0f;
        int sel = 0;
        if (!(stab >= thresh)) {
            sel = 1;
            float best = iou4[p * 4 + 1];
            if (iou4[p * 4 + 2] > best) { best = iou4[p * 4 + 2]; sel = 2; }
            if (iou4[p * 4 + 3] > best) { best = iou4[p * 4 + 3]; sel = 3; }
        }
        sel_s = sel;
        out_iou[p] = iou4[p * 4 + sel];
    }
    __syncthreads();
    const float4* src = m0 + (int64_t)sel_s * 16384;
    float4* dst = reinterpret_cast<float4*>(out_masks + (int64_t)p * 65536);
#pragma unroll 4
    for (int i = tid; i < 16384; i += 1024) dst[i] = src[i];
}

// The selection alone, for callers that read the chosen planes in place (the AMG driver: K8 takes a plane index, the m2m pass reads the
// first pass's planes through an index map): out_iou as mask_select*, out_sel[p] = chosen plane of prompt p (single-mask mode).
__global__ __launch_bounds__(1024) void mask_pick_kernel(const float* __restrict__ masks4, const float* __restrict__ iou4, int multimask,
                                                         float* __restrict__ out_iou, int* __restrict__ out_sel, float delta, float thresh,
                                                         const uint8_t* __restrict__ live) {
    __shared__ int red[2][16];
    const int p = blockIdx.x, tid = threadIdx.x;
    if (multimask) {
        if (tid < 3) out_iou[p * 3 + tid] = iou4[p * 4 + 1 + tid];
        return;
    }
    if (live && !live[p]) {      // pruned prompt: its planes were not computed; either choice fails the caller's IoU filter
        if (tid == 0) { out_iou[p] = iou4[p * 4]; if (out_sel) out_sel[p] = 0; }
        return;
    }
    const float4* m0 = reinterpret_cast<const float4*>(masks4 + (int64_t)p * 4 * 65536);
    int a = 0, u = 0;
#pragma unroll 4
    for (int i = tid; i < 16384; i += 1024) {
        const float4 v = m0[i];
        a += (v.x > delta) + (v.y > delta) + (v.z > delta) + (v.w > delta);
        u += (v.x > -delta) + (v.y > -delta) + (v.z > -delta) + (v.w > -delta);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { a += __shfl_xor(a, o, 64); u += __shfl_xor(u, o, 64); }
    if ((tid & 63) == 0) { red[0][tid >> 6] = a; red[1][tid >> 6] = u; }
    __syncthreads();
    if (tid == 0) {
        int ai = 0, au = 0;
        for (int w = 0; w < 16; ++w) { ai += red[0][w]; au += red[1][w]; }
        const float stab = au > 0 ? (float)ai / (float)au : 1.0f;
        int sel = 0;
        if (!(stab >= thresh)) {
            sel = 1;
            float best = iou4[p * 4 + 1];
            if (iou4[p * 4 + 2] > best) { best = iou4[p * 4 + 2]; sel = 2; }
            if (iou4[p * 4 + 3] > best) { best = iou4[p * 4 + 3]; sel = 3; }
        }
        out_iou[p] = iou4[p * 4 + sel];
        if (out_sel) out_sel[p] = sel;
    }
}
const char* launch_mask_pick(const float* masks4, const float* iou4, int P, int multimask, float* out_iou, int* out_sel, hipStream_t s, const uint8_t* live) {
    if (P <= 0) return nullptr;
    hipLaunchKernelGGL(mask_pick_kernel, dim3(P), dim3(multimask ? 64 : 1024), 0, s, masks4, iou4, multimask, out_iou, out_sel, 0.05f, 0.98f, live);
    return nullptr;
}
__global__ __launch_bounds__(256) void iou_live_flags_kernel(const float* __restrict__ iou4, int P, float thr, uint8_t* __restrict__ live,
                                                             unsigned long long* __restrict__ counters) {
    const int p = blockIdx.x * 256 + threadIdx.x;
    bool l = true;
    if (p < P) {
        const float4 v = *reinterpret_cast<const float4*>(iou4 + 4 * p);
        l = v.x > thr || v.y > thr || v.z > thr || v.w > thr;
        live[p] = l ? 1 : 0;
    }
    if (counters) {      // statistics: [0] += pruned prompts, [1] += prompts seen
        const unsigned long long dead = __ballot(p < P && !l), seen = __ballot(p < P);
        if ((threadIdx.x & 63) == 0) { atomicAdd(&counters[0], (unsigned long long)__popcll(dead)); atomicAdd(&counters[1], (unsigned long long)__popcll(seen)); }
    }
}
const char* launch_iou_live_flags(const float* iou4, int P, float thr, uint8_t* live, unsigned long long* counters, hipStream_t s) {
    if (P <= 0) return nullptr;
    hipLaunchKernelGGL(iou_live_flags_kernel, dim3((P + 255) / 256), dim3(256), 0, s, iou4, P, thr, live, counters);
    return nullptr;
}

const char* launch_mask_select(const float* masks4, const float* iou4, int P, int multimask, float* out_masks, float* out_iou,
                               int* counts_ws, hipStream_t s) {
    if (P <= 0) return nullptr;
    if (!multimask) hipLaunchKernelGGL(mask_select_dynamic_kernel, dim3(P), dim3(1024), 0, s, masks4, iou4, out_masks, out_iou, 0.05f, 0.98f);
    else hipLaunchKernelGGL(mask_select_kernel, dim3(16, P), dim3(256), 0, s, masks4, iou4, multimask, out_masks, out_iou, counts_ws, 0.98f);
    return nullptr;
}

// ------------------------------------------------------------------------------------------------ K8
// For each selected low-res mask: bilinear-upsample (align_corners=False) to the crop size, compare with
// thr / thr+-offset, count, track the bounding box, and bit-pack the >thr mask into full-image rows.
// One block per (mask, band of 4 full-image rows); a wave covers 64 consecutive x of one row, so
// __ballot gives two packed 32-bit words directly.  fp32 full-res logits never touch HBM.
#define MP_ROWS 64   // output rows per block (16 per wave): one set of atomics per block, not per row
// PRE: the horizontal taps (x0, weight) of every 64-pixel chunk depend on x alone; for W <= 1024 they are computed once per block
// into registers instead of once per pixel, row and mask (they were half of the VALU work of the pixel loop)
template <bool PRE>
__global__ __launch_bounds__(256) void mask_post_kernel(const float* __restrict__ lowres, const int* __restrict__ idx, int crop_x0,
                                                        int crop_y0, int crop_w, int crop_h, int H, int W, float thr, float offset,
                                                        uint32_t* __restrict__ bits, MaskStats* __restrict__ stats, const uint8_t* __restrict__ pass) {
    if (pass && !pass[blockIdx.y]) return;          // (device-side AMG: candidates the IoU filter dropped keep their initialised statistics)
    // per wave: the vertically blended source row (256 columns) lives in LDS, every output pixel is then two LDS reads
    __shared__ float vrow[4][256];
    __shared__ int red[4][8];
    const int mi = blockIdx.y;
    const float* src = lowres + (int64_t)(idx ? idx[mi] : mi) * 65536;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int W32 = (W + 31) >> 5;
    const float sy_scale = 256.0f / (float)crop_h, sx_scale = 256.0f / (float)crop_w;
    int area = 0, inter = 0, uni = 0, xmin = 1 << 30, xmax = -1, ymin = 1 << 30, ymax = -1;
    int x0a[16];
    float lxa[16];
    if (PRE) {
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            const int x = 64 * c + lane, cx = x - crop_x0;
            float sx = ((float)cx + 0.5f) * sx_scale - 0.5f;
            sx = fmaxf(sx, 0.f);
            const int x0i = min((int)sx, 255);
            x0a[c] = (x < W && cx >= 0 && cx < crop_w) ? x0i : -1;
            lxa[c] = sx - (float)x0i;
        }
    }
    // the two source rows of output row y; the pair of the NEXT row is fetched before the pixel loop of the current one (the wave walks
    // 16 rows one after the other and each needed a dependent global load -> LDS -> pixel loop chain)
    auto fetch = [&](int y, float4* a, float4* b, float* ly) -> bool {
        const int cy = y - crop_y0;
        if (y >= H || cy < 0 || cy >= crop_h) return false;       // wave-uniform
        float sy = ((float)cy + 0.5f) * sy_scale - 0.5f;
        sy = fmaxf(sy, 0.f);
        const int y0i = min((int)sy, 255), y1i = min(y0i + 1, 255);
        *ly = sy - (float)y0i;
        *a = *reinterpret_cast<const float4*>(src + y0i * 256 + lane * 4);
        *b = *reinterpret_cast<const float4*>(src + y1i * 256 + lane * 4);
        return true;
    };
    const int ybase = blockIdx.x * MP_ROWS + wave * (MP_ROWS / 4);
    float4 na = make_float4(0.f, 0.f, 0.f, 0.f), nb = na;
    float nly = 0.f;
    bool nin = fetch(ybase, &na, &nb, &nly);
    for (int r = 0; r < MP_ROWS / 4; ++r) {
        const int y = ybase + r;
        if (y >= H) break;  // wave-uniform
        uint32_t* brow = bits + ((int64_t)mi * H + y) * W32;
        const float4 a = na, b = nb;
        const float ly = nly;
        const bool in_crop = nin;
        nin = (r + 1 < MP_ROWS / 4) ? fetch(y + 1, &na, &nb, &nly) : false;
        if (!in_crop) {  // wave-uniform: rows outside the crop are all zero
            for (int wd = lane; wd < W32; wd += 64) brow[wd] = 0u;
            continue;
        }
        __builtin_amdgcn_wave_barrier();
        {
            float* vw = vrow[wave] + lane * 4;
            vw[0] = (1.0f - ly) * a.x + ly * b.x; vw[1] = (1.0f - ly) * a.y + ly * b.y;
            vw[2] = (1.0f - ly) * a.z + ly * b.z; vw[3] = (1.0f - ly) * a.w + ly * b.w;
        }
        __builtin_amdgcn_wave_barrier();
        const float* vr = vrow[wave];
        unsigned long long mybits = 0ull;   // lane k keeps the 64-pixel chunk k (k + 64j for very wide images: flushed below)
        int rowarea = 0;
#pragma unroll
        for (int cc = 0; cc < (PRE ? 16 : 1); ++cc)
        for (int xb = PRE ? 64 * cc : 0; xb < (PRE ? min(W, 64 * cc + 64) : W); xb += 64) {
            const int x = xb + lane;
            const int cx = x - crop_x0;
            bool on = false, hi = false, lo = false;
            if (PRE) {
                const int x0i = x0a[cc];
                if (x0i >= 0) {
                    const float lx = lxa[cc];
                    const float v = (1.0f - lx) * vr[x0i] + lx * vr[min(x0i + 1, 255)];
                    on = v > thr;
                    hi = v > thr + offset;
                    lo = v > thr - offset;
                }
            } else if (x < W && cx >= 0 && cx < crop_w) {
                float sx = ((float)cx + 0.5f) * sx_scale - 0.5f;
                sx = fmaxf(sx, 0.f);
                const int x0i = min((int)sx, 255);
                const int x1i = min(x0i + 1, 255);
                const float lx = sx - (float)x0i;
                const float v = (1.0f - lx) * vr[x0i] + lx * vr[x1i];
                on = v > thr;
                hi = v > thr + offset;
                lo = v > thr - offset;
            }
            const unsigned long long bm = __ballot(on);
            rowarea += __popcll(bm);
            inter += __popcll(__ballot(hi));
            uni += __popcll(__ballot(lo));
            if (bm) {
                xmin = min(xmin, xb + (int)__ffsll((long long)bm) - 1);
                xmax = max(xmax, xb + 63 - (int)__clzll((long long)bm));
            }
            const int chunk = xb >> 6;
            if (lane == (chunk & 63)) mybits = bm;
            if ((chunk & 63) == 63 || xb + 64 >= W) {   // flush: lanes 0..n-1 hold consecutive 8-byte pieces of the row
                const int c0 = chunk & ~63;
                const int wd = (c0 + lane) * 2;
                if (c0 + lane <= chunk) {
                    if (wd < W32) brow[wd] = (uint32_t)(mybits & 0xffffffffull);
                    if (wd + 1 < W32) brow[wd + 1] = (uint32_t)(mybits >> 32);
                }
            }
        }
        if (rowarea) { area += rowarea; ymin = min(ymin, y); ymax = max(ymax, y); }
    }
    if (lane == 0) {
        red[wave][0] = area; red[wave][1] = inter; red[wave][2] = uni; red[wave][3] = xmin;
        red[wave][4] = xmax; red[wave][5] = ymin; red[wave][6] = ymax;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        int a = 0, in_ = 0, u = 0, x0 = 1 << 30, x1 = -1, y0 = 1 << 30, y1 = -1;
        for (int w = 0; w < 4; ++w) {
            a += red[w][0]; in_ += red[w][1]; u += red[w][2];
            x0 = min(x0, red[w][3]); x1 = max(x1, red[w][4]); y0 = min(y0, red[w][5]); y1 = max(y1, red[w][6]);
        }
        MaskStats* st = stats + mi;
        if (a) {
            atomicAdd(&st->area, a);
            atomicMin(&st->x0, x0); atomicMax(&st->x1, x1);
            atomicMin(&st->y0, y0); atomicMax(&st->y1, y1);
        }
        if (in_) atomicAdd(&st->inter, in_);
        if (u) atomicAdd(&st->uni, u);
    }
}

// W <= 1024 (every SABER slice): the row loop with NO scalar work per 64-pixel chunk.  The previous form spent ~15 SALU instructions
// per chunk (three popcounts, ffs / clz for the box, the bit-deposit bookkeeping) beside ~14 VALU; here
//   * lane k keeps chunk k of the row's bits (v_writelane of the v_cmp result), so area / box come from ONE popcount / ffs / clz per
//     lane and row instead of per chunk, and the row is stored as is;
//   * the two stability counts are per-lane adds of the compare results, reduced across the wave once per 16 rows;
//   * pixels outside the crop read a -3e38 sentinel (taps precomputed per block), the chunk loop skips chunks the crop does not touch;
//   * vrow[256] repeats vrow[255], so the right tap is always x0 + 1 (one ds_read2).
// Same arithmetic per pixel as mask_post_kernel: bit-identical masks and counts.
__global__ __launch_bounds__(256) void mask_post_w1k_kernel(const float* __restrict__ lowres, const int* __restrict__ idx, int crop_x0,
                                                            int crop_y0, int crop_w, int crop_h, int H, int W, float thr, float offset,
                                                            uint32_t* __restrict__ bits, MaskStats* __restrict__ stats, const uint8_t* __restrict__ pass) {
    if (pass && !pass[blockIdx.y]) return;
    __shared__ float vrow[4][260];
    __shared__ int red[4][8];
    const int mi = blockIdx.y;
    const float* src = lowres + (int64_t)(idx ? idx[mi] : mi) * 65536;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int W32 = (W + 31) >> 5;
    const float sy_scale = 256.0f / (float)crop_h, sx_scale = 256.0f / (float)crop_w;
    const int c_lo = max(crop_x0, 0) >> 6, c_hi = min(crop_x0 + crop_w - 1, W - 1) >> 6;
    int x0a[16];
    float lxa[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) {
        const int x = 64 * c + lane, cx = x - crop_x0;
        float sx = ((float)cx + 0.5f) * sx_scale - 0.5f;
        sx = fmaxf(sx, 0.f);
        const int x0i = min((int)sx, 255);
        const bool valid = x < W && cx >= 0 && cx < crop_w;
        x0a[c] = valid ? x0i : 257;
        lxa[c] = valid ? sx - (float)x0i : 0.0f;
    }
    if (lane < 2) vrow[wave][257 + lane] = -3.0e38f;
    auto fetch = [&](int y, float4* a, float4* b, float* ly) -> bool {
        const int cy = y - crop_y0;
        if (y >= H || cy < 0 || cy >= crop_h) return false;       // wave-uniform
        float sy = ((float)cy + 0.5f) * sy_scale - 0.5f;
        sy = fmaxf(sy, 0.f);
        const int y0i = min((int)sy, 255), y1i = min(y0i + 1, 255);
        *ly = sy - (float)y0i;
        *a = *reinterpret_cast<const float4*>(src + y0i * 256 + lane * 4);
        *b = *reinterpret_cast<const float4*>(src + y1i * 256 + lane * 4);
        return true;
    };
    int area = 0, inter = 0, uni = 0, xmin = 1 << 30, xmax = -1, ymin = 1 << 30, ymax = -1;     // per lane; y extents wave-uniform
    const int ybase = blockIdx.x * MP_ROWS + wave * (MP_ROWS / 4);
    float4 na = make_float4(0.f, 0.f, 0.f, 0.f), nb = na;
    float nly = 0.f;
    bool nin = fetch(ybase, &na, &nb, &nly);
    for (int r = 0; r < MP_ROWS / 4; ++r) {
        const int y = ybase + r;
        if (y >= H) break;  // wave-uniform
        uint32_t* brow = bits + ((int64_t)mi * H + y) * W32;
        const float4 a = na, b = nb;
        const float ly = nly;
        const bool in_crop = nin;
        nin = (r + 1 < MP_ROWS / 4) ? fetch(y + 1, &na, &nb, &nly) : false;
        if (!in_crop) {  // wave-uniform: rows outside the crop are all zero
            for (int wd = lane; wd < W32; wd += 64) brow[wd] = 0u;
            continue;
        }
        __builtin_amdgcn_wave_barrier();
        {
            float* vw = vrow[wave] + lane * 4;
            vw[0] = (1.0f - ly) * a.x + ly * b.x; vw[1] = (1.0f - ly) * a.y + ly * b.y;
            vw[2] = (1.0f - ly) * a.z + ly * b.z; vw[3] = (1.0f - ly) * a.w + ly * b.w;
            if (lane == 63) vw[4] = vw[3];
        }
        __builtin_amdgcn_wave_barrier();
        const float* vr = vrow[wave];
        uint32_t blo = 0u, bhi = 0u;        // lane k: bits of pixels 64k .. 64k+63 of this row
        // all taps of the row first (16 ds_read2 in flight, one wait; chunks outside the crop read the sentinel), then the chunks
        float ta[16], tb[16];
#pragma unroll
        for (int cc = 0; cc < 16; ++cc) { ta[cc] = vr[x0a[cc]]; tb[cc] = vr[x0a[cc] + 1]; }
        // one 64-pixel chunk; the lane select of v_writelane_b32 must be an inline constant (one SGPR per VALU instruction)
#define MP_CHUNK(cc)                                                                                              \
        if ((cc) >= c_lo && (cc) <= c_hi) {                                                                       \
            const float lx = lxa[cc];                                                                             \
            const float v = (1.0f - lx) * ta[cc] + lx * tb[cc];                                                   \
            const unsigned long long bm = __ballot(v > thr);                                                      \
            asm("s_nop 1\n\tv_writelane_b32 %0, %1, " #cc : "+v"(blo) : "s"((uint32_t)bm));    /* VALU-written SGPR read by a VALU op: 2 wait states on gfx940+, and the hazard recogniser does not look inside asm */ \
            asm("v_writelane_b32 %0, %1, " #cc : "+v"(bhi) : "s"((uint32_t)(bm >> 32)));                          \
            inter += (v > thr + offset) ? 1 : 0;                                                                  \
            uni += (v > thr - offset) ? 1 : 0;                                                                    \
        }
        MP_CHUNK(0) MP_CHUNK(1) MP_CHUNK(2) MP_CHUNK(3) MP_CHUNK(4) MP_CHUNK(5) MP_CHUNK(6) MP_CHUNK(7)
        MP_CHUNK(8) MP_CHUNK(9) MP_CHUNK(10) MP_CHUNK(11) MP_CHUNK(12) MP_CHUNK(13) MP_CHUNK(14) MP_CHUNK(15)
#undef MP_CHUNK
        const int pc = __popc(blo) + __popc(bhi);
        area += pc;
        if (pc) {
            xmin = min(xmin, lane * 64 + (blo ? __ffs((int)blo) - 1 : 31 + __ffs((int)bhi)));
            xmax = max(xmax, lane * 64 + (bhi ? 63 - __clz((int)bhi) : 31 - __clz((int)blo)));
        }
        if (__ballot(pc != 0)) { ymin = min(ymin, y); ymax = max(ymax, y); }
        const int wd = lane * 2;
        if (wd < W32) brow[wd] = blo;
        if (wd + 1 < W32) brow[wd + 1] = bhi;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        area += __shfl_xor(area, o, 64); inter += __shfl_xor(inter, o, 64); uni += __shfl_xor(uni, o, 64);
        xmin = min(xmin, __shfl_xor(xmin, o, 64)); xmax = max(xmax, __shfl_xor(xmax, o, 64));
    }
    if (lane == 0) {
        red[wave][0] = area; red[wave][1] = inter; red[wave][2] = uni; red[wave][3] = xmin;
        red[wave][4] = xmax; red[wave][5] = ymin; red[wave][6] = ymax;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        int a = 0, in_ = 0, u = 0, x0 = 1 << 30, x1 = -1, y0 = 1 << 30, y1 = -1;
        for (int w = 0; w < 4; ++w) {
            a += red[w][0]; in_ += red[w][1]; u += red[w][2];
            x0 = min(x0, red[w][3]); x1 = max(x1, red[w][4]); y0 = min(y0, red[w][5]); y1 = max(y1, red[w][6]);
        }
        MaskStats* st = stats + mi;
        if (a) {
            atomicAdd(&st->area, a);
            atomicMin(&st->x0, x0); atomicMax(&st->x1, x1);
            atomicMin(&st->y0, y0); atomicMax(&st->y1, y1);
        }
        if (in_) atomicAdd(&st->inter, in_);
        if (u) atomicAdd(&st->uni, u);
    }
}

__global__ void mask_stats_init_kernel(MaskStats* stats, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) { MaskStats s; s.area = 0; s.inter = 0; s.uni = 0; s.x0 = 1 << 30; s.y0 = 1 << 30; s.x1 = -1; s.y1 = -1; s.pad = 0; stats[i] = s; }
}

const char* launch_mask_post(const float* lowres, const int* idx, int n, int crop_x0, int crop_y0, int crop_w, int crop_h, int H,
                             int W, float thr, float offset, uint32_t* bits, MaskStats* stats, hipStream_t s, const uint8_t* pass) {
    if (n <= 0) return nullptr;
    if (crop_w <= 0 || crop_h <= 0) return "mask_post: empty crop";
    hipLaunchKernelGGL(mask_stats_init_kernel, dim3((n + 255) / 256), dim3(256), 0, s, stats, n);
    const dim3 grid((H + MP_ROWS - 1) / MP_ROWS, n);
    if (W <= 1024 && !(g_saber_debug_flags & 8)) hipLaunchKernelGGL(mask_post_w1k_kernel, grid, dim3(256), 0, s, lowres, idx, crop_x0, crop_y0, crop_w, crop_h, H, W, thr, offset, bits, stats, pass);
    else if (W <= 1024) hipLaunchKernelGGL(mask_post_kernel<true>, grid, dim3(256), 0, s, lowres, idx, crop_x0, crop_y0, crop_w, crop_h, H, W, thr, offset, bits, stats, pass);
    else hipLaunchKernelGGL(mask_post_kernel<false>, grid, dim3(256), 0, s, lowres, idx, crop_x0, crop_y0, crop_w, crop_h, H, W, thr, offset, bits, stats, pass);
    return nullptr;
}

// ------------------------------------------------------------------------------------------------
// label plane of reference slice_by_slice (saber/segmenters/propagation.py:185-186): masks are painted
// in list order with value (position+1); a later mask overwrites an earlier one.
__global__ __launch_bounds__(256) void label_plane_kernel(const uint32_t* __restrict__ bits, const int* __restrict__ order, int n, int H,
                                                          int W, uint16_t* __restrict__ plane) {
    const int W32 = (W + 31) >> 5;
    const int64_t words = (int64_t)H * W32;
    for (int64_t wi = (int64_t)blockIdx.x * 256 + threadIdx.x; wi < words; wi += (int64_t)gridDim.x * 256) {
        const int y = (int)(wi / W32), xw = (int)(wi - (int64_t)y * W32);
        uint16_t lab[32];
#pragma unroll
        for (int b = 0; b < 32; ++b) lab[b] = 0;
        for (int i = 0; i < n; ++i) {
            const int mi = order ? order[i] : i;
            const uint32_t wd = bits[(int64_t)mi * words + wi];
            if (wd) {
#pragma unroll
                for (int b = 0; b < 32; ++b) if ((wd >> b) & 1u) lab[b] = (uint16_t)(i + 1);
            }
        }
#pragma unroll
        for (int b = 0; b < 32; ++b) {
            const int x = xw * 32 + b;
            if (x < W) plane[(int64_t)y * W + x] = lab[b];
        }
    }
}

const char* launch_label_plane(const uint32_t* bits, const int* order, int n, int H, int W, uint16_t* plane, hipStream_t s) {
    const int64_t words = (int64_t)H * ((W + 31) >> 5);
    int blocks = (int)((words + 255) / 256);
    hipLaunchKernelGGL(label_plane_kernel, dim3(blocks), dim3(256), 0, s, bits, order, n, H, W, plane);
    return nullptr;
}

// ------------------------------------------------------------------------------------------------
// dst[i] = src[idx[i]] for n bit-packed masks of `words` 32-bit words: the AMG driver's compaction of NMS survivors (one launch
// instead of one hipMemcpyAsync per kept mask: ~4.5 us each, two per mask, ~500 per slice once a few hundred masks survive)
__global__ __launch_bounds__(256) void gather_masks_kernel(const uint32_t* __restrict__ src, const int* __restrict__ idx, uint32_t* __restrict__ dst,
                                                           int64_t words) {
    const int i = blockIdx.y;
    const uint32_t* s = src + (int64_t)idx[i] * words;
    uint32_t* d = dst + (int64_t)i * words;
    if ((words & 3) == 0) {
        const int64_t n4 = words >> 2;
        for (int64_t w = (int64_t)blockIdx.x * 256 + threadIdx.x; w < n4; w += (int64_t)gridDim.x * 256)
            reinterpret_cast<uint4*>(d)[w] = reinterpret_cast<const uint4*>(s)[w];
    } else {
        for (int64_t w = (int64_t)blockIdx.x * 256 + threadIdx.x; w < words; w += (int64_t)gridDim.x * 256) d[w] = s[w];
    }
}
const char* launch_gather_masks(const uint32_t* src, const int* idx, uint32_t* dst, int n, int64_t words, hipStream_t s) {
    if (n <= 0) return nullptr;
    if (n > 65535) return "gather_masks: more than 65535 masks";
    const int64_t per = ((words & 3) == 0 ? words >> 2 : words);
    hipLaunchKernelGGL(gather_masks_kernel, dim3((unsigned)std::min<int64_t>((per + 255) / 256, 64), n), dim3(256), 0, s, src, idx, dst, words);
    return nullptr;
}

// out[r][x] = bit (x & 31) of bits[r][x >> 5] as a 0/1 byte, r over n*H rows: the bool arrays the reference's dict lists hold, made on
// the device so the host receives them ready (numpy's unpackbits + bool cast of 250 masks costs seconds on the host)
__global__ __launch_bounds__(256) void unpack_masks_kernel(const uint32_t* __restrict__ bits, int64_t rows, int W, int W32, uint8_t* __restrict__ out) {
    const int halves = (W + 15) >> 4;                        // 16-pixel pieces per row
    const int64_t total = rows * halves;
    const bool vec = (W & 15) == 0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / halves;
        const int hx = (int)(i - r * halves);
        const uint32_t w = bits[r * W32 + (hx >> 1)] >> ((hx & 1) * 16);
        uint8_t* o = out + r * W + hx * 16;
        if (vec) {
            uint32_t v[4];
#pragma unroll
            for (int k = 0; k < 4; ++k)
                v[k] = ((w >> (4 * k)) & 1u) | (((w >> (4 * k + 1)) & 1u) << 8) | (((w >> (4 * k + 2)) & 1u) << 16) | (((w >> (4 * k + 3)) & 1u) << 24);
            *reinterpret_cast<uint4*>(o) = make_uint4(v[0], v[1], v[2], v[3]);
        } else {
            const int lim = min(16, W - hx * 16);
            for (int k = 0; k < lim; ++k) o[k] = (uint8_t)((w >> k) & 1u);
        }
    }
}
const char* launch_unpack_masks(const uint32_t* bits, int n, int H, int W, uint8_t* out, hipStream_t s) {
    if (n <= 0) return nullptr;
    if (H <= 0 || W <= 0) return "unpack_masks: bad shape";
    const int64_t rows = (int64_t)n * H;
    const int64_t total = rows * ((W + 15) >> 4);
    hipLaunchKernelGGL(unpack_masks_kernel, dim3((unsigned)std::min<int64_t>((total + 255) / 256, 16384)), dim3(256), 0, s, bits, rows, W, (W + 31) / 32, out);
    return nullptr;
}

// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void unpermute_nchw_kernel(const float* __restrict__ tok, int C, int stage, float* __restrict__ out) {
    const int g = 256 >> stage;
    const int64_t total = (int64_t)g * g * C;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int c = (int)(i / ((int64_t)g * g));
        const int r = (int)(i - (int64_t)c * g * g);
        const int y = r / g, x = r - y * g;
        out[i] = tok[(int64_t)perm_index(y, x, stage) * C + c];
    }
}
const char* launch_unpermute_nchw(const float* tok, int C, int stage, float* out, hipStream_t s) {
    const int g = 256 >> stage;
    const int64_t total = (int64_t)g * g * C;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(unpermute_nchw_kernel, dim3(blocks), dim3(256), 0, s, tok, C, stage, out);
    return nullptr;
}

// ------------------------------------------------------------------------------------------------
// pairwise mask intersections on bit-packed masks: inter[i][j] = popcount(bits_i & bits_j), i <= j
// (the integer counts behind reference remove_duplicate_masks' IoU, saber/segmenters/utils.py:21-29).
__global__ __launch_bounds__(256) void pair_inter_kernel(const uint32_t* __restrict__ bits, int n, int64_t words, int* __restrict__ inter) {
    const int i = blockIdx.y, j = blockIdx.x;
    if (j < i) return;
    const uint4* a = reinterpret_cast<const uint4*>(bits + (int64_t)i * words);
    const uint4* b = reinterpret_cast<const uint4*>(bits + (int64_t)j * words);
    const int64_t nv = (words & 3) ? 0 : (words >> 2);  // rows of other masks stay 16-B aligned only when words % 4 == 0
    int acc = 0;
    for (int64_t k = threadIdx.x; k < nv; k += 256) {
        const uint4 x = a[k], y = b[k];
        acc += __popc(x.x & y.x) + __popc(x.y & y.y) + __popc(x.z & y.z) + __popc(x.w & y.w);
    }
    for (int64_t k = (nv << 2) + threadIdx.x; k < words; k += 256) acc += __popc(bits[(int64_t)i * words + k] & bits[(int64_t)j * words + k]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    __shared__ int part[4];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        const int t = part[0] + part[1] + part[2] + part[3];
        inter[(int64_t)i * n + j] = t;
        inter[(int64_t)j * n + i] = t;
    }
}

// The same counts for a PT x PT tile of pairs per block: every 16-byte word of a mask is loaded once per tile row / column instead of
// once per pair (n = 249 masks of 1024^2: 8 GB of reads become 1 GB; the slice's post-filter tail 5.3 -> 1.8 ms).  words % 4 == 0.
#define PT 8
__global__ __launch_bounds__(256) void pair_inter_tiled_kernel(const uint32_t* __restrict__ bits, int n, int64_t words, int* __restrict__ inter) {
    const int ti = blockIdx.y, tj = blockIdx.x;
    if (tj < ti) return;                                  // the tile's mirror image is written by the tile above the diagonal
    const int64_t nv = words >> 2;
    const uint4* rows_a[PT];
    const uint4* rows_b[PT];
#pragma unroll
    for (int q = 0; q < PT; ++q) {
        rows_a[q] = reinterpret_cast<const uint4*>(bits + (int64_t)min(ti * PT + q, n - 1) * words);
        rows_b[q] = reinterpret_cast<const uint4*>(bits + (int64_t)min(tj * PT + q, n - 1) * words);
    }
    int acc[PT][PT];
#pragma unroll
    for (int a = 0; a < PT; ++a)
#pragma unroll
        for (int b = 0; b < PT; ++b) acc[a][b] = 0;
    for (int64_t k = threadIdx.x; k < nv; k += 256) {
        uint4 xa[PT], xb[PT];
#pragma unroll
        for (int q = 0; q < PT; ++q) { xa[q] = rows_a[q][k]; xb[q] = rows_b[q][k]; }
#pragma unroll
        for (int a = 0; a < PT; ++a)
#pragma unroll
            for (int b = 0; b < PT; ++b)
                acc[a][b] += __popc(xa[a].x & xb[b].x) + __popc(xa[a].y & xb[b].y) + __popc(xa[a].z & xb[b].z) + __popc(xa[a].w & xb[b].w);
    }
    __shared__ int part[PT * PT];
    if (threadIdx.x < PT * PT) part[threadIdx.x] = 0;
    __syncthreads();
#pragma unroll
    for (int a = 0; a < PT; ++a)
#pragma unroll
        for (int b = 0; b < PT; ++b) {
            int v = acc[a][b];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
            if ((threadIdx.x & 63) == 0) atomicAdd(&part[a * PT + b], v);
        }
    __syncthreads();
    if (threadIdx.x < PT * PT) {
        const int i = ti * PT + threadIdx.x / PT, j = tj * PT + threadIdx.x % PT;
        if (i < n && j < n) { inter[(int64_t)i * n + j] = part[threadIdx.x]; inter[(int64_t)j * n + i] = part[threadIdx.x]; }
    }
}

const char* launch_pair_intersections(const uint32_t* bits, int n, int64_t words, int* inter, hipStream_t s) {
    if (n <= 0) return nullptr;
    if ((uintptr_t)bits & 15) return "pair_intersections: masks must be 16-byte aligned";
    if ((words & 3) == 0 && n > PT && !(g_saber_debug_flags & 65536)) {
        const int nt = (n + PT - 1) / PT;
        hipLaunchKernelGGL(pair_inter_tiled_kernel, dim3(nt, nt), dim3(256), 0, s, bits, n, words, inter);
    } else {
        hipLaunchKernelGGL(pair_inter_kernel, dim3(n, n), dim3(256), 0, s, bits, n, words, inter);
    }
    return nullptr;
}
