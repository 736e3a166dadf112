// Device side of the automatic mask generator's post-processing (K9 filters + box NMS, K10 compaction).
//
// What upstream does on the host after every decoded point batch and crop (third-party sam2, automatic_mask_generator.py:_process_batch /
// _process_crop / _generate_masks, reached from saber/adapters/sam2/predictor.py:70): keep candidates with predicted_iou > pred_iou_thresh,
// stability_score >= stability_score_thresh, boxes not near a crop edge (is_box_near_crop_edge, atol 20); torchvision batched_nms per crop
// (box IoU > box_nms_thresh, scores = predicted IoU, stable descending order); over all crops a second NMS whose score is 1 / crop area
// (smaller crops win).  Here the same decisions are taken on the device, in the same order and with the same fp32 arithmetic as the host
// restatement in amg.hip (which stays as the fallback and as the reference these kernels are tested against), so that a slice needs ONE
// host synchronisation: the one that returns the count and the records.  fp32 steps use the __f*_rn intrinsics: a contracted fma would
// change an IoU by an ulp and with it, once in a while, a decision.
#include "common.h"
#include "engine.h"
#include "kernels.h"

// ------------------------------------------------------------------------------------------------ K9a: IoU filter + plane index
// plane_mode: where K8 finds candidate k of its group: 0 plane k, 1 plane 4 (k / 3) + 1 + k % 3 (raw planes of a multimask decode),
// 2 plane 4 k + sel[k] (raw planes of a single-mask decode with dynamic selection)
__global__ __launch_bounds__(256) void amg_plane_kernel(const float* __restrict__ iou, const int* __restrict__ sel, int n, int plane_mode, float thr,
                                                        int* __restrict__ plane, uint8_t* __restrict__ pass) {
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= n) return;
    plane[k] = plane_mode == 1 ? 4 * (k / 3) + 1 + k % 3 : plane_mode == 2 ? 4 * k + sel[k] : k;
    pass[k] = (!(thr > 0.0f) || iou[k] > thr) ? 1 : 0;
}
const char* launch_amg_plane(const float* iou, const int* sel, int n, int plane_mode, float thr, int* plane, uint8_t* pass, hipStream_t s) {
    if (n <= 0) return nullptr;
    hipLaunchKernelGGL(amg_plane_kernel, dim3((n + 255) / 256), dim3(256), 0, s, iou, sel, n, plane_mode, thr, plane, pass);
    return nullptr;
}

// ------------------------------------------------------------------------------------------------ shared pieces
__device__ __forceinline__ float box_iou_rn(const float4 a, const float4 b) {      // torchvision nms: inter / (area_a + area_b - inter), fp32
    const float iarea = __fmul_rn(__fsub_rn(a.z, a.x), __fsub_rn(a.w, a.y));
    const float jarea = __fmul_rn(__fsub_rn(b.z, b.x), __fsub_rn(b.w, b.y));
    const float xx1 = fmaxf(a.x, b.x), yy1 = fmaxf(a.y, b.y), xx2 = fminf(a.z, b.z), yy2 = fminf(a.w, b.w);
    const float w = fmaxf(0.0f, __fsub_rn(xx2, xx1)), h = fmaxf(0.0f, __fsub_rn(yy2, yy1));
    const float inter = __fmul_rn(w, h);
    return __fdiv_rn(inter, __fsub_rn(__fadd_rn(iarea, jarea), inter));
}

// exclusive scan of one flag per thread over a 1024-thread block; returns this thread's offset, *total = the block's sum
__device__ __forceinline__ int block_scan_1024(int flag, int* lds16, int* total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned long long b = __ballot(flag);
    const int within = __popcll(b & ((1ull << lane) - 1ull));
    if (lane == 0) lds16[wave] = __popcll(b);
    __syncthreads();
    int base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) { const int c = lds16[w]; if (w < wave) base += c; tot += c; }
    __syncthreads();
    *total = tot;
    return base + within;
}

// Greedy NMS of n candidates already in `cand` (any order), scores in cand[i].score: stable descending order, suppress IoU > thr.
// order / removed: LDS.  On return keep order is order[a] for the a with removed[a] == 0, ascending a.  n <= AMG_NMS_MAX.
#define AMG_NMS_MAX 12288
__device__ void block_nms(const DevCand* __restrict__ cand, int n, float thr, float* s_score, unsigned short* s_order, uint8_t* s_removed) {
    const int tid = threadIdx.x;
    for (int i = tid; i < n; i += 1024) { s_score[i] = cand[i].score; s_removed[i] = 0; }
    __syncthreads();
    // stable rank: position of i among the candidates sorted by descending score, ties by ascending index
    for (int i = tid; i < n; i += 1024) {
        const float si = s_score[i];
        int r = 0;
        for (int j = 0; j < n; ++j) {
            const float sj = s_score[j];
            r += (sj > si || (sj == si && j < i)) ? 1 : 0;
        }
        s_order[r] = (unsigned short)i;
    }
    __syncthreads();
    if (!(thr < 1.0f)) return;                      // an IoU is never above 1: nothing can be suppressed
    for (int a = 0; a < n; ++a) {
        if (s_removed[a]) continue;                 // block-uniform: written before the last barrier
        const DevCand& ca = cand[s_order[a]];
        const float4 ba = make_float4(ca.box[0], ca.box[1], ca.box[2], ca.box[3]);
        for (int b = a + 1 + tid; b < n; b += 1024) {
            if (s_removed[b]) continue;
            const DevCand& cb = cand[s_order[b]];
            if (box_iou_rn(ba, make_float4(cb.box[0], cb.box[1], cb.box[2], cb.box[3])) > thr) s_removed[b] = 1;
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------ K9b + K10: one crop per workgroup
// filters -> compacted candidates (tmp, candidate order) -> NMS -> survivors in NMS order (keep) + their count
__global__ __launch_bounds__(1024) void amg_crop_kernel(const DevCrop* __restrict__ crops, const MaskStats* __restrict__ stats, const uint8_t* __restrict__ pass,
                                                        const float* __restrict__ iou, const float* __restrict__ crop_pts, float stab_thr, float nms_thr, int H, int W,
                                                        int slot_base, DevCand* __restrict__ tmp, DevCand* __restrict__ keep, int* __restrict__ counts) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* s_score = reinterpret_cast<float*>(smem);                                  // AMG_NMS_MAX floats
    unsigned short* s_order = reinterpret_cast<unsigned short*>(s_score + AMG_NMS_MAX);
    uint8_t* s_removed = reinterpret_cast<uint8_t*>(s_order + AMG_NMS_MAX);
    __shared__ int lds16[16];
    const DevCrop cr = crops[blockIdx.x];
    const int tid = threadIdx.x;
    DevCand* t = tmp + cr.kbase;
    int n = 0;
    for (int k0 = 0; k0 < cr.nm; k0 += 1024) {
        const int k = k0 + tid;
        bool ok = false;
        DevCand cd;
        if (k < cr.nm && pass[cr.kbase + k]) {
            const MaskStats st = stats[cr.kbase + k];
            const float stab = __fdiv_rn((float)st.inter, (float)st.uni);          // 0 / 0 -> nan fails the filter like upstream
            ok = !(stab_thr > 0.0f && !(stab >= stab_thr));
            if (st.area > 0) { cd.box[0] = (float)st.x0; cd.box[1] = (float)st.y0; cd.box[2] = (float)st.x1; cd.box[3] = (float)st.y1; }
            else { cd.box[0] = (float)cr.box[0]; cd.box[1] = (float)cr.box[1]; cd.box[2] = (float)cr.box[0]; cd.box[3] = (float)cr.box[1]; }
            const float cbx[4] = {(float)cr.box[0], (float)cr.box[1], (float)cr.box[2], (float)cr.box[3]};
            const float obx[4] = {0.f, 0.f, (float)W, (float)H};
            bool near = false;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const bool nc_ = fabsf(__fsub_rn(cd.box[q], cbx[q])) <= 20.0f, ni = fabsf(__fsub_rn(cd.box[q], obx[q])) <= 20.0f;
                near = near || (nc_ && !ni);
            }
            ok = ok && !near;
            cd.iou = iou[cr.kbase + k];
            cd.stab = stab;
            const int pk = k / cr.M;
            cd.pt[0] = __fadd_rn(crop_pts[2 * (cr.pt0 + pk)], (float)cr.box[0]);
            cd.pt[1] = __fadd_rn(crop_pts[2 * (cr.pt0 + pk) + 1], (float)cr.box[1]);
#pragma unroll
            for (int q = 0; q < 4; ++q) cd.crop[q] = cr.box[q];
            cd.area = st.area;
            cd.slot = slot_base + cr.kbase + k;
            cd.score = cd.iou;
            cd.pad = 0;
        }
        int tot;
        const int pos = block_scan_1024(ok ? 1 : 0, lds16, &tot);
        if (ok) t[n + pos] = cd;
        n += tot;
    }
    __threadfence_block();
    __syncthreads();
    if (n == 0) { if (tid == 0) counts[blockIdx.x] = 0; return; }
    block_nms(t, n, nms_thr, s_score, s_order, s_removed);
    __syncthreads();
    DevCand* kp = keep + cr.kbase;
    int nk = 0;
    for (int a0 = 0; a0 < n; a0 += 1024) {
        const int a = a0 + tid;
        const bool kept = a < n && !s_removed[a];
        int tot;
        const int pos = block_scan_1024(kept ? 1 : 0, lds16, &tot);
        if (kept) kp[nk + pos] = t[s_order[a]];
        nk += tot;
    }
    if (tid == 0) counts[blockIdx.x] = nk;
}

// survivors of the batch's crops, crop by crop, behind those of the earlier batches
__global__ __launch_bounds__(256) void amg_append_kernel(const DevCrop* __restrict__ crops, int n_crops, const DevCand* __restrict__ keep, const int* __restrict__ counts,
                                                         DevCand* __restrict__ surv, int* __restrict__ nsurv, int cap) {
    int base = *nsurv;
    __syncthreads();
    for (int c = 0; c < n_crops; ++c) {
        const int cnt = counts[c];
        const DevCand* src = keep + crops[c].kbase;
        for (int i = threadIdx.x; i < cnt; i += 256)
            if (base + i < cap) surv[base + i] = src[i];
        base += cnt;
    }
    __syncthreads();
    if (threadIdx.x == 0) *nsurv = base;            // may exceed cap: the host sees it and falls back
}

// cross-crop NMS (score = 1 / crop area: candidates of smaller crops first), final order, records
__global__ __launch_bounds__(1024) void amg_final_kernel(DevCand* __restrict__ surv, const int* __restrict__ nsurv, int cap, int multi_crop, float nms_thr, int max_masks,
                                                         int* __restrict__ final_slots, saber_mask_meta* __restrict__ meta, int* __restrict__ out_count) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* s_score = reinterpret_cast<float*>(smem);
    unsigned short* s_order = reinterpret_cast<unsigned short*>(s_score + AMG_NMS_MAX);
    uint8_t* s_removed = reinterpret_cast<uint8_t*>(s_order + AMG_NMS_MAX);
    __shared__ int lds16[16];
    const int tid = threadIdx.x;
    const int n = *nsurv;
    if (n > cap || n > AMG_NMS_MAX) { if (tid == 0) *out_count = -1; return; }     // host fallback
    if (n == 0) { if (tid == 0) *out_count = 0; return; }
    if (multi_crop) {
        for (int i = tid; i < n; i += 1024) {
            const float a = __fmul_rn((float)(surv[i].crop[2] - surv[i].crop[0]), (float)(surv[i].crop[3] - surv[i].crop[1]));
            surv[i].score = __fdiv_rn(1.0f, a);
        }
        __threadfence_block();
        __syncthreads();
        block_nms(surv, n, nms_thr, s_score, s_order, s_removed);
        __syncthreads();
    } else {
        for (int i = tid; i < n; i += 1024) { s_order[i] = (unsigned short)i; s_removed[i] = 0; }
        __syncthreads();
    }
    int nf = 0;
    for (int a0 = 0; a0 < n; a0 += 1024) {
        const int a = a0 + tid;
        const bool kept = a < n && !s_removed[a];
        int tot;
        const int pos = block_scan_1024(kept ? 1 : 0, lds16, &tot);
        const int o = nf + pos;
        if (kept && o < max_masks) {
            const DevCand& cd = surv[s_order[a]];
            final_slots[o] = cd.slot;
            saber_mask_meta m;
            m.area = cd.area;
            m.bbox_xywh[0] = cd.box[0]; m.bbox_xywh[1] = cd.box[1]; m.bbox_xywh[2] = __fsub_rn(cd.box[2], cd.box[0]); m.bbox_xywh[3] = __fsub_rn(cd.box[3], cd.box[1]);
            m.predicted_iou = cd.iou;
            m.stability_score = cd.stab;
            m.point_xy[0] = cd.pt[0]; m.point_xy[1] = cd.pt[1];
            m.crop_box_xywh[0] = (float)cd.crop[0]; m.crop_box_xywh[1] = (float)cd.crop[1];
            m.crop_box_xywh[2] = (float)(cd.crop[2] - cd.crop[0]); m.crop_box_xywh[3] = (float)(cd.crop[3] - cd.crop[1]);
            meta[o] = m;
        }
        nf += tot;
    }
    if (tid == 0) *out_count = nf;
}

// out[i] = scratch[final_slots[i]] for i < min(*count, max_masks)
__global__ __launch_bounds__(256) void amg_gather_kernel(const uint32_t* __restrict__ src, const int* __restrict__ slots, const int* __restrict__ count, int max_masks,
                                                         uint32_t* __restrict__ dst, int64_t words) {
    const int i = blockIdx.y;
    const int n = *count;
    if (i >= n || i >= max_masks) return;
    const uint32_t* s = src + (int64_t)slots[i] * words;
    uint32_t* d = dst + (int64_t)i * words;
    if ((words & 3) == 0) {
        const int64_t per = words >> 2;
        for (int64_t w = (int64_t)blockIdx.x * 256 + threadIdx.x; w < per; w += (int64_t)gridDim.x * 256)
            reinterpret_cast<u32x4*>(d)[w] = reinterpret_cast<const u32x4*>(s)[w];
    } else {
        for (int64_t w = (int64_t)blockIdx.x * 256 + threadIdx.x; w < words; w += (int64_t)gridDim.x * 256) d[w] = s[w];
    }
}

#define AMG_NMS_LDS (AMG_NMS_MAX * (4 + 2 + 1))

// the NMS building block on caller-supplied boxes (C-ABI saber_k_box_nms: kernel-level test against a host greedy NMS)
__global__ __launch_bounds__(1024) void amg_nms_probe_kernel(const float* __restrict__ boxes, const float* __restrict__ scores, int n, float thr, DevCand* __restrict__ tmp,
                                                             int* __restrict__ keep, int* __restrict__ count) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* s_score = reinterpret_cast<float*>(smem);
    unsigned short* s_order = reinterpret_cast<unsigned short*>(s_score + AMG_NMS_MAX);
    uint8_t* s_removed = reinterpret_cast<uint8_t*>(s_order + AMG_NMS_MAX);
    __shared__ int lds16[16];
    const int tid = threadIdx.x;
    for (int i = tid; i < n; i += 1024) {
        DevCand c;
#pragma unroll
        for (int q = 0; q < 4; ++q) { c.box[q] = boxes[4 * i + q]; c.crop[q] = 0; }
        c.iou = c.stab = 0.f; c.pt[0] = c.pt[1] = 0.f; c.area = 0; c.slot = i; c.score = scores[i]; c.pad = 0;
        tmp[i] = c;
    }
    __syncthreads();
    block_nms(tmp, n, thr, s_score, s_order, s_removed);
    __syncthreads();
    int nk = 0;
    for (int a0 = 0; a0 < n; a0 += 1024) {
        const int a = a0 + tid;
        const bool kept = a < n && !s_removed[a];
        int tot;
        const int pos = block_scan_1024(kept ? 1 : 0, lds16, &tot);
        if (kept) keep[nk + pos] = s_order[a];
        nk += tot;
    }
    if (tid == 0) *count = nk;
}
const char* launch_box_nms_probe(const float* boxes, const float* scores, int n, float thr, void* tmp, int* keep, int* count, hipStream_t s) {
    if (n < 0 || n > AMG_NMS_MAX) return "box_nms: at most 12288 boxes";
    hipLaunchKernelGGL(amg_nms_probe_kernel, dim3(1), dim3(1024), AMG_NMS_LDS, s, boxes, scores, n, thr, reinterpret_cast<DevCand*>(tmp), keep, count);
    return nullptr;
}

const char* amg_device_init() {
    hipError_t st = hipFuncSetAttribute(reinterpret_cast<const void*>(amg_crop_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, AMG_NMS_LDS);
    if (st == hipSuccess) st = hipFuncSetAttribute(reinterpret_cast<const void*>(amg_final_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, AMG_NMS_LDS);
    if (st == hipSuccess) st = hipFuncSetAttribute(reinterpret_cast<const void*>(amg_nms_probe_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, AMG_NMS_LDS);
    return st == hipSuccess ? nullptr : hipGetErrorString(st);
}
const char* launch_amg_crops(const DevCrop* crops, int n_crops, int max_nm, const MaskStats* stats, const uint8_t* pass, const float* iou, const float* crop_pts, float stab_thr,
                             float nms_thr, int H, int W, int slot_base, DevCand* tmp, DevCand* keep, int* counts, DevCand* surv, int* nsurv, int cap, hipStream_t s) {
    if (n_crops <= 0) return nullptr;
    if (max_nm > AMG_NMS_MAX) return "amg: more candidates per crop than the device NMS holds";
    hipLaunchKernelGGL(amg_crop_kernel, dim3(n_crops), dim3(1024), AMG_NMS_LDS, s, crops, stats, pass, iou, crop_pts, stab_thr, nms_thr, H, W, slot_base, tmp, keep, counts);
    hipLaunchKernelGGL(amg_append_kernel, dim3(1), dim3(256), 0, s, crops, n_crops, keep, counts, surv, nsurv, cap);
    return nullptr;
}
const char* launch_amg_final(DevCand* surv, const int* nsurv, int cap, int multi_crop, float nms_thr, int max_masks, int* final_slots, saber_mask_meta* meta, int* out_count,
                             const uint32_t* scratch, uint32_t* out_bits, int64_t words, hipStream_t s) {
    hipLaunchKernelGGL(amg_final_kernel, dim3(1), dim3(1024), AMG_NMS_LDS, s, surv, nsurv, cap, multi_crop, nms_thr, max_masks, final_slots, meta, out_count);
    if (max_masks > 0) {
        if (max_masks > 65535) return "amg: more than 65535 masks";
        const int64_t per = ((words & 3) == 0 ? words >> 2 : words);
        hipLaunchKernelGGL(amg_gather_kernel, dim3((unsigned)std::min<int64_t>((per + 255) / 256, 64), max_masks), dim3(256), 0, s, scratch, final_slots, out_count, max_masks, out_bits, words);
    }
    return nullptr;
}
