// 3-D connected components of the stitched label volume on the device.
// Replaces saber.segmenters.utils.separate_masks (saber/segmenters/utils.py:88-131; called at the end of the slice loop,
// propagation.py:189, and of the tomogram path, tomo.py:248): foreground = label != 0 (touching objects stay merged),
// 26-connectivity, components below min_mask_area * 10 voxels removed, survivors renumbered 1..K in the order scipy.ndimage.label
// meets them (C-order scan = ascending index of a component's first voxel), uint32 output.  Integer work: bit-exact.
//
// Union-find with min-index roots (a component's root IS its first voxel in scan order):
//   1. cc_init      one wave per row: every foreground voxel starts as a child of the first voxel of its x-run
//   2. cc_merge     per voxel, the 4 rows that precede it in scan order ((z,y-1), (z-1,y-1), (z-1,y), (z-1,y+1)); per row only
//                   the centre neighbour if it is foreground (its x-neighbours hang on the same run), else the two diagonals
//   3. cc_flatten   parent <- root
//   4. cc_count     voxels per root (one atomic per x-run and 64-voxel chunk), accumulated in the OUTPUT buffer at the root's index
//   5. cc_roots     roots with >= min_vol voxels are appended to a list, the others get id 0; the host sorts the (short) list
//   6. cc_assign / cc_relabel   id = rank in the sorted list + 1; out[v] = id[root(v)] in place (a root's own entry already holds its id)
#include <algorithm>
#include <vector>

#include "engine.h"

#define CC_NONE 0xffffffffu

__device__ __forceinline__ uint32_t cc_find(uint32_t* lab, uint32_t x) {
    uint32_t p = lab[x];
    while (p != x) {
        const uint32_t g = lab[p];
        if (g != p) lab[x] = g;      // path halving: only non-root entries are written, roots change by atomicMin alone
        x = p;
        p = g;
    }
    return x;
}
__device__ __forceinline__ void cc_unite(uint32_t* lab, uint32_t a, uint32_t b) {
    while (true) {
        a = cc_find(lab, a);
        b = cc_find(lab, b);
        if (a == b) return;
        if (a < b) { const uint32_t t = a; a = b; b = t; }      // hang the larger root under the smaller one
        const uint32_t old = atomicMin(&lab[a], b);
        if (old == a) return;
        a = old;                                               // somebody re-parented a meanwhile: continue from there
    }
}

__global__ __launch_bounds__(256) void cc_init_kernel(const uint16_t* __restrict__ planes, uint32_t* __restrict__ lab, int W, int64_t rows) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;                                   // wave-uniform
    const int64_t base = row * W;
    uint32_t carry = CC_NONE;                                  // start of the run that reaches the previous chunk's last voxel
    for (int x0 = 0; x0 < W; x0 += 64) {
        const int x = x0 + lane;
        const bool fg = x < W && planes[base + x] != 0;
        const unsigned long long mask = __ballot(fg);
        uint32_t start = CC_NONE;
        if (fg) {
            const unsigned long long below_bg = ~mask & ((1ull << lane) - 1ull);
            if (below_bg == 0ull) start = carry != CC_NONE ? carry : (uint32_t)(base + x0);
            else start = (uint32_t)(base + x0 + (64 - __clzll(below_bg)));
            lab[base + x] = start;
        } else if (x < W) lab[base + x] = CC_NONE;
        carry = __shfl(start, 63, 64);                         // CC_NONE when the chunk's last voxel is background / past the row
    }
}

__global__ __launch_bounds__(256) void cc_merge_kernel(const uint16_t* __restrict__ planes, uint32_t* __restrict__ lab, int Z, int H, int W) {
    const int64_t n = (int64_t)Z * H * W;
    for (int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x; v < n; v += (int64_t)gridDim.x * 256) {
        if (planes[v] == 0) continue;
        const int x = (int)(v % W);
        const int64_t r = v / W;
        const int y = (int)(r % H), z = (int)(r / H);
        // rows that precede (z, y) in scan order and touch it
        const int dz[4] = {0, -1, -1, -1}, dy[4] = {-1, -1, 0, 1};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int zz = z + dz[k], yy = y + dy[k];
            if (zz < 0 || yy < 0 || yy >= H) continue;
            const int64_t rb = ((int64_t)zz * H + yy) * W;
            if (planes[rb + x] != 0) { cc_unite(lab, (uint32_t)v, (uint32_t)(rb + x)); continue; }
            if (x > 0 && planes[rb + x - 1] != 0) cc_unite(lab, (uint32_t)v, (uint32_t)(rb + x - 1));
            if (x + 1 < W && planes[rb + x + 1] != 0) cc_unite(lab, (uint32_t)v, (uint32_t)(rb + x + 1));
        }
    }
}

__global__ __launch_bounds__(256) void cc_flatten_kernel(uint32_t* __restrict__ lab, int64_t n) {
    for (int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x; v < n; v += (int64_t)gridDim.x * 256) {
        uint32_t p = lab[v];
        if (p == CC_NONE) continue;
        while (true) { const uint32_t g = lab[p]; if (g == p) break; p = g; }
        lab[v] = p;
    }
}

// one wave per row; consecutive foreground voxels of a row share their root, so each x-run contributes one atomic per 64-voxel chunk
__global__ __launch_bounds__(256) void cc_count_kernel(const uint32_t* __restrict__ lab, uint32_t* __restrict__ sizes, int W, int64_t rows) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int64_t base = row * W;
    for (int x0 = 0; x0 < W; x0 += 64) {
        const int x = x0 + lane;
        const uint32_t r = x < W ? lab[base + x] : CC_NONE;
        const bool fg = r != CC_NONE;
        const unsigned long long mask = __ballot(fg);
        const bool head = fg && (lane == 0 || !((mask >> (lane - 1)) & 1ull));
        if (head) {
            const unsigned long long above_bg = ~mask & ~((2ull << lane) - 1ull);      // background lanes above this one
            const int end = above_bg ? __ffsll((long long)above_bg) - 1 : 64;           // first background lane after the run
            atomicAdd(&sizes[r], (uint32_t)(end - lane));
        }
    }
}

__global__ __launch_bounds__(256) void cc_roots_kernel(const uint32_t* __restrict__ lab, uint32_t* __restrict__ sizes, int64_t n, uint32_t min_vol,
                                                       uint32_t* __restrict__ list, uint32_t cap, uint32_t* __restrict__ counter) {
    for (int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x; v < n; v += (int64_t)gridDim.x * 256) {
        if (lab[v] != (uint32_t)v) continue;                   // roots only
        if (sizes[v] >= min_vol) {
            const uint32_t i = atomicAdd(counter, 1u);
            if (i < cap) list[i] = (uint32_t)v;
        } else sizes[v] = 0u;                                   // removed component: id 0
    }
}

__global__ __launch_bounds__(256) void cc_assign_kernel(const uint32_t* __restrict__ sorted_roots, uint32_t k, uint32_t* __restrict__ ids) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i < k) ids[sorted_roots[i]] = i + 1u;
}

__global__ __launch_bounds__(256) void cc_relabel_kernel(const uint32_t* __restrict__ lab, uint32_t* __restrict__ out, int64_t n) {
    for (int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x; v < n; v += (int64_t)gridDim.x * 256) {
        const uint32_t r = lab[v];
        if (r == CC_NONE) out[v] = 0u;
        else if (r != (uint32_t)v) out[v] = out[r];            // a root's own entry already holds its id and is only ever re-written with it
    }
}

#define CC_HIP(e, call)                                                                                                  \
    do {                                                                                                                 \
        hipError_t _st = (call);                                                                                         \
        if (_st != hipSuccess) { cleanup(); return eng_fail((e), SABER_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(_st)); } \
    } while (0)

extern "C" int saber_separate_masks(saber_engine* e, const uint16_t* planes_dev, int Z, int H, int W, int min_mask_area, uint32_t* out_dev,
                                    int* out_n_labels, void* stream) {
    if (!e) return SABER_ERR_INVALID;
    if (!planes_dev || !out_dev || Z <= 0 || H <= 0 || W <= 0) return eng_fail(e, SABER_ERR_INVALID, "separate_masks: bad argument");
    const int64_t n = (int64_t)Z * H * W;
    if (n >= (int64_t)0x7fffffff) return eng_fail(e, SABER_ERR_INVALID, "separate_masks: volumes of 2^31 voxels or more are not supported");
    if (out_n_labels) *out_n_labels = 0;
    hipStream_t s = (hipStream_t)stream;
    uint32_t *lab = nullptr, *list = nullptr, *counter = nullptr;
    auto cleanup = [&]() { (void)hipFree(lab); (void)hipFree(list); (void)hipFree(counter); };
    ENG_DEVICE(e);
    const uint32_t min_vol = min_mask_area > 0 ? (uint32_t)std::min<int64_t>((int64_t)min_mask_area * 10, 0x7fffffff) : 0u;   // utils.py:113
    // a kept component has >= max(min_vol, 1) voxels; isolated voxels of a 26-connected labelling are >= 2 apart in every axis
    const int64_t cap64 = min_vol > 1 ? n / min_vol + 1 : (int64_t)((Z + 1) / 2) * ((H + 1) / 2) * ((W + 1) / 2) + 1;
    const uint32_t cap = (uint32_t)cap64;
    CC_HIP(e, hipMalloc(&lab, (size_t)n * 4));
    CC_HIP(e, hipMalloc(&list, (size_t)cap * 4));
    CC_HIP(e, hipMalloc(&counter, 4));
    CC_HIP(e, hipMemsetAsync(counter, 0, 4, s));
    CC_HIP(e, hipMemsetAsync(out_dev, 0, (size_t)n * 4, s));
    const int64_t rows = (int64_t)Z * H;
    const int row_blocks = (int)((rows + 3) / 4);
    const int vox_blocks = (int)std::min<int64_t>((n + 255) / 256, 1 << 20);
    hipLaunchKernelGGL(cc_init_kernel, dim3(row_blocks), dim3(256), 0, s, planes_dev, lab, W, rows);
    hipLaunchKernelGGL(cc_merge_kernel, dim3(vox_blocks), dim3(256), 0, s, planes_dev, lab, Z, H, W);
    hipLaunchKernelGGL(cc_flatten_kernel, dim3(vox_blocks), dim3(256), 0, s, lab, n);
    hipLaunchKernelGGL(cc_count_kernel, dim3(row_blocks), dim3(256), 0, s, (const uint32_t*)lab, out_dev, W, rows);
    hipLaunchKernelGGL(cc_roots_kernel, dim3(vox_blocks), dim3(256), 0, s, (const uint32_t*)lab, out_dev, n, std::max(min_vol, 1u), list, cap, counter);
    uint32_t k = 0;
    CC_HIP(e, hipMemcpyAsync(&k, counter, 4, hipMemcpyDeviceToHost, s));
    CC_HIP(e, hipStreamSynchronize(s));
    if (k > cap) { cleanup(); return eng_fail(e, SABER_ERR_CAPACITY, "separate_masks: internal root list overflow"); }
    if (k > 0) {
        std::vector<uint32_t> roots(k);
        CC_HIP(e, hipMemcpyAsync(roots.data(), list, (size_t)k * 4, hipMemcpyDeviceToHost, s));
        CC_HIP(e, hipStreamSynchronize(s));
        std::sort(roots.begin(), roots.end());                 // ascending first-voxel index = scipy's label order
        CC_HIP(e, hipMemcpyAsync(list, roots.data(), (size_t)k * 4, hipMemcpyHostToDevice, s));
        hipLaunchKernelGGL(cc_assign_kernel, dim3((k + 255) / 256), dim3(256), 0, s, (const uint32_t*)list, k, out_dev);
        CC_HIP(e, hipStreamSynchronize(s));                    // `roots` must outlive the upload
    }
    hipLaunchKernelGGL(cc_relabel_kernel, dim3(vox_blocks), dim3(256), 0, s, (const uint32_t*)lab, out_dev, n);
    CC_HIP(e, hipGetLastError());
    CC_HIP(e, hipStreamSynchronize(s));
    cleanup();
    if (out_n_labels) *out_n_labels = (int)k;
    return SABER_OK;
}
