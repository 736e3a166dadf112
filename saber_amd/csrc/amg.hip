// Automatic mask generation driver (host control flow + small device helpers).
// Replaces SAM2AutomaticMaskGenerator.generate as the reference drives it
// (saber/adapters/sam2/predictor.py:70 -> saber/adapters/sam2/amg.py:161-183; parameters
// saber/adapters/sam2/automask.py:66-78).  Semantics restated from upstream sam2
// (automatic_mask_generator.py, utils/amg.py) - see SURVEY.md 3.3 / 8a b11-b12:
//   crops -> per-crop encode -> point grid decode (+ m2m refinement) -> pred_iou filter ->
//   stability filter -> threshold + bbox -> near-crop-edge filter -> per-crop NMS -> cross-crop NMS.
// Device work: encode/decode (engine.hip), K8 mask_post (decoder_ops.hip).  The host only sees
// per-mask scalars (iou, counts, boxes); masks stay bit-packed on the device.
#include <algorithm>
#include <array>
#include <cmath>
#include <numeric>

#include "common.h"
#include "engine.h"

#define TRY(x) do { int _r = (x); if (_r != SABER_OK) return _r; } while (0)

struct Cand {
    float box[4];       // xyxy in full-image coordinates (inclusive max edge, upstream batched_mask_to_box)
    float iou, stab;
    float pt[2];
    int crop[4];
    int area;
    size_t bits_slot;   // index into the accumulated bit-mask buffer
};

static std::vector<int> nms_host(const std::vector<Cand>& c, const std::vector<float>& scores, float thr) {
    // torchvision.ops.nms: stable sort by descending score, suppress IoU > thr (fp32 arithmetic)
    const int n = (int)c.size();
    std::vector<int> order(n);
    std::iota(order.begin(), order.end(), 0);
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return scores[a] > scores[b]; });
    std::vector<char> sup(n, 0);
    std::vector<int> keep;
    for (int _i = 0; _i < n; ++_i) {
        const int i = order[_i];
        if (sup[i]) continue;
        keep.push_back(i);
        const float* bi = c[i].box;
        const float iarea = (bi[2] - bi[0]) * (bi[3] - bi[1]);
        for (int _j = _i + 1; _j < n; ++_j) {
            const int j = order[_j];
            if (sup[j]) continue;
            const float* bj = c[j].box;
            const float xx1 = std::max(bi[0], bj[0]), yy1 = std::max(bi[1], bj[1]);
            const float xx2 = std::min(bi[2], bj[2]), yy2 = std::min(bi[3], bj[3]);
            const float w = std::max(0.0f, xx2 - xx1), h = std::max(0.0f, yy2 - yy1);
            const float inter = w * h;
            const float jarea = (bj[2] - bj[0]) * (bj[3] - bj[1]);
            const float ovr = inter / (iarea + jarea - inter);
            if (ovr > thr) sup[j] = 1;
        }
    }
    return keep;
}

static void gen_crop_boxes(int H, int W, int n_layers, float overlap_ratio, std::vector<std::array<int, 4>>& boxes, std::vector<int>& layers) {
    boxes.push_back({0, 0, W, H});
    layers.push_back(0);
    const int short_side = std::min(H, W);
    auto crop_len = [](int orig, int n, int overlap) { return (int)std::ceil((double)(overlap * (n - 1) + orig) / n); };
    for (int l = 0; l < n_layers; ++l) {
        const int ns = 1 << (l + 1);
        const int overlap = (int)((double)overlap_ratio * short_side * (2.0 / ns));
        const int cw = crop_len(W, ns, overlap), ch = crop_len(H, ns, overlap);
        for (int ix = 0; ix < ns; ++ix)        // itertools.product(x0s, y0s): x outer, y inner
            for (int iy = 0; iy < ns; ++iy) {
                const int x0 = (cw - overlap) * ix, y0 = (ch - overlap) * iy;
                boxes.push_back({x0, y0, std::min(x0 + cw, W), std::min(y0 + ch, H)});
                layers.push_back(l + 1);
            }
    }
}

extern "C" int saber_amg_last_syncs(const saber_engine* e) { return e ? e->amg_last_syncs : -1; }
extern "C" int saber_engine_set_device_amg(saber_engine* e, int enable) {
    if (!e) return SABER_ERR_INVALID;
    e->amg_device = enable != 0;
    return SABER_OK;
}
extern "C" int saber_amg_last_pruning(const saber_engine* e, int64_t* pruned, int64_t* m2m_candidates) {
    if (!e || !pruned || !m2m_candidates) return SABER_ERR_INVALID;
    *pruned = e->amg_last_pruned; *m2m_candidates = e->amg_last_m2m;
    return SABER_OK;
}

extern "C" int saber_amg_generate(saber_engine* e, const float* img_dev, int H, int W, int channels, const saber_amg_params* prm,
                                  uint32_t* out_bits_dev, int max_masks, saber_mask_meta* out_meta, int* out_count, void* stream) {
    if (!e) return SABER_ERR_INVALID;
    if (!e->finalized) return eng_fail(e, SABER_ERR_STATE, "engine not finalized");
    if (!img_dev || !prm || !out_count || H <= 0 || W <= 0 || max_masks < 0 || (max_masks > 0 && (!out_bits_dev || !out_meta)))
        return eng_fail(e, SABER_ERR_INVALID, "amg_generate: bad argument");
    if (prm->points_per_side <= 0 || prm->points_per_batch <= 0 || prm->crop_n_layers < 0 || prm->crop_n_layers > 4 ||
        prm->crop_n_points_downscale_factor <= 0)
        return eng_fail(e, SABER_ERR_INVALID, "amg_generate: bad cfgAMG value");
    ENG_DEVICE(e);
    hipStream_t s = (hipStream_t)stream;
    *out_count = 0;
    e->amg_last_syncs = 0;
    ENG_HIP(e, hipMemsetAsync(e->prune_counters, 0, 32, (hipStream_t)stream));       // pruning statistics + the overflow sentinel's counters
    const int W32 = (W + 31) >> 5;
    const size_t mask_words = (size_t)H * W32;
    const int M = prm->multimask_output ? 3 : 1;

    std::vector<std::array<int, 4>> crops;
    std::vector<int> layers;
    gen_crop_boxes(H, W, prm->crop_n_layers, prm->crop_overlap_ratio, crops, layers);
    // point grids per layer (numpy float64 linspace semantics)
    std::vector<std::vector<double>> grids(prm->crop_n_layers + 1);
    std::vector<int> grid_n(prm->crop_n_layers + 1);
    for (int l = 0; l <= prm->crop_n_layers; ++l) {
        const int n = (int)((double)prm->points_per_side / std::pow((double)prm->crop_n_points_downscale_factor, l));
        if (n <= 0) return eng_fail(e, SABER_ERR_INVALID, "amg_generate: point grid of a crop layer is empty");
        grid_n[l] = n;
        const double off = 1.0 / (2.0 * n);
        grids[l].resize(n);
        for (int i = 0; i < n; ++i) grids[l][i] = n == 1 ? off : off + (1.0 - 2.0 * off) * (double)i / (double)(n - 1);
    }
    // every decode of an encoder batch (up to max_images crops, all their layers) writes into one set of buffers, so they hold the
    // largest batch's points / candidates
    size_t max_pts = 0;
    for (int c0 = 0; c0 < (int)crops.size(); c0 += e->max_images) {
        size_t n = 0;
        for (int i = c0; i < std::min((int)crops.size(), c0 + e->max_images); ++i) n += (size_t)grid_n[layers[i]] * grid_n[layers[i]];
        max_pts = std::max(max_pts, n);
    }
    size_t max_prompts = max_pts * M;
    // two capacities: the point-indexed buffers (amg_pts, amg_low1) follow the number of grid points, the candidate-indexed ones the number
    // of prompts (points x masks per point); a later call with fewer masks per point but a denser grid must regrow the former
    if (e->amg_prompts_cap < max_prompts || e->amg_pts_cap < max_pts) {
        max_pts = std::max(max_pts, e->amg_pts_cap);
        max_prompts = std::max(max_prompts, e->amg_prompts_cap);
        eng_graphs_flush(e);                   // captured launches hold the addresses of the buffers released below
        TRY(eng_regrow(e, &e->amg_pts, max_pts * 2, s));
        TRY(eng_regrow(e, &e->amg_pts2, max_prompts * 2, s));
        // the decoder leaves all 4 low-res planes of a prompt in place (no selection copies): first pass max_pts x 4, m2m pass max_prompts x 4
        TRY(eng_regrow(e, &e->amg_low1, max_pts * 4 * 65536, s));
        TRY(eng_regrow(e, &e->amg_iou1, max_prompts, s));
        TRY(eng_regrow(e, &e->amg_low2, prm->use_m2m ? max_prompts * 4 * 65536 : 4, s));
        TRY(eng_regrow(e, &e->amg_iou2, max_prompts, s));
        TRY(eng_regrow(e, &e->amg_sel, max_prompts, s));
        e->amg_prompts_cap = max_prompts;
        e->amg_pts_cap = max_pts;
        e->amg_m2m_sized = prm->use_m2m != 0;
    } else if (prm->use_m2m && !e->amg_m2m_sized) {
        eng_graphs_flush(e);
        TRY(eng_regrow(e, &e->amg_low2, e->amg_prompts_cap * 4 * 65536, s));
        e->amg_m2m_sized = true;
    }
    if (e->amg_stats_cap < max_prompts) {
        TRY(eng_regrow(e, &e->amg_stats, max_prompts, s));
        TRY(eng_regrow(e, &e->amg_idx, max_prompts, s));
        e->amg_stats_cap = max_prompts;
    }

    std::vector<Cand> all;            // accumulated over crops (after per-crop NMS)
    size_t acc_used = 0;              // masks stored in the persistent accumulation buffer e->amg_bits
    auto acc_reserve = [&](size_t need) -> int {
        if (need * mask_words <= e->amg_bits_words) return SABER_OK;
        const size_t ncap = std::max<size_t>(need, std::max<size_t>(64, 2 * (e->amg_bits_words / mask_words)));
        uint32_t* nb = nullptr;
        TRY(eng_alloc(e, &nb, ncap * mask_words));
        if (acc_used) ENG_HIP(e, hipMemcpyAsync(nb, e->amg_bits, acc_used * mask_words * 4, hipMemcpyDeviceToDevice, s));
        { ENG_HIP(e, hipStreamSynchronize(s)); ++e->amg_last_syncs; }
        eng_free(e, e->amg_bits);
        e->amg_bits = nb; e->amg_bits_words = ncap * mask_words;
        return SABER_OK;
    };
    auto crop_reserve = [&](size_t need) -> int {  // scratch for one crop's pred_iou survivors
        if (need * mask_words <= e->amg_crop_words) return SABER_OK;
        { ENG_HIP(e, hipStreamSynchronize(s)); ++e->amg_last_syncs; }
        eng_free(e, e->amg_crop_bits);
        e->amg_crop_bits = nullptr; e->amg_crop_words = 0;
        TRY(eng_alloc(e, &e->amg_crop_bits, need * mask_words));
        e->amg_crop_words = need * mask_words;
        return SABER_OK;
    };
    // ---- device-side post-processing (the default): everything after the decodes stays on the stream; the host path below is the fallback
    // for generators with more candidates than the device NMS holds, and the reference the device path is tested against
    size_t total_cand = 0;
    int max_nm = 0;
    for (int i = 0; i < (int)crops.size(); ++i) { const int nm_ = grid_n[layers[i]] * grid_n[layers[i]] * M; total_cand += (size_t)nm_; max_nm = std::max(max_nm, nm_); }
    static const bool host_amg_env = getenv("SABER_AMD_HOST_AMG") != nullptr;
    const bool use_dev = e->amg_device && !host_amg_env && total_cand <= 12288 && total_cand * mask_words * 4 <= ((size_t)6 << 30) && max_masks > 0;
    size_t slot_base = 0;
    if (use_dev) {
        if (e->amg_dev_cap < total_cand) {
            TRY(eng_regrow(e, &e->amg_pass, total_cand, s)); TRY(eng_regrow(e, &e->amg_tmp, total_cand, s)); TRY(eng_regrow(e, &e->amg_keep, total_cand, s));
            TRY(eng_regrow(e, &e->amg_surv, total_cand, s)); TRY(eng_regrow(e, &e->amg_crops_dev, (size_t)256, s)); TRY(eng_regrow(e, &e->amg_counts, (size_t)256, s));
            TRY(eng_regrow(e, &e->amg_crop_pts, total_cand * 2, s));
            if (!e->amg_nsurv) { TRY(eng_alloc(e, &e->amg_nsurv, 1)); TRY(eng_alloc(e, &e->amg_count_dev, 1)); }
            e->amg_dev_cap = total_cand;
        }
        if (e->amg_dev_masks_cap < (size_t)max_masks) {
            TRY(eng_regrow(e, &e->amg_final_slots, (size_t)max_masks, s)); TRY(eng_regrow(e, &e->amg_meta_dev, (size_t)max_masks, s));
            e->amg_dev_masks_cap = max_masks;
        }
        if (crops.size() > 256) return eng_fail(e, SABER_ERR_INVALID, "amg_generate: more than 256 crops");
        TRY(crop_reserve(total_cand));                 // the K8 scratch holds EVERY candidate's slot: survivors are read from it at the end
        ENG_HIP(e, hipMemsetAsync(e->amg_nsurv, 0, sizeof(int), s));
    }
    std::vector<float> h_iou;
    std::vector<MaskStats> h_stats;
    std::vector<int> h_idx, h_plane, keep_src;
    std::vector<float> h_pts, h_pts2;

    // engine-owned copy of the image: the encoder pass below is replayed from a hipGraph whose launches hold its address
    {
        const size_t need = (size_t)H * W * channels;
        if (e->amg_img_elems < need) { eng_graphs_flush(e); TRY(eng_regrow(e, &e->amg_img, need, s)); e->amg_img_elems = need; }
        ENG_HIP(e, hipMemcpyAsync(e->amg_img, img_dev, need * sizeof(float), hipMemcpyDeviceToDevice, s));
    }
    auto key_of = [](std::initializer_list<long long> v) { std::string k; for (long long x : v) { k += std::to_string(x); k += ','; } return k; };
    const int nc = (int)crops.size();
    for (int c0 = 0; c0 < nc; c0 += e->max_images) {
        const int ncb = std::min(e->max_images, nc - c0);
        std::vector<int> cb(4 * ncb);
        for (int i = 0; i < ncb; ++i) for (int k = 0; k < 4; ++k) cb[4 * i + k] = crops[c0 + i][k];
        {
            const float* img = e->amg_img;
            // the crop boxes go through the pinned buffer the (possibly replayed) H2D copy of eng_encode reads at execution time
            if (!e->crops_pin) ENG_HIP(e, hipHostMalloc(reinterpret_cast<void**>(&e->crops_pin), sizeof(int) * 256 * 8));
            std::copy(cb.begin(), cb.begin() + 4 * ncb, e->crops_pin);
            // (saber_engine_set_encoder_stream: the pass runs on the engine's encoder stream - e.g. one restricted to a subset of the CUs -
            // after everything queued on s so far, and s continues after it)
            hipStream_t es = e->enc_stream ? e->enc_stream : s;
            if (es != s) {
                for (int k = 0; k < 2; ++k) if (!e->enc_ev[k]) ENG_HIP(e, hipEventCreateWithFlags(&e->enc_ev[k], hipEventDisableTiming));
                ENG_HIP(e, hipEventRecord(e->enc_ev[0], s));
                ENG_HIP(e, hipStreamWaitEvent(es, e->enc_ev[0], 0));
            }
            TRY(eng_graphed(e, "enc," + key_of({(long long)(uintptr_t)img, H, W, channels, ncb, c0, prm->crop_n_layers, (long long)(prm->crop_overlap_ratio * 1e6), (long long)(uintptr_t)es}), es,
                            [&]() { return eng_encode(e, img, H, W, channels, e->crops_pin, ncb, 0, es); }));
            if (es != s) {
                ENG_HIP(e, hipEventRecord(e->enc_ev[1], es));
                ENG_HIP(e, hipStreamWaitEvent(s, e->enc_ev[1], 0));
            }
            // host-side bookkeeping of eng_encode, which a replay does not execute
            for (int i = 0; i < ncb; ++i) { e->slot_valid[i] = 1; e->slot_shared_valid[i] = 0; e->slot_embb_valid[i] = 0; }
        }
        // ---- the crops of one layer that were encoded together are decoded as ONE batch of prompts; all groups of the encoder batch
        // are decoded back to back WITHOUT a host round trip, then the per-candidate scalars come back once (sync 1), K8 runs for every
        // crop, its per-mask statistics come back once (sync 2), and the filters / NMS run on the host: 3 synchronisations per slice
        // with the final one (24 in round 1, 7 with one pair per group).
        struct Grp { int ci, G, layer, gn, np, nm; size_t pt0, k0; int plane_mode; const float* masks; const float* ious; };
        std::vector<Grp> groups;
        size_t n_pts = 0;
        for (int ci = 0, G = 1; ci < ncb; ci += G) {
            const int layer = layers[c0 + ci];
            for (G = 1; ci + G < ncb && layers[c0 + ci + G] == layer; ++G) {}
            Grp g{ci, G, layer, grid_n[layer], grid_n[layer] * grid_n[layer], grid_n[layer] * grid_n[layer] * M, n_pts, n_pts * M, 0, nullptr, nullptr};
            groups.push_back(g);
            n_pts += (size_t)G * g.np;
        }
        const size_t n_cand = n_pts * M;
        const int first_raw = prm->multimask_output ? 1 : 0;     // the multimask first pass leaves its 4 planes per prompt in place (read through index maps)
        h_pts.resize(n_pts * 2);
        std::vector<float> crop_pts_all(n_pts * 2);
        for (const Grp& gr : groups)
            for (int g = 0; g < gr.G; ++g) {
                const auto& box = crops[c0 + gr.ci + g];
                const int cw = box[2] - box[0], chh = box[3] - box[1];
                for (int iy = 0; iy < gr.gn; ++iy)
                    for (int ix = 0; ix < gr.gn; ++ix) {
                        const float px = (float)(grids[gr.layer][ix] * (double)cw), py = (float)(grids[gr.layer][iy] * (double)chh);
                        const size_t k = gr.pt0 + (size_t)g * gr.np + (size_t)iy * gr.gn + ix;
                        crop_pts_all[2 * k] = px; crop_pts_all[2 * k + 1] = py;
                        h_pts[2 * k] = (px / (float)cw) * 1024.0f;
                        h_pts[2 * k + 1] = (py / (float)chh) * 1024.0f;
                    }
            }
        ENG_HIP(e, hipMemcpyAsync(e->amg_pts, h_pts.data(), sizeof(float) * 2 * n_pts, hipMemcpyHostToDevice, s));
        if (prm->use_m2m) {
            h_pts2.resize(n_cand * 2);
            for (size_t k = 0; k < n_pts; ++k)
                for (int m = 0; m < M; ++m) { h_pts2[2 * (k * M + m)] = h_pts[2 * k]; h_pts2[2 * (k * M + m) + 1] = h_pts[2 * k + 1]; }
            ENG_HIP(e, hipMemcpyAsync(e->amg_pts2, h_pts2.data(), sizeof(float) * 2 * n_cand, hipMemcpyHostToDevice, s));
        }
        // ---- phase 1: every decode of the encoder batch
        for (Grp& gr : groups) {
            const int ci = gr.ci, G = gr.G, np = gr.np, nm = gr.nm;
            float* pts1 = e->amg_pts + 2 * gr.pt0;
            float* low1 = e->amg_low1 + gr.pt0 * (first_raw ? 4 : 1) * 65536;
            float* iou1 = e->amg_iou1 + gr.k0;
            TRY(eng_graphed(e, "dec1," + key_of({ci, np, (long long)(uintptr_t)pts1, G * np, prm->multimask_output, (long long)(uintptr_t)low1, (long long)(uintptr_t)iou1}), s,
                            [&]() { return eng_decode_ex(e, ci, np, pts1, nullptr, G * np, prm->multimask_output, nullptr, 0, 0.f, low1, first_raw, iou1, nullptr, nullptr, s); }));
            for (int g = 0; g < G; ++g) e->slot_shared_valid[ci + g] = 1;      // (bookkeeping of the first-pass decode, for replays)
            gr.masks = low1; gr.ious = iou1; gr.plane_mode = first_raw ? 1 : 0;   // where K8 finds candidate k: 0 plane k, 1 plane 4 (k / 3) + 1 + k % 3, 2 plane 4 k + sel[k]
            if (prm->use_m2m) {
                // the predictor's clamp of the returned low-res logits to +-32 is applied where they are read back as the mask prompt
                float* pts2 = e->amg_pts2 + 2 * gr.k0;
                float* low2 = e->amg_low2 + gr.k0 * 4 * 65536;
                float* iou2 = e->amg_iou2 + gr.k0;
                int* sel2 = e->amg_sel + gr.k0;
                TRY(eng_graphed(e, "dec2," + key_of({ci, nm, (long long)(uintptr_t)pts2, G * nm, first_raw, (long long)(uintptr_t)low1, (long long)(uintptr_t)low2, (long long)(uintptr_t)iou2, (long long)(uintptr_t)sel2,
                                                     (long long)(e->iou_prune ? std::lround(prm->pred_iou_thresh * 1e6f) + 1 : 0)}), s,
                                [&]() { return eng_decode_ex(e, ci, nm, pts2, nullptr, G * nm, 0, low1, first_raw, 32.0f, low2, 1, iou2, nullptr, sel2, s, prm->pred_iou_thresh); }));
                for (int g = 0; g < G; ++g) e->slot_embb_valid[ci + g] = 1;       // (bookkeeping of the m2m decode, for replays)
                gr.masks = low2; gr.ious = iou2; gr.plane_mode = 2;
            }
        }
        if (use_dev) {
            // K9a: IoU filter + plane index per group; K8 for every crop (filtered candidates exit at once); K9b / K10: filters, NMS, compaction per crop
            const float* iou_all = prm->use_m2m ? e->amg_iou2 : e->amg_iou1;
            std::vector<DevCrop> dcs;
            for (const Grp& gr : groups) {
                ENG_KP(e, PC_ELEMENTWISE, 0.0, 0.0, launch_amg_plane(iou_all + gr.k0, gr.plane_mode == 2 ? e->amg_sel + gr.k0 : nullptr, gr.G * gr.nm, gr.plane_mode, prm->pred_iou_thresh,
                                                                  e->amg_idx + gr.k0, e->amg_pass + gr.k0, s));
                for (int g = 0; g < gr.G; ++g) {
                    const auto& box = crops[c0 + gr.ci + g];
                    DevCrop dc{{box[0], box[1], box[2], box[3]}, (int)(gr.k0 + (size_t)g * gr.nm), gr.nm, (int)(gr.pt0 + (size_t)g * gr.np), M};
                    dcs.push_back(dc);
                    ENG_KP(e, PC_MASK_POST, 0.0, (double)gr.nm * (65536.0 * 4 + (double)mask_words * 4),
                           launch_mask_post(gr.masks, e->amg_idx + dc.kbase, gr.nm, box[0], box[1], box[2] - box[0], box[3] - box[1], H, W, prm->mask_threshold,
                                            prm->stability_score_offset, e->amg_crop_bits + (slot_base + dc.kbase) * mask_words, e->amg_stats + dc.kbase, s, e->amg_pass + dc.kbase));
                }
            }
            ENG_HIP(e, hipMemcpyAsync(e->amg_crops_dev, dcs.data(), sizeof(DevCrop) * dcs.size(), hipMemcpyHostToDevice, s));
            ENG_HIP(e, hipMemcpyAsync(e->amg_crop_pts, crop_pts_all.data(), sizeof(float) * 2 * n_pts, hipMemcpyHostToDevice, s));
            ENG_KP(e, PC_MASK_POST, 0.0, 0.0, launch_amg_crops(e->amg_crops_dev, (int)dcs.size(), max_nm, e->amg_stats, e->amg_pass, iou_all, e->amg_crop_pts, prm->stability_score_thresh,
                                                            prm->box_nms_thresh, H, W, (int)slot_base, e->amg_tmp, e->amg_keep, e->amg_counts, e->amg_surv, e->amg_nsurv, (int)total_cand, s));
            slot_base += n_cand;
            continue;
        }
        // ---- phase 2: per-candidate scalars of the whole batch (sync 1)
        std::vector<float> h_iou_all(n_cand);
        std::vector<int> h_sel;
        ENG_HIP(e, hipMemcpyAsync(h_iou_all.data(), prm->use_m2m ? e->amg_iou2 : e->amg_iou1, sizeof(float) * n_cand, hipMemcpyDeviceToHost, s));
        if (prm->use_m2m) { h_sel.resize(n_cand); ENG_HIP(e, hipMemcpyAsync(h_sel.data(), e->amg_sel, sizeof(int) * n_cand, hipMemcpyDeviceToHost, s)); }
        { ENG_HIP(e, hipStreamSynchronize(s)); ++e->amg_last_syncs; }
        // pred_iou filter of every crop on the host, then K8 for ALL of them (one launch per crop) into one scratch
        struct CropRange { int first, count; };
        std::vector<std::vector<CropRange>> ranges(groups.size());
        h_idx.clear();               // candidate index within the batch (k0 + local)
        for (size_t gi = 0; gi < groups.size(); ++gi) {
            const Grp& gr = groups[gi];
            for (int g = 0; g < gr.G; ++g) {
                const size_t kbase = gr.k0 + (size_t)g * gr.nm;
                const int first = (int)h_idx.size();
                for (int k = 0; k < gr.nm; ++k)
                    if (!(prm->pred_iou_thresh > 0.0f) || h_iou_all[kbase + k] > prm->pred_iou_thresh) h_idx.push_back((int)(kbase + k));
                ranges[gi].push_back(CropRange{first, (int)h_idx.size() - first});
            }
        }
        const int ns_all = (int)h_idx.size();
        if (ns_all == 0) continue;
        if (e->amg_stats_cap < (size_t)ns_all) { TRY(eng_regrow(e, &e->amg_stats, (size_t)ns_all, s)); TRY(eng_regrow(e, &e->amg_idx, (size_t)ns_all, s)); e->amg_stats_cap = ns_all; }
        TRY(crop_reserve((size_t)ns_all));
        // K8 reads the chosen plane in place, relative to its group's buffer
        h_plane.resize(ns_all);
        for (size_t gi = 0; gi < groups.size(); ++gi) {
            const Grp& gr = groups[gi];
            for (const CropRange& r : ranges[gi])
                for (int j = r.first; j < r.first + r.count; ++j) {
                    const int k = h_idx[j] - (int)gr.k0;             // candidate within the group
                    h_plane[j] = gr.plane_mode == 1 ? 4 * (k / 3) + 1 + k % 3 : gr.plane_mode == 2 ? 4 * k + h_sel[h_idx[j]] : k;
                }
        }
        ENG_HIP(e, hipMemcpyAsync(e->amg_idx, h_plane.data(), sizeof(int) * ns_all, hipMemcpyHostToDevice, s));
        for (size_t gi = 0; gi < groups.size(); ++gi) {
            const Grp& gr = groups[gi];
            for (int g = 0; g < gr.G; ++g) {
                const CropRange& r = ranges[gi][g];
                if (r.count == 0) continue;
                const auto& box = crops[c0 + gr.ci + g];
                ENG_KP(e, PC_MASK_POST, 0.0, (double)r.count * (65536.0 * 4 + (double)mask_words * 4),
                       launch_mask_post(gr.masks, e->amg_idx + r.first, r.count, box[0], box[1], box[2] - box[0], box[3] - box[1], H, W, prm->mask_threshold,
                                        prm->stability_score_offset, e->amg_crop_bits + (size_t)r.first * mask_words, e->amg_stats + r.first, s));
            }
        }
        h_stats.resize(ns_all);
        ENG_HIP(e, hipMemcpyAsync(h_stats.data(), e->amg_stats, sizeof(MaskStats) * ns_all, hipMemcpyDeviceToHost, s));
        { ENG_HIP(e, hipStreamSynchronize(s)); ++e->amg_last_syncs; }                                       // sync 2
        // ---- phase 3: per-crop filtering and NMS, exactly as for unbatched crops
        for (size_t gi = 0; gi < groups.size(); ++gi) {
          const Grp& gr = groups[gi];
          const int M_ = M;
          for (int g = 0; g < gr.G; ++g) {
            const auto& box = crops[c0 + gr.ci + g];
            const float* crop_pts = crop_pts_all.data() + (gr.pt0 + (size_t)g * gr.np) * 2;
            const size_t kbase = gr.k0 + (size_t)g * gr.nm;              // first candidate of this crop in the batch buffers
            const CropRange& r = ranges[gi][g];
            const int ns = r.count;
            if (ns == 0) continue;
            const uint32_t* crop_bits = e->amg_crop_bits + (size_t)r.first * mask_words;
            std::vector<Cand> cand;
            std::vector<int> cand_src;  // index into crop_bits
            for (int k = 0; k < ns; ++k) {
                const MaskStats& st = h_stats[r.first + k];
                const float stab = (float)st.inter / (float)st.uni;  // 0/0 -> nan fails the filter like upstream
                if (prm->stability_score_thresh > 0.0f && !(stab >= prm->stability_score_thresh)) continue;
                Cand cd;
                if (st.area > 0) { cd.box[0] = (float)st.x0; cd.box[1] = (float)st.y0; cd.box[2] = (float)st.x1; cd.box[3] = (float)st.y1; }
                else { cd.box[0] = (float)box[0]; cd.box[1] = (float)box[1]; cd.box[2] = (float)box[0]; cd.box[3] = (float)box[1]; }
                // is_box_near_crop_edge(atol=20): near a crop edge that is not also an image edge
                const float cbx[4] = {(float)box[0], (float)box[1], (float)box[2], (float)box[3]};
                const float obx[4] = {0.f, 0.f, (float)W, (float)H};
                bool near = false;
                for (int q = 0; q < 4; ++q) {
                    const bool nc_ = std::fabs(cd.box[q] - cbx[q]) <= 20.0f, ni = std::fabs(cd.box[q] - obx[q]) <= 20.0f;
                    near = near || (nc_ && !ni);
                }
                if (near) continue;
                const int src = (int)((size_t)h_idx[r.first + k] - kbase);
                cd.iou = h_iou_all[kbase + src];
                cd.stab = stab;
                const int pk = src / M_;
                cd.pt[0] = crop_pts[2 * pk] + (float)box[0];
                cd.pt[1] = crop_pts[2 * pk + 1] + (float)box[1];
                for (int q = 0; q < 4; ++q) cd.crop[q] = box[q];
                cd.area = st.area;
                cd.bits_slot = 0;
                cand.push_back(cd);
                cand_src.push_back(k);
            }
            if (cand.empty()) continue;
            std::vector<float> sc(cand.size());
            for (size_t k = 0; k < cand.size(); ++k) sc[k] = cand[k].iou;
            const std::vector<int> keep = nms_host(cand, sc, prm->box_nms_thresh);
            for (int k : keep) {
                Cand cd = cand[k];
                cd.bits_slot = acc_used + keep_src.size();
                keep_src.push_back(r.first + cand_src[k]);           // position in the batch's K8 scratch
                all.push_back(cd);
            }
          }
        }
        // the batch's NMS survivors move from the K8 scratch to the accumulation buffer in ONE gather launch
        if (!keep_src.empty()) {
            TRY(acc_reserve(acc_used + keep_src.size()));
            ENG_HIP(e, hipMemcpyAsync(e->amg_idx, keep_src.data(), sizeof(int) * keep_src.size(), hipMemcpyHostToDevice, s));
            ENG_KP(e, PC_MASK_POST, 0.0, (double)keep_src.size() * mask_words * 8.0,
                   launch_gather_masks(e->amg_crop_bits, e->amg_idx, e->amg_bits + acc_used * mask_words, (int)keep_src.size(), (int64_t)mask_words, s));
            acc_used += keep_src.size();
            keep_src.clear();
        }
    }
    if (use_dev) {
        // cross-crop NMS, final order, records and the survivors' masks: one launch pair, then the slice's ONE synchronisation
        ENG_KP(e, PC_MASK_POST, 0.0, 0.0, launch_amg_final(e->amg_surv, e->amg_nsurv, (int)total_cand, nc > 1 ? 1 : 0, prm->crop_nms_thresh, max_masks, e->amg_final_slots, e->amg_meta_dev,
                                                        e->amg_count_dev, e->amg_crop_bits, out_bits_dev, (int64_t)mask_words, s));
        int nf = 0;
        unsigned long long h_pr[4] = {0, 0, 0, 0};
        ENG_HIP(e, hipMemcpyAsync(&nf, e->amg_count_dev, sizeof(int), hipMemcpyDeviceToHost, s));
        ENG_HIP(e, hipMemcpyAsync(out_meta, e->amg_meta_dev, sizeof(saber_mask_meta) * max_masks, hipMemcpyDeviceToHost, s));
        ENG_HIP(e, hipMemcpyAsync(h_pr, e->prune_counters, 32, hipMemcpyDeviceToHost, s));
        { ENG_HIP(e, hipStreamSynchronize(s)); ++e->amg_last_syncs; }
        e->amg_last_pruned = (int64_t)h_pr[0]; e->amg_last_m2m = (int64_t)h_pr[1];
        TRY(eng_check_finite_counts(e, reinterpret_cast<const unsigned int*>(h_pr + 2)));      // overflow sentinel: rides on the slice's one synchronisation
        if (nf < 0) return eng_fail(e, SABER_ERR_HIP, "amg_generate: device post-processing overflow (internal)");
        if (nf > max_masks) {
            *out_count = nf;
            return eng_fail(e, SABER_ERR_CAPACITY, "amg_generate: " + std::to_string(nf) + " masks exceed max_masks=" + std::to_string(max_masks));
        }
        *out_count = nf;
        return SABER_OK;
    }
    std::vector<int> final_order(all.size());
    std::iota(final_order.begin(), final_order.end(), 0);
    if (nc > 1 && !all.empty()) {
        std::vector<float> sc(all.size());
        for (size_t k = 0; k < all.size(); ++k) {
            const float a = (float)(all[k].crop[2] - all[k].crop[0]) * (float)(all[k].crop[3] - all[k].crop[1]);
            sc[k] = 1.0f / a;
        }
        final_order = nms_host(all, sc, prm->crop_nms_thresh);
    }
    const int nf = (int)final_order.size();
    if (nf > max_masks) {
        *out_count = nf;
        return eng_fail(e, SABER_ERR_CAPACITY, "amg_generate: " + std::to_string(nf) + " masks exceed max_masks=" + std::to_string(max_masks));
    }
    if (nf > 0) {       // survivors of the cross-crop NMS, in their final order, to the caller's buffer: one gather launch
        std::vector<int> slots(nf);
        for (int k = 0; k < nf; ++k) slots[k] = (int)all[final_order[k]].bits_slot;
        if (e->amg_stats_cap < (size_t)nf) { TRY(eng_regrow(e, &e->amg_stats, (size_t)nf, s)); TRY(eng_regrow(e, &e->amg_idx, (size_t)nf, s)); e->amg_stats_cap = nf; }
        ENG_HIP(e, hipMemcpyAsync(e->amg_idx, slots.data(), sizeof(int) * nf, hipMemcpyHostToDevice, s));
        ENG_KP(e, PC_MASK_POST, 0.0, (double)nf * mask_words * 8.0, launch_gather_masks(e->amg_bits, e->amg_idx, out_bits_dev, nf, (int64_t)mask_words, s));
    }
    for (int k = 0; k < nf; ++k) {
        const Cand& cd = all[final_order[k]];
        saber_mask_meta& m = out_meta[k];
        m.area = cd.area;
        m.bbox_xywh[0] = cd.box[0]; m.bbox_xywh[1] = cd.box[1]; m.bbox_xywh[2] = cd.box[2] - cd.box[0]; m.bbox_xywh[3] = cd.box[3] - cd.box[1];
        m.predicted_iou = cd.iou;
        m.stability_score = cd.stab;
        m.point_xy[0] = cd.pt[0]; m.point_xy[1] = cd.pt[1];
        m.crop_box_xywh[0] = (float)cd.crop[0]; m.crop_box_xywh[1] = (float)cd.crop[1];
        m.crop_box_xywh[2] = (float)(cd.crop[2] - cd.crop[0]); m.crop_box_xywh[3] = (float)(cd.crop[3] - cd.crop[1]);
    }
    unsigned long long h_pr[4] = {0, 0, 0, 0};
    ENG_HIP(e, hipMemcpyAsync(h_pr, e->prune_counters, 32, hipMemcpyDeviceToHost, s));
    { ENG_HIP(e, hipStreamSynchronize(s)); ++e->amg_last_syncs; }
    e->amg_last_pruned = (int64_t)h_pr[0]; e->amg_last_m2m = (int64_t)h_pr[1];
    TRY(eng_check_finite_counts(e, reinterpret_cast<const unsigned int*>(h_pr + 2)));
    *out_count = nf;
    return SABER_OK;
}
