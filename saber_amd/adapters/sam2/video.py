"""SAM2 video (memory) propagation on the MI355X engine: the host side of SURVEY.md 8f-1.

What `saber segment tomograms` runs after the seed slab (reference: saber/segmenters/tomo.py:81-139 -> SAM2Adapter.segment_volume,
saber/adapters/sam2/predictor.py:232-348, which drives the third-party video predictor: add_new_mask, propagate_in_video forwards
and backwards).  This module is the Python host of that loop; every arithmetic step is a HIP kernel behind the C-ABI:

    per frame        Hiera encoder + FPN neck                      saber_encode (engine.hip: the AMG path's kernels)
    memory attention 4 x [RoPE self-attention, RoPE cross-attention to the memory bank, MLP]
                     projections / scores / values / MLP           saber_k_gemm_ld (bf16 MFMA GEMM)
                     LayerNorm, RoPE, row softmax                  saber_k_layernorm, saber_k_rope, saber_k_softmax_rows
    SAM heads        prompt encoder (no point / mask prompt) + two-way transformer + upscaling on the memory-conditioned embedding
                                                                   saber_set_embed_tokens + saber_decode_points (+ saber_get_decoder_tokens)
    object pointer   3-layer MLP on the chosen mask token           saber_k_gemm_ld
    memory encoder   mask down-sampler, fuser, projection          saber_k_conv3x3s2, saber_k_layernorm (+GELU), saber_k_dwconv7, saber_k_gemm_ld, saber_k_axpy
    masks            up / down-sampling, "mask for memory"          saber_k_resize_plane, saber_k_conv4x4s4

torch is used for device allocation, host<->device copies and a handful of scalar read-backs (argmax of 3 IoU predictions, the sign
of the object score) - the decisions upstream also takes on the host.  Objects are tracked independently, as upstream does.
Settings follow the sam2.1 configs and the reference's construction (num_maskmem truncated to SAM2AdapterConfig.num_maskmem = 2,
predictor.py:28-34); the optional hole-filling CUDA extension of upstream is treated as absent (upstream then skips it with a warning).
"""
import ctypes as C
import math
import os
from typing import Dict, Iterator, List, Optional, Tuple

import numpy as np
import torch


NO_OBJ_SCORE = -1024.0
ACT_NONE, ACT_GELU, ACT_RELU = 0, 1, 2


def _bf16_bits(a: np.ndarray) -> np.ndarray:
    """fp32 -> bf16 bit patterns, round to nearest even (the engine's weight conversion)"""
    u = np.ascontiguousarray(a, dtype=np.float32).view(np.uint32).astype(np.uint64)
    r = (u + 0x7FFF + ((u >> 16) & 1)) >> 16
    return r.astype(np.uint16)


def _f16_bits(a: np.ndarray) -> np.ndarray:
    """fp32 -> IEEE half bit patterns, round to nearest even (the engine's weight conversion in SABER_PRECISION_FP16)"""
    a = np.ascontiguousarray(a, dtype=np.float32)
    if np.abs(a).max(initial=0.0) > 65504.0:
        raise ValueError("a memory-model weight exceeds the fp16 range (65504): run this checkpoint with bf16 operands")
    return a.astype(np.float16).view(np.uint16)


class _OperandLib:
    """The kernel-level C-ABI (saber_k_*) with the calling thread's 16-bit operand type set to fp16 around every call
    (include/saber_amd_kernels.h: saber_k_set_operand_type is thread-local; the engine-level entry points set it from the handle)."""

    def __init__(self, lib):
        self._lib = lib

    def __getattr__(self, name):
        fn = getattr(self._lib, name)
        if not name.startswith("saber_k_") or name in ("saber_k_last_error", "saber_k_set_operand_type"):
            return fn
        lib = self._lib

        def call(*args):
            prev = lib.saber_k_set_operand_type(1)
            try:
                return fn(*args)
            finally:
                lib.saber_k_set_operand_type(prev)
        setattr(self, name, call)
        return call


def _sine_pe_2d(side: int, num_pos_feats: int) -> np.ndarray:
    """upstream PositionEmbeddingSine(normalize=True, scale 2 pi, temperature 1e4): (side*side, 2*num_pos_feats), y half first"""
    eps, scale = 1e-6, 2 * math.pi
    yy, xx = np.meshgrid(np.arange(1, side + 1, dtype=np.float32), np.arange(1, side + 1, dtype=np.float32), indexing="ij")
    yy = yy / np.float32(side + eps) * np.float32(scale)
    xx = xx / np.float32(side + eps) * np.float32(scale)
    d = np.arange(num_pos_feats, dtype=np.float32)
    d = np.float32(10000.0) ** (2 * np.floor(d / 2) / np.float32(num_pos_feats))
    px, py = xx[:, :, None] / d, yy[:, :, None] / d
    px = np.stack((np.sin(px[:, :, 0::2]), np.cos(px[:, :, 1::2])), 3).reshape(side, side, -1)
    py = np.stack((np.sin(py[:, :, 0::2]), np.cos(py[:, :, 1::2])), 3).reshape(side, side, -1)
    return np.concatenate((py, px), 2).reshape(side * side, 2 * num_pos_feats).astype(np.float32)


def _sine_pe_1d(pos: np.ndarray, dim: int) -> np.ndarray:
    pe_dim = dim // 2
    dim_t = np.float32(10000.0) ** (2 * np.floor(np.arange(pe_dim, dtype=np.float32) / 2) / np.float32(pe_dim))
    e = pos.astype(np.float32)[:, None] / dim_t
    return np.concatenate([np.sin(e), np.cos(e)], -1).astype(np.float32)


def window_shares(k: int, world: int) -> List[Tuple[int, int]]:
    """[a, b) of the k frames of a window for each of `world` ranks: contiguous, ceil(k / world) frames each until the frames run out
    (the trailing ranks of a short window get nothing)."""
    per = -(-k // world) if k > 0 else 0
    return [(min(k, r * per), min(k, (r + 1) * per)) for r in range(world)]


def load_tomogram_frames(tomogram: np.ndarray, image_size: int = 1024, light_modality: bool = False) -> np.ndarray:
    """TomogramPreprocessor as the adapter applies it (saber/adapters/preprocessing.py:27-76 via predictor.py:98-105): min-max to [-1,1],
    per-slice resize to image_size (skimage.transform.resize(anti_aliasing=True): the identity at image_size, order-1 interpolation at
    pixel centres; on a DOWN-sampled axis preceded by the Gaussian anti-aliasing filter of sigma (factor - 1) / 2), then `2x - 1` once more
    (img_mean / img_std are None on this path, so frames span [-3, 1]).  Host glue on the whole volume, like the reference.
    Returns (Z, image_size, image_size) float32: one gray plane per frame (the reference's 3 identical channels)."""
    t = np.asarray(tomogram, dtype=np.float64)
    t = (t - t.min()) / (t.max() - t.min())
    t = t * 2 - 1
    Z, H, W = t.shape
    if (H, W) != (image_size, image_size):
        if H > image_size or W > image_size:
            # skimage.transform.resize(anti_aliasing=True) on a down-sampled axis: Gaussian of sigma = (factor - 1) / 2 (scipy's
            # gaussian_filter, boundary mode 'mirror' = skimage's default 'reflect', truncate 4), then order-1 interpolation at pixel centres
            from scipy import ndimage as ndi
            sig = (max(0.0, (H / image_size - 1) / 2), max(0.0, (W / image_size - 1) / 2))
            t = np.stack([ndi.gaussian_filter(t[z], sig, mode="mirror") for z in range(Z)])
        ys = np.clip((np.arange(image_size) + 0.5) * H / image_size - 0.5, 0, H - 1)
        xs = np.clip((np.arange(image_size) + 0.5) * W / image_size - 0.5, 0, W - 1)
        y0, x0 = np.floor(ys).astype(int), np.floor(xs).astype(int)
        y1, x1 = np.minimum(y0 + 1, H - 1), np.minimum(x0 + 1, W - 1)
        fy, fx = (ys - y0)[None, :, None], (xs - x0)[None, None, :]
        a = t[:, y0][:, :, x0] * (1 - fx) + t[:, y0][:, :, x1] * fx
        b = t[:, y1][:, :, x0] * (1 - fx) + t[:, y1][:, :, x1] * fx
        t = a * (1 - fy) + b * fy
    out = (2 * t.astype(np.float32) - 1).astype(np.float32)
    if light_modality:
        out = (out - out.min()) / (out.max() - out.min()) * 255
    return out


def load_tomogram_frames_device(tomogram: np.ndarray, lib, device, image_size: int = 1024, light_modality: bool = False) -> torch.Tensor:
    """load_tomogram_frames on the device: the two affine steps (min-max to [-1,1], then 2x - 1) commute with the bilinear resize, so the
    volume is uploaded once and one resize launch with the fused affine map v -> 4 (v - min) / (max - min) - 3 produces the (Z, 1024, 1024)
    frame stack in HBM (the host version spends ~25 ms per slice in numpy); tomograms larger than 1024 px go through the Gaussian filter first."""
    t = torch.from_numpy(np.ascontiguousarray(tomogram, dtype=np.float32)).to(device)
    Z, H, W = t.shape
    stream = C.c_void_p(torch.cuda.current_stream(device).cuda_stream)
    mn, mx = (float(v) for v in torch.aminmax(t))              # of the tomogram as loaded: the reference normalises before it resizes
    if H > image_size or W > image_size:
        # the Gaussian anti-aliasing of skimage's down-sampling resize (saber_k_gauss_mirror: scipy's gaussian_filter, mode 'mirror'), per
        # down-sampled axis; its taps sum to one, so it commutes with the affine normalisation applied by the resize launch below
        for axis, n in ((0, H), (1, W)):
            sigma = (n / image_size - 1) / 2
            if sigma > 0:
                tmp = torch.empty_like(t)
                if lib.saber_k_gauss_mirror(C.c_void_p(t.data_ptr()), C.c_void_p(tmp.data_ptr()), Z, H, W, axis, float(sigma), stream) != 0:
                    raise RuntimeError(lib.saber_k_last_error().decode())
                t = tmp
    a = 4.0 / (mx - mn)
    out = torch.empty((Z, image_size, image_size), dtype=torch.float32, device=device)
    if lib.saber_k_resize_plane(C.c_void_p(t.data_ptr()), Z, H, W, C.c_void_p(out.data_ptr()), image_size, image_size, 0, 3, a, -a * mn - 3.0, stream) != 0:
        raise RuntimeError(lib.saber_k_last_error().decode())
    if light_modality:
        lo, hi = torch.aminmax(out)
        out = (out - lo) / (hi - lo) * 255
    return out


class VideoPredictor:
    """add_new_mask / propagate_in_video of the SAM2 video predictor on one engine handle."""

    def __init__(self, engine, weights: Dict[str, np.ndarray], num_maskmem: int = 2):
        if num_maskmem > 7:
            raise ValueError("num_maskmem must be at most 7")
        self.eng, self.lib, self.dev = engine, engine.lib, engine.device
        # 16-bit operand type of the engine handle: every uint16 buffer of this class (GEMM weights, stored memories, attention operands)
        # holds that type's bit patterns
        self.f16 = getattr(engine, "operands", "bf16") == "fp16"
        if self.f16:
            self.lib = _OperandLib(engine.lib)
        if self.lib.saber_k_init(engine.device_index) != 0:
            raise RuntimeError(self.lib.saber_k_last_error().decode())
        self.num_maskmem = num_maskmem
        self.image_size = 1024
        self.shard_encodes = os.environ.get("SABER_AMD_VIDEO_SHARD", "1") != "0"
        # what the sharded window encodes ship through the all-gather: "fp32" (default; 16 MiB per frame, every rank ends up with the bits a
        # single process computes) or "op16" (the three feature arrays rounded to the handle's 16-bit operand type: 8 MiB per frame - SURVEY.md
        # 8f-1 sized the exchange that way; every rank, the encoding one included, then decodes from the ROUNDED features, so the ranks agree
        # bit for bit with each other but not with an unsharded run: +4e-3 (bf16) / +5e-4 (fp16) on the features).  Over xGMI the fp32 gather
        # of a 16-frame window is ~2 ms against ~55 ms of work per window, so precision is the default and bytes are the option.
        # Round 5: on an fp16 handle (the adapter's default) the 16-bit gather is the default - its +5e-4 is inside that mode's 1e-3 budget and the
        # window then moves 8 instead of 16 MiB per frame; bf16 handles keep fp32 (+4e-3 would double that mode's error).
        self.gather_dtype = os.environ.get("SABER_AMD_VIDEO_GATHER", "op16" if getattr(engine, "operands", "bf16") == "fp16" else "fp32")
        if self.gather_dtype not in ("fp32", "op16"):
            raise ValueError("SABER_AMD_VIDEO_GATHER must be 'fp32' or 'op16'")
        self._keep: List[torch.Tensor] = []
        W = weights
        missing = [k for k in ("memory_attention.norm.weight", "memory_encoder.out_proj.weight", "obj_ptr_proj.layers.0.weight", "maskmem_tpos_enc") if k not in W]
        if missing:
            raise ValueError(f"the video path needs the memory-model tensors of the checkpoint; missing {missing}")
        self.f32: Dict[str, torch.Tensor] = {}
        self.bf: Dict[str, torch.Tensor] = {}
        for k, v in W.items():
            if not (k.startswith(("memory_attention.", "memory_encoder.", "obj_ptr_proj.", "obj_ptr_tpos_proj.", "mask_downsample.")) or
                    k in ("maskmem_tpos_enc", "no_obj_ptr", "no_obj_embed_spatial", "no_mem_embed")):
                continue
            a = np.ascontiguousarray(v, dtype=np.float32)
            self.f32[k] = torch.from_numpy(a).to(self.dev)
            is_gemm_w = k.endswith(".weight") and a.ndim >= 2 and "dwconv" not in k and not k.startswith("mask_downsample.") and \
                ("mask_downsampler.encoder" not in k or k.endswith("encoder.12.weight"))
            if is_gemm_w:       # operands of the bf16 MFMA GEMM: [N][K] row-major, bf16 (RNE) like the engine's own weights
                self.bf[k] = torch.from_numpy((_f16_bits if self.f16 else _bf16_bits)(a.reshape(a.shape[0], -1))).to(self.dev)
        self.tpos = np.asarray(W["maskmem_tpos_enc"], dtype=np.float32)[:num_maskmem].reshape(num_maskmem, -1)      # predictor.py:31-32
        self.no_obj_ptr = np.asarray(W["no_obj_ptr"], dtype=np.float32).reshape(1, 256)
        self.no_obj_spatial = np.asarray(W["no_obj_embed_spatial"], dtype=np.float32).reshape(-1)
        self.neg_no_mem = torch.from_numpy(-np.asarray(W["no_mem_embed"], dtype=np.float32).reshape(1, 256)).to(self.dev)
        self.pos_no_mem = torch.from_numpy(np.asarray(W["no_mem_embed"], dtype=np.float32).reshape(1, 256).copy()).to(self.dev)
        self.curr_pos = torch.from_numpy(_sine_pe_2d(64, 128)).to(self.dev)                     # (4096,256)
        self.mem_pos = _sine_pe_2d(64, 32)                                                      # (4096,64)
        # spatial position + temporal encoding of a memory that is k slots away, once per model (device)
        self.mem_pos_t = [torch.from_numpy(self.mem_pos + self.tpos[k][None]).to(self.dev) for k in range(num_maskmem)]
        self.no_obj_ptr_dev = torch.from_numpy(self.no_obj_ptr).to(self.dev)
        self.no_obj_spatial_bias = (self.f32["memory_encoder.out_proj.bias"] + torch.from_numpy(self.no_obj_spatial).to(self.dev)).contiguous()
        # mask down-sampler convolutions with the weights laid out (3,3,Cin,Cout) for saber_k_conv3x3s2_t
        self.conv_t = {j: torch.from_numpy(np.ascontiguousarray(np.asarray(W[f"memory_encoder.mask_downsampler.encoder.{3 * j}.weight"], dtype=np.float32)
                                                                .transpose(2, 3, 1, 0))).to(self.dev) for j in range(4)}
        # depth-wise 7x7 weights of the fuser laid out (49, C) for saber_k_dwconv7_t
        self.dw_t = {i: torch.from_numpy(np.ascontiguousarray(np.asarray(W[f"memory_encoder.fuser.layers.{i}.dwconv.weight"], dtype=np.float32).reshape(256, 49).T)).to(self.dev)
                     for i in range(2)}
        self._pe1d_cache: Dict[tuple, torch.Tensor] = {}
        self._flash_ws = None
        self.hook = None
        self.images = None

    # ------------------------------------------------------------------ small wrappers over the C-ABI
    def _p(self, t):
        return None if t is None else C.c_void_p(t.data_ptr())

    def _s(self):
        return C.c_void_p(torch.cuda.current_stream().cuda_stream)

    def _ck(self, st):
        if st != 0:
            raise RuntimeError(f"saber_amd: {self.lib.saber_k_last_error().decode()}")

    def _new(self, *shape, dtype=torch.float32, zero=False):
        return (torch.zeros if zero else torch.empty)(shape, dtype=dtype, device=self.dev)

    def _gemm(self, A, Wb, bias, M, N, K, out_f32=None, out_bf=None, res=None, act=ACT_NONE, lda=None, ldw=None):
        """C = act(A[M,K] . W[N,K]^T + bias) (+ res); A, W bf16 (uint16 storage)"""
        self._ck(self.lib.saber_k_gemm_ld(self._p(A), lda or K, self._p(Wb), ldw or K, 1 if (ldw or K) % 64 == 0 else 0, self._p(bias), self._p(res),
                                          self._p(out_f32), self._p(out_bf), M, N, K, act, self._s()))

    def _to_bf(self, x, rows, Cn, y=None, y_rows=1):
        out = self._new(rows, Cn, dtype=torch.uint16)
        self._ck(self.lib.saber_k_add_to_bf16(self._p(x), self._p(y), y_rows, self._p(out), None, rows, Cn, self._s()))
        return out

    def _ln(self, x, key, rows, Cn, eps, bf=True, act=ACT_NONE):
        out = self._new(rows, Cn, dtype=torch.uint16 if bf else torch.float32)
        self._ck(self.lib.saber_k_layernorm(self._p(x), self._p(self.f32[key + ".weight"]), self._p(self.f32[key + ".bias"]), eps,
                                            None if bf else self._p(out), self._p(out) if bf else None, rows, Cn, act, self._s()))
        return out

    def _lin(self, x_bf, key, M, out_bf=False, res=None, act=ACT_NONE, bias=None):
        w = self.bf[key + ".weight"]
        N, K = w.shape
        out = self._new(M, N, dtype=torch.uint16 if out_bf else torch.float32)
        self._gemm(x_bf, w, self.f32[key + ".bias"] if bias is None else bias, M, N, K, None if out_bf else out, out if out_bf else None, res, act)
        return out

    # ------------------------------------------------------------------ state
    def init_state(self, frames, video_hw: Optional[Tuple[int, int]] = None):
        """frames: (Z, 1024, 1024) float32 gray planes as load_tomogram_frames (numpy) or load_tomogram_frames_device (tensor) return them"""
        assert frames.ndim == 3 and tuple(frames.shape[1:]) == (1024, 1024)
        self.images = frames
        self.num_frames = frames.shape[0]
        self.video_hw = video_hw or (1024, 1024)
        self.obj_ids: List[int] = []
        self.out: Dict[int, Dict[str, dict]] = {}
        self.temp: Dict[int, dict] = {}
        self._raw: Dict[int, torch.Tensor] = {}
        self._win = (0, 0)
        self._frames_dev = None
        self._points = {}                # (obj_id, frame_idx) -> (points in model pixels (k,2), labels (k,)): the clicks a frame has accumulated
        self._tracked = {}               # (obj_id, frame_idx) -> reverse flag of the propagation that last went over the frame (upstream: frames_tracked_per_obj)

    def reset_state(self):
        self.obj_ids, self.out, self.temp, self._raw, self._win = [], {}, {}, {}, (0, 0)
        self._points, self._tracked = {}, {}

    def _frame(self, t: int, reverse: bool = False) -> torch.Tensor:
        """RAW top-level features of frame t (without no_mem_embed) as (4096,256) fp32 row-major tokens.  Frames are encoded a WINDOW at a
        time: the per-frame Hiera passes are independent, so the next max_images frames in the direction of travel
        go through the encoder as one batched pass (the device copy of the frame stack is a (Z*1024, 1024) image whose crop boxes are
        the frames) and stay in the engine's slots, where the decoder finds their high-resolution features."""
        if t in self._raw:
            return self._raw[t]
        B = self.eng.max_images
        t0, t1 = (max(0, t - B + 1), t + 1) if reverse else (t, min(self.num_frames, t + B))
        if self._frames_dev is None:
            self._frames_dev = (self.images.to(self.dev, dtype=torch.float32).contiguous() if isinstance(self.images, torch.Tensor) else
                                torch.from_numpy(np.ascontiguousarray(self.images, dtype=np.float32)).to(self.dev))
        k = t1 - t0
        world, rank = self._ranks()
        if world > 1:
            self._encode_window_sharded(t0, t1, world, rank)
        else:
            self.eng.encode(self._frames_dev[t0:t1].view(k * 1024, 1024), crop_boxes=[[0, i * 1024, 1024, (i + 1) * 1024] for i in range(k)],
                            slot0=0, normalised_grey=True)
        self._raw, self._win = {}, (t0, t1)
        for i in range(k):
            emb = self._new(4096, 256)
            self.eng._check(self.lib.saber_get_embed_tokens(self.eng.h, i, self._p(emb), self._s()))
            raw = self._new(4096, 256)
            self._ck(self.lib.saber_k_add_to_bf16(self._p(emb), self._p(self.neg_no_mem), 1, None, self._p(raw), 4096, 256, self._s()))
            self._raw[t0 + i] = raw
        return self._raw[t]

    def _ranks(self) -> Tuple[int, int]:
        """(world size, rank) the window encodes shard over: the default process group when one is initialised with more than one rank and
        sharding is not switched off (shard_encodes=False / SABER_AMD_VIDEO_SHARD=0).  Every rank must then run the same tracking calls on
        the same volume: the chain itself is replicated, only the Hiera passes are divided."""
        import torch.distributed as dist
        if not self.shard_encodes or not dist.is_available() or not dist.is_initialized():
            return 1, 0
        return dist.get_world_size(), dist.get_rank()

    def _encode_window_sharded(self, t0: int, t1: int, world: int, rank: int):
        """the frames of a window are dealt to the ranks in contiguous shares (window_shares), each rank encodes its
        share into slots 0.., exports them, one all-gather per feature array (RCCL; staged through host memory under gloo) brings every
        frame to every rank, and the gathered features are imported into slots 0..k-1 - bit-identical to encoding all k here, because the
        encoder's per-image results do not depend on the batch they ran in (tests/test_gpu_graphs.py)."""
        import torch.distributed as dist
        k = t1 - t0
        shares = window_shares(k, world)
        per = shares[0][1] - shares[0][0]
        a, b = shares[rank]
        n = b - a
        if n > 0:
            self.eng.encode(self._frames_dev[t0 + a:t0 + b].view(n * 1024, 1024), crop_boxes=[[0, i * 1024, 1024, (i + 1) * 1024] for i in range(n)],
                            slot0=0, normalised_grey=True)
        sizes = (4096 * 256, 16384 * 64, 65536 * 32)
        mine = [torch.zeros((per, sz), dtype=torch.float32, device=self.dev) for sz in sizes]
        self.eng._check(self.lib.saber_export_slots(self.eng.h, 0, n, self._p(mine[0]), self._p(mine[1]), self._p(mine[2]), self._s()))
        every = [torch.empty((world * per, sz), dtype=torch.float32, device=self.dev) for sz in sizes]
        on_device = dist.get_backend() == "nccl"
        t16 = None if self.gather_dtype == "fp32" else (torch.float16 if self.f16 else torch.bfloat16)
        for src, dst in zip(mine, every):
            if t16 is not None:                      # 16-bit exchange: the same bytes through a byte view (gloo has no 16-bit collectives)
                src16 = src.to(t16).view(torch.uint8)
                dst16 = torch.empty((dst.shape[0], dst.shape[1] * 2), dtype=torch.uint8, device=self.dev)
                if on_device:
                    dist.all_gather_into_tensor(dst16, src16)
                else:
                    host = torch.empty(dst16.shape, dtype=torch.uint8)
                    dist.all_gather_into_tensor(host, src16.cpu())
                    dst16.copy_(host)
                dst.copy_(dst16.view(t16))
            elif on_device:
                dist.all_gather_into_tensor(dst, src)
            else:
                host = torch.empty(dst.shape, dtype=torch.float32)
                dist.all_gather_into_tensor(host, src.cpu())
                dst.copy_(host)
        # rank r's share sits at rows [r * per, r * per + its length) = frames [r * per, ...) of the window: already in slot order
        self.eng._check(self.lib.saber_import_slots(self.eng.h, 0, k, self._p(every[0]), self._p(every[1]), self._p(every[2]), self._s()))

    def _slot(self, t: int) -> int:
        assert self._win[0] <= t < self._win[1]
        return t - self._win[0]

    # ------------------------------------------------------------------ SAM heads on a given (4096,256) embedding
    def _sam_heads(self, embed_tokens: torch.Tensor, mask_in: Optional[torch.Tensor], multimask: bool, slot: int = 0,
                   point: Optional[Tuple[float, float]] = None, label: int = -1, points=None, labels=None):
        """returns (low (256,256) device fp32, obj logit float, obj_ptr (1,256) device fp32); slot: the engine slot that holds the frame;
        point / label: one click in the model's 1024-px frame (None: upstream's padding point with label -1); points (k,2) / labels (k,),
        k > 1: several clicks / a box's corners - decoded in the engine's exact precision mode (saber_decode_prompts), which the handle must
        have been created with (Engine(..., precision="exact"); the production precision is restored afterwards)"""
        self.eng._check(self.lib.saber_set_embed_tokens(self.eng.h, slot, self._p(embed_tokens), self._s()))
        if points is not None and len(points) > 1:
            was = self.eng.precision
            if not getattr(self.eng, "has_exact", was == "exact"):
                raise NotImplementedError("several points / a box per prompt make more than 8 decoder tokens: they are decoded in the exact precision mode - "
                                          "create the Engine with precision='exact' (it can run in bf16 between such calls: Engine.set_precision)")
            pts = torch.as_tensor(np.asarray(points, np.float32).reshape(1, -1, 2)).to(self.dev)
            lab = torch.as_tensor(np.asarray(labels, np.int32).reshape(1, -1)).to(self.dev)
            self.eng.set_precision("exact")
            try:
                low, iou, obj = self.eng.decode_prompts(pts, lab, slot=slot, multimask=multimask, mask_input=mask_in)
            finally:
                self.eng.set_precision(was)
        else:
            if points is not None and len(points) == 1:
                point, label = (float(points[0][0]), float(points[0][1])), int(labels[0])
            pts = torch.zeros(1, 2, device=self.dev) if point is None else torch.tensor([[float(point[0]), float(point[1])]], dtype=torch.float32, device=self.dev)
            lab = torch.full((1,), -1 if point is None else int(label), dtype=torch.int32, device=self.dev)
            low, iou, obj = self.eng.decode_points(pts, slot=slot, multimask=multimask, mask_input=mask_in, labels=lab)
        toks = self._new(8, 256)
        self.eng._check(self.lib.saber_get_decoder_tokens(self.eng.h, 1, self._p(toks), self._s()))
        host = torch.cat([obj.reshape(-1)[:1], iou[0].reshape(-1)]).cpu()       # the one synchronisation of a tracked frame: object score + IoUs
        self.eng.check_finite()          # overflow sentinel of the 16-bit modes (the stream has just drained: a 16-byte copy); SaberRangeError, never NaN masks
        obj_v = float(host[0])
        if self.hook is not None:
            self.hook(obj_v)
        if multimask:
            best = int(torch.argmax(host[1:]))
            low_b, tok = low[0, best], toks[3 + best:4 + best]
        else:
            low_b, tok = low[0, 0], toks[2:3]
        appearing = obj_v > 0
        if appearing:
            h = self._lin(self._to_bf(tok, 1, 256), "obj_ptr_proj.layers.0", 1, out_bf=True, act=ACT_RELU)
            h = self._lin(h, "obj_ptr_proj.layers.1", 1, out_bf=True, act=ACT_RELU)
            ptr = self._lin(h, "obj_ptr_proj.layers.2", 1)                       # (1,256) fp32, stays on the device
        else:
            low_b = torch.full((256, 256), NO_OBJ_SCORE, device=self.dev)
            ptr = self.no_obj_ptr_dev
        return low_b.contiguous(), obj_v, ptr

    def _resize(self, x, H, W, Ho, Wo, antialias=0, post=0, a=0.0, c=0.0):
        out = self._new(Ho, Wo)
        self._ck(self.lib.saber_k_resize_plane(self._p(x), 1, H, W, self._p(out), Ho, Wo, antialias, post, a, c, self._s()))
        return out

    # ------------------------------------------------------------------ memory encoder
    def _encode_memory(self, raw: torch.Tensor, mask_for_mem: torch.Tensor, appearing: bool):
        """raw (4096,256) fp32 frame features, mask_for_mem (1024,1024) fp32 already scaled (sigmoid or binary, * 20 - 10).
        Returns the spatial memory (4096,64) as stored: bf16 bits."""
        x, H, Cin = mask_for_mem, 1024, 1
        pre = "memory_encoder.mask_downsampler.encoder."
        for j in range(4):
            Cout = Cin * 4
            y = self._new((H // 2) * (H // 2), Cout)
            self._ck(self.lib.saber_k_conv3x3s2_t(self._p(x), H, H, Cin, self._p(self.conv_t[j]), self._p(self.f32[f"{pre}{3 * j}.bias"]), Cout, self._p(y), self._s()))
            H //= 2
            x = self._ln(y, f"{pre}{3 * j + 1}", H * H, Cout, 1e-6, bf=(j == 3), act=ACT_GELU)
            Cin = Cout
        m = self._lin(x, pre + "12", 4096)                                                  # 1x1 conv 256 -> 256
        p = self._lin(self._to_bf(raw, 4096, 256), "memory_encoder.pix_feat_proj", 4096, res=m)
        for i in range(2):
            f = f"memory_encoder.fuser.layers.{i}."
            h = self._new(4096, 256)
            self._ck(self.lib.saber_k_dwconv7_t(self._p(p), 64, 64, 256, self._p(self.dw_t[i]), self._p(self.f32[f + "dwconv.bias"]), self._p(h), self._s()))
            hn = self._ln(h, f + "norm", 4096, 256, 1e-6)
            h1 = self._lin(hn, f + "pwconv1", 4096, out_bf=True, act=ACT_GELU)
            h2 = self._lin(h1, f + "pwconv2", 4096)
            p2 = self._new(4096, 256)
            self._ck(self.lib.saber_k_axpy(self._p(p), self._p(h2), self._p(self.f32[f + "gamma"]), 1.0, 4096, 256, self._p(p2), self._s()))
            p = p2
        # no_obj_embed_spatial on frames where the object is predicted absent
        bias = self.f32["memory_encoder.out_proj.bias"] if appearing else self.no_obj_spatial_bias
        return self._lin(self._to_bf(p, 4096, 256), "memory_encoder.out_proj", 4096, out_bf=True, bias=bias)

    # ------------------------------------------------------------------ prompts
    @torch.inference_mode()
    def add_new_mask(self, frame_idx: int, obj_id: int, mask: np.ndarray):
        """upstream add_new_mask: the mask becomes this frame's output as it is (+-10 logits), its object pointer comes from the SAM heads
        prompted with the mask; the memory of the frame is encoded by the preflight of the next propagate_in_video."""
        if self.images is None:
            raise RuntimeError("call init_state() first")
        if obj_id not in self.obj_ids:
            self.obj_ids.append(obj_id)
            self.out[obj_id] = {"cond": {}, "non_cond": {}}
            self.temp[obj_id] = {}
        m = np.ascontiguousarray(np.squeeze(np.asarray(mask)), dtype=np.float32)
        md = torch.from_numpy(m).to(self.dev)
        if m.shape != (1024, 1024):
            md = self._resize(md, m.shape[0], m.shape[1], 1024, 1024, antialias=1, post=4, a=0.5)
        appearing = bool((md > 0).any().item())
        raw = self._frame(frame_idx)
        high = self._resize(md, 1024, 1024, 1024, 1024, post=3, a=20.0, c=-10.0)
        low = self._resize(high, 1024, 1024, 256, 256, antialias=1)
        mdown = self._new(256, 256)
        self._ck(self.lib.saber_k_conv4x4s4(self._p(md), 1024, 1024, self._p(self.f32["mask_downsample.weight"]), self._p(self.f32["mask_downsample.bias"]), self._p(mdown), self._s()))
        _, _, ptr = self._sam_heads(raw, mdown.view(1, 256, 256), multimask=False, slot=self._slot(frame_idx))
        # the pointer has been through the decoder's own object score (_forward_sam_heads); upstream then blends once more with the
        # appearance the MASK itself says (_use_mask_as_output)
        tok_ptr = ptr if appearing else self.no_obj_ptr_dev
        self.temp[obj_id][frame_idx] = {"pred_masks": low, "obj_ptr": tok_ptr, "obj": 10.0 if appearing else -10.0, "mem": None, "raw": raw}
        return frame_idx, list(self.obj_ids), low

    @torch.inference_mode()
    def add_new_points_or_box(self, frame_idx: int, obj_id: int, points=None, labels=None, clear_old_points: bool = True,
                              normalize_coords: bool = True, box=None):
        """upstream SAM2VideoPredictor.add_new_points_or_box on a frame that has not been tracked yet: the SAM heads run on the frame's own
        features (+ no_mem_embed: an initial conditioning frame sees no memory) with the accumulated clicks as the point prompt and the
        frame's previous output, if any, as the mask prompt (clamped to +-32).  A box is its two corners with labels 2 / 3 in front of the
        clicks (only with clear_old_points, as upstream); clicks accumulate over calls unless clear_old_points.  At most one point =>
        multimask output, the best mask by predicted IoU becomes the frame's output (the production bf16 decoder: 8 tokens per prompt);
        more points (or a box) => the single-mask output with dynamic selection, decoded in the engine's exact precision mode (the handle
        must have been created with precision="exact").  On a frame a propagation has already gone over the call is a CORRECTION, as upstream:
        the frame's features are conditioned on the memory bank (in the direction the frame was tracked), its previous output is the mask
        prompt, and the new output replaces the frame's tracked (non-conditioning) output at the next propagation's preflight."""
        if self.images is None:
            raise RuntimeError("call init_state() first")
        if (points is not None) != (labels is not None):
            raise ValueError("points and labels must be provided together")
        if points is None and box is None:
            raise ValueError("at least one of points or box must be provided as input")
        pts = np.zeros((0, 2), np.float32) if points is None else np.asarray(points, dtype=np.float32).reshape(-1, 2)
        lab = np.zeros((0,), np.int32) if labels is None else np.asarray(labels).reshape(-1).astype(np.int32)
        if len(pts) != len(lab):
            raise ValueError("points and labels must have the same length")
        if box is not None:
            if not clear_old_points:
                raise ValueError("cannot add box without clearing old points, since box prompt must be provided before any point prompt "
                                 "(please use clear_old_points=True instead)")
            pts = np.concatenate([np.asarray(box, np.float32).reshape(2, 2), pts], 0)
            lab = np.concatenate([np.array([2, 3], np.int32), lab], 0)
        if obj_id not in self.obj_ids:
            self.obj_ids.append(obj_id)
            self.out[obj_id] = {"cond": {}, "non_cond": {}}
            self.temp[obj_id] = {}
        Hv, Wv = self.video_hw
        xy = pts / np.array([Wv, Hv], np.float32) if normalize_coords else pts
        xy = (xy * np.float32(self.image_size)).astype(np.float32)
        held = self.__dict__.setdefault("_points", {})
        old = None if clear_old_points else held.get((obj_id, frame_idx))
        if old is not None:
            xy, lab = np.concatenate([old[0], xy], 0), np.concatenate([old[1], lab], 0)
        if len(xy) > 9:
            raise NotImplementedError("at most 9 points per object and frame (16 decoder tokens)")
        held[(obj_id, frame_idx)] = (xy, lab)
        # a frame a propagation has already gone over is CORRECTED (upstream: is_init_cond_frame False): features conditioned on the memory bank
        # in the direction the frame was tracked, and the new output stays a non-conditioning one (add_all_frames_to_correct_as_cond False)
        is_init = (obj_id, frame_idx) not in self._tracked
        prev = self.temp[obj_id].get(frame_idx) or self.out[obj_id]["cond"].get(frame_idx) or self.out[obj_id]["non_cond"].get(frame_idx)
        mask_in = prev["pred_masks"].clamp(-32.0, 32.0).view(1, 256, 256).contiguous() if prev is not None else None
        reverse = False if is_init else self._tracked[(obj_id, frame_idx)]
        raw = self._frame(frame_idx, reverse)
        if is_init:
            emb = self._new(4096, 256)
            self._ck(self.lib.saber_k_add_to_bf16(self._p(raw), self._p(self.pos_no_mem), 1, None, self._p(emb), 4096, 256, self._s()))
        else:
            emb = self._memory_conditioned(obj_id, frame_idx, raw, reverse)
        low, obj_v, ptr = self._sam_heads(emb, mask_in, multimask=len(xy) <= 1, slot=self._slot(frame_idx), points=xy, labels=lab)
        self.temp[obj_id][frame_idx] = {"pred_masks": low, "obj_ptr": ptr, "obj": obj_v, "mem": None, "raw": raw, "is_cond": is_init}
        return frame_idx, list(self.obj_ids), self._resize(low, 256, 256, Hv, Wv)[None, None]

    NO_OBJ_SCORE = -1024.0

    def _frame_output(self, frame_idx: int):
        """(frame_idx, obj_ids, video-resolution mask logits (n_obj,1,Hv,Wv)) of the frame as the state holds it now: an object without an
        output on the frame reads NO_OBJ_SCORE, like upstream's consolidated outputs."""
        Hv, Wv = self.video_hw
        outs = []
        for oid in self.obj_ids:
            o = self.temp[oid].get(frame_idx) or self.out[oid]["cond"].get(frame_idx) or self.out[oid]["non_cond"].get(frame_idx)
            if o is None:
                outs.append(torch.full((Hv, Wv), self.NO_OBJ_SCORE, dtype=torch.float32, device=self.dev))
            else:
                outs.append(self._resize(o["pred_masks"], 256, 256, Hv, Wv))
        masks = torch.stack(outs, 0)[:, None] if outs else torch.empty((0, 1, Hv, Wv), dtype=torch.float32, device=self.dev)
        return frame_idx, list(self.obj_ids), masks

    @torch.inference_mode()
    def clear_all_prompts_in_frame(self, frame_idx: int, obj_id: int, need_output: bool = True):
        """upstream SAM2VideoPredictor.clear_all_prompts_in_frame: the object's inputs on the frame go; if the frame was a conditioning frame
        its output is kept as a plain tracked (non-conditioning) output, since it no longer receives inputs."""
        if obj_id not in self.obj_ids:
            raise RuntimeError(f"Cannot clear prompts for object id {obj_id}: it does not exist. All existing object ids: {self.obj_ids}.")
        self.temp[obj_id].pop(frame_idx, None)
        self.__dict__.setdefault("_points", {}).pop((obj_id, frame_idx), None)
        out = self.out[obj_id]["cond"].pop(frame_idx, None)
        if out is not None:
            self.out[obj_id]["non_cond"][frame_idx] = out
        if not need_output:
            return None
        return self._frame_output(frame_idx)

    @torch.inference_mode()
    def remove_object(self, obj_id: int, strict: bool = False, need_output: bool = True):
        """upstream SAM2VideoPredictor.remove_object: the object leaves the tracking state at any time.  Returns (remaining object ids,
        [(frame_idx, video-resolution masks of the remaining objects) for every frame the removed object had received inputs on])."""
        if obj_id not in self.obj_ids:
            if not strict:
                return list(self.obj_ids), []
            raise RuntimeError(f"Cannot remove object id {obj_id}: it does not exist. All existing object ids: {self.obj_ids}.")
        input_frames = sorted(set(self.temp[obj_id]) | set(self.out[obj_id]["cond"]))
        self.obj_ids.remove(obj_id)
        self.out.pop(obj_id, None)
        self.temp.pop(obj_id, None)
        self._points = {k: v for k, v in self.__dict__.get("_points", {}).items() if k[0] != obj_id}
        self._tracked = {k: v for k, v in self._tracked.items() if k[0] != obj_id}
        if not need_output or not self.obj_ids:
            return list(self.obj_ids), []
        return list(self.obj_ids), [(t, self._frame_output(t)[2]) for t in input_frames]

    def _preflight(self):
        for oid in self.obj_ids:
            for t, o in self.temp[oid].items():
                if o["mem"] is None:
                    mfm = self._resize(o["pred_masks"], 256, 256, 1024, 1024, antialias=0, post=2, a=20.0, c=-10.0)     # binarised: is_mask_from_pts
                    o["mem"] = self._encode_memory(o["raw"], mfm, o["obj"] > 0)
                    o["raw"] = None
                if o.pop("is_cond", True):
                    self.out[oid]["cond"][t] = o
                else:
                    self.out[oid]["non_cond"][t] = o
            for t in self.out[oid]["cond"]:              # upstream keeps the two dictionaries disjoint (a corrected conditioning frame keeps its old output)
                self.out[oid]["non_cond"].pop(t, None)
            self.temp[oid] = {}
            if not self.out[oid]["cond"]:
                raise RuntimeError("No input points or masks are provided for any object; please add inputs first.")

    # ------------------------------------------------------------------ memory attention
    def _memory_conditioned(self, oid: int, t: int, raw: torch.Tensor, reverse: bool) -> torch.Tensor:
        st = self.out[oid]
        mems, pos = [], []
        for tc, o in st["cond"].items():
            mems.append(o["mem"]); pos.append(self.mem_pos_t[self.num_maskmem - 1])
        for t_pos in range(1, self.num_maskmem):
            t_rel = self.num_maskmem - t_pos
            o = st["non_cond"].get(t + t_rel if reverse else t - t_rel)
            if o is None:
                continue
            mems.append(o["mem"]); pos.append(self.mem_pos_t[self.num_maskmem - t_pos - 1])
        max_ptrs = min(self.num_frames, 16)
        sign = -1 if reverse else 1
        offs, ptrs = [], []
        for tc, o in st["cond"].items():
            if (tc >= t) if reverse else (tc <= t):
                offs.append((t - tc) * sign); ptrs.append(o["obj_ptr"])
        for d in range(1, max_ptrs):
            tt = t + d if reverse else t - d
            if tt < 0 or tt >= self.num_frames:
                break
            o = st["non_cond"].get(tt)
            if o is not None:
                offs.append(d); ptrs.append(o["obj_ptr"])
        n_spatial = 4096 * len(mems)
        n_ptr_tok = 4 * len(ptrs)
        Nk = n_spatial + n_ptr_tok
        Nkp = (Nk + 63) // 64 * 64
        # memory tokens (bf16 as stored) and their position encodings (fp32): [spatial memories ..., pointer tokens], assembled on the device
        mem_bf = self._new(Nkp, 64, dtype=torch.uint16, zero=True)
        pos_d = self._new(Nk, 64)
        for i, (mm, pp) in enumerate(zip(mems, pos)):
            mem_bf[4096 * i:4096 * (i + 1)].copy_(mm)
            pos_d[4096 * i:4096 * (i + 1)].copy_(pp)
        if ptrs:
            P = torch.cat(ptrs, 0).contiguous()                                                # (n,256) fp32, device
            # pointer tokens enter the bank in fp32 upstream; the GEMM operand is bf16 either way
            mem_bf[n_spatial:Nk].copy_(self._to_bf(P, len(ptrs), 256).view(4 * len(ptrs), 64))
            key = (tuple(offs), max_ptrs)
            pe_bf = self._pe1d_cache.get(key)
            if pe_bf is None:
                pe = _sine_pe_1d(np.asarray(offs, np.float32) / np.float32(max_ptrs - 1), 256)
                pe_bf = self._to_bf(torch.from_numpy(pe).to(self.dev), len(ptrs), 256)
                if len(self._pe1d_cache) < 256:
                    self._pe1d_cache[key] = pe_bf
            pe_d = self._lin(pe_bf, "obj_ptr_tpos_proj", len(ptrs))                           # (n,64)
            pos_d[n_spatial:].copy_(pe_d.repeat_interleave(4, dim=0))
        mem_f = self._new(Nk, 64)
        self._ck(self.lib.saber_k_bf16_to_f32(self._p(mem_bf), Nk * 64, self._p(mem_f), self._s()))
        kin_bf = self._to_bf(mem_f, Nk, 64, pos_d, Nk)                                          # bf16(memory + position)
        # ---- 4 layers
        x = self._new(4096, 256)
        self._ck(self.lib.saber_k_axpy(self._p(raw), self._p(self.curr_pos), None, 0.1, 4096, 256, self._p(x), self._s()))
        scale = 1.0 / 16.0

        def attend(q_bf, k_bf, v_bf, n_keys, bv):
            # flash-style (csrc/flash256.hip): no materialised (4096, n_keys) scores; the value bias is added after the product (softmax
            # rows sum to 1)
            O = self._new(4096, 256, dtype=torch.uint16)
            if self._flash_ws is None:
                self._flash_ws = self._new(64 * 8 * 64 * 258)
            self._ck(self.lib.saber_k_flash256(self._p(q_bf), self._p(k_bf), self._p(v_bf), 4096, n_keys, scale, self._p(bv), self._p(O),
                                               self._p(self._flash_ws), self._flash_ws.numel(), self._s()))
            return O

        def rope(x_f32, rows, n_rot):
            out = self._new(rows, 256, dtype=torch.uint16)
            self._ck(self.lib.saber_k_rope(self._p(x_f32), rows, n_rot, 256, 64, 10000.0, None, self._p(out), self._s()))
            return out

        for i in range(4):
            L = f"memory_attention.layers.{i}."
            tb = self._ln(x, L + "norm1", 4096, 256, 1e-5)
            q = rope(self._lin(tb, L + "self_attn.q_proj", 4096), 4096, 4096)
            k = rope(self._lin(tb, L + "self_attn.k_proj", 4096), 4096, 4096)
            v = self._new(4096, 256, dtype=torch.uint16)
            self._gemm(tb, self.bf[L + "self_attn.v_proj.weight"], None, 4096, 256, 256, out_bf=v)
            O = attend(q, k, v, 4096, self.f32[L + "self_attn.v_proj.bias"])
            x = self._lin(O, L + "self_attn.out_proj", 4096, res=x)
            tb = self._ln(x, L + "norm2", 4096, 256, 1e-5)
            q = rope(self._lin(tb, L + "cross_attn_image.q_proj", 4096), 4096, 4096)
            kf = self._new(Nkp, 256, zero=True)
            self._gemm(kin_bf, self.bf[L + "cross_attn_image.k_proj.weight"], self.f32[L + "cross_attn_image.k_proj.bias"], Nk, 256, 64, out_f32=kf)
            k = rope(kf, Nkp, n_spatial)
            v = self._new(Nk, 256, dtype=torch.uint16)
            self._gemm(mem_bf, self.bf[L + "cross_attn_image.v_proj.weight"], None, Nk, 256, 64, out_bf=v)
            O = attend(q, k, v, Nk, self.f32[L + "cross_attn_image.v_proj.bias"])
            x = self._lin(O, L + "cross_attn_image.out_proj", 4096, res=x)
            tb = self._ln(x, L + "norm3", 4096, 256, 1e-5)
            h = self._lin(tb, L + "linear1", 4096, out_bf=True, act=ACT_RELU)
            x = self._lin(h, L + "linear2", 4096, res=x)
        return self._ln(x, "memory_attention.norm", 4096, 256, 1e-5, bf=False)

    # ------------------------------------------------------------------ tracking
    def _track(self, oid: int, t: int, reverse: bool) -> dict:
        raw = self._frame(t, reverse)
        cond = self._memory_conditioned(oid, t, raw, reverse)
        low, obj_v, ptr = self._sam_heads(cond, None, multimask=True, slot=self._slot(t))
        mfm = self._resize(low, 256, 256, 1024, 1024, antialias=0, post=1, a=20.0, c=-10.0)       # sigmoid(high-res logits) * 20 - 10
        mem = self._encode_memory(raw, mfm, obj_v > 0)
        return {"pred_masks": low, "obj_ptr": ptr, "obj": obj_v, "mem": mem}

    @torch.inference_mode()
    def propagate_in_video(self, start_frame_idx: int, max_frame_num_to_track: Optional[int] = None, reverse: bool = False) -> Iterator:
        """yields (frame_idx, obj_ids, video_res_mask_logits (n_obj, 1, Hv, Wv) device fp32) like upstream"""
        self._preflight()
        n = self.num_frames
        if max_frame_num_to_track is None:
            max_frame_num_to_track = n
        if reverse:
            end = max(start_frame_idx - max_frame_num_to_track, 0)
            order = range(start_frame_idx, end - 1, -1) if start_frame_idx > 0 else []
        else:
            end = min(start_frame_idx + max_frame_num_to_track, n - 1)
            order = range(start_frame_idx, end + 1)
        Hv, Wv = self.video_hw
        for t in order:
            outs = []
            for oid in self.obj_ids:
                self._tracked[(oid, t)] = reverse
                if t in self.out[oid]["cond"]:
                    low = self.out[oid]["cond"][t]["pred_masks"]
                else:
                    o = self._track(oid, t, reverse)
                    self.out[oid]["non_cond"][t] = o
                    low = o["pred_masks"]
                outs.append(self._resize(low, 256, 256, Hv, Wv))
            yield t, list(self.obj_ids), torch.stack(outs, 0)[:, None]
