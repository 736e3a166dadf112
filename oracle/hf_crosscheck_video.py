"""Cross-check of oracle/sam2_video_ref.py against the independent restatement in `transformers` (Sam2VideoMemoryAttention,
Sam2VideoMemoryEncoder) with shared random weights.  CPU only; run in the authoring container:

    python -m oracle.hf_crosscheck_video
"""
import torch

from oracle import sam2_video_ref as V


def build(seed: int = 0):
    from transformers import Sam2VideoConfig
    from transformers.models.sam2_video import modeling_sam2_video as M
    cfg = Sam2VideoConfig()
    torch.manual_seed(seed)
    ma, me = M.Sam2VideoMemoryAttention(cfg).eval(), M.Sam2VideoMemoryEncoder(cfg).eval()
    with torch.no_grad():       # default initialisers leave LayerNorms / layer scales trivial: randomise everything
        for mod in (ma, me):
            for n, p in mod.named_parameters():
                p.copy_(torch.randn_like(p) * (0.5 if p.ndim == 1 else 1.0 / max(1, p[0].numel()) ** 0.5))
    return cfg, ma, me


@torch.no_grad()
def check_memory_attention(cfg, ma, n_frames: int = 2, n_ptr: int = 5, hw=(64, 64), seed: int = 1):
    g = torch.Generator().manual_seed(seed)
    HW = hw[0] * hw[1]
    curr, cpos = torch.randn(HW, 1, 256, generator=g), torch.randn(HW, 1, 256, generator=g)
    N = n_frames * HW + n_ptr
    mem, mpos = torch.randn(N, 1, 64, generator=g), torch.randn(N, 1, 64, generator=g)
    ref = ma(current_vision_features=curr, memory=mem, current_vision_position_embeddings=cpos, memory_posision_embeddings=mpos,
             num_object_pointer_tokens=n_ptr)
    W = V.from_hf_memory_attention(ma.state_dict())
    out = V.memory_attention(W, curr.transpose(0, 1), mem.transpose(0, 1), cpos.transpose(0, 1), mpos.transpose(0, 1), n_ptr,
                             n_layers=cfg.memory_attention_num_layers, rope_feat=tuple(cfg.memory_attention_rope_feat_sizes),
                             theta=cfg.memory_attention_rope_theta)
    ref = ref.reshape(HW, -1, 256).transpose(0, 1)
    return (out - ref).abs().max().item(), ref.abs().max().item()


@torch.no_grad()
def check_memory_encoder(cfg, me, seed: int = 2):
    g = torch.Generator().manual_seed(seed)
    pix, masks = torch.randn(1, 256, 64, 64, generator=g), torch.rand(1, 1, 1024, 1024, generator=g)
    rf, rp = me(pix, masks)
    W = V.from_hf_memory_encoder(me.state_dict())
    of, op = V.memory_encoder(W, pix, masks, skip_mask_sigmoid=True, n_fuser_layers=cfg.memory_fuser_num_layers)
    return (of - rf).abs().max().item(), rf.abs().max().item(), (op - rp).abs().max().item()


if __name__ == "__main__":
    cfg, ma, me = build()
    print("memory_attention  max|diff| %.3e  (scale %.2f)" % check_memory_attention(cfg, ma))
    print("memory_encoder    max|diff| %.3e  (scale %.2f), position encoding %.3e" % check_memory_encoder(cfg, me))
