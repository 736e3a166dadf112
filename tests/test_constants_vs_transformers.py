"""CPU: the model constants the product AND the oracle share (saber_amd.model_config is imported by oracle/sam2_ref.py, so a wrong value there
is invisible to every engine-vs-oracle test: VERDICT r03 weak #2) against an independent source - the installed `transformers` package's
Sam2Config defaults (its default vision backbone is the sam2.1 hiera-tiny trunk) and its ImageNet statistics."""
import pytest

transformers = pytest.importorskip("transformers")


def test_decoder_constants_match_transformers_sam2config():
    from transformers import Sam2Config
    from saber_amd import model_config as mc
    c = Sam2Config()
    md, pe = c.mask_decoder_config, c.prompt_encoder_config
    assert md.hidden_size == mc.DEC_DIM == pe.hidden_size
    assert md.num_attention_heads == mc.DEC_HEADS
    assert md.mlp_dim == mc.DEC_MLP
    assert md.num_hidden_layers == mc.DEC_DEPTH
    assert md.num_multimask_outputs + 1 == mc.NUM_MASK_TOKENS
    assert md.attention_downsample_rate == 2            # the 128-wide cross attentions of the two-way transformer (engine.hip: F.attn(.., 128, ..))
    assert md.iou_head_depth == 3 and md.iou_head_hidden_dim == 256
    assert md.dynamic_multimask_via_stability is True
    assert md.dynamic_multimask_stability_delta == mc.DYN_MULTIMASK_DELTA
    assert md.dynamic_multimask_stability_thresh == mc.DYN_MULTIMASK_THRESH
    assert pe.image_size == 1024 and pe.patch_size == 16 and pe.mask_input_channels == 16 and pe.num_point_embeddings == 4
    assert pe.layer_norm_eps == 1e-6


def test_tiny_trunk_matches_transformers_default_backbone():
    from transformers import Sam2Config
    from saber_amd.model_config import get_config
    v = Sam2Config().vision_config
    b = v.backbone_config
    t = get_config("tiny")
    assert tuple(b.blocks_per_stage) == t.stages
    assert tuple(b.global_attention_blocks) == t.global_att_blocks
    assert tuple(b.window_size_per_stage) == t.window_spec
    assert list(b.embed_dim_per_stage) == t.stage_dims
    assert list(b.num_attention_heads_per_stage) == t.stage_heads
    assert tuple(b.window_positional_embedding_background_size) == t.pos_embed_bkg
    assert b.num_query_pool_stages == t.q_pool
    assert list(b.image_size) == [t.image_size] * 2
    assert list(b.patch_kernel_size) == [7, 7] and list(b.patch_stride) == [4, 4] and list(b.patch_padding) == [3, 3]
    assert b.layer_norm_eps == t.ln_eps and b.mlp_ratio == 4.0
    assert v.fpn_hidden_size == t.fpn_dim and tuple(v.fpn_top_down_levels) == t.fpn_top_down_levels
    assert list(v.backbone_channel_list) == t.stage_dims[::-1]


def test_image_statistics_match_imagenet_defaults():
    from transformers.image_utils import IMAGENET_DEFAULT_MEAN, IMAGENET_DEFAULT_STD
    from saber_amd import model_config as mc
    assert tuple(IMAGENET_DEFAULT_MEAN) == mc.IMAGE_MEAN and tuple(IMAGENET_DEFAULT_STD) == mc.IMAGE_STD
