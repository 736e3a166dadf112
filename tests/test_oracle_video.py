"""Oracle of the next SURVEY 8(f) row (SAM2 video propagation): the memory attention and memory encoder restatements in
oracle/sam2_video_ref.py against the independent `transformers` modules with shared random weights (CPU).  Tolerances: fp32
reassociation only - 1e-5 absolute on outputs of magnitude 3-7."""
import pytest
import torch

pytest.importorskip("transformers")


@pytest.fixture(scope="module")
def hf():
    try:
        from oracle import hf_crosscheck_video as X
        return X, X.build(0)
    except ImportError as e:      # a transformers build without the sam2_video model
        pytest.skip(str(e))


def test_memory_attention_matches_hf(hf):
    X, (cfg, ma, me) = hf
    for n_frames, n_ptr in ((1, 0), (2, 5)):
        d, scale = X.check_memory_attention(cfg, ma, n_frames=n_frames, n_ptr=n_ptr)
        assert scale > 1.0 and d <= 1e-5, (n_frames, n_ptr, d)


def test_memory_encoder_matches_hf(hf):
    X, (cfg, ma, me) = hf
    d, scale, dpos = X.check_memory_encoder(cfg, me)
    assert scale > 1.0 and d <= 1e-5 and dpos <= 1e-6


def test_rope_is_a_rotation():
    from oracle import sam2_video_ref as V
    cos, sin = V.rope_table(8, 8, 32)
    x = torch.randn(2, 64, 32)
    y = V.rope_rotate(x, cos, sin)
    assert torch.allclose(y.norm(dim=-1), x.norm(dim=-1), atol=1e-5)          # norms of the rotated pairs are kept
    assert torch.allclose(y[:, 0], x[:, 0])                                     # position (0, 0): identity
    # relative property along x: <R(p) q, R(p') k> depends on p - p' only (first half of the channels)
    q, k = torch.zeros(64, 32), torch.zeros(64, 32)
    q[:, :16], k[:, :16] = torch.randn(16), torch.randn(16)
    s = V.rope_rotate(q, cos, sin) @ V.rope_rotate(k, cos, sin).T
    assert torch.allclose(s[1, 3], s[2, 4], atol=1e-4) and torch.allclose(s[9, 11], s[1, 3], atol=1e-4)


def test_tracking_loop_matches_transformers_sam2_video_model():
    """The whole tracking loop of the oracle (mask prompt, forward + backward propagation, num_maskmem = 2: which memories and object
    pointers a frame attends to, their temporal encodings, pointer projection, object-score gating, memory encoding) against the independent
    `transformers` Sam2VideoModel + Sam2VideoInferenceSession with shared weights (oracle/hf_crosscheck_tracking.py).  A semantic slip
    (wrong temporal index, a pointer too many, a missed gate) moves the logits by O(1); agreeing implementations differ by the bf16 storage
    of the memories meeting different fp32 summation orders: measured 2e-4 of the logit scale, object scores to 2e-4 absolute."""
    pytest.importorskip("transformers")
    from oracle import hf_crosscheck_tracking as H
    assert H.crosscheck("tiny", Z=4, start=1, verbose=False) < 1e-3
