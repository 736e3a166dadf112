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


def test_frames_larger_than_the_model_are_antialiased_like_skimage():
    """Tomogram slices above 1024 px (saber/adapters/preprocessing.py:21: skimage.transform.resize(img, (1024, 1024), anti_aliasing=True)).
    skimage is absent, so the recipe is restated from its published source: Gaussian of sigma (factor - 1) / 2 per down-sampled axis
    (scipy.ndimage.gaussian_filter, mode 'mirror', truncate 4), then order-1 interpolation at pixel centres.  Known answers checked by
    hand: a constant plane stays constant; factor 2 gives sigma 0.5, radius 2, taps exp(-2 k^2) / sum = [2.6e-4, 0.10645, 0.78657, ...]
    and output pixel o averages filtered pixels 2 o and 2 o + 1; the product's host glue equals the oracle."""
    import numpy as np
    from oracle import sam2_video_ref as V
    from saber_amd.adapters.sam2.video import load_tomogram_frames
    rng = np.random.default_rng(0)
    tomo = rng.normal(0, 1, (2, 96, 80)).astype(np.float32)
    ref = V.load_tomogram_frames(tomo, image_size=48)[:, 0].numpy()            # H factor 2 (sigma 0.5), W factor 5/3 (sigma 1/3)
    got = load_tomogram_frames(tomo, image_size=48)
    assert got.shape == (2, 48, 48) and np.abs(got - ref).max() < 2e-5        # (float64 host glue vs the float32 oracle)
    # hand computation of one interior output pixel of a ramp along y (x constant): the symmetric filter leaves a linear ramp unchanged,
    # the bilinear sample at (2 o + 0.5) is the mean of rows 2 o and 2 o + 1
    ramp = np.repeat(np.arange(96, dtype=np.float32)[None, :, None], 80, axis=2)
    out = V.load_tomogram_frames(ramp, image_size=48)[0, 0].numpy()
    norm = lambda v: 2 * (2 * (v - 0.0) / 95.0 - 1) - 1                         # min-max to [-1, 1], then 2 x - 1
    assert abs(out[10, 7] - norm(20.5)) < 1e-5 and abs(out[30, 40] - norm(60.5)) < 1e-5
    w = np.exp(-0.5 * np.arange(-2, 3) ** 2 / 0.25); w /= w.sum()
    assert abs(w[2] - 0.78657) < 1e-4 and abs(w[1] - 0.10645) < 1e-4
    const = np.full((1, 70, 70), 3.0, np.float32); const[0, 0, 0] = 2.0; const[0, -1, -1] = 4.0       # (min != max)
    c = V.load_tomogram_frames(const, image_size=32)[0, 0].numpy()
    assert abs(c[16, 16] - (2 * (2 * 0.5 - 1) - 1)) < 1e-6
