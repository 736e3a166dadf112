"""fp8 (e4m3) weight format (BASELINE configs[4]): the engine's own quantiser against the oracle run with the SAME
quantised weights (the bf16-operand bounds of tests/test_gpu_engine.py must hold unchanged: quantisation is the only thing that differs),
and the price of the format against the fp32 oracle with the original weights, reported and bounded."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def rel_rms(a, b):
    return ((a - b).pow(2).mean().sqrt() / b.pow(2).mean().sqrt()).item()


def test_fp8_weights_match_oracle_with_the_same_quantised_weights():
    from oracle import fp8_ref, saber_ref, sam2_ref
    from saber_amd.engine import Engine
    from saber_amd.model_config import get_config
    from saber_amd.utils import preprocessing  # noqa: F401
    from saber_amd.weights import seeded_weights
    cfg = get_config("tiny")
    W = seeded_weights(cfg, 0)
    eng8 = Engine("tiny", weights=W, max_images=1, max_prompts=8, weight_format="fp8")
    eng16 = Engine("tiny", weights=W, max_images=1, max_prompts=8)
    img = saber_ref.prepare(saber_ref.synthetic_slice(seed=2).astype(np.float32))
    Wq = sam2_ref.to_torch(fp8_ref.quantise_encoder_weights(W, cfg))
    W0 = sam2_ref.to_torch(W)
    with torch.no_grad():
        pix = sam2_ref.sam2_transforms(np.repeat(img[..., None], 3, 2))
        fq = sam2_ref.encode_image(Wq, cfg, pix)
        f0 = sam2_ref.encode_image(W0, cfg, pix)
    t = torch.from_numpy(img).cuda()
    eng8.encode(t); g8 = {k: v.cpu() for k, v in eng8.get_features(0).items()}
    eng16.encode(t); g16 = {k: v.cpu() for k, v in eng16.get_features(0).items()}
    torch.cuda.synchronize()
    same = {k: rel_rms(g8[k], fq[k][0]) for k in g8}            # engine(fp8) vs oracle(same quantised weights): bf16-operand residual only
    base = {k: rel_rms(g16[k], f0[k][0]) for k in g16}          # engine(bf16) vs oracle(original weights): the same residual
    cost = {k: rel_rms(fq[k][0], f0[k][0]) for k in g8}         # what the format itself costs (fp32 oracle, quantised vs original weights)
    print("fp8 engine vs oracle with quantised weights:", same)
    print("bf16 engine vs oracle:", base)
    print("price of e4m3 weights (oracle vs oracle):", cost)
    for k in same:
        assert same[k] < 2.0 * max(base[k], 2e-3), (k, same[k], base[k])
    # stage-0/1 features never see a quantised weight
    assert torch.equal(g8["feat_s0"], g16["feat_s0"])
    assert 1e-3 < cost["image_embed"] < 0.12
    # the two engines really differ by the quantisation
    assert rel_rms(g8["image_embed"], g16["image_embed"]) > 0.5 * cost["image_embed"]
