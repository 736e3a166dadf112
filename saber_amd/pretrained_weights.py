"""Checkpoint naming contract of the reference (saber/pretrained_weights.py:174-202): the strings
tiny/small/base/large map to (configs/sam2.1/sam2.1_hiera_{t,s,b+,l}.yaml, sam2.1_hiera_*.pt).
There is no network here, so nothing is downloaded: a present file is loaded, an absent one raises -
unless SABER_AMD_SEEDED_WEIGHTS=1 explicitly selects the deterministic synthetic weights used by the
tests and the benchmark."""
import os

_NAMES = {
    "tiny": ("configs/sam2.1/sam2.1_hiera_t.yaml", "sam2.1_hiera_tiny.pt"),
    "small": ("configs/sam2.1/sam2.1_hiera_s.yaml", "sam2.1_hiera_small.pt"),
    "base": ("configs/sam2.1/sam2.1_hiera_b+.yaml", "sam2.1_hiera_base_plus.pt"),
    "large": ("configs/sam2.1/sam2.1_hiera_l.yaml", "sam2.1_hiera_large.pt"),
}


def checkpoint_dir() -> str:
    return os.environ.get("SABER_AMD_CHECKPOINTS", os.path.join(os.path.dirname(os.path.abspath(__file__)), "checkpoints"))


def get_sam2_checkpoint(sam2_cfg: str):
    """-> (config name, checkpoint path).  ValueError for an unknown trunk, like the reference (:193-195)."""
    if sam2_cfg not in _NAMES:
        raise ValueError(f"Invalid SAM2 Config: {sam2_cfg}. Valid options are: {list(_NAMES)}")
    cfg, fname = _NAMES[sam2_cfg]
    return cfg, os.path.join(checkpoint_dir(), fname)


def resolve_weights(sam2_cfg: str, checkpoint: str = None):
    """-> dict(checkpoint=path) or dict(seed=int) for saber_amd.engine.Engine."""
    _, path = get_sam2_checkpoint(sam2_cfg)
    path = checkpoint or path
    if os.path.exists(path):
        return {"checkpoint": path}
    if os.environ.get("SABER_AMD_SEEDED_WEIGHTS", "0") == "1":
        return {"seed": int(os.environ.get("SABER_AMD_SEED", "0"))}
    if os.environ.get("SABER_AMD_SEEDED_WEIGHTS", "0") == "fitted":      # the seeded encoder with the fitted mask decoder (saber_amd.weights.fitted_decoder_weights; tests / bench only)
        return {"seed": 0, "fitted_decoder": 1}
    raise FileNotFoundError(f"SAM2.1 checkpoint '{path}' not found (no network: nothing is downloaded). Place the file there, "
                            f"or set SABER_AMD_SEEDED_WEIGHTS=1 to run with deterministic synthetic weights.")


def load_weights(sam2_cfg: str, checkpoint: str = None, video: bool = False):
    """-> {upstream checkpoint key: fp32 numpy array} from the resolved source (checkpoint file, or the seeded tensors when
    SABER_AMD_SEEDED_WEIGHTS=1); video=True adds the memory model of the video predictor (memory attention / encoder, object pointers)."""
    from saber_amd.model_config import get_config
    from saber_amd.weights import load_checkpoint, seeded_weights
    src = resolve_weights(sam2_cfg, checkpoint)
    cfg = get_config(sam2_cfg)
    if "checkpoint" in src:
        return load_checkpoint(src["checkpoint"], cfg, video=video)
    if src.get("fitted_decoder"):
        from saber_amd.weights import fitted_decoder_weights
        return fitted_decoder_weights(cfg, src["seed"], video=video)
    return seeded_weights(cfg, src["seed"], video=video)
