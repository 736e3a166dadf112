"""SAM2.1 image-model hyper-parameters for the four trunks SABER can name.

The reference selects a trunk by the strings ``tiny/small/base/large``
(reference: saber/adapters/base.py:28-33, saber/pretrained_weights.py:174-202) and
hands the matching ``configs/sam2.1/sam2.1_hiera_{t,s,b+,l}.yaml`` to the third-party
``sam2`` package.  That package is absent here, so the hyper-parameters of those four
yaml files are restated as data (cross-checked against the
independent HF restatement's parameter counts in oracle/hf_crosscheck.py).
"""
from dataclasses import dataclass, field
from typing import List, Tuple


@dataclass(frozen=True)
class HieraConfig:
    name: str
    embed_dim: int
    num_heads: int
    stages: Tuple[int, ...]
    global_att_blocks: Tuple[int, ...]
    window_spec: Tuple[int, ...]
    pos_embed_bkg: Tuple[int, int] = (7, 7)
    q_pool: int = 3
    image_size: int = 1024
    fpn_dim: int = 256
    fpn_top_down_levels: Tuple[int, ...] = (2, 3)
    ln_eps: float = 1e-6

    @property
    def depth(self) -> int:
        return sum(self.stages)

    @property
    def stage_dims(self) -> List[int]:
        return [self.embed_dim * (2 ** i) for i in range(len(self.stages))]

    @property
    def stage_heads(self) -> List[int]:
        return [self.num_heads * (2 ** i) for i in range(len(self.stages))]

    @property
    def stage_ends(self) -> List[int]:
        out, s = [], 0
        for n in self.stages:
            s += n
            out.append(s - 1)
        return out

    def block_specs(self):
        """Per block: (dim_in, dim_out, heads, window, q_stride) following upstream
        Hiera: first block of a stage takes dim and window from the previous stage and
        pools q (stages 1..q_pool); global blocks have window 0."""
        specs = []
        q_pool_blocks = [e + 1 for e in self.stage_ends[:-1]][: self.q_pool]
        dims, heads = self.stage_dims, self.stage_heads
        cur_stage = 0
        for i in range(self.depth):
            if i - 1 in self.stage_ends:
                cur_stage += 1
            first = cur_stage > 0 and (i - 1) in self.stage_ends
            dim_out = dims[cur_stage]
            dim_in = dims[cur_stage - 1] if first else dim_out
            window = self.window_spec[cur_stage - 1] if first else self.window_spec[cur_stage]
            if i in self.global_att_blocks:
                window = 0
            q_stride = 2 if i in q_pool_blocks else 1
            specs.append((dim_in, dim_out, heads[cur_stage], window, q_stride))
        return specs


HIERA_CONFIGS = {
    "tiny": HieraConfig("tiny", 96, 1, (1, 2, 7, 2), (5, 7, 9), (8, 4, 14, 7)),
    "small": HieraConfig("small", 96, 1, (1, 2, 11, 2), (7, 10, 13), (8, 4, 14, 7)),
    "base": HieraConfig("base", 112, 2, (2, 3, 16, 3), (12, 16, 20), (8, 4, 14, 7), (14, 14)),
    "large": HieraConfig("large", 144, 2, (2, 6, 36, 4), (23, 33, 43), (8, 4, 16, 8)),
}

# Decoder / prompt-encoder constants shared by all sam2.1 checkpoints.
DEC_DIM = 256
DEC_HEADS = 8
DEC_MLP = 2048
DEC_DEPTH = 2
NUM_MASK_TOKENS = 4
IMAGE_MEAN = (0.485, 0.456, 0.406)
IMAGE_STD = (0.229, 0.224, 0.225)
DYN_MULTIMASK_DELTA = 0.05
DYN_MULTIMASK_THRESH = 0.98


def get_config(name: str) -> HieraConfig:
    if name not in HIERA_CONFIGS:
        raise ValueError(f"cfg must be one of tiny/small/base/large, got '{name}'")
    return HIERA_CONFIGS[name]
