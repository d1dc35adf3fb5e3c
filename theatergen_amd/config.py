"""UNet plans (the diffusers ``config.json`` constants) for the checkpoints TheaterGen uses.

The reference never stores these (they live in the HF hub config of each checkpoint downloaded at
``generate.py:58-62,105-109``); values from SURVEY.md Appendix A.3.  Field names are the diffusers
``UNet2DConditionModel`` config names (reference ctor ``models/unet_2d_condition.py:209-262``) because
callers read them: ``unet.config.in_channels`` / ``.cross_attention_dim`` / ``.block_out_channels`` /
``.sample_size`` (reference ``ip_adapter/ip_adapter.py:89,96-107``, ``utils/latents.py:261-264``).
"""
from dataclasses import dataclass, field, asdict
from typing import Optional, Tuple, Union


@dataclass
class UNetConfig:
    sample_size: int = 64
    in_channels: int = 4
    out_channels: int = 4
    flip_sin_to_cos: bool = True
    freq_shift: int = 0
    down_block_types: Tuple[str, ...] = ("CrossAttnDownBlock2D", "CrossAttnDownBlock2D", "CrossAttnDownBlock2D", "DownBlock2D")
    up_block_types: Tuple[str, ...] = ("UpBlock2D", "CrossAttnUpBlock2D", "CrossAttnUpBlock2D", "CrossAttnUpBlock2D")
    block_out_channels: Tuple[int, ...] = (320, 640, 1280, 1280)
    layers_per_block: int = 2
    norm_num_groups: int = 32
    norm_eps: float = 1e-5
    cross_attention_dim: int = 768
    transformer_layers_per_block: Union[int, Tuple[int, ...]] = 1
    attention_head_dim: Union[int, Tuple[int, ...]] = 8      # used as NUMBER OF HEADS (unet_2d_blocks.py:202-205)
    use_linear_projection: bool = False
    addition_embed_type: Optional[str] = None
    addition_time_embed_dim: Optional[int] = None
    projection_class_embeddings_input_dim: Optional[int] = None
    prediction_type: str = "epsilon"                         # scheduler-side, kept with the plan
    name: str = "sd15"

    def get(self, k, default=None):
        return getattr(self, k, default)

    def to_dict(self):
        return asdict(self)

    @property
    def time_embed_dim(self):
        return self.block_out_channels[0] * 4

    def per_block(self, v):
        n = len(self.block_out_channels)
        return tuple(v) if isinstance(v, (tuple, list)) else (v,) * n


def sd15():
    """runwayml/stable-diffusion-v1-5 (reference ``generate.py:58``)."""
    return UNetConfig(name="sd15")


def sd21():
    """stabilityai/stable-diffusion-2-1 (768-v): heads (5,10,20,20) -> d=64, ctx 1024, linear projections."""
    return UNetConfig(name="sd21", sample_size=96, cross_attention_dim=1024, attention_head_dim=(5, 10, 20, 20),
                      use_linear_projection=True, prediction_type="v_prediction")


def sdxl():
    """stabilityai/stable-diffusion-xl-base-1.0 (reference ``generate.py:105``, text_time cond
    ``ip_adapter/unet_2d_condition.py:937-954``)."""
    return UNetConfig(name="sdxl", sample_size=128,
                      down_block_types=("DownBlock2D", "CrossAttnDownBlock2D", "CrossAttnDownBlock2D"),
                      up_block_types=("CrossAttnUpBlock2D", "CrossAttnUpBlock2D", "UpBlock2D"),
                      block_out_channels=(320, 640, 1280), cross_attention_dim=2048,
                      transformer_layers_per_block=(1, 2, 10), attention_head_dim=(5, 10, 20),
                      use_linear_projection=True, addition_embed_type="text_time",
                      addition_time_embed_dim=256, projection_class_embeddings_input_dim=2816)


def tiny(name="tiny", ctx=64, linear=False, xl=False):
    """Small plan with the same block structure (for parity tests the CPU oracle finishes in seconds)."""
    if xl:
        return UNetConfig(name=name, sample_size=16,
                          down_block_types=("DownBlock2D", "CrossAttnDownBlock2D", "CrossAttnDownBlock2D"),
                          up_block_types=("CrossAttnUpBlock2D", "CrossAttnUpBlock2D", "UpBlock2D"),
                          block_out_channels=(64, 128, 256), cross_attention_dim=ctx,
                          transformer_layers_per_block=(1, 1, 2), attention_head_dim=(1, 2, 4),
                          use_linear_projection=True, addition_embed_type="text_time",
                          addition_time_embed_dim=32, projection_class_embeddings_input_dim=32 * 6 + 64)
    return UNetConfig(name=name, sample_size=16, block_out_channels=(64, 128, 256, 256),
                      cross_attention_dim=ctx, attention_head_dim=(2, 2, 4, 4) if linear else 2,
                      use_linear_projection=linear)


PLANS = {"sd15": sd15, "sd21": sd21, "sdxl": sdxl, "tiny": tiny}
