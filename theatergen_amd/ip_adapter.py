"""IP-Adapter object = the plug-in boundary of the hot path (reference ``ip_adapter/ip_adapter.py``).

Same surface as the reference: ``IPAdapter(sd_pipe, image_encoder_path, ip_ckpt, device, num_tokens=4)``
(:68-85), ``init_proj`` (:87-93), ``set_ip_adapter`` (:95-125, the name-driven processor table),
``load_ip_adapter`` (:127-140, ``image_proj.*`` / ``ip_adapter.*`` split, ``.bin`` or ``.safetensors``),
``get_image_embeds(pil_image=None, clip_image_embeds=None) -> (cond, uncond)`` (:142-153), ``set_scale``
(:155-158); ``IPAdapterPlus`` (:289-317), ``IPAdapterFull`` (:320-328), ``IPAdapterXL`` (:225), ``IPAdapterPlusXL``
(:331-359).  ``sd_pipe`` is any object with ``.unet`` (a ``theatergen_amd.UNet2DConditionModel`` or anything with
the diffusers ``attn_processors`` / ``set_attn_processor`` / ``config`` surface) — ``pipelines.SDPipe`` is the
minimal one.

The CLIP vision encoder is a third-party ``transformers`` model (out of the hot path):
it is only constructed when ``image_encoder_path`` is given; with ``image_encoder_path=None`` the adapter takes
pre-computed ``clip_image_embeds`` (what the benchmark's synthetic image tokens stand for).
"""
import os
import re

import torch

from .attention_processor import AttnProcessor, CNAttnProcessor, IPAttnProcessor
from .resampler import ImageProjModel, MLPProjModel, Resampler


_LAYER_NAME = re.compile(r"^(mid_block|up_blocks|down_blocks)(?:\.(\d+))?\.")


class IPAdapter:
    def __init__(self, sd_pipe, image_encoder_path, ip_ckpt, device, num_tokens=4, dtype=None):
        self.device = device
        self.image_encoder_path = image_encoder_path
        self.ip_ckpt = ip_ckpt
        self.num_tokens = num_tokens
        self.pipe = sd_pipe.to(self.device) if hasattr(sd_pipe, "to") else sd_pipe
        self.dtype = dtype if dtype is not None else self.pipe.unet.dtype
        self.set_ip_adapter()
        self.image_encoder = None
        self.clip_image_processor = None
        if image_encoder_path is not None:
            # the checkpoint is read with transformers (file format / config), the tower itself then runs on the native
            # kernels (theatergen_amd/clip.py: same call surface and state-dict names as the library's class)
            from transformers import CLIPImageProcessor, CLIPVisionModelWithProjection as HFCLIP  # third-party, optional
            from .clip import CLIPVisionConfig, CLIPVisionModelWithProjection
            hf = HFCLIP.from_pretrained(image_encoder_path)
            c = hf.config
            cfg = CLIPVisionConfig(hidden_size=c.hidden_size, intermediate_size=c.intermediate_size,
                                   num_hidden_layers=c.num_hidden_layers, num_attention_heads=c.num_attention_heads,
                                   image_size=c.image_size, patch_size=c.patch_size, projection_dim=c.projection_dim,
                                   hidden_act=c.hidden_act, layer_norm_eps=c.layer_norm_eps)
            self.image_encoder = CLIPVisionModelWithProjection.from_state_dict(cfg, hf.state_dict(), device=self.device, dtype=self.dtype)
            del hf
            self.clip_image_processor = CLIPImageProcessor()
        self.image_proj_model = self.init_proj()
        if ip_ckpt is not None:
            self.load_ip_adapter()

    # dimensions of the (possibly absent) CLIP encoder: ViT-H/14 defaults (projection 1024, hidden 1280)
    @property
    def _clip_projection_dim(self):
        return self.image_encoder.config.projection_dim if self.image_encoder is not None else 1024

    @property
    def _clip_hidden_size(self):
        return self.image_encoder.config.hidden_size if self.image_encoder is not None else 1280

    def init_proj(self):
        return ImageProjModel(cross_attention_dim=self.pipe.unet.config.cross_attention_dim,
                              clip_embeddings_dim=self._clip_projection_dim,
                              clip_extra_context_tokens=self.num_tokens).to(self.device, dtype=self.dtype)

    def _processor_table(self):
        """(name, hidden width or None) per attention layer, in ``unet.attn_processors`` order.  The contract of reference :95-113:
        ``attn1`` layers are self-attention (no width needed), every other layer is a cross-attention of its block's width —
        ``block_out_channels`` indexed forwards for ``down_blocks.<i>``, backwards for ``up_blocks.<i>``, last entry for ``mid_block``."""
        widths = list(self.pipe.unet.config.block_out_channels)
        pick = {"mid_block": lambda i: widths[-1], "up_blocks": lambda i: widths[::-1][i], "down_blocks": lambda i: widths[i]}
        table = []
        for name in self.pipe.unet.attn_processors.keys():
            if name.endswith("attn1.processor"):
                table.append((name, None))
                continue
            m = _LAYER_NAME.match(name)
            if m is None:
                raise RuntimeError(f"IPAdapter: cannot place attention layer {name!r} in a UNet block")
            table.append((name, pick[m.group(1)](int(m.group(2) or 0))))
        return table

    def set_ip_adapter(self):
        unet = self.pipe.unet
        ctx = unet.config.cross_attention_dim
        current = unet.attn_processors
        procs = {}
        for name, width in self._processor_table():
            if width is None:
                procs[name] = AttnProcessor()
                continue
            old = current[name]
            keep = isinstance(old, IPAttnProcessor) and old.num_tokens == self.num_tokens and old.hidden_size == width
            procs[name] = old if keep else IPAttnProcessor(hidden_size=width, cross_attention_dim=ctx, scale=1.0,      # keep loaded IP weights
                                                           num_tokens=self.num_tokens).to(self.device, dtype=self.dtype)
        unet.set_attn_processor(procs)
        controlnet = getattr(self.pipe, "controlnet", None)
        for net in (getattr(controlnet, "nets", None) or ([controlnet] if controlnet is not None else [])):
            net.set_attn_processor(CNAttnProcessor(num_tokens=self.num_tokens))        # reference :120-125 (Multi- or single ControlNet)

    def load_ip_adapter(self):
        """checkpoint -> the two sub-dicts ``image_proj`` / ``ip_adapter`` (reference :127-140): a ``.safetensors`` file stores them flat
        under those two prefixes, a ``.bin`` already nested"""
        if os.path.splitext(self.ip_ckpt)[-1] == ".safetensors":
            from safetensors import safe_open
            parts = {"image_proj": {}, "ip_adapter": {}}
            with safe_open(self.ip_ckpt, framework="pt", device="cpu") as f:
                for key in f.keys():
                    prefix, _, rest = key.partition(".")
                    if prefix in parts:
                        parts[prefix][rest] = f.get_tensor(key)
        else:
            parts = torch.load(self.ip_ckpt, map_location="cpu")
        self.load_state_dicts(parts["image_proj"], parts["ip_adapter"])

    def load_state_dicts(self, image_proj_sd, ip_adapter_sd):
        self.image_proj_model.load_state_dict(image_proj_sd)
        ip_layers = torch.nn.ModuleList(self.pipe.unet.attn_processors.values())    # keys "1.to_k_ip.weight", "3...."
        ip_layers.load_state_dict(ip_adapter_sd)

    @torch.inference_mode()
    def get_image_embeds(self, pil_image=None, clip_image_embeds=None):
        if pil_image is not None:
            if self.image_encoder is None:
                raise RuntimeError("IPAdapter was built without a CLIP image encoder: pass clip_image_embeds=")
            if not isinstance(pil_image, (list, tuple)):
                pil_image = [pil_image]
            clip_image = self.clip_image_processor(images=pil_image, return_tensors="pt").pixel_values
            clip_image_embeds = self.image_encoder(clip_image.to(self.device, dtype=self.dtype)).image_embeds
        else:
            clip_image_embeds = clip_image_embeds.to(self.device, dtype=self.dtype)
        image_prompt_embeds = self.image_proj_model(clip_image_embeds)
        zeros = torch.zeros(clip_image_embeds.shape, dtype=clip_image_embeds.dtype, device=clip_image_embeds.device)
        uncond_image_prompt_embeds = self.image_proj_model(zeros)
        return image_prompt_embeds, uncond_image_prompt_embeds

    def set_scale(self, scale):
        for attn_processor in self.pipe.unet.attn_processors.values():
            if isinstance(attn_processor, IPAttnProcessor):
                attn_processor.scale = scale


class IPAdapterXL(IPAdapter):
    """SDXL (reference :225-286; generation itself lives in the caller's pipeline)."""


class IPAdapterPlus(IPAdapter):
    """IP-Adapter with fine-grained features: Perceiver Resampler over the penultimate CLIP hidden state."""

    _resampler_dim_from_unet = True
    _heads = 12

    def init_proj(self):
        unet_cfg = self.pipe.unet.config
        dim = unet_cfg.cross_attention_dim if self._resampler_dim_from_unet else 1280
        return Resampler(dim=dim, depth=4, dim_head=64, heads=self._heads, num_queries=self.num_tokens,
                         embedding_dim=self._clip_hidden_size, output_dim=unet_cfg.cross_attention_dim,
                         ff_mult=4).to(self.device, dtype=self.dtype)

    def _zero_image_states(self, like):
        """CLIP penultimate hidden states of an all-zero image (reference :313-315): a constant of the encoder, computed once"""
        if self.image_encoder is None:
            raise RuntimeError("IPAdapterPlus.get_image_embeds: the unconditional branch is the CLIP hidden state of an all-zero image "
                               "(ip_adapter/ip_adapter.py:313-315); this adapter was built without an image encoder, so pass "
                               "uncond_clip_image_embeds= next to clip_image_embeds=")
        cached = getattr(self, "_zero_states", None)
        if cached is None:
            size = self.image_encoder.config.image_size
            zeros = torch.zeros((1, 3, size, size), device=self.device, dtype=self.dtype)
            cached = self._zero_states = self.image_encoder(zeros, output_hidden_states=True).hidden_states[-2]
        return cached.expand(like.shape[0], -1, -1)

    @torch.inference_mode()
    def get_image_embeds(self, pil_image=None, clip_image_embeds=None, uncond_clip_image_embeds=None):
        """Reference signature ``(pil_image=None, clip_image_embeds=None)`` (:305-317; there only ``pil_image`` is honoured).  Here
        ``clip_image_embeds`` = precomputed penultimate hidden states [b, 257, hidden] (``hidden_states[-2]``, :310) is honoured too;
        the unconditional states come from ``uncond_clip_image_embeds`` (extension) or are computed from a zero image with the
        adapter's own encoder."""
        if pil_image is not None:
            if self.image_encoder is None:
                raise RuntimeError("IPAdapterPlus was built without a CLIP image encoder: pass clip_image_embeds=")
            if not isinstance(pil_image, (list, tuple)):
                pil_image = [pil_image]
            clip_image = self.clip_image_processor(images=pil_image, return_tensors="pt").pixel_values
            clip_image = clip_image.to(self.device, dtype=self.dtype)
            clip_image_embeds = self.image_encoder(clip_image, output_hidden_states=True).hidden_states[-2]
        elif clip_image_embeds is None:
            raise RuntimeError("IPAdapterPlus.get_image_embeds: pass pil_image= or clip_image_embeds=")
        clip_image_embeds = clip_image_embeds.to(self.device, dtype=self.dtype)
        if uncond_clip_image_embeds is None:
            uncond_clip_image_embeds = self._zero_image_states(clip_image_embeds)
        # the Resampler is ~40 launch-bound kernels per call: replayed from a hipGraph when it has one to offer (Resampler.graphed)
        proj = getattr(self.image_proj_model, "graphed", None) if getattr(self, "use_graph", True) else None
        proj = proj if proj is not None else self.image_proj_model
        image_prompt_embeds = proj(clip_image_embeds)
        uncond_image_prompt_embeds = proj(uncond_clip_image_embeds.to(self.device, dtype=self.dtype).contiguous())
        return image_prompt_embeds, uncond_image_prompt_embeds


class IPAdapterFull(IPAdapterPlus):
    def init_proj(self):
        return MLPProjModel(cross_attention_dim=self.pipe.unet.config.cross_attention_dim,
                            clip_embeddings_dim=self._clip_hidden_size).to(self.device, dtype=self.dtype)


class IPAdapterPlusXL(IPAdapterPlus):
    """SDXL Plus: Resampler dim 1280, 20 heads, output = cross_attention_dim 2048 (reference :334-345)."""
    _resampler_dim_from_unet = False
    _heads = 20
