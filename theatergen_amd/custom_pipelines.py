"""The denoising loop of the reference's SDXL pipeline on the MI355X engine.

Reference: ``ip_adapter/custom_pipelines.py::StableDiffusionXLCustomPipeline`` — ``set_scale`` (:18-21) and the loop of ``__call__``
(:308-367): CFG batch = [negative ; positive] (:299-302), ``added_cond_kwargs = {"text_embeds", "time_ids"}`` (:341), per-step gating of
the IP-Adapter scale ``set_scale(0.0)`` outside ``[control_guidance_start, control_guidance_end]`` and ``set_scale(conditioning_scale)``
inside (:322-333), ``noise_uncond + guidance_scale * (noise_text - noise_uncond)`` (:352-354), ``scheduler.step`` (:361).

Only the hot path lives here: the caller passes EMBEDDINGS (prompt / negative prompt / pooled), exactly the tensors the reference computes with
``encode_prompt`` before its loop, and gets latents back (``output_type="latent"``; ``theatergen_amd.vae`` decodes).  The loop runs on
``DenoiseEngine``: one captured hipGraph per step, the IP scale read from a device scalar, so the gating costs one 4-byte fill per step and
never re-captures.  Unsupported arguments raise (never silently ignored)."""
import torch

from .attention_processor import IPAttnProcessor
from .pipelines import DenoiseEngine
from .scheduler import DDIMScheduler


class StableDiffusionXLCustomPipeline:
    def __init__(self, unet, scheduler=None):
        self.unet = unet
        self.scheduler = scheduler if scheduler is not None else DDIMScheduler(prediction_type=unet.config.prediction_type)
        self.vae_scale_factor = 8
        self._engines = {}

    def set_scale(self, scale):                                  # :18-21
        for attn_processor in self.unet.attn_processors.values():
            if isinstance(attn_processor, IPAttnProcessor):
                attn_processor.scale = scale

    @staticmethod
    def _get_add_time_ids(original_size, crops_coords_top_left, target_size, dtype):
        return torch.tensor([list(original_size + crops_coords_top_left + target_size)], dtype=dtype)

    def _engine(self, n, height, width, steps, guidance_scale, enc_len):
        key = (n, height, width, steps, float(guidance_scale), enc_len)
        if key not in self._engines:
            self._engines[key] = DenoiseEngine(self.unet, self.scheduler, n_img=n, height=height, width=width, num_inference_steps=steps,
                                               guidance_scale=guidance_scale, enc_len=enc_len)
        return self._engines[key]

    @torch.no_grad()
    def __call__(self, prompt=None, prompt_2=None, height=None, width=None, num_inference_steps=50, denoising_end=None, guidance_scale=5.0,
                 negative_prompt=None, negative_prompt_2=None, num_images_per_prompt=1, eta=0.0, generator=None, latents=None,
                 prompt_embeds=None, negative_prompt_embeds=None, pooled_prompt_embeds=None, negative_pooled_prompt_embeds=None,
                 output_type="latent", return_dict=True, callback=None, callback_steps=1, cross_attention_kwargs=None, guidance_rescale=0.0,
                 original_size=None, crops_coords_top_left=(0, 0), target_size=None, negative_original_size=None,
                 negative_crops_coords_top_left=(0, 0), negative_target_size=None, control_guidance_start=0.0, control_guidance_end=1.0):
        if prompt is not None or prompt_2 is not None or negative_prompt is not None or negative_prompt_2 is not None:
            raise NotImplementedError("text encoding is outside the hot path: pass prompt_embeds / negative_prompt_embeds / pooled_* "
                                      "(theatergen_amd.clip.CLIPTextModel produces them)")
        if prompt_embeds is None or negative_prompt_embeds is None or pooled_prompt_embeds is None or negative_pooled_prompt_embeds is None:
            raise ValueError("prompt_embeds, negative_prompt_embeds, pooled_prompt_embeds and negative_pooled_prompt_embeds are required")
        for name, bad in (("denoising_end", denoising_end is not None), ("eta", eta != 0.0), ("guidance_rescale", guidance_rescale != 0.0),
                          ("cross_attention_kwargs", cross_attention_kwargs is not None), ("num_images_per_prompt", num_images_per_prompt != 1),
                          ("negative_original_size / negative_target_size", negative_original_size is not None or negative_target_size is not None),
                          ("guidance_scale <= 1 (no classifier-free guidance)", guidance_scale <= 1.0)):
            if bad:
                raise NotImplementedError(f"StableDiffusionXLCustomPipeline: {name} is not on the TheaterGen hot path")
        if output_type != "latent":
            raise NotImplementedError("only output_type='latent' (decode with theatergen_amd.vae.AutoencoderKL)")
        unet, dev, dt = self.unet, self.unet.device, self.unet.dtype
        height = height or unet.config.sample_size * self.vae_scale_factor
        width = width or unet.config.sample_size * self.vae_scale_factor
        original_size = original_size or (height, width)
        target_size = target_size or (height, width)
        n = prompt_embeds.shape[0]
        # CFG batch: negatives first (:299-302)
        enc = torch.cat([negative_prompt_embeds, prompt_embeds], dim=0).to(dev, dt)
        text_embeds = torch.cat([negative_pooled_prompt_embeds, pooled_prompt_embeds], dim=0).to(dev, dt)
        time_ids = self._get_add_time_ids(tuple(original_size), tuple(crops_coords_top_left), tuple(target_size), torch.float32)
        time_ids = time_ids.to(dev).repeat(2 * n, 1)             # negative ids = positive ids (:292-297 when no negative sizes are given)
        h8, w8 = height // self.vae_scale_factor, width // self.vae_scale_factor
        if latents is None:                                      # prepare_latents: randn * init_noise_sigma (1.0 for DDIM)
            latents = torch.randn((n, unet.config.in_channels, h8, w8), generator=generator, dtype=torch.float32)
        eng = self._engine(n, height, width, num_inference_steps, guidance_scale, enc.shape[1])
        eng.set_conditioning(enc, {"text_embeds": text_embeds, "time_ids": time_ids})
        conditioning_scale = None                                # :322-326
        for p in unet.attn_processors.values():
            if isinstance(p, IPAttnProcessor):
                conditioning_scale = p.scale
                break
        steps = num_inference_steps

        def before_step(i):                                      # :329-333
            if conditioning_scale is not None:
                self.set_scale(0.0 if (i / steps < control_guidance_start or (i + 1) / steps > control_guidance_end) else conditioning_scale)
            if callback is not None and i > 0 and (i - 1) % callback_steps == 0:
                callback(i - 1, int(eng.timesteps[i - 1]), eng.history[i])
        hist = eng.run(latents, before_step=before_step)
        if callback is not None and (steps - 1) % callback_steps == 0:
            callback(steps - 1, int(eng.timesteps[steps - 1]), hist[steps])
        out = hist[-1].to(dt)
        if not return_dict:
            return (out,)
        from types import SimpleNamespace
        return SimpleNamespace(images=out)
