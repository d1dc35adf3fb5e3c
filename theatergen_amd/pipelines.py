"""The two denoising loops that call the hot path, restructured for MI355X (reference ``models/pipelines.py``).

Reference loop bodies: stage 1 ``generate_semantic_guidance`` :406-453 (``cat([latents]*2)`` -> UNet -> CFG :441-442
-> ``scheduler.step`` :447 -> ``latents_all.append(latents.cpu())`` :449-453) and stage 2
``final_image_generation`` :742-835 (same + frozen-mask replace :833-834); IP embeds ``prepare_ip_embeds`` :860-950.

Here one step = ONE captured hipGraph (UNet CFG call + fused CFG/DDIM/mask epilogue): the timestep table, DDIM
coefficient table and step counter live on the device, the epilogue writes the next UNet input and the per-step
history row itself, so the 50-step loop is 50 graph replays with no host<->device traffic (the reference syncs
and copies latents to the CPU every step).  Any number of independent character images ride in the batch
dimension (CFG batch 2*n: all uncond rows first, then all cond rows = ``noise_pred.chunk(2)`` order).
"""
import os
import weakref

import torch

from . import ops
from .attention_processor import register_replay_stream, tensor_version, unregister_replay_stream
from .scheduler import DDIMScheduler
from .unet import DeviceSchedule


DEFAULT_GUIDANCE_ATTN_KEYS = [("mid", 0, 0, 0), ("up", 1, 0, 0), ("up", 1, 1, 0), ("up", 1, 2, 0)]     # reference pipelines.py:21


class SDPipe:
    """Minimal pipeline object: what ``IPAdapter`` and the latent utilities read from ``adapter.pipe``
    (``.unet``, ``.scheduler``, ``.controlnet``; reference ``ip_adapter.py:74``, ``utils/latents.py:261``)."""

    def __init__(self, unet, scheduler=None, controlnet=None):
        self.unet = unet
        self.scheduler = scheduler if scheduler is not None else DDIMScheduler(prediction_type=unet.config.prediction_type)
        self.controlnet = controlnet
        self.vae_scale_factor = 8

    def to(self, device, dtype=None):
        self.unet.to(device=device, dtype=dtype) if dtype is not None else self.unet.to(device)
        return self


def prepare_ip_embeds(prompt_embeds, negative_prompt_embeds, image_prompt_embeds, uncond_image_prompt_embeds):
    """Text + image tokens along the sequence axis, negatives first (reference pipelines.py:230-233, 369-370, 937-948).
    Inputs [n, 77, D] / [n, T, D]; returns [2n, 77+T, D] = [all negative rows ; all positive rows]."""
    pos = torch.cat([prompt_embeds, image_prompt_embeds.to(prompt_embeds.dtype)], dim=1)
    neg = torch.cat([negative_prompt_embeds, uncond_image_prompt_embeds.to(prompt_embeds.dtype)], dim=1)
    return torch.cat([neg, pos], dim=0)


_engine_ids = iter(range(1, 1 << 30))


class DenoiseEngine:
    """Batched CFG denoiser for ``n_img`` independent character images on one GPU.

    ``run()`` returns ``latents_all`` fp32 [steps+1, n_img, C, h, w] (row 0 = the input latents), the
    ``torch.stack(latents_all)`` of reference pipelines.py:488 for every image of the batch."""

    def __init__(self, unet, scheduler=None, n_img=1, height=512, width=512, num_inference_steps=50, guidance_scale=7.5,
                 enc_len=81, use_graph=True, controlnet=None, controlnet_enc_len=77, timesteps=None):
        """``timesteps`` (optional): a SUBSET of ``scheduler.set_timesteps(num_inference_steps)`` to walk instead of all of them — the reference's fast
        schedule (``utils/schedule.py:4-8`` applied at ``models/pipelines.py:383-384``); as in the reference the DDIM update of a kept step still jumps by
        the full schedule's stride (``prev_t = t - 1000 // num_inference_steps``)."""
        self.unet = unet
        self.ws_slot = next(_engine_ids)      # private kernel scratch: engines may replay concurrently on different streams
        # stage 2 (reference pipelines.py:759-818): every step the ControlNet runs on the same model input with the TEXT
        # embeddings and the (step-invariant) control image; its residuals enter the UNet call
        self.controlnet = controlnet
        self.scheduler = scheduler if scheduler is not None else DDIMScheduler(prediction_type=unet.config.prediction_type)
        self.n_img = n_img
        self.h, self.w = height // 8, width // 8
        self.steps = num_inference_steps
        self.g = guidance_scale
        self.use_graph = use_graph
        dev, dt, cfg = unet.device, unet.dtype, unet.config
        self.dev, self.dt = dev, dt
        C = cfg.in_channels
        self.scheduler.set_timesteps(num_inference_steps)
        self.timesteps = self.scheduler.timesteps.clone() if timesteps is None else torch.as_tensor(timesteps, dtype=torch.int64).clone()
        num_inference_steps = self.steps = int(self.timesteps.numel())
        self.t_table = self.timesteps.to(device=dev, dtype=torch.float32)
        self.coef = self.scheduler.coef_table(self.timesteps).to(dev)
        self.step_idx = torch.zeros(1, dtype=torch.int32, device=dev)
        self.sched = DeviceSchedule(self.t_table, self.step_idx)
        self.latents = torch.zeros((n_img, C, self.h, self.w), dtype=torch.float32, device=dev)
        self.model_in = torch.zeros((2 * n_img, C, self.h, self.w), dtype=dt, device=dev)
        self.enc = torch.zeros((2 * n_img, enc_len, cfg.cross_attention_dim), dtype=dt, device=dev)
        self.history = torch.zeros((num_inference_steps + 1, n_img, C, self.h, self.w), dtype=torch.float32, device=dev)
        self.cn_enc = self.cn_cond = None
        self.cn_scale = 1.0
        if controlnet is not None:
            self.cn_enc = torch.zeros((2 * n_img, controlnet_enc_len, cfg.cross_attention_dim), dtype=dt, device=dev)
            self.cn_cond = torch.zeros((2 * n_img, 3, height, width), dtype=dt, device=dev)
        self.frozen = None
        self.frozen_mask = None
        self.frozen_steps = 0
        self.added = None
        self.graph = None
        self._graph_sig = None
        self.pred_type = 0 if self.scheduler.config.prediction_type == "epsilon" else 1
        self._tproj_table = None
        self._tproj_key = None

    # ---- conditioning -----------------------------------------------------------------------------------
    def set_conditioning(self, encoder_hidden_states, added_cond_kwargs=None):
        """[2*n_img, L, D] (negatives first).  Copied into the engine's static buffer; the step-invariant text /
        image K, V^T of every IP cross-attention layer are re-projected once here (not once per step)."""
        if tuple(encoder_hidden_states.shape) != tuple(self.enc.shape):
            raise ValueError(f"conditioning shape {tuple(encoder_hidden_states.shape)} != engine shape {tuple(self.enc.shape)}")
        self.enc.copy_(encoder_hidden_states)
        if added_cond_kwargs is not None:
            if self.added is None:
                self.added = {k: v.to(self.dev).clone() for k, v in added_cond_kwargs.items()}
            else:
                for k, v in added_cond_kwargs.items():
                    self.added[k].copy_(v)
        self._refresh_kv()

    def set_control(self, controlnet_prompt_embeds, control_image, conditioning_scale=1.0):
        """ControlNet inputs of reference pipelines.py:761-778: text embeddings [2*n_img, L, D] (negatives first) and the
        prepared control image [2*n_img, 3, H, W] in [0, 1]; both are copied into static buffers (graph replay) and the
        conditioning embedding is recomputed once here, not once per step."""
        if self.controlnet is None:
            raise RuntimeError("DenoiseEngine was built without a controlnet")
        self.cn_enc.copy_(controlnet_prompt_embeds)
        self.cn_cond.copy_(control_image)
        if self.graph is not None and float(conditioning_scale) != self.cn_scale:
            self.graph = None          # the scale is a launch constant of the zero-conv epilogues
        self.cn_scale = float(conditioning_scale)
        self.controlnet.cond_embedding(self.cn_cond, static=True)

    def _refresh_kv(self):
        self.unet.register_conditioning(self.enc)              # cache keyed by the tensor OBJECT self.enc (never its address)

    def set_frozen(self, frozen_latents, frozen_mask, frozen_steps):
        """Stage-2 frozen-mask replace (reference pipelines.py:733-738, 833-834): ``frozen_latents`` fp32
        [steps+1, n_img, C, h, w], ``frozen_mask`` fp32 [h, w] or [n_img, h, w]."""
        if self.graph is not None and (self.frozen is None) != (frozen_latents is None):
            self.graph = None          # epilogue signature changed: re-capture
        if frozen_latents is None:
            self.frozen = self.frozen_mask = None
            self.frozen_steps = 0
            return
        if self.frozen is None:
            self.frozen = torch.empty_like(self.history)
            self.frozen_mask = torch.empty((self.n_img, self.h, self.w), dtype=torch.float32, device=self.dev)
        self.frozen.copy_(frozen_latents)
        self.frozen_mask.copy_(frozen_mask.to(torch.float32).expand(self.n_img, self.h, self.w))
        if self.graph is not None and frozen_steps != self.frozen_steps:
            self.graph = None          # frozen_steps is a launch constant
        self.frozen_steps = int(frozen_steps)

    def _ensure_time_proj(self):
        """The UNet's time path (sinusoidal embedding -> TimestepEmbedding MLP -> SiLU -> all ResBlock ``time_emb_proj`` as one GEMM: models/unet_2d_condition.py:
        315-328, 829-856, diffusers ResnetBlock2D) depends on the timestep ONLY: one row per step of this engine's schedule is computed once, with the very launches
        ``unet.time_embed`` issues per step (same shapes: bit-identical rows), and a step gathers its row by the device step counter — 1 launch instead of 5 per step.
        Rebuilt when the schedule or any of the weights involved change; SDXL's ``text_time`` conditioning enters the embedding, there the per-step path stays."""
        unet = self.unet
        if unet.config.addition_embed_type is not None or os.environ.get("TG_TPROJ_TABLE", "1") == "0":
            if self._tproj_table is not None:
                self._tproj_table, self._tproj_key, self.graph = None, None, None
            return
        te = unet.time_embedding
        ps = [te.linear_1.weight, te.linear_1.bias, te.linear_2.weight, te.linear_2.bias]
        for r in unet._resnets:
            ps += [r.time_emb_proj.weight, r.time_emb_proj.bias]
        key = tuple((p.data_ptr(), tensor_version(p)) for p in ps) + (self.t_table.data_ptr(), tensor_version(self.t_table), unet.dtype)
        if key == self._tproj_key:
            return
        B = 2 * self.n_img
        idx = torch.zeros(1, dtype=torch.int32, device=self.dev)
        rows = []
        with torch.no_grad(), ops.workspace_slot(self.ws_slot):
            for s in range(self.steps):
                idx.fill_(s)
                rows.append(unet.time_embed(DeviceSchedule(self.t_table, idx), B, None)[1].clone())
        self._tproj_table = torch.stack(rows).contiguous()                # [steps, B, sum(cout)]
        self._tproj_key = key
        self.graph = None                                                  # a captured step holds the old table's address

    # ---- one step ---------------------------------------------------------------------------------------
    def _step(self):
        with ops.workspace_slot(self.ws_slot):
            self._step_body()

    def _step_body(self):
        down = mid = None
        if self.controlnet is not None:
            down, mid = self.controlnet(self.model_in, self.sched, self.cn_enc, self.cn_cond, conditioning_scale=self.cn_scale,
                                        return_dict=False, token_major=True)
        tp = None
        if self._tproj_table is not None:
            tp = torch.index_select(self._tproj_table, 0, self.step_idx)[0]        # this step's row, by the DEVICE step counter (graph-replayable)
        noise_pred = self.unet(self.model_in, self.sched, self.enc, added_cond_kwargs=self.added, return_dict=False,
                               out_dtype=torch.float32, down_block_additional_residuals=down,
                               mid_block_additional_residual=mid, time_proj=tp)[0]
        ops.step_epilogue(noise_pred, self.latents, self.g, self.coef, self.step_idx, advance=True,
                          prediction_type=self.pred_type, frozen=self.frozen,
                          frozen_mask=self.frozen_mask, frozen_steps=self.frozen_steps, history=self.history,
                          model_in=self.model_in)

    def _signature(self):
        """Launch constants baked into a captured graph: the frozen-mask configuration and the guidance scale.  The IP scales
        are NOT among them: ``IPAttnProcessor.scale`` is read from a device scalar by the attention kernel, so
        ``IPAdapter.set_scale`` between characters (reference pipelines.py:196, 213) and the per-step gating of
        ``ip_adapter/custom_pipelines.py:328-333`` replay the same graph."""
        return self.frozen is not None, self.frozen_steps, self.g

    def _reset(self, latents):
        self.latents.copy_(latents.to(device=self.dev, dtype=torch.float32))
        self.history[0].copy_(self.latents)
        self.model_in[:self.n_img].copy_(self.latents)           # dtype cast on copy = the `.half()` of pipelines.py:414
        self.model_in[self.n_img:].copy_(self.latents)
        self.step_idx.zero_()

    def _capture(self):
        s = torch.cuda.Stream(device=self.dev)
        s.wait_stream(torch.cuda.current_stream(self.dev))
        with torch.cuda.stream(s):
            self._step()                                          # warm-up: allocator, packed weights, K/V caches
        torch.cuda.current_stream(self.dev).wait_stream(s)
        torch.cuda.synchronize(self.dev)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            self._step()
        self.graph = g

    def _ensure_graph(self, latents):
        self._ensure_time_proj()
        sig = self._signature()
        if self.use_graph and (self.graph is None or sig != self._graph_sig):
            self._reset(latents)
            self._capture()
            self._graph_sig = sig

    @staticmethod
    def run_concurrent(engines, latents_list, before_step=None):
        """Independent denoising chains (e.g. the two halves of a story's character batch) replayed CONCURRENTLY, one HIP
        stream per engine: the second chain's kernels fill the partially filled grid rounds, small-grid layers and
        launch gaps of the first (+4 % images/s for 2 x 4 images vs 1 x 8 on MI355X).  Engines may share one UNet; each
        has its own conditioning buffers, K / V^T caches, kernel scratch slot and captured graph.  Returns the histories.
        The IP scale is ONE device scalar per processor (= per UNet).  ``before_step(i)`` (optional) runs on the host before round i of replays is
        queued — the per-step ``set_scale`` gating of ``ip_adapter/custom_pipelines.py:328-333``: the assignment fences its 4-byte fill against the
        engine streams (``attention_processor.fenced_fill``), so round i - 1 still reads the old value and round i the new one."""
        dev = engines[0].dev
        with torch.no_grad():
            for e, lat in zip(engines, latents_list):
                if not e.use_graph:
                    raise RuntimeError("run_concurrent needs graph engines")
                e._ensure_graph(lat)
            cur = torch.cuda.current_stream(dev)
            streams = [e._stream() for e in engines]
            for e, lat, st in zip(engines, latents_list, streams):
                st.wait_stream(cur)
                with torch.cuda.stream(st):
                    e._reset(lat)
            for i in range(engines[0].steps):
                if before_step is not None:
                    before_step(i)
                for e, st in zip(engines, streams):
                    with torch.cuda.stream(st):
                        e.graph.replay()
            for st in streams:
                cur.wait_stream(st)
        return [e.history for e in engines]

    def _stream(self):
        if getattr(self, "_own_stream", None) is None:
            # registered: host-side writes of device scalars the captured step reads (IPAttnProcessor.scale) fence against it
            self._own_stream = register_replay_stream(torch.cuda.Stream(device=self.dev), self.dev)
            weakref.finalize(self, unregister_replay_stream, self._own_stream)
        return self._own_stream

    def run(self, latents, before_step=None):
        """latents [n_img, C, h, w] (any float dtype / device) -> latents_all fp32 [steps+1, n_img, C, h, w] on the GPU.
        ``before_step(i)`` (optional) runs on the host before step i is launched — the place of the per-step
        ``set_scale(0.0)`` / ``set_scale(s)`` gating of reference ``ip_adapter/custom_pipelines.py:328-333``; the IP scale is a
        device scalar, so the same captured graph replays whatever it is set to."""
        with torch.no_grad():
            self._ensure_graph(latents)
            self._reset(latents)
            for i in range(self.steps):
                if before_step is not None:
                    before_step(i)
                if self.use_graph:
                    self.graph.replay()
                else:
                    self._step()
        return self.history


def denoise_single_object(adapter, prompt_embeds, negative_prompt_embeds, input_latents, scale, clip_image_embeds=None,
                          image_prompt_embeds=None, uncond_image_prompt_embeds=None, num_inference_steps=50,
                          guidance_scale=7.5, engine=None):
    """Stage-1 per-character generation core (reference ``generate_semantic_guidance`` SD-1.5 branch, :183-247 image
    prompt + scale, :369-453 loop, :488 stack): returns ``(latents [1,C,h,w], latents_all [steps+1,1,C,h,w])``."""
    adapter.set_scale(scale)
    if image_prompt_embeds is None:
        image_prompt_embeds, uncond_image_prompt_embeds = adapter.get_image_embeds(clip_image_embeds=clip_image_embeds)
    enc = prepare_ip_embeds(prompt_embeds.to(adapter.pipe.unet.dtype), negative_prompt_embeds.to(adapter.pipe.unet.dtype),
                            image_prompt_embeds, uncond_image_prompt_embeds)
    n = prompt_embeds.shape[0]
    if engine is None:
        engine = DenoiseEngine(adapter.pipe.unet, adapter.pipe.scheduler, n_img=n, height=input_latents.shape[-2] * 8,
                               width=input_latents.shape[-1] * 8, num_inference_steps=num_inference_steps,
                               guidance_scale=guidance_scale, enc_len=enc.shape[1])
    engine.set_conditioning(enc)
    latents_all = engine.run(input_latents)
    return latents_all[-1], latents_all


# ---- the reference's two stage functions, same signatures and return tuples, on the engine ------------------------------------------------------
SINGLE_OBJECT_NEGATIVE_PROMPT = "background, multiple objects, incomplete, lowres, bad anatomy, low quality, obscured"      # models/pipelines.py:225
PLACEHOLDER_IMAGE = "model.png"                                                                                            # models/pipelines.py:195


def _own_unet(adapter, what):
    from .unet import UNet2DConditionModel
    unet = adapter.pipe.unet
    if not isinstance(unet, UNet2DConditionModel):
        raise TypeError(f"theatergen_amd.pipelines.{what}: adapter.pipe.unet must be a theatergen_amd.unet.UNet2DConditionModel (INTEGRATION.md level 1: "
                        f"UNet2DConditionModel.from_state_dict(config.sd15(), diffusers_unet.state_dict(), ...)), got {type(unet).__name__}")
    return unet


def _engine_of(adapter, **kw):
    """one engine (static buffers + captured step graph) per loop geometry, kept on the adapter: the characters of a run replay the same graph"""
    cache = adapter.__dict__.setdefault("_tg_engines", {})
    ts = kw.get("timesteps")
    key = tuple(sorted((k, (tuple(v.tolist()) if k == "timesteps" and v is not None else (id(v) if k == "controlnet" else v))) for k, v in kw.items()))
    eng = cache.get(key)
    if eng is None or eng.unet is not adapter.pipe.unet:
        eng = cache[key] = DenoiseEngine(adapter.pipe.unet, adapter.pipe.scheduler, **kw)
    return eng


def _to_pil(image):
    """``image_processor.postprocess(image, output_type='pil', do_denormalize=[True])[0]`` (models/pipelines.py:475): NCHW in [-1, 1] -> PIL RGB"""
    from PIL import Image
    import numpy as np
    arr = (image[0].float() / 2 + 0.5).clamp(0, 1).permute(1, 2, 0).cpu().numpy()
    return Image.fromarray((arr * 255).round().astype(np.uint8))


def _text_and_image_rows(adapter, prompt, negative_prompt, pil_image):
    """[negative rows ; positive rows] = text tokens followed by the IP-Adapter image tokens (models/pipelines.py:204-233, 369-370, 864-948)"""
    img, unc = adapter.get_image_embeds(pil_image=pil_image, clip_image_embeds=None)
    pos, neg = adapter.pipe.encode_prompt(prompt, device=adapter.device, num_images_per_prompt=1, do_classifier_free_guidance=True,
                                          negative_prompt=negative_prompt)[:2]
    return prepare_ip_embeds(pos, neg, img.to(pos.device), unc.to(pos.device))


def generate_semantic_guidance(task, fg_seed_now, basever, ip_prompt, database_path, id, adapter, model_dict, latents, input_embeddings,
                               num_inference_steps, bboxes, phrases, object_positions, guidance_scale=7.5, semantic_guidance_kwargs=None,
                               return_cross_attn=False, return_saved_cross_attn=False, saved_cross_attn_keys=None, return_cond_ca_only=False,
                               return_token_ca_only=None, offload_guidance_cross_attn_to_cpu=False, offload_cross_attn_to_cpu=False,
                               offload_latents_to_cpu=True, return_box_vis=False, show_progress=True, save_all_latents=False,
                               dynamic_num_inference_steps=False, fast_after_steps=None, fast_rate=2, use_boxdiff=False, use_adapter=False,
                               obj_id=False, have_reffer=0):
    """Stage 1, one character (reference ``models/pipelines.py:175-490``; caller ``theatergen.py:108-135``): same arguments, same side effects (the
    character's first image becomes its reference PNG, :476-477), same return tuple ``(latents, image[, saved_attns][, image][, latents_all])``.

    What runs where.  Host, as in the reference: the reference image / placeholder and its IP scale (:183-199), the prompt strings (:216-225),
    ``adapter.get_image_embeds`` and ``adapter.pipe.encode_prompt``.  Device: the whole loop :406-453 is ``num_inference_steps`` replays of ONE captured
    step (UNet CFG call + fused CFG / DDIM epilogue + history row) — no ``.cpu()`` per step; the 51 latents come back in one copy when
    ``offload_latents_to_cpu`` asks for them on the host.  ``cross_attention_kwargs`` is ``None`` in the reference's UNet call (:428), so no map is
    ever saved: ``saved_attns`` is the reference's list of empty dicts.  Unsupported here, loudly: the ``'xl'`` branch (multi-device ``.to('cuda:1')``
    placement; the SDXL loop lives in ``theatergen_amd.custom_pipelines``) and ``return_cross_attn`` (an attribute no diffusers output has)."""
    from PIL import Image
    from . import schedule as tg_schedule
    if basever == "xl":
        raise NotImplementedError("theatergen_amd.pipelines.generate_semantic_guidance: the 'xl' branch (models/pipelines.py:261-366, 466-470) is not on "
                                  "this path; SDXL runs through theatergen_amd.custom_pipelines.StableDiffusionXLCustomPipeline")
    if return_cross_attn:
        raise NotImplementedError("return_cross_attn reads unet_output.cross_attention_probs_* (models/pipelines.py:431-434), which the UNet call of "
                                  "the reference never fills; the saved-map side channel is save_attn_to_dict")
    if not obj_id:
        raise RuntimeError("generate_semantic_guidance: obj_id is required (without it the reference has no prompt embeddings to run on, models/pipelines.py:183-233)")
    unet = _own_unet(adapter, "generate_semantic_guidance")
    # ---- image prompt of the character: its database PNG, or the placeholder with the adapter switched off (:183-199)
    try:
        image = Image.open(database_path + str(obj_id) + ".png")
        scale, have_reffer = 0.4, 1
    except (OSError, ValueError):
        image = Image.open(PLACEHOLDER_IMAGE)
        scale, have_reffer = 0, 0
    adapter.set_scale(scale)
    prompt = ("single object, " if task == "editing" else "full-body picture of ") + str(ip_prompt)
    enc = _text_and_image_rows(adapter, prompt, SINGLE_OBJECT_NEGATIVE_PROMPT, image)
    # ---- the loop
    scheduler = adapter.pipe.scheduler
    timesteps = None
    if fast_after_steps is not None:
        scheduler.set_timesteps(num_inference_steps)
        timesteps = tg_schedule.get_fast_schedule(scheduler.timesteps, fast_after_steps, fast_rate)
    h8, w8 = latents.shape[-2], latents.shape[-1]
    eng = _engine_of(adapter, n_img=latents.shape[0], height=8 * h8, width=8 * w8, num_inference_steps=num_inference_steps,
                     guidance_scale=guidance_scale, enc_len=enc.shape[1], timesteps=timesteps)
    eng.set_conditioning(enc.to(eng.dev, eng.dt))
    hist = eng.run(latents)
    out = hist[-1].to(unet.dtype)                                # the reference's `latents.half()` (:461), in the UNet's storage dtype
    vae = adapter.pipe.vae
    decoded = vae.decode(out / vae.config.scaling_factor, return_dict=False)[0].detach()
    post = getattr(getattr(adapter.pipe, "image_processor", None), "postprocess", None)
    images = post(decoded, output_type="pil", do_denormalize=[True])[0] if post is not None else _to_pil(decoded)
    if have_reffer == 0:
        images.save(database_path + str(obj_id) + ".png")       # the character's first appearance becomes its reference (:476-477)
    ret = [out, images]
    if return_saved_cross_attn:
        ret.append([{} for _ in range(eng.steps)])
    if return_box_vis:
        ret.append(images)
    if save_all_latents:
        allv = hist.to(latents.dtype).clone()
        ret.append(allv.cpu() if offload_latents_to_cpu else allv)
    return tuple(ret)


def _control_image(controlnetpipe, control_image, width, height, device, dtype):
    """``controlnetpipe.prepare_image(...)`` of :712-722 (CFG: the image twice) or, for a pipeline object without it, the same preparation:
    resize to (width, height) with LANCZOS, [0, 1], NCHW"""
    if hasattr(controlnetpipe, "prepare_image"):
        return controlnetpipe.prepare_image(image=control_image, width=width, height=height, batch_size=1, num_images_per_prompt=1, device=device,
                                            dtype=dtype, do_classifier_free_guidance=True, guess_mode=False)
    import numpy as np
    from PIL import Image
    if torch.is_tensor(control_image):
        t = control_image.to(device=device, dtype=dtype)
        t = t if t.dim() == 4 else t[None]
    else:
        pil = control_image if isinstance(control_image, Image.Image) else Image.fromarray(np.asarray(control_image))
        arr = np.asarray(pil.convert("RGB").resize((width, height), resample=Image.LANCZOS)).astype(np.float32) / 255.0
        t = torch.from_numpy(arr).permute(2, 0, 1)[None].to(device=device, dtype=dtype)
    return torch.cat([t] * 2)


@torch.no_grad()
def final_image_generation(basever, processor, controlnetpipe, tpipe, overall_prompt, overall_negative_prompt, bg_prompt, single_obj_img_list, objects,
                           repeat_ind, height, width, bg_seed, inp_mask, input_img, adapter, model_dict, latents_all, frozen_mask, bg_input_embeddings,
                           input_embeddings, num_inference_steps, frozen_steps, guidance_scale=7.5, bboxes=None, phrases=None, object_positions=None,
                           semantic_guidance_kwargs=None, offload_guidance_cross_attn_to_cpu=False, use_boxdiff=False):
    """Stage 2, the final image (reference ``models/pipelines.py:592-857``, SD-1.5 branch; caller ``theatergen.py:448-484``): same arguments, returns
    ``(latents, images uint8 [1, H, W, 3])``.  As in the reference: the frozen latents are the pasted image VAE-encoded and re-noised at ALL timesteps
    (:617-631) and OVERWRITE the caller's ``latents_all`` (:737-738); the frozen mask is the inverted, re-binarised 64 x 64 resize of ``inp_mask``
    (:605-614), not the ``frozen_mask`` argument; the start latents are fresh background noise (:632); IP scale 0.1 on the first character's image
    (:700-701); ControlNet at scale 1 on ``processor(input_img)`` every step (:705-731, 762-778).  All random draws come from the DEVICE generator
    seeded with ``bg_seed`` (:594, 625-632), in the reference's order.  Device: one captured step (ControlNet + UNet + CFG / DDIM / frozen-mask
    replace) replayed ``num_inference_steps`` times; no per-step ``.cpu()`` (:835)."""
    import numpy as np
    if basever == "xl":
        raise NotImplementedError("theatergen_amd.pipelines.final_image_generation: the 'xl' branch (T2I-Adapter on 'cuda:2', models/pipelines.py:634-697) "
                                  "is not on this path")
    unet = _own_unet(adapter, "final_image_generation")
    from .controlnet import ControlNetModel
    controlnet = controlnetpipe.controlnet
    if not isinstance(controlnet, ControlNetModel):
        raise TypeError("theatergen_amd.pipelines.final_image_generation: controlnetpipe.controlnet must be a theatergen_amd.controlnet.ControlNetModel "
                        f"(INTEGRATION.md, stage 2), got {type(controlnet).__name__}")
    vae, scheduler, dtype, dev = adapter.pipe.vae, adapter.pipe.scheduler, unet.dtype, unet.device
    generator = torch.Generator(dev).manual_seed(bg_seed)
    text_embeddings, uncond_embeddings, cond_embeddings = input_embeddings
    scheduler.set_timesteps(num_inference_steps)
    h8, w8 = int(height / 8), int(width / 8)
    # frozen mask: 1 where a character was pasted (inp_mask is 0 there)
    m = np.array(inp_mask.resize((h8, w8)).convert("L")).astype(np.float32) / 255.0
    m[m > 0] = 1
    my_mask = torch.from_numpy(1 - m).to(device=dev, dtype=torch.float32)
    # frozen latents: the pasted image through the VAE encoder, re-noised at every timestep
    img = torch.from_numpy(np.array(input_img).astype(np.float32) / 255.0)[None].permute(0, 3, 1, 2)
    myimage = (2.0 * img - 1.0).to(device=dev, dtype=dtype)
    init_latents = vae.config.scaling_factor * vae.encode(myimage).latent_dist.sample(generator=generator)
    noise = torch.randn(init_latents.shape, generator=generator, device=dev, dtype=dtype)
    my_latents = scheduler.add_noise(init_latents, noise, scheduler.timesteps).unsqueeze(1)                    # [steps, 1, C, h, w]
    my_bg = torch.randn((1, unet.config.in_channels, h8, w8), generator=generator, device=dev, dtype=dtype) * float(scheduler.init_noise_sigma)
    # conditioning: overall prompt + the first character's image tokens at scale 0.1; ControlNet sees the text rows only
    ip_embeds = _text_and_image_rows(adapter, overall_prompt, overall_negative_prompt, single_obj_img_list[0])
    adapter.set_scale(0.1)
    imagetight = _control_image(controlnetpipe, processor(np.array(input_img)), width, height, dev, dtype)
    latents_all[0] = my_bg.to(latents_all.device, latents_all.dtype)
    latents_all[1:] = my_latents.to(latents_all.device, latents_all.dtype)
    eng = _engine_of(adapter, n_img=1, height=int(imagetight.shape[-2]), width=int(imagetight.shape[-1]), num_inference_steps=num_inference_steps,
                     guidance_scale=guidance_scale, enc_len=ip_embeds.shape[1], controlnet=controlnet, controlnet_enc_len=text_embeddings.shape[1])
    eng.set_conditioning(ip_embeds.to(dev, dtype))
    eng.set_control(text_embeddings.to(dev, dtype), imagetight.to(dev, dtype), 1.0)
    eng.set_frozen(latents_all.to(dev, torch.float32), my_mask, frozen_steps)
    latents = eng.run(my_bg)[-1].to(dtype)
    image = vae.decode(1 / 0.18215 * latents).sample
    image = (image.float() / 2 + 0.5).clamp(0, 1).detach().cpu().permute(0, 2, 3, 1).numpy()
    return latents, (image * 255).round().astype("uint8")


def latent_backward_guidance(*args, **kwargs):
    """reference ``models/pipelines.py:62-128`` (same signature); implemented in ``theatergen_amd.backward``"""
    from .backward import latent_backward_guidance as impl
    return impl(*args, **kwargs)
