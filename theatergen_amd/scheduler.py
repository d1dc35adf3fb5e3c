"""DDIM scheduler (host side) with the diffusers call surface the reference loops use
(``set_timesteps`` / ``timesteps`` / ``init_noise_sigma`` / ``scale_model_input`` / ``step(...).prev_sample`` /
``add_noise`` / ``alphas_cumprod``: reference ``models/pipelines.py:382, 409-410, 447, 629-631``; constructed
at ``generate.py:68-76``).  The schedule constants are host scalars; the per-step update itself runs on the
GPU in ``tg_step_epilogue`` (CFG + DDIM + frozen-mask blend fused), fed by ``coef_table()``.

diffusers==0.21.4 semantics (not under /root/reference -> parity unpinned, see oracle/ddim.py): betas =
linspace(sqrt(b0), sqrt(b1), T, fp32)**2, "leading" timestep spacing + steps_offset, eta = 0, no clipping.
"""
from types import SimpleNamespace

import numpy as np
import torch

from . import ops


class DDIMScheduler:
    order = 1

    def __init__(self, num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear",
                 clip_sample=False, set_alpha_to_one=False, steps_offset=1, prediction_type="epsilon"):
        if clip_sample:
            raise ValueError("clip_sample=True is not supported (the reference uses clip_sample=False)")
        if beta_schedule == "scaled_linear":
            betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float32) ** 2
        elif beta_schedule == "linear":
            betas = torch.linspace(beta_start, beta_end, num_train_timesteps, dtype=torch.float32)
        else:
            raise ValueError(f"unknown beta_schedule {beta_schedule}")
        self.betas = betas
        self.alphas_cumprod = torch.cumprod(1.0 - betas, dim=0)
        self.final_alpha_cumprod = torch.tensor(1.0) if set_alpha_to_one else self.alphas_cumprod[0]
        self.init_noise_sigma = 1.0
        self.config = SimpleNamespace(num_train_timesteps=num_train_timesteps, steps_offset=steps_offset,
                                      prediction_type=prediction_type, beta_start=beta_start, beta_end=beta_end,
                                      beta_schedule=beta_schedule, clip_sample=clip_sample, set_alpha_to_one=set_alpha_to_one)
        self.num_inference_steps = None
        self.timesteps = torch.from_numpy(np.arange(0, num_train_timesteps)[::-1].copy().astype(np.int64))
        self._coef_cache = {}

    def set_timesteps(self, num_inference_steps, device=None):
        T = self.config.num_train_timesteps
        if num_inference_steps > T:
            raise ValueError("num_inference_steps > num_train_timesteps")
        self.num_inference_steps = num_inference_steps
        ratio = T // num_inference_steps
        ts = (np.arange(0, num_inference_steps) * ratio).round()[::-1].copy().astype(np.int64) + self.config.steps_offset
        self.timesteps = torch.from_numpy(ts)
        self._coef_cache = {}
        return self.timesteps

    def scale_model_input(self, sample, timestep=None):
        return sample

    def _alphas(self, t):
        t = int(t)
        prev_t = t - self.config.num_train_timesteps // self.num_inference_steps
        a_t = self.alphas_cumprod[t]
        a_prev = self.alphas_cumprod[prev_t] if prev_t >= 0 else self.final_alpha_cumprod
        return a_t, a_prev

    def coef_table(self, timesteps=None):
        """fp32 [n_steps, 4] = sqrt(a_t), sqrt(1-a_t), sqrt(a_prev), sqrt(1-a_prev) (fp32 math like diffusers)."""
        ts = self.timesteps if timesteps is None else timesteps
        rows = []
        for t in ts.tolist():
            a_t, a_prev = self._alphas(t)
            rows.append(torch.stack([a_t ** 0.5, (1 - a_t) ** 0.5, a_prev ** 0.5, (1 - a_prev) ** 0.5]))
        return torch.stack(rows).to(torch.float32).contiguous()

    def step(self, model_output, timestep, sample, eta=0.0, return_dict=True, **kw):
        """Drop-in ``scheduler.step``: one DDIM update on the GPU (no CFG, no mask)."""
        if eta != 0.0:
            raise ValueError("only eta = 0 (deterministic DDIM) is supported")
        dev = sample.device
        t = int(timestep)
        key = (t, dev)
        if key not in self._coef_cache:
            a_t, a_prev = self._alphas(t)
            coef = torch.stack([a_t ** 0.5, (1 - a_t) ** 0.5, a_prev ** 0.5, (1 - a_prev) ** 0.5]).reshape(1, 4)
            self._coef_cache[key] = (coef.to(dev, torch.float32), torch.zeros(1, dtype=torch.int32, device=dev))
        coef, idx = self._coef_cache[key]
        out = sample.detach().to(torch.float32).clone()
        ops.step_epilogue(model_output.to(torch.float32).contiguous(), out, 0.0, coef, idx, has_cfg=False, advance=False,
                          prediction_type=0 if self.config.prediction_type == "epsilon" else 1)
        out = out.to(sample.dtype)
        if not return_dict:
            return (out,)
        return SimpleNamespace(prev_sample=out)

    def add_noise_coeffs(self, timesteps):
        a = self.alphas_cumprod[timesteps]
        return a ** 0.5, (1 - a) ** 0.5

    def add_noise(self, original_samples, noise, timesteps):
        """diffusers ``add_noise``: ``sqrt(a_t) x0 + sqrt(1 - a_t) noise`` broadcast over a vector of timesteps, as the
        stage-2 path calls it with ALL 50 timesteps at once (reference models/pipelines.py:629-631: [1, C, h, w] latents
        -> [50, C, h, w]).  One ``tg_add_noise`` launch on the device; result in the dtype of ``original_samples``.

        Deliberate deviation: diffusers 0.21.4 casts ``alphas_cumprod`` to the sample dtype and rounds both products
        and their sum in half precision; here the two products and the sum are formed in fp32 and rounded ONCE to the sample
        dtype — at most one storage-dtype ulp away from the reference, closer to the exact value (the oracle,
        ``oracle/ddim.py::add_noise``, is fp32 as well; parity tests state 1 ulp / rel 2^-8 bf16, 2^-11 fp16 for this op)."""
        ts = torch.as_tensor(timesteps).reshape(-1).to("cpu", torch.long)
        ca, cb = self.add_noise_coeffs(ts)
        dev = original_samples.device
        x0 = original_samples.detach().to(torch.float32).contiguous()
        nz = noise.detach().to(device=dev, dtype=torch.float32).contiguous()
        if x0.shape[0] not in (1, ts.numel()) or nz.shape != x0.shape:
            raise ValueError("add_noise: samples / noise must have batch 1 (broadcast over the timesteps) or one row per timestep")
        if x0.shape[0] == 1:
            out = ops.add_noise(x0[0], nz[0], ca.to(dev, torch.float32), cb.to(dev, torch.float32))
        else:
            rows = [ops.add_noise(x0[i], nz[i], ca[i:i + 1].to(dev, torch.float32), cb[i:i + 1].to(dev, torch.float32))[0]
                    for i in range(ts.numel())]
            out = torch.stack(rows)
        return out.to(original_samples.dtype)
