"""Oracle (test infrastructure): DDIM scheduler + the per-step loop-body arithmetic.

PARITY UNPINNED: ``DDIMScheduler`` is third-party ``diffusers==0.21.4`` (constructed at
reference ``generate.py:68-76``: 1000 train steps, beta 0.00085 -> 0.012 "scaled_linear",
``clip_sample=False``, ``set_alpha_to_one=False``, ``steps_offset=1``; default
``timestep_spacing="leading"``, ``prediction_type="epsilon"``, eta = 0).  Restated from the
documented 0.21.4 semantics; float64 on the host for the schedule constants.

Loop body it mirrors: reference ``models/pipelines.py:406-453`` (stage 1: CFG :441-442,
``scheduler.step`` :447) and ``:742-835`` (stage 2: frozen-mask replace :833-834).
"""
import numpy as np
import torch


class DDIMSchedule:
    def __init__(self, num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012,
                 beta_schedule="scaled_linear", set_alpha_to_one=False, steps_offset=1,
                 prediction_type="epsilon"):
        if beta_schedule == "scaled_linear":
            # diffusers builds this in torch.float32: linspace(fp32) ** 2
            betas = (torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float32) ** 2).numpy()
        elif beta_schedule == "linear":
            betas = torch.linspace(beta_start, beta_end, num_train_timesteps, dtype=torch.float32).numpy()
        else:
            raise ValueError(beta_schedule)
        alphas = 1.0 - torch.from_numpy(betas)
        self.alphas_cumprod = torch.cumprod(alphas, dim=0)          # fp32, as diffusers
        self.final_alpha_cumprod = torch.tensor(1.0) if set_alpha_to_one else self.alphas_cumprod[0]
        self.num_train_timesteps = num_train_timesteps
        self.steps_offset = steps_offset
        self.prediction_type = prediction_type
        self.init_noise_sigma = 1.0
        self.timesteps = None
        self.num_inference_steps = None

    def set_timesteps(self, n):
        """"leading" spacing: (arange(n) * (T // n)).round()[::-1] + steps_offset"""
        self.num_inference_steps = n
        ratio = self.num_train_timesteps // n
        ts = (np.arange(0, n) * ratio).round()[::-1].copy().astype(np.int64) + self.steps_offset
        self.timesteps = torch.from_numpy(ts)
        return self.timesteps

    def coeffs(self, t):
        prev_t = int(t) - self.num_train_timesteps // self.num_inference_steps
        a_t = self.alphas_cumprod[int(t)]
        a_prev = self.alphas_cumprod[prev_t] if prev_t >= 0 else self.final_alpha_cumprod
        return a_t, a_prev

    def step(self, model_output, t, sample):
        """eta = 0, no clipping.  epsilon / v_prediction."""
        a_t, a_prev = self.coeffs(t)
        b_t = 1 - a_t
        if self.prediction_type == "epsilon":
            x0 = (sample - b_t ** 0.5 * model_output) / a_t ** 0.5
            eps = model_output
        elif self.prediction_type == "v_prediction":
            x0 = a_t ** 0.5 * sample - b_t ** 0.5 * model_output
            eps = a_t ** 0.5 * model_output + b_t ** 0.5 * sample
        else:
            raise ValueError(self.prediction_type)
        direction = (1 - a_prev) ** 0.5 * eps
        return a_prev ** 0.5 * x0 + direction

    def add_noise(self, x0, noise, t):
        a = self.alphas_cumprod[int(t)]
        return a ** 0.5 * x0 + (1 - a) ** 0.5 * noise


def cfg_combine(noise_pred, guidance_scale):
    """reference models/pipelines.py:441-442 — uncond first, cond second."""
    u, c = noise_pred.chunk(2)
    return u + guidance_scale * (c - u)


def step_epilogue(sched, noise_pred, t, latents, guidance_scale, frozen_latents=None, frozen_mask=None):
    """CFG + DDIM step (+ frozen-mask replace, reference models/pipelines.py:833-834:
    ``latents = latents_all[index+1] * mask + latents * (1 - mask)``)."""
    eps = cfg_combine(noise_pred, guidance_scale)
    new = sched.step(eps, t, latents)
    if frozen_latents is not None:
        new = frozen_latents * frozen_mask + new * (1.0 - frozen_mask)
    return new
