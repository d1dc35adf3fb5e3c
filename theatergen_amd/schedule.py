"""Timestep schedule helpers (reference ``utils/schedule.py``): ``get_fast_schedule`` :4-8 keeps the first
``fast_after_steps`` timesteps and then every ``fast_rate``-th one (off by default in the flow, theatergen.py:319)."""
import torch


def get_fast_schedule(origial_timesteps, fast_after_steps, fast_rate):
    if fast_after_steps >= len(origial_timesteps) - 1:
        return origial_timesteps
    return torch.cat((origial_timesteps[:fast_after_steps], origial_timesteps[fast_after_steps + 1::fast_rate]), dim=0)
