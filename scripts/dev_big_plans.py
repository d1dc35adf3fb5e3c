"""dev helper: the two other BASELINE.json plans at their native resolution (SD-2.1 768^2, SDXL 1024^2 with 16 image
tokens), CFG batch 2, random weights: finite output, eager step time, and the VAE decode time at 512^2."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from theatergen_amd import config, weights
from theatergen_amd.unet import UNet2DConditionModel
dev, dt = "cuda:0", torch.bfloat16
def timeit(fn, iters=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3
for name, T in (("sd21", 4), ("sdxl", 16)):
    cfg = config.PLANS[name]()
    t0 = time.perf_counter()
    sd = weights.random_unet_state_dict(cfg, seed=0)
    unet = UNet2DConditionModel.from_state_dict(cfg, sd, device=dev, dtype=dt, num_tokens=T, ip_scale=0.4)
    del sd
    s = cfg.sample_size
    g = torch.Generator().manual_seed(0)
    x = torch.randn(2, 4, s, s, generator=g).to(dev, dt)
    enc = (torch.randn(2, 77 + T, cfg.cross_attention_dim, generator=g) * 0.5).to(dev, dt)
    added = None
    if cfg.addition_embed_type:
        added = {"text_embeds": torch.randn(2, 1280, generator=g).to(dev, dt), "time_ids": torch.tensor([[1024., 1024., 0., 0., 1024., 1024.]] * 2, device=dev)}
    with torch.no_grad():
        out = unet(x, 500, enc, added_cond_kwargs=added).sample
        ok = bool(torch.isfinite(out.float()).all())
        ms = timeit(lambda: unet(x, 500, enc, added_cond_kwargs=added))
    print(f"{name}: latent {s}x{s}, out {tuple(out.shape)} finite={ok} std={float(out.float().std()):.3f}  {ms:.1f} ms per CFG-batch-2 call (build {time.perf_counter() - t0:.0f} s)", flush=True)
    del unet
    torch.cuda.empty_cache()
from theatergen_amd.vae import AutoencoderKL, sd_vae_config
cfg = sd_vae_config()
vae = AutoencoderKL.from_state_dict(cfg, weights.random_vae_decoder_state_dict(cfg, seed=2), device=dev, dtype=dt)
lat = torch.randn(8, 4, 64, 64, device=dev) * cfg.scaling_factor
with torch.no_grad():
    ms = timeit(lambda: vae.decode_latents(lat))
print(f"vae decode 8 x 512x512: {ms:.1f} ms = {ms / 8:.2f} ms per image ({8 * 2.48e12 / (ms * 1e-3) / 1e12:.0f} TFLOP/s)")
