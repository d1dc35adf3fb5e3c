"""dev experiment: one engine of 8 images vs two engines of 4 images replayed concurrently on two streams (do the
second stream's kernels fill the tails / small grids of the first?)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from theatergen_amd import config, weights
from theatergen_amd.unet import UNet2DConditionModel
from theatergen_amd.pipelines import DenoiseEngine
dev, dt = torch.device("cuda:0"), torch.bfloat16
cfg = config.sd15()
sd = weights.random_unet_state_dict(cfg, seed=0)
STEPS = 10
def make(n):
    u = UNet2DConditionModel.from_state_dict(cfg, sd, device=dev, dtype=dt, num_tokens=4, ip_scale=0.4)
    e = DenoiseEngine(u, None, n_img=n, height=512, width=512, num_inference_steps=STEPS, guidance_scale=7.5, enc_len=81)
    e.set_conditioning(torch.randn(2 * n, 81, 768, device=dev).to(dt) * 0.5)
    lat = torch.randn(n, 4, 64, 64, device=dev)
    e.run(lat)          # capture + warm
    return e, lat
def time_it(fn, iters=3):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters
e8, l8 = make(8)
t8 = time_it(lambda: e8.run(l8))
print(f"1 x 8 images: {t8 * 1e3:.1f} ms per {STEPS} steps -> {8 / (t8 * 50 / STEPS):.3f} img/s (50-step equiv)", flush=True)
for n, k in ((4, 2), (2, 4), (1, 8)):
    engs = [make(n) for _ in range(k)]
    streams = [torch.cuda.Stream(device=dev) for _ in range(k)]
    def seq():
        for e, l in engs: e.run(l)
    def par():
        cur = torch.cuda.current_stream(dev)
        for (e, l), s in zip(engs, streams):
            s.wait_stream(cur)
            with torch.cuda.stream(s):
                e._reset(l)
        for _ in range(STEPS):
            for (e, l), s in zip(engs, streams):
                with torch.cuda.stream(s):
                    e.graph.replay()
        for s in streams: cur.wait_stream(s)
    ts, tp = time_it(seq), time_it(par)
    print(f"{k} x {n} images: sequential {ts * 1e3:.1f} ms -> {8 / (ts * 50 / STEPS):.3f} img/s; concurrent streams {tp * 1e3:.1f} ms -> {8 / (tp * 50 / STEPS):.3f} img/s", flush=True)
    del engs
    torch.cuda.empty_cache()
