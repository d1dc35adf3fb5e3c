"""dev: where does a bench 'story' step spend time outside the 50 graph-replayed denoise steps?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from theatergen_amd import story
from theatergen_amd.pipelines import DenoiseEngine
dev, dt = torch.device("cuda:0"), torch.bfloat16
torch.cuda.set_device(dev)
cfg, sd, unet, adapter = bench.build_model("sd15", dt, dev, num_tokens=4)
ns, cb, T = 2, 8, 4
sub = cb // ns
engines = [DenoiseEngine(unet, None, n_img=sub, height=512, width=512, num_inference_steps=50, guidance_scale=7.5, enc_len=81) for _ in range(ns)]
shared = story.shared_conditioning(cfg.cross_attention_dim, T, dt, dev)
jobs = story.story_jobs(0)
char_ids = sorted({j.char_id for j in jobs})
img_tok = story.character_image_tokens(char_ids, cfg.cross_attention_dim, T, dt, dev)
cidx = {c: i for i, c in enumerate(char_ids)}
enc = story.job_conditioning(jobs, shared, img_tok, cidx, cfg.cross_attention_dim, dt, dev)
lat = story.job_latents(jobs, adapter)
def sync_t():
    torch.cuda.synchronize(); return time.perf_counter()
for rep in range(3):
    t0 = sync_t()
    lats = []
    for k, e in enumerate(engines):
        rows = list(range(k * sub, (k + 1) * sub)) + list(range(cb + k * sub, cb + (k + 1) * sub))
        e.set_conditioning(enc[rows])
        lats.append(lat[k * sub:(k + 1) * sub])
    t1 = sync_t()
    hists = DenoiseEngine.run_concurrent(engines, lats)
    t2 = sync_t()
    fin = torch.stack([h[-1] for h in hists])
    t3 = sync_t()
    print(f"rep {rep}: set_conditioning {1e3 * (t1 - t0):.1f} ms, run_concurrent {1e3 * (t2 - t1):.1f} ms, collect {1e3 * (t3 - t2):.2f} ms", flush=True)
