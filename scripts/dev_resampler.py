"""Resampler (SDXL-Plus and SD-1.5-Plus geometry, batch 2 = cond + zero image): generic launch-per-op path vs the skinny latent path, eager and hipGraph-replayed."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from theatergen_amd import weights as W  # noqa: E402
from theatergen_amd.resampler import Resampler  # noqa: E402

DEV = "cuda:0"


def ev_us(fn, iters=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return round(e0.elapsed_time(e1) * 1e3 / iters, 1)


for name, kw, dt in [("sdxl_plus", dict(dim=1280, depth=4, dim_head=64, heads=20, num_queries=16, embedding_dim=1280, output_dim=2048, ff_mult=4), torch.float16),
                     ("sd15_plus", dict(dim=768, depth=4, dim_head=64, heads=12, num_queries=16, embedding_dim=1280, output_dim=768, ff_mult=4), torch.bfloat16)]:
    out = dict(config=name, dtype=str(dt))
    x = torch.randn(2, 257, 1280, generator=torch.Generator().manual_seed(1)).to(DEV, dt)
    x[1].zero_()
    if os.environ.get("RS_ONLY") and os.environ["RS_ONLY"] != name:
        continue
    for mode in (os.environ.get("RS_MODES", "01")):
        os.environ["TG_RESAMPLER_SKINNY"] = mode
        rs = Resampler(**kw)
        rs.load_state_dict(W.random_resampler_state_dict(seed=401, **kw))
        rs = rs.to(DEV, dt)
        with torch.no_grad():
            y = rs(x)
            tag = "skinny" if mode == "1" else "generic"
            out[tag + "_eager_us"] = ev_us(lambda: rs(x))
            rs.graphed(x)
            out[tag + "_graph_us"] = ev_us(lambda: rs.graphed(x))
            out[tag + "_y"] = y.float().cpu()
    if "skinny_y" in out and "generic_y" in out:
        out["rel_l2_between_paths"] = float((out["skinny_y"] - out["generic_y"]).norm() / out["generic_y"].norm())
    out.pop("skinny_y", None)
    out.pop("generic_y", None)
    params = sum(p.numel() for p in rs.parameters())
    out["weights_MB"] = round(params * 2 / 1e6, 1)
    out["hbm_floor_us_at_8TBs"] = round(params * 2 / 8e6, 1)
    print(json.dumps(out))
