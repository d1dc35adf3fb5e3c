"""dev helper: full SD-1.5 UNet on the GPU vs the CPU oracle, with timestamps (not part of the test suite)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
t00 = time.time()
def log(*a):
    print(f"[{time.time()-t00:7.1f}s]", *a, flush=True)
from theatergen_amd import config, weights
from theatergen_amd.unet import UNet2DConditionModel
dtype = torch.bfloat16
cfg = config.sd15()
sd = weights.random_unet_state_dict(cfg, seed=0); log("weights")
unet = UNet2DConditionModel.from_state_dict(cfg, sd, device="cuda:0", dtype=dtype, num_tokens=4, ip_scale=0.4); log("unet built")
g = torch.Generator().manual_seed(2)
x = torch.randn(2, 4, 64, 64, generator=g); enc = torch.randn(2, 81, 768, generator=g) * 0.5
out = unet(x.to("cuda:0", dtype), 981, enc.to("cuda:0", dtype), out_dtype=torch.float32).sample
torch.cuda.synchronize(); log("gpu fwd 1", out.float().std().item())
t0 = time.time()
for _ in range(5):
    out = unet(x.to("cuda:0", dtype), 981, enc.to("cuda:0", dtype), out_dtype=torch.float32).sample
torch.cuda.synchronize(); log("gpu fwd eager avg ms", (time.time()-t0)/5*1e3)
if "--oracle" in sys.argv:
    from oracle import unet as ou
    torch.set_num_threads(min(32, os.cpu_count() or 8))
    sd_r = {k: v.to(dtype).float() for k, v in sd.items()}
    ref = ou.unet_forward(cfg, sd_r, x.to(dtype).float(), 981, enc.to(dtype).float(), ip_scale=0.4, num_tokens=4); log("oracle fwd")
    err = (out.cpu() - ref).abs().max().item() / ref.abs().max().item()
    log("rel max err", err)
