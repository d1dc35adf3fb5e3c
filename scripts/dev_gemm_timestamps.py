"""dev (needs the temporarily instrumented kernel): s_memtime stamps inside gemm_glds_kernel -> phase durations per block."""
import os, sys
os.environ["TG_DEBUG_TS"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from theatergen_amd import ops
dev, dt = "cuda:0", torch.bfloat16
for (M, N, K) in [(65536, 320, 64), (65536, 320, 320), (16384, 640, 640), (4096, 1280, 1280), (65536, 2560, 320)]:
    a = torch.randn(M, K, device=dev).to(dt); w = (torch.randn(N, K, device=dev) / K ** 0.5).to(dt); b = torch.randn(N, device=dev).to(dt)
    r = torch.randn(M, N, device=dev).to(dt)
    for _ in range(3):
        ops.linear(a, w, b, res=r, out_scale=1.000001)
    torch.cuda.synchronize()
    ws = ops.workspace(64 << 20, a.device)
    tiles = (M // 128) * ((N + 127) // 128)
    ts = ws[: tiles * 16].view(torch.int64).reshape(tiles, 8).cpu().double()
    t0 = ts[:, 0].min()
    d = lambda i, j: (ts[:, j] - ts[:, i])
    span = (ts[:, 4].max() - t0)
    print(f"M={M} N={N} K={K}: bias-load {d(0,1).mean():.0f} | i=0: lds-bounce {d(1,2).mean():.0f} res-wait {d(2,3).mean():.0f} math+stores {d(3,4).mean():.0f} | "
          f"i=1: lds-bounce {d(4,5).mean():.0f} res-wait {d(5,6).mean():.0f} math+stores {d(6,7).mean():.0f} | total {d(0,7).mean():.0f}", flush=True)
