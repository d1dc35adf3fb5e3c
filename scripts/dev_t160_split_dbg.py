"""dev: where does a K-split 128 x 160 launch go wrong?  error map by (32-row block, 32-column block) of the first tiles"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from theatergen_amd import ops
dev, dt = "cuda:0", torch.bfloat16
torch.manual_seed(0)
for (M, N, K, ft, fs) in [(256, 320, 1024, 23, 2), (256, 320, 1024, 21, 2), (256, 320, 1024, 1, 2), (16384, 640, 2560, 23, 3), (16384, 640, 2560, 0, 0)]:
    a = torch.randn(M, K, device=dev).to(dt); w = (torch.randn(N, K, device=dev) / K ** 0.5).to(dt)
    ref = a.float() @ w.float().t()
    try:
        out = ops.gemm(a, w, M, N, K, force_tile=ft, force_split_k=fs).float()
    except RuntimeError as e:
        print(M, N, K, ft, fs, "ERR", str(e)[:100]); continue
    err = (out - ref).abs()
    print(f"M{M} N{N} K{K} ft{ft} fs{fs} plan {ops.gemm(a, w, M, N, K, force_tile=ft, force_split_k=fs, plan_only=True)} max err {err.max().item():.3e} ref max {ref.abs().max().item():.2f}")
    blk = err[:256, :320].reshape(8, 32, 10, 32).amax(dim=(1, 3))
    for r in range(8):
        print("   " + " ".join(f"{v:7.2e}" for v in blk[r].tolist()))
