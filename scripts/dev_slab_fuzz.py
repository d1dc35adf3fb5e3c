"""dev: randomized shapes through the slab conv kernel (force_tile 11 / 12, with / without GroupNorm prologue, bias / vector / residual on
or off) against the halo / implicit-GEMM path on the same inputs: hang or mismatch hunting.  Run under `timeout`."""
import math, os, random, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from theatergen_amd import ops
dev = "cuda:0"
random.seed(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
n_ok = 0
for it in range(int(sys.argv[2]) if len(sys.argv) > 2 else 60):
    dt = random.choice([torch.bfloat16, torch.float16])
    w = random.choice([16, 32, 64, 96, 128, 96, 128])     # 96 / 128: patch tiles (round 3): 4 x 32 and 2 x 64 patches
    rows_per_tile = {96: 4, 128: 2}.get(w, 128 // w)
    h = rows_per_tile * random.randint(1, max(1, (64 if w <= 64 else 24) // rows_per_tile // 2))
    B = random.randint(1, 5)
    cin = 64 * random.randint(1, 6)
    c1 = 64 * random.randint(0, 3) if random.random() < 0.4 else 0
    cout = 320 * random.randint(1, 3)
    ft = random.choice([11, 12])
    if ft == 12 and ((cin + c1) // 64 < 2 or w > 64):      # patch tiles are never split
        ft = 11
    g = torch.Generator().manual_seed(it)
    M = B * h * w
    x0 = torch.randn(M, cin, generator=g).to(dev, dt)
    x1 = torch.randn(M, c1, generator=g).to(dev, dt) if c1 else None
    W = (torch.randn(cout, 9 * (cin + c1), generator=g) / math.sqrt(9 * (cin + c1))).to(dev, dt)
    kw = {}
    if random.random() < 0.7: kw["bias"] = torch.randn(cout, generator=g).to(dev, dt)
    if random.random() < 0.5: kw.update(bvec=torch.randn(B, cout, generator=g).to(dev, dt), rows_per_batch=h * w)
    if random.random() < 0.5: kw["res"] = torch.randn(M, cout, generator=g).to(dev, dt)
    pro = random.random() < 0.5
    desc = f"{str(dt)[6:]} B={B} {h}x{w} {cin}+{c1}->{cout} ft={ft} pro={pro} {sorted(kw)}"
    if pro and (cin + c1) % 32 == 0:
        gam, bet = torch.randn(cin + c1, generator=g).to(dev, dt), torch.randn(cin + c1, generator=g).to(dev, dt)
        coef = ops.groupnorm_coef(x0, B, h * w, 32, 1e-5, gam, bet, x1=x1)
        out = ops.conv3x3(x0, W, B, h, w, cin, x1=x1, c1=c1, a_coef=coef, a_silu=True, force_tile=ft, **kw)
        hn = ops.groupnorm(x0, B, h * w, 32, 1e-5, gam, bet, silu=True, x1=x1)
        os.environ["TG_GEMM_FLAGS"] = "128"
        ref = ops.conv3x3(hn, W, B, h, w, cin + c1, **kw)
    else:
        out = ops.conv3x3(x0, W, B, h, w, cin, x1=x1, c1=c1, force_tile=ft, **kw)
        os.environ["TG_GEMM_FLAGS"] = "128"
        ref = ops.conv3x3(x0, W, B, h, w, cin, x1=x1, c1=c1, **kw)
    del os.environ["TG_GEMM_FLAGS"]
    torch.cuda.synchronize()
    err = ((out.float() - ref.float()).norm() / ref.float().norm()).item()
    tol = 6e-3 if dt == torch.bfloat16 else 8e-4
    status = "ok" if err < tol and torch.isfinite(out.float()).all() else "MISMATCH"
    n_ok += status == "ok"
    print(f"{it:3d} {status} rel-L2 {err:.2e}  {desc}", flush=True)
print("passed", n_ok)
