"""dev: tg_rc_front (GroupNorm + proj_in + LayerNorm1 + q|k|v in one launch) vs fp32 and vs the current launches"""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from theatergen_amd import ops
from theatergen_amd.weights_pack import pack_ln_linear, rc_pack_tiles
from dev_rc_linear import timeit

dev = "cuda"
torch.manual_seed(0)


def run(B, N, dtype, time_it=True):
    C = 320
    M = B * N
    x = (torch.randn(M, C, device=dev) * 1.3 + 0.4).to(dtype)
    gg = (1 + 0.2 * torch.randn(C, device=dev)).to(dtype); gb = (0.1 * torch.randn(C, device=dev)).to(dtype)
    win = (torch.randn(C, C, device=dev) / C ** 0.5).to(dtype); bin_ = (0.1 * torch.randn(C, device=dev)).to(dtype)
    wq = (torch.randn(3 * C, C, device=dev) / C ** 0.5).to(dtype)
    lg = (1 + 0.2 * torch.randn(C, device=dev)).to(dtype); lb = (0.1 * torch.randn(C, device=dev)).to(dtype)
    xg = F.group_norm(x.float().reshape(B, N, C).permute(0, 2, 1), 32, gg.float(), gb.float(), 1e-6).permute(0, 2, 1).reshape(M, C)
    y = xg @ win.float().T + bin_.float()
    qkv = F.layer_norm(y, (C,), lg.float(), lb.float(), 1e-5) @ wq.float().T
    coef = ops.groupnorm_coef(x, B, N, 32, 1e-6, gg, gb)
    winp = rc_pack_tiles(win, bin_.float())
    Wp, u, v = pack_ln_linear(wq, None, lg, lb)
    wqp = rc_pack_tiles(Wp, v, u)
    gy, gqk, gvt, ldt = ops.rc_front(x, coef, winp, wqp, N, 1e-5)
    torch.cuda.synchronize()
    rel = lambda a, b: ((a.float() - b).norm() / b.norm()).item()
    row = {"B": B, "N": N, "dtype": str(dtype), "y": rel(gy, y), "qk": rel(gqk, qkv[:, :640]),
           "v": rel(gvt[:, :, :N].permute(0, 2, 1).reshape(M, C), qkv[:, 640:])}
    if time_it:
        row["rc_front_us"] = round(timeit(lambda i: ops.rc_front(x, coef, winp, wqp, N, 1e-5, y=gy, qk=gqk, vt=gvt), n=10), 1)
        for dbg in (256, 7, 256 | 7):
            row[f"dbg{dbg}"] = round(timeit(lambda i: ops.rc_front(x, coef, winp, wqp, N, 1e-5, y=gy, qk=gqk, vt=gvt, dbg=dbg), n=10), 1)
        qk2 = torch.empty(M, 640, device=dev, dtype=dtype); vt2 = torch.empty(B, C, ldt, device=dev, dtype=dtype)
        from theatergen_amd.weights_pack import rc_pack
        wlin = rc_pack(win, bin_.float())
        def old(i):
            n = ops.groupnorm(x, B, N, 32, 1e-6, gg, gb)
            yy = ops.rc_linear(n, wlin, C)
            ops.gemm(yy, Wp, M, 3 * C, C, rows_per_batch=N, out=qk2, n_split=640, out_t=vt2, ldt=ldt, ln=(u, v, 1e-5))
        old(0)
        row["old_us"] = round(timeit(old, n=10), 1)
        row["coef_us"] = round(timeit(lambda i: ops.groupnorm_coef(x, B, N, 32, 1e-6, gg, gb), n=10), 1)
        row["groupnorm_us"] = round(timeit(lambda i: ops.groupnorm(x, B, N, 32, 1e-6, gg, gb), n=10), 1)
    print(json.dumps(row), flush=True)
    return row


if __name__ == "__main__":
    run(2, 256, torch.bfloat16, False)
    run(3, 128, torch.float16, False)
    run(16, 4096, torch.bfloat16)
    run(2, 4096, torch.float16)
