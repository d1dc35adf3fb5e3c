"""dev: tg_rc_ff (norm3 + GEGLU feed-forward + residual (+ proj_out + residual) in one launch) vs fp32 and vs the current launches"""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from theatergen_amd import ops, rowchain
from theatergen_amd.weights_pack import pack_geglu, pack_ln_linear, rc_pack_tiles
from dev_rc_linear import timeit

dev = "cuda"
torch.manual_seed(0)


def run(M, dtype, proj, time_it=True, inner=1280):
    C = 320
    h = (torch.randn(M, C, device=dev) * 1.2 + 0.2).to(dtype)
    x0 = torch.randn(M, C, device=dev).to(dtype)
    w1 = (torch.randn(2 * inner, C, device=dev) / C ** 0.5).to(dtype); b1 = (0.2 * torch.randn(2 * inner, device=dev)).to(dtype)
    w2 = (torch.randn(C, inner, device=dev) / inner ** 0.5).to(dtype); b2 = (0.1 * torch.randn(C, device=dev)).to(dtype)
    wp = (torch.randn(C, C, device=dev) / C ** 0.5).to(dtype); bp = (0.1 * torch.randn(C, device=dev)).to(dtype)
    gamma = (1 + 0.2 * torch.randn(C, device=dev)).to(dtype); beta = (0.1 * torch.randn(C, device=dev)).to(dtype)
    x = F.layer_norm(h.float(), (C,), gamma.float(), beta.float(), 1e-5)
    pr = x @ w1.float().T + b1.float()
    hid = pr[:, :inner] * F.gelu(pr[:, inner:])
    h3 = hid @ w2.float().T + b2.float() + h.float()
    ref = h3 @ wp.float().T + bp.float() + x0.float() if proj else h3
    s1, s2, bb2 = rowchain.pack_ff(w1, b1, gamma, beta, w2, b2)
    wpo = rc_pack_tiles(wp, bp.float()) if proj else None
    got = ops.rc_ff(h, s1, s2, bb2, inner, 1e-5, wpo=wpo, res0=x0 if proj else None)
    torch.cuda.synchronize()
    row = {"M": M, "proj": proj, "dtype": str(dtype), "rel_l2": ((got.float() - ref).norm() / ref.norm()).item(),
           "max": ((got.float() - ref).abs().max() / ref.abs().max()).item()}
    if time_it:
        out = torch.empty_like(got)
        row["rc_ff_us"] = round(timeit(lambda i: ops.rc_ff(h, s1, s2, bb2, inner, 1e-5, wpo=wpo, res0=x0 if proj else None, out=out), n=10), 1)
        for dbg in (1, 2, 4, 3, 6, 7, 7 | 8, 7 | 8 | 16):
            row[f"dbg{dbg}_us"] = round(timeit(lambda i: ops.rc_ff(h, s1, s2, bb2, inner, 1e-5, wpo=wpo, res0=x0 if proj else None, out=out, dbg=dbg), n=5), 1)
        # current path: layernorm + GEGLU GEMM + net.2 GEMM (+ proj_out GEMM)
        w1p, b1p = pack_geglu(w1, b1)
        hb = torch.empty(M, inner, device=dev, dtype=dtype); h3b = torch.empty(M, C, device=dev, dtype=dtype); ob = torch.empty(M, C, device=dev, dtype=dtype)
        def old(i):
            n = ops.layernorm(h, gamma, beta, 1e-5)
            ops.gemm(n, w1p, M, 2 * inner, C, bias=b1p, geglu=True, out=hb)
            ops.linear(hb, w2, b2, res=h, out=h3b)
            if proj:
                ops.linear(h3b, wp, bp, res=x0, out=ob)
        old(0); torch.cuda.synchronize()
        o = ob if proj else h3b
        row["old_rel_l2"] = ((o.float() - ref).norm() / ref.norm()).item()
        row["old_us"] = round(timeit(old, n=10), 1)
    print(json.dumps(row), flush=True)
    return row


if __name__ == "__main__":
    res = [run(256, torch.bfloat16, False, time_it=False, inner=128), run(1000, torch.bfloat16, True, time_it=False, inner=256),
           run(512, torch.float16, True, time_it=False), run(32768, torch.bfloat16, True)]
    os.makedirs("../gpurun_out", exist_ok=True)
    json.dump(res, open("../gpurun_out/dev_rc_ff.json", "w"), indent=1)
