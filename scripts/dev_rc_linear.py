"""dev: row-chain projection (tg_rc_linear) vs the LDS-tiled GEMM (tg_gemm) on the first-level shapes: parity vs fp32 + HIP-event timing
with rotating operands (working set > L2 / Infinity Cache per launch)."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from theatergen_amd import ops
from theatergen_amd.weights_pack import rc_pack, pack_ln_linear

dev = "cuda"
torch.manual_seed(0)


def timeit(fn, n=20, warm=3):
    for _ in range(warm):
        fn(0)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n):
        fn(i)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def case(M, N, K, res, ln, dtype, variants, out):
    ROT = 6
    xs = [(torch.randn(M, K, device=dev) * 1.5 + 0.3).to(dtype) for _ in range(ROT)]
    rs = [torch.randn(M, N, device=dev).to(dtype) for _ in range(ROT)] if res else [None] * ROT
    W = (torch.randn(N, K, device=dev) / K ** 0.5).to(dtype)
    bias = torch.randn(N, device=dev).to(dtype)
    gamma = (1 + 0.2 * torch.randn(K, device=dev)).to(dtype)
    beta = (0.1 * torch.randn(K, device=dev)).to(dtype)
    if ln:
        Wp, u, v = pack_ln_linear(W, bias, gamma, beta)
        wpk = rc_pack(Wp, v, u)
        ref = torch.nn.functional.linear(torch.nn.functional.layer_norm(xs[0].float(), (K,), gamma.float(), beta.float(), 1e-5), W.float(), bias.float())
    else:
        wpk = rc_pack(W, bias.float())
        ref = torch.nn.functional.linear(xs[0].float(), W.float(), bias.float())
    if res:
        ref = ref + rs[0].float()
    outs = [torch.empty(M, N, device=dev, dtype=dtype) for _ in range(ROT)]
    row = {"M": M, "N": N, "K": K, "res": res, "ln": ln, "dtype": str(dtype)}
    for v_ in variants:
        try:
            got = ops.rc_linear(xs[0], wpk, N, res=rs[0], ln_eps=1e-5 if ln else None, variant=v_)
            torch.cuda.synchronize()
            err = ((got.float() - ref).norm() / ref.norm()).item()
            t = timeit(lambda i: ops.rc_linear(xs[i % ROT], wpk, N, res=rs[i % ROT], ln_eps=1e-5 if ln else None, out=outs[i % ROT], variant=v_))
            row[f"rc_v{v_}"] = {"us": round(t, 1), "rel_l2": err}
        except RuntimeError as e:
            row[f"rc_v{v_}"] = {"error": str(e)[:200]}
    # the existing path
    if ln:
        kw = dict(ln=(u, v, 1e-5))
        t = timeit(lambda i: ops.linear(xs[i % ROT], Wp, None, out=outs[i % ROT], **kw))
        got = ops.linear(xs[0], Wp, None, **kw)
    else:
        t = timeit(lambda i: ops.linear(xs[i % ROT], W, bias, res=rs[i % ROT], out=outs[i % ROT]))
        got = ops.linear(xs[0], W, bias, res=rs[0])
    torch.cuda.synchronize()
    row["tg_gemm"] = {"us": round(t, 1), "rel_l2": ((got.float() - ref).norm() / ref.norm()).item()}
    flops = 2.0 * M * N * K
    byts = 2.0 * (M * K + M * N * (2 if res else 1))
    row["hbm_floor_us_at_6TBs"] = round(byts / 6e12 * 1e6, 1)
    row["mfma_floor_us"] = round(flops / 2.5e15 * 1e6, 1)
    print(json.dumps(row), flush=True)
    out.append(row)


if __name__ == "__main__":
    variants = [0, 1, 2, 3]
    res = []
    bf = torch.bfloat16
    case(65536, 320, 320, True, False, bf, variants, res)
    case(65536, 320, 320, False, False, bf, variants, res)
    case(65536, 960, 320, False, True, bf, variants, res)
    case(65536, 320, 320, False, True, bf, variants, res)
    case(65536, 2560, 320, False, False, bf, variants, res)
    case(65536, 320, 320, True, False, torch.float16, [0, 1], res)
    case(1000, 320, 320, True, True, bf, [0, 1], res)
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump(res, open("gpurun_out/dev_rc_linear.json", "w"), indent=1)
