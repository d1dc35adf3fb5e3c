"""dev: tg_rc_linear at K = 640 (the 32 x 32 level's projections) vs fp32 and vs tg_gemm"""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from theatergen_amd import ops
from theatergen_amd.weights_pack import pack_ln_linear, rc_pack_tiles
from dev_rc_linear import timeit

dev = "cuda"
torch.manual_seed(0)


def case(M, N, res, ln, dtype):
    K, ROT = 640, 4
    xs = [(torch.randn(M, K, device=dev) * 1.5 + 0.3).to(dtype) for _ in range(ROT)]
    rs = [torch.randn(M, N, device=dev).to(dtype) for _ in range(ROT)] if res else [None] * ROT
    W = (torch.randn(N, K, device=dev) / K ** 0.5).to(dtype)
    bias = torch.randn(N, device=dev).to(dtype)
    gamma = (1 + 0.2 * torch.randn(K, device=dev)).to(dtype); beta = (0.1 * torch.randn(K, device=dev)).to(dtype)
    if ln:
        Wp, u, v = pack_ln_linear(W, bias, gamma, beta)
        wpk = rc_pack_tiles(Wp, page=False)
        ref = F.linear(F.layer_norm(xs[0].float(), (K,), gamma.float(), beta.float(), 1e-5), W.float(), bias.float())
    else:
        wpk, v, u = rc_pack_tiles(W, page=False), bias.float().contiguous(), None
        ref = F.linear(xs[0].float(), W.float(), bias.float())
    if res:
        ref = ref + rs[0].float()
    outs = [torch.empty(M, N, device=dev, dtype=dtype) for _ in range(ROT)]
    got = ops.rc_linear(xs[0], wpk, N, res=rs[0], ln_eps=1e-5 if ln else None, v=v, u=u)
    torch.cuda.synchronize()
    row = {"M": M, "N": N, "res": res, "ln": ln, "dtype": str(dtype), "rel_l2": ((got.float() - ref).norm() / ref.norm()).item()}
    row["rc_us"] = round(timeit(lambda i: ops.rc_linear(xs[i % ROT], wpk, N, res=rs[i % ROT], ln_eps=1e-5 if ln else None, v=v, u=u, out=outs[i % ROT])), 1)
    if ln:
        row["tg_gemm_us"] = round(timeit(lambda i: ops.linear(xs[i % ROT], Wp, None, ln=(u, v, 1e-5), out=outs[i % ROT])), 1)
    else:
        row["tg_gemm_us"] = round(timeit(lambda i: ops.linear(xs[i % ROT], W, bias, res=rs[i % ROT], out=outs[i % ROT])), 1)
    print(json.dumps(row), flush=True)


if __name__ == "__main__":
    bf = torch.bfloat16
    case(1000, 128, True, True, bf)
    case(16384, 640, True, False, bf)
    case(16384, 640, False, False, bf)
    case(16384, 640, False, True, bf)
    case(16384, 640, True, False, torch.float16)
