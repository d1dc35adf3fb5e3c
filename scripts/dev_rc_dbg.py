import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from theatergen_amd import ops
from theatergen_amd.weights_pack import rc_pack
from dev_rc_linear import timeit
dev = "cuda"
bf = torch.bfloat16
for (M, N) in [(65536, 2560), (65536, 320)]:
    K = 320
    ROT = 4
    xs = [torch.randn(M, K, device=dev).to(bf) for _ in range(ROT)]
    W = (torch.randn(N, K, device=dev) / K ** 0.5).to(bf)
    wpk = rc_pack(W, torch.zeros(N, device=dev))
    outs = [torch.empty(M, N, device=dev, dtype=bf) for _ in range(ROT)]
    for base in (1, 5):
        row = {}
        for dbg in (0, 1, 2, 4, 8, 1 | 2, 1 | 8, 1 | 2 | 8, 1 | 2 | 4 | 8, 1 | 4):
            v_ = base | (dbg << 8)
            t = timeit(lambda i: ops.rc_linear(xs[i % ROT], wpk, N, out=outs[i % ROT], variant=v_))
            row[dbg] = round(t, 1)
        print(M, N, "variant", base, "dbg(1 nostore,2 nodma,4 nobarrier,8 nomfma) ->", row, flush=True)
