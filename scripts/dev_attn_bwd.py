"""dev helper (round 5): self-attention reverse pass, recompute kernel (tg_attention_bwd) vs the materialised per-(item, head) path, SD-2.1 768^2 shapes."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from theatergen_amd import ops, backward
from theatergen_amd.attention_processor import Attention, AttnProcessor
dev, dt = "cuda:0", torch.bfloat16
def timeit(fn, iters=5):
    fn(); torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(iters): fn()
    torch.cuda.synchronize()
    return (time.time() - t0) / iters * 1e3
for (B, N, heads, d) in [(2, 9216, 5, 64), (2, 2304, 10, 64), (2, 576, 20, 64), (2, 4096, 10, 64), (2, 1024, 20, 64)]:
    C = heads * d
    q, k, v, do = [torch.randn(B * N, C, device=dev).to(dt) for _ in range(4)]
    t_new = timeit(lambda: ops.attention_bwd(q, k, v, do, B, N, heads, d, d ** -0.5))
    fl = 2.0 * B * heads * N * N * d * 7
    attn = Attention(query_dim=C, heads=heads, dim_head=d).to(dev, dt)
    h = torch.randn(B * N, C, device=dev).to(dt)
    backward.FLASH_BWD = True
    t_a = timeit(lambda: backward.attention_input_grad(attn, AttnProcessor(), h, B, N, None, do, None), 3)
    backward.FLASH_BWD = False
    t_b = timeit(lambda: backward.attention_input_grad(attn, AttnProcessor(), h, B, N, None, do, None), 2)
    print(f"B={B} N={N} heads={heads} d={d}: attention_bwd {t_new:7.3f} ms ({fl / t_new / 1e9:6.1f} TFLOP/s executed)   input_grad flash {t_a:7.3f} ms  materialised {t_b:7.3f} ms", flush=True)
