"""dev helper: achievable HBM write / copy bandwidth at GEMM-output sizes (is a small-K GEMM write-bound?)."""
import torch
dev = "cuda:0"
def timeit(fn, iters=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
for mb in [10, 42, 84, 168, 335, 1000]:
    n = mb * 1000 * 1000 // 2
    x = torch.empty(n, device=dev, dtype=torch.bfloat16); y = torch.empty_like(x)
    t_fill = timeit(lambda: x.zero_())
    t_copy = timeit(lambda: y.copy_(x))
    print(f"{mb:5d} MB  zero_: {mb / t_fill / 1e3:6.2f} TB/s ({t_fill * 1e3:7.1f} us)   copy_: {2 * mb / t_copy / 1e3:6.2f} TB/s r+w ({t_copy * 1e3:7.1f} us)", flush=True)
