"""HBM bandwidth probes with stock torch ops (copy = read+write, fill = write only, sum = read only)."""
import torch
dev = torch.device("cuda", 0)


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


for mb in (77, 154, 308, 1024, 4096):
    n = mb * 1024 * 1024 // 2
    x = torch.empty(n, dtype=torch.float16, device=dev).normal_()
    y = torch.empty_like(x)
    c = timeit(lambda: y.copy_(x))
    f = timeit(lambda: y.fill_(1.0))
    s = timeit(lambda: x.view(torch.int32).sum())
    g = lambda b, ms: b / 1024 / 1024 / ms * 1e3 / 1e6 * 1.048576
    print(f"{mb:5d} MB: copy {c*1e3:7.1f} us {g(2*mb*2**20, c):.2f} TB/s (r+w) | fill {f*1e3:7.1f} us {g(mb*2**20, f):.2f} TB/s | "
          f"sum {s*1e3:7.1f} us {g(mb*2**20, s):.2f} TB/s")
