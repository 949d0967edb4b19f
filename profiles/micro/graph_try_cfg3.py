"""HIP-graph replay of the vae B=32 bf16 step (BASELINE configs[2]) against eager launches: is that configuration bound by
the host's launch rate (about 180 launches of ~10 us per step)?"""
import sys, time, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from genconvit_amd import synth, _lib
dev = torch.device("cuda", 0)
torch.set_grad_enabled(False)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
model, sds = bench.build_models("vae", torch.bfloat16, B, dev)
x = synth.make_frames(B, name="bench_frames_r0").to(dev).bfloat16()
eps = synth.make_eps(B, name="bench_eps_r0").to(dev)
def step():
    return _lib.vote(model(x, eps=eps))
def timeit(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): out = fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
print("eager: %.3f ms/step" % timeit(step))
t0 = time.perf_counter()
for _ in range(50): step()
t1 = time.perf_counter(); torch.cuda.synchronize()
print("host time to enqueue one step: %.3f ms" % ((t1 - t0) / 50 * 1e3))
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3): step()
torch.cuda.current_stream().wait_stream(s)
try:
    with torch.cuda.graph(g):
        out = step()
    print("graph: %.3f ms/step" % timeit(g.replay))
except Exception as e:
    print("graph capture failed:", repr(e)[:300])
