import sys, time, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from genconvit_amd import synth, _lib
from genconvit_amd.model.genconvit import GenConViT
dev = torch.device("cuda", 0)
torch.set_grad_enabled(False)
model, sds = bench.build_models("genconvit", torch.float16, 128, dev)
x = synth.make_frames(128, name="bench_frames_r0").to(dev).half()
eps = synth.make_eps(128, name="bench_eps_r0").to(dev)
def step():
    return _lib.vote(model(x, eps=eps))
def timeit(fn, n=20):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): out = fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
for conc in (True, False):
    GenConViT.concurrent = conc
    print("eager  concurrent=%s: %.3f ms/step" % (conc, timeit(step)))
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3): step()
    torch.cuda.current_stream().wait_stream(s)
    try:
        with torch.cuda.graph(g):
            out = step()
        print("graph  concurrent=%s: %.3f ms/step" % (conc, timeit(g.replay)))
    except Exception as e:
        print("graph capture failed:", repr(e)[:300])
