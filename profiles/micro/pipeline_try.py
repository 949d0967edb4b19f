"""How much would cross-step pipelining buy?  ED and VAE forwards of K consecutive batches on two streams with no join
between steps (upper bound for an asynchronous serving queue) against the synchronous two-stream step of bench.py."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from genconvit_amd import synth, _lib
dev = torch.device("cuda", 0)
torch.set_grad_enabled(False)
model, sds = bench.build_models("genconvit", torch.float16, 128, dev)
x = synth.make_frames(128, name="bench_frames_r0").to(dev).half()
eps = synth.make_eps(128, name="bench_eps_r0").to(dev)
K = 20
def sync_steps():
    for _ in range(K):
        out = _lib.vote(model(x, eps=eps))
    return out
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def free_running():
    cur = torch.cuda.current_stream()
    s1.wait_stream(cur); s2.wait_stream(cur)
    for _ in range(K):
        with torch.cuda.stream(s1):
            a = model.model_ed(x)
        with torch.cuda.stream(s2):
            b = model.model_vae(x, eps=eps, want_recon=False)[0]
    cur.wait_stream(s1); cur.wait_stream(s2)
    return a, b
for name, fn in (("synchronous steps", sync_steps), ("free-running streams", free_running)):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter(); fn(); torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / K * 1e3
    print(f"{name:24s}: {dt:.3f} ms per batch of 128  ({128 / dt * 1e3:.0f} frames/s)")
