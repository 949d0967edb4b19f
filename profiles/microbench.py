#!/usr/bin/env python3
"""Micro-benchmarks of single kernels at the genconvit B=128 fp16 shapes (HIP-event timing), for use
under rocprofv3 --pmc.   usage: microbench.py <dwconv96|dwconv192|mlp96|mlp192|pw1_384|pw2_384> [iters]"""
import ctypes, math, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from genconvit_amd import _lib

which = sys.argv[1]
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
dt, code = torch.float16, _lib.GCV_F16
dev = torch.device("cuda", 0)
lib = _lib.load()
st = lambda: _lib.current_stream_ptr(dev)
R = lambda *s: (torch.rand(*s, device=dev) * 2 - 1)
nimg = int(os.environ.get("GCV_MB_NIMG", "256"))


def run(fn, flops, bytes_):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    print(f"{which}: {ms*1e3:.1f} us/launch  {flops/ms/1e9:.1f} TFLOP/s  {bytes_/ms/1e6:.1f} GB/s (algorithmic)")


if which.startswith("dwconv"):
    C = int(which[6:])
    H = {96: 56, 192: 28, 384: 14, 768: 7}[C]
    x = R(nimg, H, H, C).to(dt)
    y = torch.empty_like(x)
    w, b, lw, lb = R(49, C), R(C), R(C) + 1.5, R(C)
    fn = lambda: _lib.check(lib.gcv_k_dwconv7_ln(code, x.data_ptr(), w.data_ptr(), b.data_ptr(), lw.data_ptr(),
                                                 lb.data_ptr(), y.data_ptr(), nimg, H, H, C, 1e-6, st()), "dw")
    n = nimg * H * H * C
    run(fn, 2.0 * 49 * n, 4.0 * n)
elif which.startswith("mlp"):
    C = int(which[3:])
    H = {96: 56, 192: 28, 384: 14}[C]
    M = nimg * H * H
    x = R(M, C).to(dt)
    res = R(M, C).to(dt)
    w1 = (R(4 * C, C) / math.sqrt(C)).to(dt)
    w2 = R(C, 4 * C) / math.sqrt(4 * C)
    b1, b2, g = R(4 * C), R(C), R(C)
    # weights packed once inside the library, the MLP kernel(s) timed alone with HIP events (gcv_k_fused_mlp_timed)
    ms3 = (ctypes.c_float * 3)()
    _lib.check(lib.gcv_k_fused_mlp_timed(code, C, x.data_ptr(), w1.data_ptr(), b1.data_ptr(), w2.data_ptr(), b2.data_ptr(),
                                         g.data_ptr(), res.data_ptr(), res.data_ptr(), M, iters, ms3, st()), "mlp")
    ms = ms3[0]
    extra = f"  (pw1 {ms3[1]*1e3:.1f} us, pw2 {ms3[2]*1e3:.1f} us)" if ms3[1] > 0 else ""
    print(f"{which}: {ms*1e3:.1f} us/launch  {16.0*M*C*C/ms/1e9:.1f} TFLOP/s  {6.0*M*C/ms/1e6:.1f} GB/s (algorithmic){extra}")
else:
    kind, C = which.split("_")
    C = int(C)
    H = {96: 56, 192: 28, 384: 14, 768: 7}[C]
    M = nimg * H * H
    if kind == "pw1":
        N, K, epi, act = 4 * C, C, _lib.EPI_BIAS_ACT, _lib.ACT_GELU
    else:
        N, K, epi, act = C, 4 * C, _lib.EPI_RESID, _lib.ACT_NONE
    A = R(M, K).to(dt)
    W = (R(N, K) / math.sqrt(K)).to(dt)
    Cm = R(M, N).to(dt)
    bias, gamma = R(N), R(N)
    g = _lib.GemmArgs(A.data_ptr(), W.data_ptr(), Cm.data_ptr(), bias.data_ptr(), gamma.data_ptr(), Cm.data_ptr(), None,
                      M, N, K, K, N, act, 1, 0, 0, 0, 0, 0)
    fn = lambda: _lib.check(lib.gcv_k_gemm(code, _lib.A_PLAIN, epi, ctypes.byref(g), st()), "gemm")
    run(fn, 2.0 * M * N * K, 2.0 * (M * K + N * K + M * N * (2 if kind == "pw2" else 1)))
