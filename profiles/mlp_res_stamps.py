#!/usr/bin/env python3
"""Diagnostic: s_memtime stamps of fused_mlp_res_kernel (library built with -DGCV_MLP_STAMPS=1, path in GCV_LIB_PATH):
wave 0 of workgroups 0..63, third tile of the wave.  Prints median tick deltas between phases."""
import ctypes, math, os, sys, statistics
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from genconvit_amd import _lib
lib = _lib.load()
dev = torch.device("cuda", 0)
C, H = 96, 56
M = 256 * H * H
R = lambda *s: (torch.rand(*s, device=dev) * 2 - 1)
x, res = R(M, C).half(), R(M, C).half()
w1 = (R(4 * C, C) / math.sqrt(C)).half(); w2 = R(C, 4 * C) / math.sqrt(4 * C)
b1, b2, g = R(4 * C), R(C), R(C)
for _ in range(3):
    _lib.check(lib.gcv_k_fused_mlp(_lib.GCV_F16, C, x.data_ptr(), w1.data_ptr(), b1.data_ptr(), w2.data_ptr(), b2.data_ptr(),
                                   g.data_ptr(), res.data_ptr(), res.data_ptr(), M, _lib.current_stream_ptr(dev)), "mlp")
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * (64 * 16))()
raw = ctypes.CDLL(_lib.LIB_PATH)
assert raw.gcv_debug_read_stamps(buf, 64 * 16) == 0
names = {0: "tile start (x loads issued)", 1: "x landed, GEMM1(0), GEMM1(1) issued", 2: "h(0) done, first step starts",
         3: "after step g=1", 4: "after step g=2", 5: "after step g=10 (loop end)", 7: "tail done", 6: "epilogue stores issued"}
order = [0, 1, 2, 3, 4, 5, 7, 6]
rows = [[buf[b * 16 + i] for i in range(16)] for b in range(64)]
rows = [r for r in rows if r[0] and r[6]]
prev = 0
for i in order:
    med = statistics.median(r[i] - r[0] for r in rows)
    print(f"{names[i]:40s} t = {med:9.0f} ticks (+{med - prev:7.0f})")
    prev = med
rt = [(r[6] - r[0]) / max(r[9] - r[8], 1) * 100e6 / 1e9 for r in rows if r[9] > r[8]]
if rt:
    print(f"in-kernel shader clock over the stamped tile: {statistics.median(rt):.2f} GHz (s_memtime / s_memrealtime x 100 MHz)")
for w, off in (("wave 0", 0), ("wave 7", 3)):
    v = [(r[11 + off] - r[10 + off], r[12 + off] - r[10 + off]) for r in rows if r[12 + off]]
    if v:
        print(f"{w}: weights in LDS after {statistics.median(a for a, b in v):.0f} ticks, wave done after {statistics.median(b for a, b in v):.0f} ticks")
