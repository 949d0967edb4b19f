#!/usr/bin/env python3
"""Diagnostic: s_memtime timeline of xs_mlp_kernel workgroups 0..63, wave 0, first pass (library built with
-DGCV_XM_STAMPS=1, path in GCV_LIB_PATH).  usage: xm_stamps.py <96|192>"""
import ctypes, math, os, sys, statistics
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from genconvit_amd import _lib
lib = _lib.load()
dev = torch.device("cuda", 0)
C = int(sys.argv[1]) if len(sys.argv) > 1 else 192
H = {96: 56, 192: 28}[C]; M = int(os.environ.get("GCV_MB_NIMG", "256")) * H * H
R = lambda *s: (torch.rand(*s, device=dev) * 2 - 1)
x, res = R(M, C).half(), R(M, C).half()
w1 = (R(4 * C, C) / math.sqrt(C)).half(); w2 = R(C, 4 * C) / math.sqrt(4 * C)
b1, b2, g = R(4 * C), R(C), R(C)
for _ in range(3):
    _lib.check(lib.gcv_k_fused_mlp(_lib.GCV_F16, C, x.data_ptr(), w1.data_ptr(), b1.data_ptr(), w2.data_ptr(), b2.data_ptr(),
                                   g.data_ptr(), res.data_ptr(), res.data_ptr(), M, _lib.current_stream_ptr(dev)), "mlp")
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * (64 * 64))()
raw = ctypes.CDLL(_lib.LIB_PATH)
assert raw.gcv_debug_read_xm_stamps(buf, 64 * 64) == 0
rows = [[buf[b * 64 + i] for i in range(64)] for b in range(64)]
med = lambda i, j: statistics.median([r[i] - r[j] for r in rows if r[i] and r[j]])
names = ["x loads issued", "pass 0: x landed, first step starts", "two steps done", "main loop done", "drain done", "epilogue done (pass 0)", "kernel end"]
for i in range(1, 7):
    print(f"{names[i]:40s} +{med(i, i - 1):9.0f} cycles  (t = {med(i, 0):9.0f})")
