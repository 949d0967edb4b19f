#!/usr/bin/env python3
"""Diagnostic: read s_memtime stamps of fused_mlp_kernel workgroups (library built with -DGCV_MLP_STAMPS=1,
path in GCV_LIB_PATH).  Prints median cycle deltas between phases (100 MHz... s_memtime ticks = shader cycles)."""
import ctypes, math, os, sys, statistics
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from genconvit_amd import _lib
lib = _lib.load()
dev = torch.device("cuda", 0)
C = int(sys.argv[1]) if len(sys.argv) > 1 else 96
H = {96: 56, 192: 28}[C]; M = 256 * H * H
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
names = {0: "start", 1: "prologue done (x frags, chunk 0 in LDS)", 7: "chunk0: after GEMM1", 8: "chunk0: after GELU+GEMM2",
         9: "chunk0: after stash", 2: "chunk0 barrier", 3: "chunk1 done", 4: "chunk2 done", 5: "chunk3 done", 6: "epilogue done"}
order = [0, 1, 7, 8, 9, 2, 3, 4, 5, 6]
rows = [[buf[b * 16 + i] for i in range(16)] for b in range(64)]
prev = None
for i in order:
    vals = [r[i] - r[0] for r in rows if r[i] and r[0]]
    med = statistics.median(vals)
    print(f"{names[i]:45s} t = {med:9.0f} cycles   (+{med - (prev or 0):7.0f})")
    prev = med
