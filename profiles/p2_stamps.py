#!/usr/bin/env python3
"""Diagnostic: s_memtime timeline of pw2f_kernel workgroups 0..63, one wave (library built with -DGCV_P2_STAMPS=1
[-DGCV_P2_STAMP_WAVE=w], path in GCV_LIB_PATH).  Per step k = 32..39 (round 4: the first 27 steps are the straight-line trickle steps): before wait, after wait+barrier, after DMA issue,
after the chunk's MFMAs; 40 = ring prologue starts, 41 = K loop done, 42 = epilogue done."""
import ctypes, math, os, sys, statistics
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from genconvit_amd import _lib
lib = _lib.load()
dev = torch.device("cuda", 0)
C = 384; M = int(os.environ.get("GCV_MB_NIMG", "256")) * 196
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
assert raw.gcv_debug_read_p2_stamps(buf, 64 * 64) == 0
rows = [[buf[b * 64 + i] for i in range(64)] for b in range(64)]
med = lambda i, j: statistics.median([r[i] - r[j] for r in rows if r[i] and r[j]])
print(f"K loop {med(41, 40):9.0f} cycles; epilogue {med(42, 41):7.0f}; prologue -> step 32 {med(0, 40):7.0f}")
for k in range(8):
    nxt = med(4 * k + 4, 4 * k + 3) if k < 7 else float('nan')
    print(f"step {k + 32:2d}: wait+barrier {med(4 * k + 1, 4 * k):6.0f}  issue {med(4 * k + 2, 4 * k + 1):6.0f}  reads+MFMA {med(4 * k + 3, 4 * k + 2):6.0f}  (loop overhead {nxt:4.0f})")
