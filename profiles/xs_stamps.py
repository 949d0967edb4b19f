#!/usr/bin/env python3
"""Diagnostic: s_memtime timeline of xs_pw1_kernel workgroups 0..63, wave 0 (library built with -DGCV_XS_STAMPS=1, path in
GCV_LIB_PATH; profiles/build_variant.sh xsstamps "-DGCV_XS_STAMPS=1" mlp_pair_f16.hip).  Slots: 0 start, 1 last step done,
2 end, 2k / 2k+1 = before / after the wait + barrier of step k (k = 8..15)."""
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
assert raw.gcv_debug_read_xs_stamps(buf, 64 * 64) == 0
rows = [[buf[b * 64 + i] for i in range(64)] for b in range(64)]
med = lambda i, j: statistics.median([r[i] - r[j] for r in rows if r[i] and r[j]])
print(f"whole kernel      : {med(2, 0):9.0f} cycles; main loop ends at {med(1, 0):9.0f}")
print(f"start -> step 8   : {med(16, 0):9.0f}")
for k in range(8, 16):
    nxt = med(2 * k + 2, 2 * k + 1) if k < 15 else float('nan')
    print(f"step {k:2d}: wait+barrier {med(2 * k + 1, 2 * k):7.0f}   body {nxt:7.0f}")
for k in (10, 11):
    base = 2 * k + 1
    prev = base
    line = []
    for j in range(6):
        i = 32 + 8 * (k - 10) + j
        line.append(f"{med(i, prev):5.0f}")
        prev = i
    print(f"step {k}: sub-blocks after the barrier: " + " ".join(line))
