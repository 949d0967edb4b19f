#!/usr/bin/env python3
"""Diagnostic: s_memtime stamps of dwconv7_ln_roll_kernel (library built with EXTRA=-DGCV_DW_STAMPS=1, path in
GCV_LIB_PATH): tap wave 0 and staging wave 0 of workgroups 0..63 in step 30.  usage: dw_stamps.py [C]"""
import ctypes, os, sys, statistics
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from genconvit_amd import _lib
lib = _lib.load()
dev = torch.device("cuda", 0)
C = int(sys.argv[1]) if len(sys.argv) > 1 else 96
H = {96: 56, 192: 28, 384: 14, 768: 7}[C]
nimg = 256
R = lambda *s: (torch.rand(*s, device=dev) * 2 - 1)
x = R(nimg, H, H, C).half(); y = torch.empty_like(x)
w, b, lw, lb = R(49, C), R(C), R(C) + 1.5, R(C)
for _ in range(5):
    _lib.check(lib.gcv_k_dwconv7_ln(_lib.GCV_F16, x.data_ptr(), w.data_ptr(), b.data_ptr(), lw.data_ptr(), lb.data_ptr(),
                                     y.data_ptr(), nimg, H, H, C, 1e-6, _lib.current_stream_ptr(dev)), "dw")
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * (64 * 32))()
raw = ctypes.CDLL(_lib.LIB_PATH)
assert raw.gcv_debug_read_dw_stamps(buf, 64 * 32) == 0
rows = [[buf[b * 32 + i] for i in range(32)] for b in range(64)]
rows = [r for r in rows if r[0] and r[4] and r[8] and r[14]]
names = {0: "tap: step start", 1: "tap: 13 LDS values landed", 2: "tap: FMAs done", 3: "tap: sval written", 4: "tap: barrier passed",
         8: "stg: step start", 9: "stg: row it+2 written to LDS", 10: "stg: loads of row it+3 issued", 11: "stg: LN values in registers",
         12: "stg: statistics done", 13: "stg: stores issued", 14: "stg: barrier passed",
         16: "tap wave 5: step start", 17: "tap wave 5: LDS values landed", 18: "tap wave 5: FMAs done", 19: "tap wave 5: sval written",
         20: "tap wave 5: barrier passed", 21: "last tap wave: step start", 22: "last tap wave: LDS values landed",
         23: "last tap wave: FMAs done", 24: "last tap wave: sval written", 25: "last tap wave: barrier passed",
         26: "last stg wave: step start", 27: "last stg wave: stores issued", 28: "last stg wave: barrier passed"}
print(f"C={C}: {len(rows)} workgroups, ticks relative to the tap wave's step start (median)")
for i in [0, 1, 2, 3, 4, 16, 17, 18, 19, 20, 21, 22, 23, 24, 25, 8, 9, 10, 11, 12, 13, 14, 26, 27, 28]:
    print(f"  {names[i]:36s} {statistics.median(r[i] - r[0] for r in rows):8.0f}")
ck = [(r[6] - r[4]) / max(r[7] - r[5], 1) * 100e6 / 1e9 for r in rows if r[7] > r[5]]
if ck:
    print(f"  10 steps = {statistics.median(r[6] - r[4] for r in rows):.0f} ticks; shader clock {statistics.median(ck):.2f} GHz")
