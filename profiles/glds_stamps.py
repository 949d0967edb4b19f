#!/usr/bin/env python3
"""Diagnostic: per-workgroup timeline of gemm_glds_kernel (library built with -DGCV_GLDS_STAMPS=1, path in
GCV_LIB_PATH).  usage: glds_stamps.py <pw1|pw2>_<C>.  Prints phase medians and the per-CU concurrency."""
import ctypes, math, os, sys, statistics, collections
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from genconvit_amd import _lib
lib = _lib.load()
dev = torch.device("cuda", 0)
kind, C = sys.argv[1].split("_"); C = int(C)
H = {96: 56, 192: 28, 384: 14, 768: 7}[C]; M = 256 * H * H
R = lambda *s: (torch.rand(*s, device=dev) * 2 - 1)
if kind == "pw1":
    N, K, epi, act = 4 * C, C, _lib.EPI_BIAS_ACT, _lib.ACT_GELU
else:
    N, K, epi, act = C, 4 * C, _lib.EPI_RESID, _lib.ACT_NONE
A = R(M, K).half(); W = (R(N, K) / math.sqrt(K)).half(); Cm = R(M, N).half(); bias, gamma = R(N), R(N)
g = _lib.GemmArgs(A.data_ptr(), W.data_ptr(), Cm.data_ptr(), bias.data_ptr(), gamma.data_ptr(), Cm.data_ptr(), None,
                  M, N, K, K, N, act, 1, 0, 0, 0, 0, 0)
for _ in range(3):
    _lib.check(lib.gcv_k_gemm(_lib.GCV_F16, _lib.A_PLAIN, epi, ctypes.byref(g), _lib.current_stream_ptr(dev)), "gemm")
torch.cuda.synchronize()
ntiles = min(4096, ((M + 127) // 128) * (N // 192))
if os.environ.get('GLDS2'):
    ntiles = 256          # persistent kernel: one row per workgroup (stamps of its third tile)
buf = (ctypes.c_ulonglong * (4096 * 8))()
raw = ctypes.CDLL(_lib.LIB_PATH)
assert raw.gcv_debug_read_glds_stamps(buf, 4096 * 8) == 0
rows = [[buf[b * 8 + i] for i in range(8)] for b in range(ntiles)]
rows = [r for r in rows if r[0] and r[6]]
names = ["start", "first stage landed", "mainloop done", "epilogue math+LDS stage done", "stores issued", None, "stores acked"]
prev = 0
for i in (1, 2, 3, 4, 6):
    med = statistics.median(r[i] - r[0] for r in rows)
    print(f"{names[i]:32s} t = {med:8.0f} cycles (+{med - prev:7.0f})")
    prev = med
# per-CU occupancy: key = (xcc, se, sh, cu)
percu = collections.defaultdict(list)
for b, r in enumerate(rows):
    hw, xcc = r[5] & 0xffffffff, (r[5] >> 32) & 0xf
    key = (xcc, (hw >> 13) & 7, (hw >> 12) & 1, (hw >> 8) & 15)
    percu[key].append((r[0], r[6], b))
print("distinct CUs seen:", len(percu), " tiles:", ntiles)
conc = []
for key, iv in percu.items():
    ev = sorted([(s, 1) for s, e, _ in iv] + [(e, -1) for s, e, _ in iv])
    cur = 0; last = ev[0][0]; acc = collections.Counter()
    for t, d in ev:
        acc[cur] += t - last; last = t; cur += d
    tot = sum(acc.values())
    conc.append({k: v / tot for k, v in acc.items()})
avg = collections.Counter()
for c in conc:
    for k, v in c.items():
        avg[k] += v / len(conc)
print("time share by number of co-resident workgroups per CU:", {k: round(v, 3) for k, v in sorted(avg.items()) if 0 <= k <= 8})
key0 = sorted(percu)[0]
print("timeline of CU", key0)
t0 = min(s for s, e, b in percu[key0])
for s, e, b in sorted(percu[key0])[:14]:
    r = rows[b]
    print(f"  wg {b:5d}: start {s - t0:8d}  landed +{r[1]-r[0]:6d}  main +{r[2]-r[1]:6d}  epi +{r[3]-r[2]:6d}  store +{r[4]-r[3]:6d}  ack +{r[6]-r[4]:6d}  total {e - s:7d}")
span = max(e for iv in percu.values() for s, e, b in iv) - min(s for iv in percu.values() for s, e, b in iv)
print("kernel span (cycles, first start to last ack, unsynchronised across XCDs):", span)
