#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace CSV: per-launch durations of the last forward step and
totals per kernel symbol.  usage: summarize_trace.py <kernel_trace.csv> [--per-launch]"""
import csv, re, subprocess, sys
from collections import OrderedDict, defaultdict

def short(n):
    n = re.sub(r'^_ZN3gcv', '', n)
    n = re.sub(r'EvNS_8GemmArgsE$', '', n)
    n = n.replace('gemm_kernelI', 'gemm<').replace('DF16_', 'f16,').replace('DF16b', 'bf16,')
    return n[:70]

rows = list(csv.DictReader(open(sys.argv[1])))
g = [r for r in rows if r['Kernel_Name'].startswith('_ZN3gcv') and 'pack_mu' not in r['Kernel_Name']]
g.sort(key=lambda r: int(r['Start_Timestamp']))
# a step starts with the ED encoder's first conv (POOL = true: 'Lb1'), VALU or matrix-pipe variant
starts = [i for i, r in enumerate(g) if 'conv3_first_' in r['Kernel_Name'] and 'Lb1' in r['Kernel_Name']]
last = g[starts[-1]:] if starts else g
tot = 0.0
agg = OrderedDict()
for r in last:
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    tot += d
    k = short(r['Kernel_Name'])
    a = agg.setdefault(k, [0, 0.0])
    a[0] += 1; a[1] += d
    if '--per-launch' in sys.argv:
        print(f"{k:72s} {d:9.1f} us grid={r['Grid_Size_X']:>9} wg={r['Workgroup_Size_X']:>4} vgpr={r['VGPR_Count']:>4} lds={r['LDS_Block_Size']}")
print(f"# last step: {len(last)} launches, {tot/1e3:.3f} ms of kernel time")
for k, (n, d) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{k:72s} n={n:4d} total={d/1e3:8.3f} ms  avg={d/n:9.1f} us  {100*d/tot:5.1f}%")
