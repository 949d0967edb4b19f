#!/usr/bin/env python3
"""Timeline of the CONCURRENT genconvit step from a rocprofv3 --kernel-trace CSV: which stream (queue) runs what, when the
GPU has 0 / 1 / 2+ kernels in flight, where each network's chain starts and ends.  usage: concurrent_timeline.py <kernel_trace.csv>"""
import csv, re, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if r['Kernel_Name'].startswith('_ZN3gcv')]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
starts = [i for i, r in enumerate(rows) if 'conv3_first_' in r['Kernel_Name'] and 'Lb1' in r['Kernel_Name']]
# bench.py's timed (two-stream) steps come after the warm-up and before the profiled serial passes: take step number
# argv[2] (default 4) and everything launched from 150 us before its first ED kernel up to the next step's
which = int(sys.argv[2]) if len(sys.argv) > 2 else 4
b0 = int(rows[starts[which]]['Start_Timestamp']) - 150000
b1 = int(rows[starts[which + 1]]['Start_Timestamp']) - 150000
step = [r for r in rows if b0 <= int(r['Start_Timestamp']) < b1]
t0 = min(int(r['Start_Timestamp']) for r in step)
t1 = max(int(r['End_Timestamp']) for r in step)
qs = sorted(set(r['Queue_Id'] for r in step))
print(f"# step wall {(t1 - t0) / 1e6:.3f} ms, {len(step)} launches, queues {qs}")
for q in qs:
    k = [r for r in step if r['Queue_Id'] == q]
    busy = sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in k)
    print(f"queue {q}: {len(k):3d} launches, first starts +{(int(k[0]['Start_Timestamp']) - t0) / 1e3:8.1f} us, last ends +{(max(int(r['End_Timestamp']) for r in k) - t0) / 1e3:8.1f} us, "
          f"kernel time {busy / 1e6:.3f} ms")
# concurrency histogram
ev = []
for r in step:
    ev.append((int(r['Start_Timestamp']), 1)); ev.append((int(r['End_Timestamp']), -1))
ev.sort()
lvl = 0; prev = t0; hist = {}
for t, d in ev:
    hist[lvl] = hist.get(lvl, 0) + (t - prev); prev = t; lvl += d
for l in sorted(hist):
    print(f"  {l} kernel(s) in flight: {hist[l] / 1e6:7.3f} ms ({100 * hist[l] / (t1 - t0):5.1f} %)")
# coarse timeline: per 250 us slice, time share of each queue
def short(n):
    n = re.sub(r'^_ZN3gcv\d+', '', n); return re.sub(r'I(DF16_|DF16b|f).*', '', n)[:26]
sl = 250000
print("# per 0.25 ms slice: busy fraction per queue and the kernel that took most of the slice on it")
for s0 in range(t0, t1, sl):
    line = f"+{(s0 - t0) / 1e6:5.2f} ms "
    for q in qs:
        best = ("", 0); tot = 0
        for r in step:
            if r['Queue_Id'] != q: continue
            a, b = max(int(r['Start_Timestamp']), s0), min(int(r['End_Timestamp']), s0 + sl)
            if b > a:
                tot += b - a
                if b - a > best[1]: best = (short(r['Kernel_Name']), b - a)
        line += f"| q{q} {100 * tot / sl:4.0f}% {best[0]:26s} "
    print(line)
