#!/bin/bash
# Per-kernel average durations of one command under rocprofv3 (kernel trace, no counters).
# usage: kernel_times.sh <outdir> <python script + args ...>     (set GCV_LIB_PATH etc. in the environment)
O=$(realpath -m $1); shift
R=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p $O; export TMPDIR=/tmp
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 "$@" > $O/run.log 2>&1 )
S=$(ls $O/trace/*/*kernel_stats.csv 2>/dev/null | head -1)
[ -z "$S" ] && { tail -5 $O/run.log; exit 1; }
python3 - "$S" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]
    if "gcv" not in n or "pack_" in n: continue
    print(f'{n[:80]:80s} n={r["Calls"]:>4s} avg={float(r["AverageNs"])/1e3:8.1f} us  min={float(r["MinNs"])/1e3:8.1f}')
PY
rm -rf $O/trace
