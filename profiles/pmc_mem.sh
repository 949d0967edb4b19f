#!/bin/bash
# usage: pmc_mem.sh <outdir> <microbench target> — HBM/L2 traffic counters in separate passes (TCC: FETCH_SIZE costs 3
# slots, WRITE_SIZE 2).  gfx950: FETCH_SIZE reads 1/2 of a wide coalesced stream (MI355X_MICROARCH.md §HBM) — doubled below.
out=$1; tgt=$2; R=${GRAFT_REPO_ROOT:-$PWD}; mkdir -p $R/$out
cd /tmp && export TMPDIR=/tmp
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $R/$out/mem$i -- python3 $R/profiles/microbench.py $tgt 3 > $R/$out/mem$i.log 2>&1
done
python3 - "$R/$out" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob(out + "/mem*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "gcv" not in k: continue
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
if not agg: sys.exit("no *counter_collection.csv with gcv kernels found under " + out)
for k, d in agg.items():
    print("==", k[:110])
    for c, v in sorted(d.items()):
        print(f"   {c:24s} {v/cnt[(k,c)]:16.1f}  per dispatch")
    if "FETCH_SIZE" in d:
        f = d["FETCH_SIZE"]/cnt[(k,"FETCH_SIZE")]; w = d.get("WRITE_SIZE",0)/max(cnt[(k,"WRITE_SIZE")],1)
        print(f"   -> HBM-side traffic per dispatch: read {2*f*1024/1e6:.1f} MB (FETCH_SIZE KB x2, gfx950 correction), write {w*1024/1e6:.1f} MB")
PY
