#!/bin/bash
# HBM traffic of the bench step per kernel family, from rocprofv3 PMC counters (separate passes for FETCH_SIZE and
# WRITE_SIZE: they do not fit one TCC pass; no trace domains mixed in).  Writes <out>/traffic.json.
# gfx950: FETCH_SIZE (KB) reads 1/2 of a wide coalesced stream -> doubled (MI355X_MICROARCH.md §HBM); WRITE_SIZE is exact.
# usage: pmc_bench_traffic.sh <outdir> [bench.py args, e.g. --net vae --batch 32 --dtype bf16]
out=$1; shift; ARGS="$@"; R=${GRAFT_REPO_ROOT:-$PWD}; mkdir -p $R/$out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/$out/fetch -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --profile-steps 1 $ARGS > $R/$out/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/$out/write -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --profile-steps 1 $ARGS > $R/$out/write.log 2>&1
python3 - "$R/$out" $ARGS <<'PY'
import csv, glob, json, sys, collections
out = sys.argv[1]
import argparse
ap = argparse.ArgumentParser(); ap.add_argument("--net", default="genconvit"); ap.add_argument("--batch", type=int, default=128)
ap.add_argument("--dtype", default="f16")
cfg, _ = ap.parse_known_args(sys.argv[2:])
def load(sub, name):
    d = collections.defaultdict(lambda: [0, 0.0])
    for f in glob.glob(f"{out}/{sub}/*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != name or "gcv" not in r["Kernel_Name"] or "pack" in r["Kernel_Name"]: continue
            k = r["Kernel_Name"].split("(")[0]
            d[k][0] += 1; d[k][1] += float(r["Counter_Value"])
    return d
fe, wr = load("fetch", "FETCH_SIZE"), load("write", "WRITE_SIZE")
fam = lambda k: "mfma_gemm" if ("gemm_glds_kernel" in k or "fused_mlp" in k or ("gemm_kernel" in k and "ELi0ELi0E" in k) or ("gemm_kernel" in k and "ELi0ELi1E" in k)) else ("dwconv7_ln" if "dwconv7_ln" in k else "other")
res = {}
for k in set(fe) | set(wr):
    f = res.setdefault(fam(k), {"dispatches": 0, "read_bytes": 0.0, "write_bytes": 0.0})
    f["dispatches"] += fe[k][0]
    f["read_bytes"] += 2.0 * fe[k][1] * 1024.0
    f["write_bytes"] += wr[k][1] * 1024.0
for f in res.values():
    n = max(f["dispatches"], 1)
    f["hbm_bytes_per_launch"] = (f["read_bytes"] + f["write_bytes"]) / n
if not res: sys.exit("no *counter_collection.csv found under " + out)
json.dump({"config": {"net": cfg.net, "batch": cfg.batch, "dtype": cfg.dtype}, "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE over `bench.py --steps 2 --warmup 1 --profile-steps 1` "
                   "(4 forward steps), FETCH_SIZE x2 per the gfx950 correction", "families": res}, open(out + "/traffic.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
