#!/bin/bash
# HBM traffic of the bench step per kernel family, from rocprofv3 PMC counters (separate passes for FETCH_SIZE and
# WRITE_SIZE: they do not fit one TCC pass; no trace domains mixed in).  Writes <out>/traffic.json.
# gfx950: FETCH_SIZE (KB) reads 1/2 of a wide coalesced stream -> doubled (MI355X_MICROARCH.md §HBM); WRITE_SIZE is exact.
# The runs use --no-concurrent: one stream and the merged two-segment VAE pass in every step, i.e. the schedule bench.py's
# roofline pass times (the split VAE schedule launches the backbone twice: 78 instead of 60 MLP launches per step).
# usage: pmc_bench_traffic.sh <outdir> [bench.py args, e.g. --net vae --batch 32 --dtype bf16]
out=$1; shift; ARGS="$@"; R=${GRAFT_REPO_ROOT:-$PWD}; mkdir -p $R/$out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/$out/fetch -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-concurrent --profile-steps 1 $ARGS > $R/$out/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/$out/write -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-concurrent --profile-steps 1 $ARGS > $R/$out/write.log 2>&1
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
import re
# Families = the launch tags bench.py aggregates (cnx.pw1_gelu + cnx.pw2_scale_res + cnx.fused_mlp; cnx.dwconv7_ln), by the
# kernels those tags launch in a 16-bit run: the MLP kernels proper, and gemm_glds_kernel only in the two instantiations
# run_convnext uses for pw1 (bias + GELU: EPI 0, ACT 2) and pw2 (layer-scale residual: EPI 1) — NOT the down-sampling /
# head GEMMs (EPI 0, ACT 0 / 1), which an earlier version of this script swept in.
MLP = ("xs_pw1_kernel", "pw2f_kernel", "xs_mlp_kernel", "fused_mlp_res_kernel", "fused_mlp_kernel", "fused_mlp_ring_kernel")
def fam(k):
    if any(m in k for m in MLP): return "mfma_gemm"
    # (the counter CSV carries mangled names: gemm_glds_kernelI<T>Li<EPI>ELi<ACT>ELi<BKB>EE; demangled ones are matched too)
    if "gemm_glds_kernel" in k and (re.search(r"gemm_glds_kernelI\w+?_Li1E", k) or re.search(r"gemm_glds_kernelI\w+?_Li0ELi2E", k) or
                                    re.search(r"gemm_glds_kernel<[^,]+, 1,", k) or re.search(r"gemm_glds_kernel<[^,]+, 0, 2,", k)):
        return "mfma_gemm"
    if "dwconv7_ln" in k: return "dwconv7_ln"
    # the split-K GEMM (EPI 4: the 8th template argument of gemm_kernel) is the mu GEMM of the VAE encoder (the var GEMM runs
    # only when the KL term is asked for, which bench.py does not)
    if re.search(r"gemm_kernelI\w+?(?:Li\d+E){6}Li4E", k) or re.search(r"gemm_kernel<(?:[^,]+, ){7}4,", k): return "mu_gemm"
    return "other"
res = {}
for k in set(fe) | set(wr):
    f = res.setdefault(fam(k), {"dispatches": 0, "write_pass_dispatches": 0, "read_bytes": 0.0, "write_bytes": 0.0, "kernels": []})
    f["dispatches"] += fe[k][0]
    f["write_pass_dispatches"] += wr[k][0]
    f["read_bytes"] += 2.0 * fe[k][1] * 1024.0
    f["write_bytes"] += wr[k][1] * 1024.0
    f["kernels"].append(k)
# the bench line of the FETCH pass says how many launches per step each family has: 5 forward steps were recorded
# (1 warm-up + 2 timed + 1 untimed profiled + 1 profiled), so dispatches must be exactly 4 x launches_per_step — anything else means the
# classification above and bench.py's tags have drifted apart, and the file would mislead
STEPS = 5
line = None
for ln in open(out + "/fetch.log"):
    ln = ln.strip()
    if ln.startswith("{") and '"roofline"' in ln:
        line = json.loads(ln)
if line is None: sys.exit("no bench line in " + out + "/fetch.log")
expected = {}
for e in [line.get("roofline")] + (line.get("roofline_families") or []):
    if not e: continue
    key = "mfma_gemm" if e["kernel"].startswith("mfma_gemm") else ("dwconv7_ln" if "dwconv" in e["kernel"] else
                                                                   ("mu_gemm" if "mu_gemm" in e["kernel"] else None))
    if key: expected[key] = e["launches_per_step"]
bad = []
for key, lps in expected.items():
    f = res.get(key)
    if f is None or f["dispatches"] != STEPS * lps or f["write_pass_dispatches"] != STEPS * lps:
        bad.append(f"{key}: {None if f is None else (f['dispatches'], f['write_pass_dispatches'])} dispatches recorded, bench.py launches {lps} per step x {STEPS} steps")
if bad:
    for key, f in res.items():
        print(key, f["dispatches"], f["write_pass_dispatches"])
        for k in sorted(set(f["kernels"])): print("     ", fe[k][0], wr[k][0], k[:120])
    sys.exit("dispatch count mismatch, traffic.json NOT written:\n  " + "\n  ".join(bad))
for key, f in res.items():
    n = max(f["dispatches"], 1)
    f["hbm_bytes_per_launch"] = (f["read_bytes"] + f["write_bytes"]) / n
    f["steps"] = STEPS
    if key in expected: f["launches_per_step"] = expected[key]
    f["kernels"] = sorted(set(f["kernels"]))
if not res: sys.exit("no *counter_collection.csv found under " + out)
json.dump({"config": {"net": cfg.net, "batch": cfg.batch, "dtype": cfg.dtype}, "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE over `bench.py --steps 2 --warmup 1 --profile-steps 1` "
                   "(5 forward steps), FETCH_SIZE x2 per the gfx950 correction; dispatches checked against bench.py's launches_per_step", "families": res}, open(out + "/traffic.json", "w"), indent=1)
print(json.dumps({k: {kk: vv for kk, vv in v.items() if kk != "kernels"} for k, v in res.items()}, indent=1))
PY
