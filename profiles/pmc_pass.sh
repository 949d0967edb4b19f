#!/bin/bash
# usage: pmc_pass.sh <outdir> <microbench target> — three PMC passes (8 SQ counters each) over profiles/microbench.py
# rocprofv3 --pmc runs are kept separate from kernel-trace runs (gpurun refuses mixing with sys/hip traces).
out=$1; tgt=$2; R=${GRAFT_REPO_ROOT:-$PWD}; mkdir -p $R/$out
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_MFMA" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_SMEM"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $R/$out/pass$i -- python3 $R/profiles/microbench.py $tgt 3 > $R/$out/pass$i.log 2>&1
done
python3 - "$R/$out" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob(out + "/pass*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "gcv" not in k: continue
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[(k, r["Counter_Name"])] += 1
for k, d in agg.items():
    print("==", k[:100])
    for c, v in sorted(d.items()):
        n = cnt[(k, c)]
        print(f"   {c:28s} {v/n:16.0f}  (avg over {n} dispatches)")
PY
