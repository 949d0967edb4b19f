#!/bin/bash
# Round-4 measurement set on one MI355X box (gpurun): bench lines for the BASELINE.json configs, the rocprofv3
# kernel-trace summary of the serial fp16 step, PMC traffic for the headline config (checked against bench.py's launch
# counts), SQ counters of the new MFMA kernels.  Outputs under gpurun_out/r04/; copy what is to be kept into profiles/.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r04; mkdir -p $O
cd $R
python3 bench.py --steps 20 --warmup 5 --swin 2> $O/bench_f16.log | tail -1 > $O/bench_f16.json && echo f16 done
python3 bench.py --steps 20 --warmup 5 --no-concurrent --no-cpu-baseline 2>> $O/bench_f16.log | tail -1 > $O/bench_f16_serial.json
python3 bench.py --steps 20 --warmup 5 --dtype bf16 --no-cpu-baseline 2>> $O/bench_f16.log | tail -1 > $O/bench_bf16.json
python3 bench.py --steps 10 --warmup 3 --dtype f32 --no-cpu-baseline 2>> $O/bench_f16.log | tail -1 > $O/bench_f32.json
python3 bench.py --steps 20 --warmup 5 --net ed --batch 32 --dtype f32 --no-cpu-baseline 2>> $O/bench_f16.log | tail -1 > $O/bench_cfg2_ed_b32_f32.json
python3 bench.py --steps 20 --warmup 5 --net vae --batch 32 --dtype bf16 --no-cpu-baseline 2>> $O/bench_f16.log | tail -1 > $O/bench_cfg3_vae_b32_bf16.json
GCV_BENCH_FORCE_DIST=1 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>> $O/bench_f16.log | tail -1 > $O/bench_f16_force_dist_1rank.json
echo benches done
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-concurrent > $O/trace.log 2>&1 )
T=$(ls $O/trace/*/*kernel_trace.csv | head -1); S=$(ls $O/trace/*/*kernel_stats.csv | head -1)
python3 $R/profiles/summarize_trace.py $T > $O/bench_f16_serial_last_step_by_kernel.txt && cp $S $O/bench_f16_serial_kernel_stats.csv
rm -rf $O/trace
echo trace done
bash profiles/pmc_bench_traffic.sh gpurun_out/r04/pmc > $O/pmc.log 2>&1 && cp $O/pmc/traffic.json $O/traffic_genconvit_b128_f16.json
rm -rf $O/pmc/fetch $O/pmc/write
echo traffic done
for t in mlp384 pw1_768 pw2_768 dwconv384; do bash profiles/pmc_pass.sh gpurun_out/r04/sq_$t $t > $O/pmc_sq_$t.txt 2>&1; rm -rf $O/sq_$t; done
echo all done
# roctx ranges of the profiled pass (gcv_profile_enable): marker trace of a short serial run, ranges counted per tag
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --marker-trace --kernel-trace --output-format csv -d $O/marker -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-concurrent --profile-steps 1 > $O/marker.log 2>&1 )
python3 - $O <<'PY'
import csv, glob, sys, collections
o = sys.argv[1]
c = collections.Counter()
for f in glob.glob(o + "/marker/*/*marker_api_trace.csv"):
    for r in csv.DictReader(open(f)):
        c[r.get("Function") or r.get("Name") or "?"] += 1
open(o + "/roctx_ranges_by_tag.txt", "w").write("# roctx ranges recorded by rocprofv3 --marker-trace over one profiled genconvit B=128 fp16 step (serial schedule)\n" +
    "".join(f"{n:6d}  {k}\n" for k, n in sorted(c.items(), key=lambda kv: -kv[1])))
print(open(o + "/roctx_ranges_by_tag.txt").read()[:1500])
PY
rm -rf $O/marker
echo marker done
