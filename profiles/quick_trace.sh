#!/bin/bash
# rocprofv3 kernel trace of the serial fp16 step -> per-kernel table of the last step (gpurun_out/trace_last_step.txt)
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/tr; mkdir -p $O
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-concurrent > $O/trace.log 2>&1 )
T=$(ls $O/trace/*/*kernel_trace.csv | head -1)
python3 $R/profiles/summarize_trace.py $T > $R/gpurun_out/trace_last_step.txt
rm -rf $O/trace
head -${1:-45} $R/gpurun_out/trace_last_step.txt
