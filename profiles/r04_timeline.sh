R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r4o; mkdir -p $O
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --profile-steps 0 > $O/trace.log 2>&1 )
T=$(ls $O/trace/*/*kernel_trace.csv | head -1)
python3 $R/profiles/concurrent_timeline.py $T | tee $O/timeline.txt
rm -rf $O/trace
