O=$GRAFT_REPO_ROOT/gpurun_out/s4; mkdir -p $O
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_parity_gpu.py -m gpu -x -q -s -k "16bit or fp16 or bf16 or config" 2>&1 | grep -E "\||passed|failed|=" | tail -30 > $O/deltas.txt; tail -25 $O/deltas.txt
python3 bench.py --steps 20 --warmup 5 --net vae --batch 32 --dtype bf16 --no-cpu-baseline 2> $O/b.log | tail -1 > $O/bench_cfg3.json
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --net vae --batch 32 --dtype bf16 > $O/trace.log 2>&1 )
T=$(ls $O/trace/*/*kernel_trace.csv | head -1)
python3 profiles/summarize_trace.py $T > $O/cfg3_last_step_by_kernel.txt
rm -rf $O/trace
python3 -c "
import json; d=json.load(open('$O/bench_cfg3.json')); print(d['value'], d['ms_per_step'])"
head -50 $O/cfg3_last_step_by_kernel.txt
