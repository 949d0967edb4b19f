mkdir -p gpurun_out/r4p
python -m pytest tests/test_parity_gpu.py -m gpu -x -q -k "genconvit" > gpurun_out/r4p/tests.log 2>&1; tail -2 gpurun_out/r4p/tests.log
for rep in 1 2 3; do for v in default edfirst; do
  L=""; [ $v != default ] && L="GCV_LIB_PATH=genconvit_amd/lib/libgenconvit_hip_$v.so"
  env $L python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline 2> gpurun_out/r4p/b_$v.log | tail -1 > gpurun_out/r4p/bench_$v.json
  python3 -c "
import json
d=json.load(open('gpurun_out/r4p/bench_$v.json')); print('$v', d['value'], 'pipelined', d['ms_per_step'], 'synchronised', d['ms_per_step_synchronised'])"
done; done
