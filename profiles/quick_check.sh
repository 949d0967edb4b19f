mkdir -p gpurun_out/s3
python -m pytest tests -m gpu -x -q > gpurun_out/s3/gpu_tests.log 2>&1; tail -3 gpurun_out/s3/gpu_tests.log
for k in mlp96 mlp192 mlp384; do python3 profiles/microbench.py $k 50; done 2>&1 | tee gpurun_out/s3/mb.txt
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/s3/bench_f16.json 2> gpurun_out/s3/bench.log; python3 -c "
import json; d=json.loads(open('gpurun_out/s3/bench_f16.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['breakdown_ms_per_step'])"
