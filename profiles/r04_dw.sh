mkdir -p gpurun_out/r4e
python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "dwconv7x7 or fused_mlp" > gpurun_out/r4e/k_tests.log 2>&1; tail -2 gpurun_out/r4e/k_tests.log
for c in 384 192 96 768; do python3 profiles/microbench.py dwconv$c 50; done 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4e/mb.txt
for n in 256 160 40; do GCV_MB_NIMG=$n python3 profiles/microbench.py mlp384 50; done 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r4e/mb.txt
python -m pytest tests -m gpu -x -q > gpurun_out/r4e/gpu_tests.log 2>&1; tail -2 gpurun_out/r4e/gpu_tests.log
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r4e/bench_f16.json 2> gpurun_out/r4e/bench.log; python3 -c "
import json; d=json.loads(open('gpurun_out/r4e/bench_f16.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['breakdown_ms_per_step'])"
python3 bench.py --steps 20 --warmup 5 --net vae --batch 32 --dtype bf16 --no-cpu-baseline 2>> gpurun_out/r4e/bench.log | tail -1 > gpurun_out/r4e/cfg3.json; python3 -c "
import json; d=json.load(open('gpurun_out/r4e/cfg3.json')); print(d['value'], d['ms_per_step'], d['roofline']['breakdown_ms_per_step'])"
