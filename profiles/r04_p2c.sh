mkdir -p gpurun_out/r4h
python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "fused_mlp" > gpurun_out/r4h/k_tests.log 2>&1; tail -2 gpurun_out/r4h/k_tests.log
for n in 256 160 40; do GCV_MB_NIMG=$n python3 profiles/microbench.py mlp384 50; done 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4h/mb.txt
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r4h/bench_f16.json 2> gpurun_out/r4h/bench.log; python3 -c "
import json; d=json.loads(open('gpurun_out/r4h/bench_f16.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['breakdown_ms_per_step'])"
