# round 4: pw2f residual trickle + tiny-map dw kernel: parity, then kernels alone and the bench line
mkdir -p gpurun_out/r4c
python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "fused_mlp or dwconv7x7" > gpurun_out/r4c/k_tests.log 2>&1; tail -3 gpurun_out/r4c/k_tests.log
for n in 256 160 40; do GCV_MB_NIMG=$n python3 profiles/microbench.py mlp384 50; done 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4c/mb.txt
python -m pytest tests -m gpu -x -q > gpurun_out/r4c/gpu_tests.log 2>&1; tail -3 gpurun_out/r4c/gpu_tests.log
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r4c/bench_f16.json 2> gpurun_out/r4c/bench.log; python3 -c "
import json; d=json.loads(open('gpurun_out/r4c/bench_f16.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['breakdown_ms_per_step'])"
