# round 4, first GPU call: the new kernel-level LN-patchify epilogue tests, then the whole GPU suite and a baseline bench line
mkdir -p gpurun_out/r4a
python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "layernorm_patchify_epilogue" > gpurun_out/r4a/lnp_tests.log 2>&1; tail -5 gpurun_out/r4a/lnp_tests.log
python -m pytest tests -m gpu -x -q > gpurun_out/r4a/gpu_tests.log 2>&1; tail -3 gpurun_out/r4a/gpu_tests.log
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r4a/bench_f16.json 2> gpurun_out/r4a/bench.log; python3 -c "
import json; d=json.loads(open('gpurun_out/r4a/bench_f16.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['breakdown_ms_per_step'])"
