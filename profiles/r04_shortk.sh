mkdir -p gpurun_out/r4n
GCV_LIB_PATH=genconvit_amd/lib/libgenconvit_hip_shortk.so python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "conv3x3" > gpurun_out/r4n/k_tests.log 2>&1; tail -2 gpurun_out/r4n/k_tests.log
for rep in 1 2; do for v in default shortk; do
  L=""; [ $v != default ] && L="GCV_LIB_PATH=genconvit_amd/lib/libgenconvit_hip_$v.so"
  env $L python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline 2> gpurun_out/r4n/b_$v.log | tail -1 > gpurun_out/r4n/bench_$v.json
  python3 -c "
import json
d=json.load(open('gpurun_out/r4n/bench_$v.json')); b=d['roofline']['breakdown_ms_per_step']; print('$v', d['value'], d['ms_per_step'], 'ed.enc', b.get('ed.enc_conv3_relu_pool'), 'vae.enc', b.get('vae.enc_conv3s2_bn_leaky'))"
done; done
