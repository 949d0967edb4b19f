mkdir -p gpurun_out/r4f
python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "gemm_lds_dma" > gpurun_out/r4f/k_tests.log 2>&1; tail -2 gpurun_out/r4f/k_tests.log
for rep in 1 2; do for v in default dwmall; do
  L=""; [ $v != default ] && L="GCV_LIB_PATH=genconvit_amd/lib/libgenconvit_hip_$v.so"
  env $L python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline 2> gpurun_out/r4f/b_$v.log | tail -1 > gpurun_out/r4f/bench_$v.json
  python3 -c "
import json
d=json.load(open('gpurun_out/r4f/bench_$v.json')); b=d['roofline']['breakdown_ms_per_step']; print('$v', d['value'], d['ms_per_step'], 'dw', b.get('cnx.dwconv7_ln'), 'pw2', b.get('cnx.pw2_scale_res'), 'pw1', b.get('cnx.pw1_gelu'))"
done; done
