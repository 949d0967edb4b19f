# round 4: split-K plan of the mu GEMM (784 / 392 / 686 workgroups on 512 slots), headline config and vae B=32 bf16
mkdir -p gpurun_out/r4d
for v in default musk4 musk7; do
  L=""; [ $v != default ] && L="GCV_LIB_PATH=genconvit_amd/lib/libgenconvit_hip_$v.so"
  env $L python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline 2> gpurun_out/r4d/b_$v.log | tail -1 > gpurun_out/r4d/bench_$v.json
  env $L python3 bench.py --steps 20 --warmup 5 --net vae --batch 32 --dtype bf16 --no-cpu-baseline 2>> gpurun_out/r4d/b_$v.log | tail -1 > gpurun_out/r4d/cfg3_$v.json
  python3 -c "
import json
for f in ('bench','cfg3'):
    d=json.load(open('gpurun_out/r4d/%s_$v.json'%f)); print('$v', f, d['value'], d['ms_per_step'], d['roofline']['breakdown_ms_per_step'].get('vae.mu_gemm_splitk'), d['roofline']['breakdown_ms_per_step'].get('vae.reparam'))"
done
