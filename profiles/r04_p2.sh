# round 4: pw2f with split DMA streams — parity cases, then pw1 / pw2 timed alone (ED-sized and VAE-sized launches), A/B nt
mkdir -p gpurun_out/r4b
python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "fused_mlp" > gpurun_out/r4b/mlp_tests.log 2>&1; tail -3 gpurun_out/r4b/mlp_tests.log
for n in 256 160 40; do
  echo "== nimg $n (product: nt hidden)"; GCV_MB_NIMG=$n python3 profiles/microbench.py mlp384 50
  echo "== nimg $n (variant: no nt)"; GCV_LIB_PATH=genconvit_amd/lib/libgenconvit_hip_p2nt0.so GCV_MB_NIMG=$n python3 profiles/microbench.py mlp384 50
done 2>&1 | tee gpurun_out/r4b/mb.txt
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r4b/bench_f16.json 2> gpurun_out/r4b/bench.log; python3 -c "
import json; d=json.loads(open('gpurun_out/r4b/bench_f16.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['breakdown_ms_per_step'])"
GCV_LIB_PATH=genconvit_amd/lib/libgenconvit_hip_p2nt0.so python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r4b/bench_f16_nt0.json 2> gpurun_out/r4b/bench_nt0.log; python3 -c "
import json; d=json.loads(open('gpurun_out/r4b/bench_f16_nt0.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['breakdown_ms_per_step'])"
