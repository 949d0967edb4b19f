#!/bin/bash
# vae B=32 bf16 (BASELINE configs[2]): MLP / dw kernel tests, then the bench line
O=gpurun_out/s4; mkdir -p $O
python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "fused_mlp or dwconv" 2>&1 | tail -3
python -m pytest tests/test_parity_gpu.py -m gpu -x -q -k "config or vae" 2>&1 | tail -2
for i in 1 2; do python3 bench.py --steps 20 --warmup 5 --net vae --batch 32 --dtype bf16 --no-cpu-baseline 2> $O/b.log | tail -1 > $O/bench_cfg3.json
python3 -c "
import json; d=json.load(open('$O/bench_cfg3.json')); print(d['value'], d['ms_per_step'], d['roofline']['breakdown_ms_per_step'])"; done
