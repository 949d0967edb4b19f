#!/bin/bash
# dw7x7 + LayerNorm: parity cases, then the 256-image microbenchmarks of the product build and of variant builds
python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "dwconv" 2>&1 | tail -15
for rep in 1 2; do for k in dwconv96 dwconv192; do
  printf "%-8s" new; python3 profiles/microbench.py $k 50 2>/dev/null | tail -1
  for V in $@; do printf "%-8s" $V; GCV_LIB_PATH=$PWD/genconvit_amd/lib/libgenconvit_hip_$V.so python3 profiles/microbench.py $k 50 2>/dev/null | tail -1; done
done; done
