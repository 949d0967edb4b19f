#!/bin/bash
# A/B of library builds on one box: the single-kernel microbenchmarks (50 launches), alternating between the builds.
# usage: quick_ab.sh "<variant> [<variant> ...]" [kernels...]   (variant = genconvit_amd/lib/libgenconvit_hip_<name>.so;
# build one with profiles/build_variant.sh or `make BUILD=build_<name> OUT=../lib/libgenconvit_hip_<name>.so`)
VS=$1; shift
K=${@:-mlp96 mlp192 mlp384}
R=${GRAFT_REPO_ROOT:-$PWD}
for rep in 1 2; do
for k in $K; do
  printf "%-8s" new; python3 profiles/microbench.py $k 50 2>/dev/null | tail -1
  for V in $VS; do
    printf "%-8s" $V; GCV_LIB_PATH=$R/genconvit_amd/lib/libgenconvit_hip_$V.so python3 profiles/microbench.py $k 50 2>/dev/null | tail -1
  done
done
done
