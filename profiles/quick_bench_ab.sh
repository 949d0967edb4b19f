#!/bin/bash
# A/B of whole-step throughput on one box: bench.py with the product library and with variant builds, alternating.
# usage: quick_bench_ab.sh "<variant> ..." [bench args]
VS=$1; shift
R=${GRAFT_REPO_ROOT:-$PWD}
one() { python3 bench.py --steps 30 --warmup 8 --no-cpu-baseline "$@" 2>/dev/null | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); b=d['roofline']['breakdown_ms_per_step']; print('%8.1f fps %.3f ms  dw %.3f  mlp %.3f pw1 %.3f pw2 %.3f' % (d['value'], d['ms_per_step'], b.get('cnx.dwconv7_ln',0), b.get('cnx.fused_mlp',0), b.get('cnx.pw1_gelu',0), b.get('cnx.pw2_scale_res',0)))"; }
for rep in 1 2 3; do
  printf "%-8s" new; one "$@"
  for V in $VS; do printf "%-8s" $V; GCV_LIB_PATH=$R/genconvit_amd/lib/libgenconvit_hip_$V.so one "$@"; done
done
