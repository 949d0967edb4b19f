for k in dwconv96 dwconv192; do
  printf "%-8s" new; python3 profiles/microbench.py $k 50 2>/dev/null | tail -1
  for V in dwa1 dwa2 dwa4 old; do printf "%-8s" $V; GCV_LIB_PATH=$PWD/genconvit_amd/lib/libgenconvit_hip_$V.so python3 profiles/microbench.py $k 50 2>/dev/null | tail -1; done
done
