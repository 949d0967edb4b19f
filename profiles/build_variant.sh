#!/bin/bash
# Diagnostic library variants: recompile the named translation units with extra -D flags and link them with the
# objects of the normal build.   usage: build_variant.sh <name> "<flags>" <tu.hip> [<tu.hip> ...]
# -> genconvit_amd/lib/libgenconvit_hip_<name>.so  (select it with GCV_LIB_PATH)
set -e
R=$(cd "$(dirname "$0")/.." && pwd); C=$R/genconvit_amd/csrc; N=$1; F=$2; shift 2
# (the normal build must be current: run `make -C genconvit_amd/csrc` first — not from here, so that several variants can be
#  built in parallel without racing on the build directory)
mkdir -p $C/build_$N
OBJS=""
for o in $C/build/*.o; do
  case $o in *-hip-amdgcn-*|*-host-*) continue;; esac      # -save-temps by-products
  b=$(basename $o .o); keep=1
  for t in "$@"; do [ "$b" == "$(basename $t .hip)" ] && keep=0; done
  [ $keep == 1 ] && OBJS="$OBJS $o"
done
for t in "$@"; do
  b=$(basename $t .hip)
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fvisibility=hidden -DGCV_BUILD -mcode-object-version=5 -w $F -c $C/$b.hip -o $C/build_$N/$b.o
  OBJS="$OBJS $C/build_$N/$b.o"
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -no-hip-rt -o $R/genconvit_amd/lib/libgenconvit_hip_$N.so $OBJS
echo built $R/genconvit_amd/lib/libgenconvit_hip_$N.so
