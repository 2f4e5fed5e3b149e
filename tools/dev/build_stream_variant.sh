#!/bin/bash
# usage: build_stream_variant.sh <name> <file.hip (in csrc)> <extra hipcc flags...>
#   -> flowtrack/pytorch_amd/libflowtrack_hip_<name>.so: ONE translation unit rebuilt with the extra flags, every other object taken
#      from the default build (flowtrack/pytorch_amd/build/*.o).  FT_LIB_PATH selects the library (A/B runs inside one gpurun call).
R=$(cd $(dirname $0)/../.. && pwd); P=$R/flowtrack/pytorch_amd; name=$1; src=$2; shift; shift
mkdir -p $P/build_$name
base=$(basename $src .hip)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -I$R/include -I$P/csrc "$@" -c $P/csrc/$src -o $P/build_$name/$base.o || exit 1
objs=""
for o in $P/build/*.o; do
  if [ $(basename $o) = $base.o ]; then objs="$objs $P/build_$name/$base.o"; else objs="$objs $o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -Wl,--version-script=$P/csrc/exports.map $objs -o $P/libflowtrack_hip_$name.so && echo built $P/libflowtrack_hip_$name.so
