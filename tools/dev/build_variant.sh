#!/bin/bash
# usage: build_variant.sh <name> <extra hipcc flags...>  -> flowtrack/pytorch_amd/libflowtrack_hip_<name>.so (A/B builds, FT_LIB_PATH selects)
R=$(cd $(dirname $0)/../.. && pwd); P=$R/flowtrack/pytorch_amd; name=$1; shift
mkdir -p $P/build_$name
pids=()
for s in conv_igemm conv_igemm8 bottleneck bottleneck_rstat bottleneck_stream bottleneck_cluster conv_direct conv_wstat aux_ops flow_ops crop_ops runtime; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -I$R/include -I$P/csrc "$@" -c $P/csrc/$s.hip -o $P/build_$name/$s.o &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p || exit 1; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -Wl,--version-script=$P/csrc/exports.map $P/build_$name/*.o -o $P/libflowtrack_hip_$name.so && echo built $P/libflowtrack_hip_$name.so
