#!/bin/bash
# usage: ab_libs.sh <variant names...>: the default library and every libflowtrack_hip_<name>.so on the default pose bench, twice each, interleaved
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for rep in 1 2; do
  for v in default "$@"; do
    if [ $v = default ]; then unset FT_LIB_PATH; else export FT_LIB_PATH=$R/flowtrack/pytorch_amd/libflowtrack_hip_$v.so; fi
    timeout 300 python bench.py --no-extras --no-cpu-baseline --steps 400 ${AB_ARGS} 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', d['value'], d['ms_per_step'], d.get('roofline',{}).get('frac'))"
  done
done
