# dev A/B (one gpurun call = one box): correlation rows64 forms at configs[3] shapes, and the DIRECT form's ablations
set -x
timeout 900 python -m pytest tests/test_flow_gpu.py -x -q -k "correlation" 2>&1 | tail -3
for c in "0 8 0" "1 6 0" "1 8 0" "1 6 1" "1 6 2" "1 6 3" "1 8 1" "1 8 2"; do set -- $c; echo "== FT_CORR_DIRECT=$1 WAVES=$2 DBG=$3"; FT_CORR_DIRECT=$1 FT_CORR_DIRECT_WAVES=$2 FT_CORR_DBG=$3 timeout 300 python tools/dev/flow_ops_prof.py 2>&1 | grep -i corr | cut -c1-120; done
