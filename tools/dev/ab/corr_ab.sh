# dev A/B (one gpurun call = one box): correlation rows64 forms at configs[3] shapes
#   FT_CORR_DIRECT=0            the window-column tiles (round 3)
#   FT_CORR_PIPE=0              DIRECT columns, 2-byte band stores behind the products (FT_CORR_DIRECT_WAVES=8: R = 4; FT_CORR_DBG: ablations)
#   default                     DIRECT columns, software-pipelined (FT_CORR_LTR=1: band through wave-private LDS tiles, 16-byte stores)
set -x
for l in 0 1; do FT_CORR_LTR=$l timeout 900 python -m pytest tests/test_flow_gpu.py -x -q -k "correlation" 2>&1 | tail -3; done
for c in "0 1 0" "1 0 0" "1 1 0" "1 1 1" "0 1 0" "1 0 0" "1 1 0" "1 1 1"; do set -- $c; echo "== FT_CORR_DIRECT=$1 PIPE=$2 LTR=$3"; FT_CORR_DIRECT=$1 FT_CORR_PIPE=$2 FT_CORR_LTR=$3 timeout 300 python tools/dev/flow_ops_prof.py 2>&1 | grep -i corr | cut -c1-120; done
