# dev A/B (one gpurun call = one box): correlation rows64 forms at configs[3] shapes
#   FT_CORR_DIRECT=0            the window-column tiles (round 3)
#   FT_CORR_DIRECT_WAVES=8      DIRECT columns, eight waves / R = 4 (192 workgroups at 16 x 48 rows)
#   default                     DIRECT columns, six waves / R = 3 (256 workgroups)
#   FT_CORR_DBG=<mask>          ablations of the DIRECT form (timing only): 1 stores out of range, 2 no MFMAs, 4 no fragment reads,
#                               8 no ring loads, 16 no barriers, 32 no band store instructions
set -x
timeout 900 python -m pytest tests/test_flow_gpu.py -x -q -k "correlation" 2>&1 | tail -3
for c in "0 8 0" "1 8 0" "1 6 0" "1 6 1" "1 6 32" "1 6 34" "1 6 38" "1 6 46" "1 6 62"; do set -- $c; echo "== FT_CORR_DIRECT=$1 WAVES=$2 DBG=$3"; FT_CORR_DIRECT=$1 FT_CORR_DIRECT_WAVES=$2 FT_CORR_DBG=$3 timeout 300 python tools/dev/flow_ops_prof.py 2>&1 | grep -i corr | cut -c1-120; done
