# dev A/B (one gpurun call = one box): the mean fold of FlowNet2S and the direct-column correlation tiles
set -x
timeout 900 python -m pytest tests/test_flow_gpu.py -x -q 2>&1 | tail -5
timeout 900 python -m pytest tests/test_fullsize_gpu.py -x -q -k "flow or Flow" 2>&1 | tail -5
for i in 1 2; do
for f in 0 1; do echo "== FT_MEAN_FOLD=$f"; FT_MEAN_FOLD=$f timeout 300 python tools/dev/net_bench.py FlowNet2S 16 384 512 fp16 2>&1 | grep -E "^ +[0-2] |sum of|graph replay"; done
done
for c in "0 8" "1 8" "1 6" "0 8" "1 8" "1 6"; do set -- $c; echo "== FT_CORR_DIRECT=$1 WAVES=$2"; FT_CORR_DIRECT=$1 FT_CORR_DIRECT_WAVES=$2 timeout 300 python tools/dev/flow_ops_prof.py 2>&1 | grep -i corr; done
