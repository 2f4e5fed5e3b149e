# dev A/B (one gpurun call = one box): weight-step register slots of the 256-plane streamed block (FT_BNS_SLOTS / FT_BNS_XH_SLOTS = 3 / 4 / 6)
cd ${GRAFT_REPO_ROOT:-/root/repo}
echo "== R50 256x192 batch 64"; bash tools/dev/ab_libs.sh ns4 ns6
echo "== R101 384x288 batch 16"; AB_ARGS="--backbone resnet101 --res 384x288 --batch 16" bash tools/dev/ab_libs.sh ns4 ns6
echo "== flow tests on the default library"; timeout 900 python -m pytest tests/test_flow_gpu.py -x -q 2>&1 | tail -3
