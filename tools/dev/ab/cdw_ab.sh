# dev A/B (one gpurun call = one box): weight-step register slots of conv_direct_kernel's unrolled walks (-DFT_CD_WSLOTS=3 | 4)
cd ${GRAFT_REPO_ROOT:-/root/repo}
echo "== conv_direct tests on the 4-slot build"; FT_LIB_PATH=$PWD/flowtrack/pytorch_amd/libflowtrack_hip_cdw4.so timeout 900 python -m pytest tests/test_conv_direct_gpu.py -x -q 2>&1 | tail -3
echo "== R50 256x192 batch 64"; bash tools/dev/ab_libs.sh cdw4
echo "== R101 384x288 batch 16"; AB_ARGS="--backbone resnet101 --res 384x288 --batch 16" bash tools/dev/ab_libs.sh cdw4
echo "== FlowNet2S"; AB_ARGS="--workload flow" bash tools/dev/ab_libs.sh cdw4
