# the round's last GPU call: every profiles/r04_* artifact on this box (round_artifacts.sh) + the launch-by-launch tables
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
bash tools/dev/round_artifacts.sh r04 > gpurun_out/r04_round_artifacts.log 2>&1; tail -c 1500 gpurun_out/r04_round_artifacts.log
NB_HOT=1 timeout 300 python tools/dev/net_bench.py resnet101 16 384 288 fp16 > gpurun_out/r04_netbench_r101_b16.txt 2>&1
timeout 300 python tools/dev/net_bench.py resnet50 64 256 192 fp16 > gpurun_out/r04_netbench_r50.txt 2>&1
timeout 300 python tools/dev/net_bench.py FlowNet2S 16 384 512 fp16 > gpurun_out/r04_netbench_flownet2s.txt 2>&1
