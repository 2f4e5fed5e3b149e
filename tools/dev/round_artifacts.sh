#!/bin/bash
# usage: round_artifacts.sh <tag>   -- everything profiles/README.md cites for one round, written to gpurun_out/<tag>_*:
#   prof_round.sh (kernel-trace stats, FETCH / WRITE and MFMA-busy PMC passes, flow-op rooflines), the default bench line with
#   its flow / fp32 / parity sub-records, the R101 384x288 lines, the parity reports and the 300-frame clip log
tag=${1:-r03}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
timeout 1500 tools/dev/prof_round.sh $tag > gpurun_out/${tag}_prof_round.log 2>&1
export FT_TILE_CACHE=$R/gpurun_out/tile_cache_$tag.json
timeout 1200 python bench.py > gpurun_out/${tag}_pose_bench_line.json 2> gpurun_out/${tag}_pose_bench_line.err
timeout 300 python bench.py --workload flow --no-extras > gpurun_out/${tag}_flow_bench_line.json 2>/dev/null
for b in 16 64; do
  timeout 300 python bench.py --backbone resnet101 --res 384x288 --batch $b --no-extras --no-cpu-baseline --steps 200 > gpurun_out/${tag}_pose_r101_384x288_b${b}_bench.json 2>/dev/null
done
timeout 300 python bench.py --batch 128 --no-extras --no-cpu-baseline --steps 200 > gpurun_out/${tag}_pose_r50_b128_bench.json 2>/dev/null
timeout 600 python tests/parity_report.py > gpurun_out/${tag}_pose_parity_report.json 2>/dev/null
timeout 600 python tests/parity_report.py --workload flow > gpurun_out/${tag}_flow_parity_report.json 2>/dev/null
timeout 600 python tools/tracking/demo.py --frames 300 --fp16 > gpurun_out/${tag}_clip_300frames.log 2>&1
cp $FT_TILE_CACHE gpurun_out/${tag}_tile_benchmark_picks.json
for m in FlowNet2C FlowNet2CS FlowNet2CSS FlowNet2SD FlowNet2; do
  timeout 300 python bench.py --workload flow --flow-model $m --no-extras --no-cpu-baseline --no-roofline --steps 100 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$m', d['value'], d['ms_per_step'])"
done > gpurun_out/${tag}_flow_models.txt 2>&1
tail -c 1500 gpurun_out/${tag}_pose_bench_line.json; tail -5 gpurun_out/${tag}_clip_300frames.log; cat gpurun_out/${tag}_flow_models.txt
