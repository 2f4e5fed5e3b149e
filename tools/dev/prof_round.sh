#!/bin/bash
# usage: prof_round.sh <round tag, e.g. r01>   -- regenerates every file under profiles/ for pose and flow:
#   1. one plain bench run per workload that benchmarks the tiles and persists the picks (FT_TILE_CACHE)
#   2. rocprofv3 --kernel-trace --stats of the same command (kernels = the bench's kernels, no tuning launches)
#   3. separate --pmc FETCH_SIZE / WRITE_SIZE passes of a short fixed-length run -> HBM bytes per conv launch
tag=${1:-r01}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
export FT_TILE_CACHE=$R/gpurun_out/tile_cache_$tag.json
rm -f $FT_TILE_CACHE
for w in pose flow; do
  timeout 300 python bench.py --workload $w --no-cpu-baseline --no-extras --steps 200 > gpurun_out/${tag}_${w}_bench.json 2> gpurun_out/${tag}_${w}_bench.err
  # kernel-level passes run the launch list in order (FT_NO_BRANCHES=1): inside parallel graph branches two kernels share
  # the GPU and each one's traced duration stretches, which is not what roofline.avg_launch_us (per-kernel, in order) means
  export FT_NO_BRANCHES=1
  timeout 600 tools/dev/prof_trace.sh ${tag}_${w}_bench python bench.py --workload $w --steps 20 --warmup 3 --no-cpu-baseline --no-roofline --no-extras --fixed-warmup --rotate 1 > /dev/null 2>&1
  timeout 900 tools/dev/prof_traffic.sh ${tag}_${w} python bench.py --workload $w --steps 4 --warmup 2 --no-cpu-baseline --no-roofline --no-extras --fixed-warmup --rotate 1
  # forwards in that run: the first call runs the launch list twice eagerly (plain + after the tile picks), then 1
  # warm-up replay + 4 timed replays
  python tools/dev/pmc_traffic.py gpurun_out/traffic_${tag}_${w} 7 gpurun_out/${tag}_${w}_hbm_traffic_pmc.json
  # 4. MFMA-busy fraction per kernel (its own --pmc pass)
  ( cd /tmp && export TMPDIR=/tmp && cd $R && timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d gpurun_out/mfma_${tag}_${w} -o pmc --output-format csv -- python bench.py --workload $w --steps 4 --warmup 2 --no-cpu-baseline --no-roofline --no-extras --fixed-warmup --rotate 1 > gpurun_out/mfma_${tag}_${w}.log 2>&1 )
  python tools/dev/pmc_mfma.py gpurun_out/mfma_${tag}_${w} gpurun_out/${tag}_${w}_mfma_busy_pmc.json
  unset FT_NO_BRANCHES
done
# 5. the custom operators of the FLOW path at configs[3] shapes (correlation / warp+concat / resample2d / channelnorm):
#    kernel trace + FETCH / WRITE passes of tools/dev/flow_ops_prof.py
timeout 300 tools/dev/prof_trace.sh ${tag}_flow_ops python tools/dev/flow_ops_prof.py > /dev/null 2>&1
timeout 600 tools/dev/prof_traffic.sh ${tag}_flow_ops python tools/dev/flow_ops_prof.py
python tools/dev/pmc_traffic.py gpurun_out/traffic_${tag}_flow_ops 22 gpurun_out/${tag}_flow_ops_hbm_traffic_pmc.json
python tools/dev/flow_ops_prof.py --json > gpurun_out/${tag}_flow_ops_rooflines.json 2>/dev/null
tail -c 600 gpurun_out/${tag}_pose_bench.json
