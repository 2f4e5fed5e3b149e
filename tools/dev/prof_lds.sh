#!/bin/bash
# usage: prof_lds.sh <pose|flow>  -- LDS / wait counters per kernel of the bench step (own --pmc passes), summary to gpurun_out/lds_<w>.txt
w=${1:-pose}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp && cd $R
export FT_NO_BRANCHES=1
cmd="python bench.py --workload $w --steps 3 --warmup 2 --no-cpu-baseline --no-roofline --no-extras --fixed-warmup"
rm -rf gpurun_out/lds_$w; mkdir -p gpurun_out/lds_$w
timeout 400 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES -d gpurun_out/lds_$w/a -o pmc --output-format csv -- $cmd > gpurun_out/lds_$w/a.log 2>&1
timeout 400 rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS -d gpurun_out/lds_$w/b -o pmc --output-format csv -- $cmd > gpurun_out/lds_$w/b.log 2>&1
timeout 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VMEM_RD -d gpurun_out/lds_$w/c -o pmc --output-format csv -- $cmd > gpurun_out/lds_$w/c.log 2>&1
python tools/dev/pmc_summary.py gpurun_out/lds_$w > gpurun_out/lds_$w.txt
