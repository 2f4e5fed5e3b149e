#!/bin/bash
# usage: prof_pmc.sh <outdir> <cmd...>   -- separate PMC passes (no trace domains combined), summaries under gpurun_out/
out=$1; shift
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out/$out
run() { tag=$1; shift; timeout 180 rocprofv3 --pmc "$@" -d $R/gpurun_out/$out/$tag -o pmc --output-format csv -- "${CMD[@]}" > $R/gpurun_out/$out/$tag.log 2>&1; }
CMD=("$@")
cd $R
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES
run sq2 SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_WAVES
run tcc1 TCC_HIT_sum TCC_MISS_sum
run tcc2 FETCH_SIZE
run tcc3 WRITE_SIZE
find $R/gpurun_out/$out -name "*.csv" | head -20
