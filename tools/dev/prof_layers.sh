#!/bin/bash
# PMC counters per single layer (one process per layer so kernel names do not mix)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for L in "$@"; do
  out=$R/gpurun_out/pmc_$L; mkdir -p $out
  cd $R
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES -d $out/sq1 -o pmc --output-format csv -- env ITERS=3 python tools/dev/conv_bench.py pose fp16 $L > $out/sq1.log 2>&1
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_WAVES -d $out/sq2 -o pmc --output-format csv -- env ITERS=3 python tools/dev/conv_bench.py pose fp16 $L > $out/sq2.log 2>&1
  rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE -d $out/tcc1 -o pmc --output-format csv -- env ITERS=3 python tools/dev/conv_bench.py pose fp16 $L > $out/tcc1.log 2>&1
  rocprofv3 --pmc FETCH_SIZE -d $out/tcc2 -o pmc --output-format csv -- env ITERS=3 python tools/dev/conv_bench.py pose fp16 $L > $out/tcc2.log 2>&1
  rocprofv3 --pmc WRITE_SIZE -d $out/tcc3 -o pmc --output-format csv -- env ITERS=3 python tools/dev/conv_bench.py pose fp16 $L > $out/tcc3.log 2>&1
  echo "=== $L"; python tools/dev/pmc_summary.py $out | grep -A30 "conv_igemm" | head -34
done
