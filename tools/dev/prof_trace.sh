#!/bin/bash
# rocprofv3 --kernel-trace --stats of bench.py (the same command the bench line comes from); copies the
# per-kernel summary to gpurun_out/<tag>_kernel_stats.csv
tag=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
cd $R
timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$tag -o trace --output-format csv -- "$@" > $R/gpurun_out/prof_$tag.log 2>&1
f=$(find $R/gpurun_out/prof_$tag -name "*kernel_stats.csv" | head -1)
cp "$f" $R/gpurun_out/${tag}_kernel_stats.csv
head -25 $R/gpurun_out/${tag}_kernel_stats.csv
# drop the bulky raw trace, keep the summary
find $R/gpurun_out/prof_$tag -name "*kernel_trace.csv" -size +8M -delete
