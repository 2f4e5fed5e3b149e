#!/bin/bash
# HBM traffic of one bench.py forward: separate --pmc passes (FETCH_SIZE, WRITE_SIZE), no trace domains combined.
tag=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
cd $R
mkdir -p gpurun_out/traffic_$tag
timeout 300 rocprofv3 --pmc FETCH_SIZE -d gpurun_out/traffic_$tag/fetch -o pmc --output-format csv -- "$@" > gpurun_out/traffic_$tag/fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE -d gpurun_out/traffic_$tag/write -o pmc --output-format csv -- "$@" > gpurun_out/traffic_$tag/write.log 2>&1
