#!/bin/bash
# usage: tools/pmc.sh <tag> "<counters>" <bench args...>   -- one rocprofv3 --pmc pass of bench.py (run on the GPU box)
TAG=$1; CNT=$2; shift 2
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/gpurun_out
timeout -s KILL 170 rocprofv3 --pmc $CNT -d $R/gpurun_out/$TAG --output-format csv -- python3 $R/bench.py "$@" > $R/gpurun_out/$TAG.log 2>&1
echo exit=$?
