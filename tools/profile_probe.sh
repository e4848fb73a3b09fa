#!/bin/bash
# Run on the GPU box (gpurun): kernel-trace stats + HBM traffic counters (separate --pmc passes) of one probe program.
# usage: tools/profile_probe.sh <tag> <python script under tools/> [args]      -> gpurun_out/<tag>_*
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 280 rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_trace --output-format csv -- python3 $R/tools/$@ > $OUT/${TAG}_trace.log 2>&1 || exit 1
echo "trace done"
timeout -s KILL 170 rocprofv3 --pmc FETCH_SIZE -d $OUT/${TAG}_fetch --output-format csv -- python3 $R/tools/$@ > $OUT/${TAG}_fetch.log 2>&1 || exit 1
echo "fetch done"
timeout -s KILL 170 rocprofv3 --pmc WRITE_SIZE -d $OUT/${TAG}_write --output-format csv -- python3 $R/tools/$@ > $OUT/${TAG}_write.log 2>&1 || exit 1
echo "write done"
