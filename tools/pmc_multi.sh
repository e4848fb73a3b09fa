#!/bin/bash
# usage (GPU box): tools/pmc_multi.sh <tag> <script + args> -- <counter set 1> -- <counter set 2> ...   (one rocprofv3 --pmc run per set)
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
CMD=(); while [ "$1" != "--" ] && [ $# -gt 0 ]; do CMD+=("$1"); shift; done
cd /tmp && export TMPDIR=/tmp
i=0
while [ $# -gt 0 ]; do
  shift; SET=(); while [ "$1" != "--" ] && [ $# -gt 0 ]; do SET+=("$1"); shift; done
  timeout -s KILL 120 rocprofv3 --pmc "${SET[@]}" -d $R/gpurun_out/${TAG}_$i --output-format csv -- python3 $R/"${CMD[0]}" "${CMD[@]:1}" > $R/gpurun_out/${TAG}_$i.log 2>&1 || exit 1
  python3 $R/tools/pmc_sum.py $R/gpurun_out/${TAG}_$i | grep -E "k_caprot|k_raster|k_cover"
  i=$((i+1))
done
