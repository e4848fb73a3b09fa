#!/bin/bash
# usage (GPU box): tools/ko_pmc.sh <variant...>  -- SQ_INSTS_VALU of k_raster per diagnostic build
R=${GRAFT_REPO_ROOT:-$(pwd)}
for v in "$@"; do
  lib=$R/smrender_amd/libsmr_hip.so; [ "$v" != main ] && lib=$R/tools/libsmr_hip_$v.so
  export SMR_HIP_LIBRARY=$lib
  (cd /tmp && TMPDIR=/tmp timeout -s KILL 170 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU -d $R/gpurun_out/ko_$v --output-format csv -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-other-charts --captions 0 > $R/gpurun_out/ko_$v.log 2>&1) || exit 1
  echo $v $(python3 $R/tools/pmc_sum.py $R/gpurun_out/ko_$v | grep k_raster)
done
