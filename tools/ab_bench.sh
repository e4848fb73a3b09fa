#!/bin/bash
# usage (GPU box): tools/ab_bench.sh <variant names...>   -- stage times of bench.py per diagnostic build
for v in "$@"; do
  SMR_HIP_LIBRARY=$PWD/tools/libsmr_hip_$v.so timeout -k 10 150 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-other-charts --captions 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', d['stage_ms'], d['counts']['cover_prims'], d['counts']['deltas'], 'checksum', d['checksum'])" || exit 1
done
