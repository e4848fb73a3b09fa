#!/bin/bash
# Diagnostic (GPU box): bench.py's C2 step with diagnostic builds of the library (k_blend's batch size / loads in flight).
cd ${GRAFT_REPO_ROOT:-.}
for v in "$@"; do
  SMR_HIP_LIBRARY=$PWD/tools/lib_$v.so timeout -k 10 200 python bench.py --captions 0 > gpurun_out/bv_$v.json 2> gpurun_out/bv_$v.err
  rc=$?
  if [ $rc -ne 0 ]; then echo "$v: exit $rc"; tail -3 gpurun_out/bv_$v.err; exit $rc; fi
  python -c "
import json; d=json.load(open('gpurun_out/bv_$v.json')); print('$v', d['ms_per_step'], d['stage_ms']['raster_ms'], d['other_charts']['target_32k']['ms'])"
done
