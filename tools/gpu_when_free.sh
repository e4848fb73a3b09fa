#!/bin/bash
# Runs ONE gpurun call, waiting while the pool has no free box (exit 3 = nothing ran, nothing charged).
# usage: tools/gpu_when_free.sh <timeout_s> <outfile> '<command>'
T=$1; OUT=$2; shift 2
for i in $(seq 1 40); do
  /usr/local/graft/bin/gpurun --timeout "$T" -- "$@" > "$OUT" 2>&1
  rc=$?
  if [ $rc -ne 3 ]; then echo "gpurun rc=$rc (attempt $i)" >> "$OUT"; exit $rc; fi
  sleep 90
done
echo "gave up: no box" >> "$OUT"; exit 3
