#!/bin/bash
# Run on the GPU box: tools/soak.py over several seeds and modes (GPU against the oracle); one summary line per run.
# A run that ends abnormally (time limit, abort, fault) ends the whole script: nothing more goes to the GPU after it.
cd ${GRAFT_REPO_ROOT:-.}
for spec in "101 60" "102 60" "103 60 big" "104 60 big" "105 60 dir" "106 60 dash" "107 60" "108 60 dirbig"; do
  log=gpurun_out/soak_$(echo $spec | tr ' ' '_')${SOAK_TAG}.log
  timeout -k 10 250 python -u tools/soak.py $spec > $log 2>&1
  rc=$?
  echo "soak $spec: $(tail -1 $log)"
  if [ $rc -ge 124 ] || [ $rc -lt 0 ]; then echo "soak $spec: exit $rc, stopping"; tail -3 $log; exit $rc; fi
done
