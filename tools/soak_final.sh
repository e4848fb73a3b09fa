#!/bin/bash
# Run on the GPU box: tools/soak.py over further seeds and modes (GPU against the oracle); one summary line per run.
# A run that ends abnormally (time limit, abort, fault) ends the whole script: nothing more goes to the GPU after it.
cd ${GRAFT_REPO_ROOT:-.}
for spec in "301 100" "302 100 dash" "303 100 big" "304 100 dir" "305 100 dash" "306 100 dashbig" "307 100 dirbig" "308 100 dash" "309 100" "310 100 dash"; do
  log=gpurun_out/soakF_$(echo $spec | tr ' ' '_').log
  timeout -k 10 280 python -u tools/soak.py $spec > $log 2>&1
  rc=$?
  echo "soak $spec: $(tail -1 $log)"
  if [ $rc -ge 124 ] || [ $rc -lt 0 ]; then echo "soak $spec: exit $rc, stopping"; tail -3 $log; exit $rc; fi
done
