#!/bin/bash
# Run on the GPU box: tools/soak.py over several seeds and modes (GPU against the oracle); one summary line per run.
cd ${GRAFT_REPO_ROOT:-.}
for spec in "101 60" "102 60" "103 60 big" "104 60 big" "105 60 dir" "106 60 dash" "107 60" "108 60 dirbig"; do
  timeout -k 10 250 python -u tools/soak.py $spec > gpurun_out/soak_$(echo $spec | tr ' ' '_').log 2>&1 || { echo "soak $spec: exit $?"; tail -3 gpurun_out/soak_$(echo $spec | tr ' ' '_').log; }
  echo "soak $spec: $(tail -1 gpurun_out/soak_$(echo $spec | tr ' ' '_').log)"
done
