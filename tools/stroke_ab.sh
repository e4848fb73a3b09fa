cd $GRAFT_REPO_ROOT
for m in pieces polygon; do
  SMR_HIP_STROKE=$m ORC_STROKE_OUTLINE=$([ $m = polygon ] && echo 1 || echo "") timeout -k 10 300 python -u bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-other-charts --no-replay --captions 0 > gpurun_out/stroke_$m.json 2> gpurun_out/stroke_$m.err || exit 1
  python3 -c "
import json; d=json.load(open('gpurun_out/stroke_$m.json')); print('$m', d['ms_per_step'], d['stage_ms'], d['counts'])"
done
