#!/bin/bash
# usage (GPU box): tools/quick_bench.sh  -- C2 stage times + checksum of the in-tree library (no CPU baseline, no other charts)
timeout -k 10 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-other-charts --no-replay 2>gpurun_out/quick.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['stage_ms'], d.get('counts'), 'checksum', d['checksum'], 'OK' if d['checksum']==4490173931545220 else 'CHECKSUM DIFFERS')"
