#!/usr/bin/env python3
"""Copies the summaries of one tools/profile_round.sh run from gpurun_out/ (scratch) into profiles/ (tracked).
usage: tools/profiles_collect.py <tag in gpurun_out> <name in profiles>"""
import glob, json, os, shutil, subprocess, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, name = sys.argv[1], sys.argv[2]
G, P = os.path.join(R, "gpurun_out"), os.path.join(R, "profiles")
st = glob.glob(f"{G}/{tag}_trace/**/*kernel_stats.csv", recursive=True)[0]
shutil.copy(st, f"{P}/{name}_kernel_stats.csv")
for line in open(f"{G}/{tag}_trace.log"):
    if line.startswith('{"metric'):
        open(f"{P}/{name}_bench_under_rocprof.json", "w").write(line)
raw = json.loads(subprocess.run([sys.executable, f"{R}/tools/pmc_traffic.py", f"{G}/{tag}_fetch", f"{G}/{tag}_write"],
                                capture_output=True, text=True, check=True).stdout)
json.dump(raw, open(f"{P}/{name}_pmc_fetch_write_raw.json", "w"), indent=1)
out = {"_note": "HBM bytes per launch = FETCH_SIZE[KiB]*1024*2 (gfx950: 128-B requests tallied at 64 B, MI355X_MICROARCH.md HBM section) "
                "+ WRITE_SIZE[KiB]*1024; separate --pmc passes of `bench.py --steps 3 --warmup 1 --no-cpu-baseline` (tools/profile_round.sh)",
       "workload": "C2 1000000 ways 16384x16384 @300dpi", "kernels": {}}
for k, v in raw.items():
    f, w = v["fetch_raw_per_launch"] * 1024 * 2, v["write_raw_per_launch"] * 1024
    out["kernels"][k] = {"fetch_bytes": f, "write_bytes": w, "hbm_bytes": f + w}
json.dump(out, open(f"{P}/{name}_hbm_traffic.json", "w"), indent=1)
print("k_raster", out["kernels"].get("k_raster"))
