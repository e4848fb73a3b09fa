#!/usr/bin/env python3
"""Per-launch SQ counters of one rocprofv3 --pmc pass (gpurun_out/<dir>) as profiles/<name>_valu_counters.json (read by bench.py's
roofline_valu).  usage: tools/valu_collect.py <dir in gpurun_out> <name in profiles> "<the command's arguments>" """
import csv, glob, json, os, sys, collections
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
d, name, cmd = sys.argv[1], sys.argv[2], sys.argv[3]
tot = collections.defaultdict(lambda: collections.defaultdict(float)); launches = collections.defaultdict(set)
for f in glob.glob(os.path.join(R, "gpurun_out", d, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        tot[k][r["Counter_Name"]] += float(r["Counter_Value"]); launches[k].add(r["Dispatch_Id"])
out = {"_note": "per launch, one rocprofv3 --pmc pass of `bench.py " + cmd + "` (tools/pmc.sh); SQ_INSTS_* count wave64 instructions; "
                "SQ_WAVE_CYCLES / SQ_ACTIVE_INST_* count quad-cycles per wave (MI355X_MICROARCH.md). Peak VALU issue: 256 CUs x 4 SIMD-32 x "
                "2.4 GHz / 2 cycles per wave64 instruction = 1.2288e12 /s; the CU's one scalar unit at one instruction per cycle: 256 x 2.4e9 = 6.144e11 /s",
       "workload": "C2 1000000 ways 16384x16384 @300dpi", "kernels": {}}
for k in sorted(tot):
    n = max(1, len(launches[k]))
    out["kernels"][k] = {c: round(v / n) for c, v in sorted(tot[k].items())}
    out["kernels"][k]["launches"] = n
json.dump(out, open(os.path.join(R, "profiles", name + "_valu_counters.json"), "w"), indent=1)
print({k: out["kernels"][k] for k in ("k_cover", "k_blend", "k_prim_setup") if k in out["kernels"]})
