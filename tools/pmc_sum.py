#!/usr/bin/env python3
"""Sum rocprofv3 --pmc counter_collection.csv per (kernel, counter). usage: pmc_sum.py <dir>"""
import csv, glob, sys, collections
tot = collections.defaultdict(float); n = collections.Counter()
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = (r["Kernel_Name"].split("(")[0][:40], r["Counter_Name"])
        tot[k] += float(r["Counter_Value"]); n[k] += 1
ker = sorted({k[0] for k in tot})
for kn in ker:
    print(kn, {c: (round(v), n[(k, c)]) for (k, c), v in sorted(tot.items()) if k == kn})
