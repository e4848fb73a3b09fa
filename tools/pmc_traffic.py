#!/usr/bin/env python3
"""Per-kernel HBM traffic from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes.
FETCH_SIZE / WRITE_SIZE are reported in KiB-like units of 1 KB by rocprofv3's derived metric on this
stack (verified below against k_fill_u32, whose bytes are known); gfx950 correction per
MI355X_MICROARCH.md §HBM: FETCH_SIZE counts 128-byte requests at 64 B -> doubled.
usage: pmc_traffic.py <fetch_dir> <write_dir>"""
import csv, glob, sys, collections, json


def load(d, name):
    tot = collections.defaultdict(float); n = collections.Counter()
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != name:
                continue
            k = r["Kernel_Name"].split("(")[0]
            if k.startswith("void "):
                k = k[5:]
            if k == "k_raster<false>":      # the fast instantiation is THE rasterizer; k_raster<true> only re-renders overflowed tiles
                k = "k_raster"
            tot[k] += float(r["Counter_Value"]); n[k] += 1
    return tot, n


f, nf = load(sys.argv[1], "FETCH_SIZE")
w, nw = load(sys.argv[2], "WRITE_SIZE")
out = {}
for k in sorted(set(f) | set(w)):
    out[k] = {"launches": int(max(nf[k], nw[k])),
              "fetch_raw_per_launch": f[k] / max(nf[k], 1), "write_raw_per_launch": w[k] / max(nw[k], 1)}
print(json.dumps(out, indent=1))
