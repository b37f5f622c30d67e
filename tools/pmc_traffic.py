"""Reduce two rocprofv3 --pmc CSV runs (FETCH_SIZE, WRITE_SIZE; separate passes) of bench.py to HBM bytes per
launch per kernel.  FETCH_SIZE / WRITE_SIZE are in KiB (x1024).  On gfx950 FETCH_SIZE under-reports wide
coalesced streaming reads by 2x (MI355X_MICROARCH.md, HBM); these kernels mostly issue 4-byte and scattered
loads, for which the guide gives no calibration, so the raw reading is kept and the file says so.
usage: pmc_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json>"""
import csv
import json
import sys
from collections import defaultdict


def per_kernel(path, counter):
    tot, cnt = defaultdict(float), defaultdict(int)
    for row in csv.DictReader(open(path)):
        if row.get("Counter_Name") != counter:
            continue
        name = row["Kernel_Name"]
        tot[name] += float(row["Counter_Value"])
        cnt[name] += 1
    return {k: (tot[k] * 1024.0 / cnt[k], cnt[k]) for k in tot}


fetch, write = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE")
out = {"unit": "bytes per launch", "note": "FETCH_SIZE*1024 + WRITE_SIZE*1024, raw (no gfx950 wide-load x2 correction: access widths here are uncalibrated)", "kernels": {}}
for k in sorted(set(fetch) | set(write)):
    f, w = fetch.get(k, (0.0, 0)), write.get(k, (0.0, 0))
    out["kernels"][k] = {"fetch": f[0], "write": w[0], "traffic": f[0] + w[0], "launches": max(f[1], w[1])}
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(f"{len(out['kernels'])} kernels -> {sys.argv[3]}")
