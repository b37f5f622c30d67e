"""Reduce two rocprofv3 --pmc CSV runs (FETCH_SIZE, WRITE_SIZE; separate passes) of bench.py to HBM bytes per
launch per kernel.  The counters are in KiB (x1024).  On gfx950 FETCH_SIZE reports half the bytes of coalesced
streaming reads (MI355X_MICROARCH.md, HBM); tools/dev/fetch_calib.hip confirms exactly 0.5 for this library's own
patterns (4-, 8- and 16-byte loads per lane, and the 256-byte tile walk of g_decode_seg), so FETCH_SIZE is doubled
here.  For scattered 4-byte gathers that may over-state the reads by up to 2x; WRITE_SIZE is taken as is.
usage: pmc_traffic.py <workload> <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json>"""
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


workload = sys.argv[1]
fetch, write = per_kernel(sys.argv[2], "FETCH_SIZE"), per_kernel(sys.argv[3], "WRITE_SIZE")
out = {"workload": workload, "unit": "bytes per launch (average over the launches of the run)",
       "note": "2 x FETCH_SIZE x 1024 (gfx950 correction, calibrated with tools/dev/fetch_calib.hip) + WRITE_SIZE x 1024", "kernels": {}}
for k in sorted(set(fetch) | set(write)):
    f, w = fetch.get(k, (0.0, 0)), write.get(k, (0.0, 0))
    out["kernels"][k] = {"fetch": 2 * f[0], "write": w[0], "traffic": 2 * f[0] + w[0], "launches": max(f[1], w[1])}
json.dump(out, open(sys.argv[4], "w"), indent=1)
print(f"{len(out['kernels'])} kernels -> {sys.argv[4]}")
