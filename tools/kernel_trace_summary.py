#!/usr/bin/env python
"""Tooling: per-kernel average duration over the LAST n calls of a rocprofv3 kernel trace CSV (the timed
region of bench.py), printed as JSON; optionally deletes the (large) CSV afterwards."""
import csv, collections, glob, json, os, sys
d, n = sys.argv[1], int(sys.argv[2])
out = {}
for path in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    per = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        per[r["Kernel_Name"].split("(")[0]].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
    for k, v in per.items():
        v.sort()
        w = v[-n:]
        out[k] = {"calls": len(v), "avg_us_last": sum(e - s for s, e in w) / len(w) / 1e3,
                  "min_us_last": min(e - s for s, e in w) / 1e3, "max_us_last": max(e - s for s, e in w) / 1e3}
    if len(sys.argv) > 3 and sys.argv[3] == "--rm":
        os.remove(path)
print(json.dumps(out, indent=1))
