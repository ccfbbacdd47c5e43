#!/usr/bin/env python
"""Tooling: fold one gpurun_out/<dir> of rocprofv3 passes (kt/, fetch/, write/, sq/, lds/ and the
bench/phase/census JSON lines written next to them) into profiles/r01_final and
profiles/r01_pmc_hbm.json.  Usage: python tools/collect_profiles.py gpurun_out/final7 "<kernel state>"."""
import collections
import csv
import json
import os
import shutil
import sys

src, state = sys.argv[1], sys.argv[2]
dst = "profiles/r01_final"


def agg(path):
    d = collections.defaultdict(lambda: collections.defaultdict(list))
    with open(path) as f:
        for r in csv.DictReader(f):
            d[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return d


def avg(v):
    v = v[len(v) // 3:]              # drop the warm-up third
    return sum(v) / len(v)


out = json.load(open("profiles/r01_pmc_hbm.json"))
b = out["blockout"]
f, w = agg(f"{src}/fetch/r01_counter_collection.csv"), agg(f"{src}/write/r01_counter_collection.csv")
env = [k for k in f if k.startswith("irbpp_env_kernel")][0]
pol = [k for k in f if "policy" in k][0]
fk, wk = avg(f[env]["FETCH_SIZE"]), avg(w[env]["WRITE_SIZE"])
b["FETCH_SIZE_KB_avg"], b["WRITE_SIZE_KB_avg"] = fk, wk
b["calibration"]["policy_kernel_FETCH_SIZE_KB"] = avg(f[pol]["FETCH_SIZE"])
b["hbm_bytes_per_launch"] = (2 * fk + wk) * 1024
for name, key in (("sq", "sq_per_launch_avg"), ("lds", "lds_grbm_per_launch_avg")):
    path = f"{src}/{name}/r01_counter_collection.csv"
    if os.path.exists(path):
        b[key] = {c: avg(v) for c, v in agg(path)[env].items()}
b["kernel_state"] = state
json.dump(out, open("profiles/r01_pmc_hbm.json", "w"), indent=1)

rows = [r for r in csv.DictReader(open(f"{src}/kt/r01_kernel_trace.csv")) if r["Kernel_Name"].startswith("irbpp_env_kernel")]
d = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows]
summ = {"irbpp_env_kernel_calls": len(d), "avg_ns_all_calls": sum(d) / len(d),
        "avg_ns_last_200_calls (the timed region of bench.py)": sum(d[-200:]) / 200,
        "bench_live_kernel_ms_same_run": json.load(open(f"{src}/bench_under_rocprof.json"))["roofline"]["kernel_ms"]}
json.dump(summ, open(f"{dst}/kernel_trace_timed_region.json", "w"), indent=1)
shutil.copy(f"{src}/kt/r01_kernel_stats.csv", f"{dst}/kernel_stats.csv")
os.makedirs(f"{dst}/other_workloads", exist_ok=True)
for n in os.listdir(src):
    if n in ("bench_default.json", "bench_under_rocprof.json") or n.startswith(("phase_", "census_")):
        shutil.copy(f"{src}/{n}", f"{dst}/{n}")
    elif n.startswith("bench_") or n == "vecenv.txt":
        shutil.copy(f"{src}/{n}", f"{dst}/other_workloads/{n}")
print(json.dumps(summ))
print(json.dumps({k: v for k, v in b.items() if k != "calibration"}, indent=1))
