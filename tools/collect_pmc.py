#!/usr/bin/env python
"""Tooling: fold rocprofv3 --pmc passes of bench.py into profiles/pmc_hbm.json, stamped with the hash of
the kernel sources they were taken on (bench.py quotes them only while that hash still matches).

    python tools/collect_pmc.py <dir with fetch/ write/ sq/ [sq2/] sub-directories> <workload> <bins> [<git rev>]

Each sub-directory holds the counter_collection.csv of ONE pass (MI355X_MICROARCH.md: FETCH_SIZE and
WRITE_SIZE do not fit one pass; SQ has eight slots).  A step is a chain of kernels ([apply,] transition, trace,
polygon, emit), each launched once per step of ONE group of bins (the passes run bench.py --groups 1): per-step figures are sums of the kernels' per-launch averages over the
second half of the run (prefill and warm-up dropped).  HBM bytes = 2 x FETCH_SIZE + WRITE_SIZE (KB -> bytes):
gfx950's FETCH_SIZE reports half the bytes of wide coalesced reads (same guide, HBM section); the emit kernel,
whose reads and writes per bin are known exactly, is kept as the calibration of both factors."""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from irbpp_amd.build import source_hash  # noqa: E402

src, workload, bins = sys.argv[1], sys.argv[2], int(sys.argv[3])
rev = sys.argv[4] if len(sys.argv) > 4 else ""
STEP_KERNELS = ("irbpp_apply_kernel", "irbpp_env_kernel", "irbpp_trace_kernel", "irbpp_polygon_kernel", "irbpp_emit_kernel",
                "irbpp_emit_wave_kernel")


def agg(sub):
    d = collections.defaultdict(lambda: collections.defaultdict(list))
    paths = sorted(glob.glob(os.path.join(src, sub, "**", "*counter_collection.csv"), recursive=True))
    if len(paths) > 1:      # gpurun_out/ is merged into, never cleaned: a pass of an earlier round must not be averaged in
        raise SystemExit(f"{os.path.join(src, sub)}: more than one counter_collection.csv ({paths}): remove the stale ones")
    for path in paths:
        with open(path) as f:
            for r in csv.DictReader(f):
                d[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    summ = os.path.join(src, sub + "_summary.json")          # written by tools/gpu_profile.sh when the CSV was too large to keep
    if not d and os.path.exists(summ):
        for k, cs in json.load(open(summ)).items():
            for c, v in cs.items():
                d[k][c] = [v["avg_second_half"]] * 2
    return d


def tail_avg(v):
    v = v[len(v) // 2:]
    return sum(v) / len(v)


def step_kernels(d):
    return [k for k in d if k.startswith(STEP_KERNELS)]


out_path = os.path.join(ROOT, "profiles", "pmc_hbm.json")
doc = json.load(open(out_path)) if os.path.exists(out_path) else {}
if doc.get("kernel_source_sha") != source_hash():
    doc = {"kernel_source_sha": source_hash(), "git_rev": rev, "workloads": {}}
f, w = agg("fetch"), agg("write")
per_kernel = {}
for k in step_kernels(f):
    fk = tail_avg(f[k]["FETCH_SIZE"])
    wk = tail_avg(w[k]["WRITE_SIZE"]) if k in w else 0.0
    per_kernel[k] = {"FETCH_SIZE_KB_avg": fk, "WRITE_SIZE_KB_avg": wk, "hbm_bytes_per_launch": (2 * fk + wk) * 1024}
total = sum(v["hbm_bytes_per_launch"] for v in per_kernel.values())
entry = {"bins": bins, "kernels": per_kernel, "hbm_bytes_per_launch": total, "hbm_bytes_per_bin_step": total / bins,
         "note": "hbm_bytes_per_launch = one step = one launch of each of the listed kernels",
         "command": "rocprofv3 --pmc <counters> --output-format csv -- python bench.py --bins %d --workload %s "
                    "--no-cpu-baseline --no-extra (each counter group in its own pass)" % (bins, workload)}
emit_name = next((k for k in per_kernel if k.startswith("irbpp_emit")), None)
if emit_name is not None and len(sys.argv) > 5:
    r_ac, sel = int(sys.argv[5]), 500
    true_r, true_w = bins * (r_ac * 8 + r_ac // 16 * 4 + 32), bins * (5 * sel * 4 + sel * 4 + 4)
    e = per_kernel[emit_name]
    entry["calibration"] = {"kernel": emit_name, "true_read_bytes": true_r, "true_write_bytes": true_w,
                            "two_x_FETCH_SIZE_over_true_reads": 2 * e["FETCH_SIZE_KB_avg"] * 1024 / true_r,
                            "WRITE_SIZE_over_true_writes": e["WRITE_SIZE_KB_avg"] * 1024 / true_w}
sq = collections.defaultdict(dict)
for sub in ("sq", "sq2"):
    d = agg(sub)
    for k in step_kernels(d):
        for c, v in d[k].items():
            sq[k][c] = tail_avg(v)
if sq:
    entry["sq_per_launch_avg"] = sq
    tot = collections.defaultdict(float)
    for k, cs in sq.items():
        for c, v in cs.items():
            tot[c] += v
    wc = tot.get("SQ_WAVE_CYCLES")
    if wc:
        entry["issue"] = {k2: tot[k1] / wc for k1, k2 in (("SQ_ACTIVE_INST_VALU", "valu_share_of_wave_cycles"),
                                                           ("SQ_ACTIVE_INST_SCA", "salu_share_of_wave_cycles"),
                                                           ("SQ_ACTIVE_INST_LDS", "lds_share_of_wave_cycles"),
                                                           ("SQ_WAIT_ANY", "wait_share_of_wave_cycles"),
                                                           ("SQ_WAIT_INST_ANY", "issue_stall_share_of_wave_cycles"))
                          if k1 in tot}
        if "SQ_INSTS_VALU" in tot:
            entry["issue"]["valu_insts_per_bin_step"] = tot["SQ_INSTS_VALU"] / bins
        if "SQ_INSTS_SALU" in tot:
            entry["issue"]["salu_insts_per_bin_step"] = tot["SQ_INSTS_SALU"] / bins
        entry["issue"]["per_kernel_wait_share"] = {k: cs["SQ_WAIT_ANY"] / cs["SQ_WAVE_CYCLES"] for k, cs in sq.items()
                                                   if "SQ_WAIT_ANY" in cs and cs.get("SQ_WAVE_CYCLES")}
doc["workloads"][workload] = entry
json.dump(doc, open(out_path, "w"), indent=1)
print(json.dumps(entry, indent=1)[:3000])
