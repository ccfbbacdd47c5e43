#!/usr/bin/env python
"""Tooling: fold rocprofv3 --pmc passes of bench.py into profiles/pmc_hbm.json, stamped with the hash of
the kernel sources they were taken on (bench.py quotes them only while that hash still matches).

    python tools/collect_pmc.py <dir with fetch/ write/ sq/ [sq2/] sub-directories> <workload> <bins> [<git rev>]

Each sub-directory holds the counter_collection.csv of ONE pass (MI355X_MICROARCH.md: FETCH_SIZE and
WRITE_SIZE do not fit one pass; SQ has eight slots).  HBM bytes per launch = 2 x FETCH_SIZE + WRITE_SIZE
(KB -> bytes): gfx950's FETCH_SIZE reports half the bytes of wide coalesced reads (same guide, HBM section);
the policy kernel, whose read volume is known exactly, is kept as the calibration of that factor."""
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


def agg(sub):
    d = collections.defaultdict(lambda: collections.defaultdict(list))
    for path in glob.glob(os.path.join(src, sub, "**", "*counter_collection.csv"), recursive=True):
        with open(path) as f:
            for r in csv.DictReader(f):
                d[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return d


def tail_avg(v):
    v = v[len(v) // 2:]              # the timed half: prefill and warm-up launches dropped
    return sum(v) / len(v)


out_path = os.path.join(ROOT, "profiles", "pmc_hbm.json")
doc = json.load(open(out_path)) if os.path.exists(out_path) else {}
if doc.get("kernel_source_sha") != source_hash():
    doc = {"kernel_source_sha": source_hash(), "git_rev": rev, "workloads": {}}
f, w = agg("fetch"), agg("write")
env = [k for k in f if k.startswith("irbpp_env_kernel")][0]
pol = [k for k in f if "policy" in k]
fk, wk = tail_avg(f[env]["FETCH_SIZE"]), tail_avg(w[env]["WRITE_SIZE"])
entry = {"bins": bins, "kernel": env, "FETCH_SIZE_KB_avg": fk, "WRITE_SIZE_KB_avg": wk,
         "hbm_bytes_per_launch": (2 * fk + wk) * 1024,
         "hbm_bytes_per_bin_step": (2 * fk + wk) * 1024 / bins,
         "command": "rocprofv3 --pmc <counters> --output-format csv -- python bench.py --bins %d --workload %s "
                    "--no-cpu-baseline --no-extra (each counter group in its own pass)" % (bins, workload)}
if pol:
    entry["calibration"] = {"policy_kernel_FETCH_SIZE_KB": tail_avg(f[pol[0]]["FETCH_SIZE"]),
                            "policy_kernel_true_read_bytes": bins * 2500 * 4,
                            "note": "the policy kernel reads every 128-B line of the [bins][2500] f32 candidate block"}
sq = {}
for sub in ("sq", "sq2"):
    for c, v in agg(sub).get(env, {}).items():
        sq[c] = tail_avg(v)
if sq:
    entry["sq_per_launch_avg"] = sq
    wc = sq.get("SQ_WAVE_CYCLES")
    if wc:
        entry["issue"] = {k2: sq[k1] / wc for k1, k2 in (("SQ_ACTIVE_INST_VALU", "valu_share_of_wave_cycles"),
                                                          ("SQ_ACTIVE_INST_SCA", "salu_share_of_wave_cycles"),
                                                          ("SQ_ACTIVE_INST_LDS", "lds_share_of_wave_cycles"),
                                                          ("SQ_WAIT_ANY", "wait_share_of_wave_cycles"),
                                                          ("SQ_WAIT_INST_ANY", "issue_stall_share_of_wave_cycles"))
                          if k1 in sq}
        if "SQ_INSTS_VALU" in sq:
            entry["issue"]["valu_insts_per_bin_step"] = sq["SQ_INSTS_VALU"] / bins
        if "SQ_INSTS_SALU" in sq:
            entry["issue"]["salu_insts_per_bin_step"] = sq["SQ_INSTS_SALU"] / bins
doc["workloads"][workload] = entry
json.dump(doc, open(out_path, "w"), indent=1)
print(json.dumps(entry, indent=1))
