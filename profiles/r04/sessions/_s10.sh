#!/bin/bash
# Tooling: round-4 session 10: arg-max keys pre-reduced per row of lanes (DPP) before the LDS atomic
O=gpurun_out/r04_s10; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -q > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.txt; tail -5 $O/pytest_gpu.txt | cut -c1-200
bash tools/gpu_kernel_stats.sh r04_s10 blockout general 2>&1 | grep irbpp | cut -c1-110
timeout 300 python tools/ab_matrix.py --repeat 2 blockout:4096:1:0 general:4096:1:0 blockout_k10:1024:1:0 > $O/ab_matrix.jsonl 2> $O/ab_matrix.err; cat $O/ab_matrix.jsonl | cut -c1-150
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d /tmp/pmc_lds -o x -- python $R/bench.py --no-cpu-baseline --no-extra --steps 30 --warmup 5 --prefill 150 --min-seconds 0 > /dev/null 2>&1
python - <<'PY'
import csv, collections, glob, json
d = collections.defaultdict(lambda: collections.defaultdict(list))
for path in glob.glob("/tmp/pmc_lds/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        d[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
summ = {k: {c: sum(v[len(v)//2:]) / max(1, len(v[len(v)//2:])) for c, v in cs.items()} for k, cs in d.items() if k.startswith("irbpp")}
json.dump(summ, open("/root/repo/gpurun_out/r04_s10/lds_summary.json", "w"), indent=1)
for k, cs in summ.items():
    print(k, {c: round(v) for c, v in cs.items()}, "conflict share", round(cs.get("SQ_LDS_BANK_CONFLICT", 0) / max(1, cs.get("SQ_LDS_IDX_ACTIVE", 1)), 3))
PY
