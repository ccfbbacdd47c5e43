#!/bin/bash
# Tooling: the rocprofv3 passes behind profiles/ (run on the GPU box through gpurun).
#   tools/gpu_profile.sh <out dir under gpurun_out/> <workload> <bins for the PMC passes>
# kernel trace + stats at the bench's default size; FETCH_SIZE, WRITE_SIZE and two SQ groups each in their own
# pass (MI355X_MICROARCH.md: TCC slots; never combined with other trace domains) at <bins> bins per launch, chosen
# so that the working set exceeds the 256 MiB Infinity Cache.
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/$1; WL=${2:-blockout}; BINS=${3:-16384}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --workload $WL --no-cpu-baseline --no-extra --groups 1"     # one launch group: per-launch counters = per-step counters
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt" -o r06 -- $B --steps 200 --warmup 20 --min-seconds 0 > "$OUT/bench_under_rocprof.json" 2> "$OUT/kt.err"
python $R/tools/kernel_trace_summary.py "$OUT/kt" 200 --rm > "$OUT/kernel_trace_timed_region.json"
f=$(find "$OUT/kt" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" "$OUT/kernel_stats.csv"
P="$B --bins $BINS --steps 60 --warmup 10 --prefill 150 --min-seconds 0"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -o r06 -- $P > "$OUT/bench_pmc_fetch.json" 2> "$OUT/fetch.err"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -o r06 -- $P > /dev/null 2> "$OUT/write.err"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INSTS_VALU --output-format csv -d "$OUT/sq" -o r06 -- $P > /dev/null 2> "$OUT/sq.err"
rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES --output-format csv -d "$OUT/sq2" -o r06 -- $P > /dev/null 2> "$OUT/sq2.err"
# the raw per-dispatch CSVs are large: keep per-kernel averages only
python - "$OUT" <<'PY'
import csv, collections, glob, json, os, sys
out = sys.argv[1]
for sub in ("fetch", "write", "sq", "sq2"):
    for path in glob.glob(os.path.join(out, sub, "**", "*counter_collection.csv"), recursive=True):
        d = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(path)):
            d[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        summ = {k: {c: {"n": len(v), "avg_second_half": sum(v[len(v) // 2:]) / max(1, len(v[len(v) // 2:]))} for c, v in cs.items()}
                for k, cs in d.items()}
        json.dump(summ, open(os.path.join(out, sub + "_summary.json"), "w"), indent=1)
        if os.path.getsize(path) > 4 << 20:
            os.remove(path)
PY
