#!/bin/bash
# Tooling: dynamic instruction counts of the transition kernel's phases.  The -DIRBPP_ABLATE build (tools/build_variant.sh
# ablate -DIRBPP_ABLATE) runs one phase twice per IRBPP_DEBUG_REPEAT bit; the difference of the SQ instruction counters
# of a PMC pass with and without the bit is the phase's own count.  Uncapped run-time-path build (--tuning 2): the capped
# builds spill under the extra loops.   tools/env_phase_pmc.sh <out dir under gpurun_out/> [workload] [bins]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/$1; WL=${2:-blockout}; BINS=${3:-4096}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
export IRBPP_LIBRARY=$R/irbpp_amd/libirbpp_var_ablate.so
for bit in ${BITS:-0 32 64 4 128 256 512 1024 2048 4096 8192 16384}; do
  rm -rf /tmp/pmc_$bit
  IRBPP_DEBUG_REPEAT=$bit rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU \
    --output-format csv -d /tmp/pmc_$bit -o x -- python $R/bench.py --workload $WL --bins $BINS --tuning 2 --no-cpu-baseline --no-extra \
    --steps 30 --warmup 5 --prefill 120 --min-seconds 0 > "$OUT/bench_$bit.json" 2> "$OUT/err_$bit.txt"
  python - "$OUT" $bit <<'PY'
import csv, collections, glob, json, sys
out, bit = sys.argv[1], sys.argv[2]
d = collections.defaultdict(lambda: collections.defaultdict(list))
for path in glob.glob(f"/tmp/pmc_{bit}/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        d[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
summ = {k: {c: sum(v[len(v) // 2:]) / max(1, len(v[len(v) // 2:])) for c, v in cs.items()} for k, cs in d.items() if k.startswith("irbpp_env")}
json.dump(summ, open(f"{out}/pmc_{bit}.json", "w"), indent=1)
for k, cs in summ.items():
    w = max(cs.get("SQ_WAVES", 1), 1)
    print(bit, k, {c: round(v / w, 1) for c, v in cs.items() if c.startswith("SQ_INSTS")}, "waves", int(w))
PY
done
