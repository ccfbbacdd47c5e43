#!/bin/bash
# Tooling: fold gpurun_out/final (written by profiles/r06/sessions/_final.sh on the GPU box) into profiles/r06/final and
# profiles/pmc_hbm.json.   bash tools/fold_final.sh [<git rev the session ran on>]
set -e
cd "$(dirname "$0")/.."
S=gpurun_out/final; D=profiles/r06/final; REV=${1:-$(git rev-parse --short HEAD)}
# (gpurun_out/ is merged into by every session and never cleaned: passes of earlier rounds would be averaged in)
find $S -name "r0[0-9]_*" ! -name "r06_*" -delete
mkdir -p $D/other_workloads
for f in bench_default.json bench_driver_style.json pytest_gpu.txt trace_blockout.json phase_blockout.json phase_general.json \
         phase_abc_fine.json phase_cube.json scaling_points.jsonl; do cp $S/$f $D/$f; done
for f in bench_under_rocprof.json kernel_stats.csv kernel_trace_timed_region.json fetch_summary.json write_summary.json \
         sq_summary.json sq2_summary.json; do cp $S/prof_blockout/$f $D/$f; done
for wl in general abc_fine; do
  for f in fetch write sq sq2; do cp $S/prof_$wl/${f}_summary.json $D/other_workloads/${wl}_${f}_summary.json; done
  cp $S/prof_$wl/kernel_stats.csv $D/other_workloads/kernel_stats_$wl.csv
  cp $S/prof_$wl/kernel_trace_timed_region.json $D/other_workloads/kernel_trace_timed_region_$wl.json
done
cp $S/kernel_stats_*.csv $D/other_workloads/
python tools/collect_pmc.py $S/prof_blockout blockout 16384 $REV > /dev/null
python tools/collect_pmc.py $S/prof_general general 8192 $REV > /dev/null
python tools/collect_pmc.py $S/prof_abc_fine abc_fine 8192 $REV > /dev/null
python - <<'PY'
import json
d = json.load(open("profiles/pmc_hbm.json"))
print(d["kernel_source_sha"], d["git_rev"])
for k, v in d["workloads"].items():
    print(k, v["bins"], round(v["hbm_bytes_per_bin_step"]), "B per bin-step")
PY
