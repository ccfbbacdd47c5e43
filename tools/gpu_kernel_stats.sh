#!/bin/bash
# Tooling: rocprofv3 --kernel-trace --stats of a short bench run per workload; keeps only the per-kernel statistics.
#   tools/gpu_kernel_stats.sh <out dir under gpurun_out/> <workload> [more workloads]   (IRBPP_LIBRARY selects a variant build)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/$1; shift; mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
for wl in "$@"; do
  rm -rf /tmp/kt_$wl
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_$wl -o x -- python $R/bench.py --workload $wl --no-cpu-baseline --no-extra --groups 1 --steps 100 --warmup 10 --min-seconds 0 > "$O/bench_under_rocprof_$wl.json" 2> /dev/null
  f=$(find /tmp/kt_$wl -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp "$f" "$O/kernel_stats_$wl.csv" && head -7 "$O/kernel_stats_$wl.csv" | cut -c1-150
done
