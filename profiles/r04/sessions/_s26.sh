#!/bin/bash
# Tooling: round-4 session 26: per-kernel durations at 1024 bins (BlockOut, buffered BlockOut k = 10) and 2048 bins (abc_fine)
O=$PWD/gpurun_out/r04_s26; mkdir -p $O
R=$PWD
cd /tmp && export TMPDIR=/tmp
for spec in blockout:1024 blockout_k10:1024 abc_fine:2048 blockout:512; do
  wl=${spec%%:*}; bins=${spec##*:}
  rm -rf /tmp/kt_$wl$bins
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_$wl$bins -o x -- python $R/bench.py --workload $wl --bins $bins --no-cpu-baseline --no-extra --steps 100 --warmup 10 --min-seconds 0 > $O/bench_$wl$bins.json 2>/dev/null
  f=$(find /tmp/kt_$wl$bins -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && head -8 "$f" > $O/kernel_stats_$wl$bins.csv && echo $spec && grep irbpp $O/kernel_stats_$wl$bins.csv | cut -d, -f1,2,4,6,7
done
