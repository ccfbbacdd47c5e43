#!/bin/bash
# Tooling: round-4 session 1 on the GPU box: parity suite, small-N A/B matrix (groups x candidates per trace wave), k10 @1024 kernel stats
O=gpurun_out/r04_s1; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.txt; tail -5 $O/pytest_gpu.txt
timeout 600 python tools/ab_matrix.py \
  blockout_k10:1024:1:0 blockout_k10:1024:1:64 blockout_k10:1024:1:128 \
  blockout_k10:1024:2:0 blockout_k10:1024:2:64 blockout_k10:1024:2:128 \
  blockout_k10:1024:4:0 blockout_k10:1024:4:128 \
  abc_fine:2048:1:0 abc_fine:2048:2:0 abc_fine:2048:4:0 abc_fine:2048:2:64 \
  blockout:4096:1:0 blockout:4096:1:64 blockout:2048:1:0 blockout:2048:1:64 blockout:2048:1:128 blockout:1024:1:0 blockout:1024:1:128 \
  general:4096:1:0 general:4096:2:0 blockout:4096:2:0 blockout:4096:4:0 \
  > $O/ab_matrix.jsonl 2> $O/ab_matrix.err; cat $O/ab_matrix.jsonl | cut -c1-170
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_k10 -o x -- python $R/bench.py --workload blockout_k10 --bins 1024 --no-cpu-baseline --no-extra --steps 100 --warmup 10 --min-seconds 0 > $R/$O/bench_k10_1024_rocprof.json 2>/dev/null
f=$(find /tmp/kt_k10 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $R/$O/kernel_stats_k10_1024.csv && head -8 $R/$O/kernel_stats_k10_1024.csv | cut -c1-120
