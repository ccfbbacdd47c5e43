#!/bin/bash
# round-5 session 16: single-GPU points of the strong-scaling projection (cfg 4: 8192 k=10 bins / n GPUs; cfg 5: 16384 fine bins / n GPUs),
# each at the number of groups groups_for recommends; kernel durations of the BlockOut step at 4096 bins
O=gpurun_out/r05_s16; mkdir -p $O
timeout 600 python tools/ab_matrix.py --repeat 1 --min-seconds 0.4 blockout_k10:8192:2:0 blockout_k10:4096:2:0 blockout_k10:2048:1:0 blockout_k10:2048:2:0 blockout_k10:1024:1:0 \
  abc_fine:16384:2:0 abc_fine:8192:2:0 abc_fine:4096:2:0 abc_fine:2048:2:0 blockout:8192:2:0 blockout:4096:2:0 blockout:2048:2:0 blockout:1024:1:0 2>/dev/null | tee $O/ab.jsonl | cut -c1-150
R=$PWD; cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/kt4096
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt4096 -o x -- python $R/bench.py --bins 4096 --groups 1 --no-cpu-baseline --no-extra --steps 100 --warmup 10 --min-seconds 0 > /dev/null 2>&1
f=$(find /tmp/kt4096 -name "*kernel_stats.csv" | head -1); cp "$f" $R/$O/kernel_stats_blockout_4096.csv; head -7 $R/$O/kernel_stats_blockout_4096.csv | cut -c1-120
