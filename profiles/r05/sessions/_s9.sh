#!/bin/bash
# round-5 session 9: the same bins as two / four groups on their own streams (default hardware queues): how much, how repeatable?
O=gpurun_out/r05_s9; mkdir -p $O
for i in 1 2 3; do
timeout 300 python tools/ab_matrix.py --repeat 1 --min-seconds 0.3 blockout:8192:1:0 blockout:8192:2:0 blockout:8192:4:0 blockout:4096:2:0 cube:8192:2:0 general:4096:2:0 \
  blockout_k10:1024:2:0 abc_fine:2048:2:0 abc_fine:2048:4:0 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    j = json.loads(l); print('proc $i', j['spec'], j['Msteps_per_s'])" | tee -a $O/groups.txt
done
