#!/bin/bash
# round-5 session 11: two groups on streams of different priority: repeatable within a process?
O=gpurun_out/r05_s11; mkdir -p $O
python -c "import torch; print(torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream,'priority_range') else 'n/a')"
for i in 1 2; do
timeout 300 python tools/ab_matrix.py --repeat 4 --min-seconds 0.25 blockout:2048:2:0 blockout_k10:8192:2:0 blockout:4096:2:0 blockout:8192:2:0 general:4096:2:0 abc_fine:2048:2:0 blockout:8192:4:0 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    j = json.loads(l); print('proc $i', j['spec'], j['Msteps_per_s'])" | tee -a $O/groups.txt
done
