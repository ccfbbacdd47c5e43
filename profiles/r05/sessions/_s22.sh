#!/bin/bash
# round-5 session 22: the two groups a fraction of a step out of phase (a one-off spin kernel on the second group's stream after reset):
# does one group's transition kernel then run beside the other's trace / polygon kernels, and does the shift last?
O=gpurun_out/r05_s22; mkdir -p $O
for cyc in 0 60000 120000 200000; do
IRBPP_EXPERIMENT_STAGGER_CYCLES=$cyc timeout 300 python tools/ab_matrix.py --repeat 2 --min-seconds 0.5 blockout:8192:2:0 general:4096:2:0 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    j = json.loads(l); print('stagger $cyc', j['spec'], j['Msteps_per_s'])" | tee -a $O/stagger.txt
done
