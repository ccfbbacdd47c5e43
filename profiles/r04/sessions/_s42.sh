#!/bin/bash
# Tooling: round-4 session 42: is the spread of four-group stepping (stream -> hardware queue mapping) a matter of GPU_MAX_HW_QUEUES?  Three processes each way.
O=gpurun_out/r04_s42; mkdir -p $O
for q in default 8 16; do
  for i in 1 2 3; do
    if [ $q = default ]; then unset GPU_MAX_HW_QUEUES; else export GPU_MAX_HW_QUEUES=$q; fi
    timeout 120 python tools/ab_matrix.py --repeat 1 blockout:4096:4:0 general:4096:2:0 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    j=json.loads(l); print('$q', $i, j['spec'], j['Msteps_per_s'])" | tee -a $O/queues.txt
  done
done
