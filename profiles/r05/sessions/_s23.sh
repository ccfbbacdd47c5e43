#!/bin/bash
# round-5 session 23: the default bench line five times (five processes): how repeatable is `value`?
O=gpurun_out/r05_s23; mkdir -p $O
for i in 1 2 3 4 5; do
timeout 200 python bench.py --no-extra --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('run $i', round(d['value']/1e6, 2), 'M', d['config']['groups_per_gpu'], 'groups', round(d['ms_per_step'], 4), 'ms', d['steps'], 'steps')" | tee -a $O/repeat.txt
done
