#!/bin/bash
# round-6 session 6: CHAIN builds (one kernel per observation at launches of up to 2048 bins) -- the whole GPU suite (every small-N test now takes
# that form by default), then A/B against the split pipeline (tuning 2097152 = IRBPP_TUNE_NO_CHAIN) at the sharded configs' per-GPU sizes
O=gpurun_out/r06_s6; mkdir -p $O
timeout 2000 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.txt 2>&1; echo "rc=$?" >> $O/pytest_gpu.txt; tail -6 $O/pytest_gpu.txt
timeout 900 python tools/ab_matrix.py --min-seconds 0.4 blockout_k10:1024:1:0 blockout_k10:1024:1:2097152 blockout:1024:1:0 blockout:1024:1:2097152 blockout:2048:1:0 blockout:2048:1:2097152 \
   blockout_k10:2048:1:0 blockout_k10:2048:1:2097152 blockout_k10:512:1:0 blockout_k10:512:1:2097152 cube:1024:1:0 cube:1024:1:2097152 general:1024:1:0 general:1024:1:2097152 \
   blockout_r8:1024:1:0 blockout_r8:1024:1:2097152 blockout:4096:1:1048576 blockout:4096:1:0 2>/dev/null | tee $O/ab.jsonl | python -c "
import sys, json
for l in sys.stdin:
    j = json.loads(l); print(j['spec'], j['Msteps_per_s'])"
