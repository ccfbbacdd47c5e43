#!/bin/bash
# round-6 session 9: two waves per bin (irbpp_env_kernel_s1_w128, tuning 2097152) against four -- parity, A/B, kernel durations
O=gpurun_out/r06_s9; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_features.py -q -x -k "specialised" > $O/pytest.txt 2>&1; echo "rc=$?" >> $O/pytest.txt; tail -4 $O/pytest.txt
timeout 600 python tools/ab_matrix.py --min-seconds 0.4 blockout:8192:2:0 blockout:8192:2:2097152 blockout:8192:1:0 blockout:8192:1:2097152 blockout:4096:2:0 blockout:4096:2:2097152 \
   blockout:16384:2:0 blockout:16384:2:2097152 blockout:1024:1:0 blockout:1024:1:2097152 blockout_k10:8192:2:0 blockout_k10:8192:2:2097152 blockout_k10:1024:1:0 blockout_k10:1024:1:2097152 2>/dev/null | tee $O/ab.jsonl | python -c "
import sys, json
for l in sys.stdin:
    j = json.loads(l); print(j['spec'], j['Msteps_per_s'], j['kernel_ms'])"
