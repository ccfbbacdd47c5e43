#!/bin/bash
# round-6 session 2: the lane-refill trace kernel -- parity (every launch shape against the default, large forms, RCCL tests again), then A/B against
# the one-candidate-per-lane form (tuning 32 = IRBPP_TUNE_TRACE_CPW64) and per-kernel durations
O=gpurun_out/r06_s2; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_features.py -q -x -k "trace_launch or specialised" > $O/pytest_trace.txt 2>&1; echo "rc=$?" >> $O/pytest_trace.txt; tail -4 $O/pytest_trace.txt
timeout 1200 python -m pytest tests/test_gpu_large_forms.py tests/test_gpu_rccl.py tests/test_gpu_boundary.py -q -x > $O/pytest_large.txt 2>&1; echo "rc=$?" >> $O/pytest_large.txt; tail -4 $O/pytest_large.txt
timeout 600 python tools/ab_matrix.py --min-seconds 0.4 blockout:8192:1:0 blockout:8192:1:32 blockout:8192:2:0 blockout:8192:2:32 blockout:4096:1:0 blockout:4096:1:32 \
   general:4096:2:0 general:4096:2:32 cube:8192:2:0 cube:8192:2:32 abc_fine:2048:2:0 abc_fine:2048:2:32 blockout_k10:8192:2:0 blockout_k10:8192:2:32 2>/dev/null | tee $O/ab.jsonl | python -c "
import sys, json
for l in sys.stdin:
    j = json.loads(l); print(j['spec'], j['Msteps_per_s'])"
bash tools/gpu_kernel_stats.sh r06_s2 blockout 2>&1 | tail -7
