#!/bin/bash
# Tooling: round-4 session 4: borders approximated per bin in the emit kernel (no polygon kernel), ticket-dealt generic tasks
O=gpurun_out/r04_s4; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -q > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.txt; tail -12 $O/pytest_gpu.txt | cut -c1-200
bash tools/gpu_kernel_stats.sh r04_s4 blockout general 2>&1 | grep irbpp | cut -c1-110
timeout 400 python tools/ab_matrix.py --repeat 2 blockout:4096:1:0 blockout:8192:1:0 blockout_k10:1024:1:0 blockout_k10:4096:1:0 general:4096:1:0 abc_fine:2048:1:0 cube:4096:1:0 blockout:1024:1:0 \
   > $O/ab_matrix.jsonl 2> $O/ab_matrix.err; cat $O/ab_matrix.jsonl | cut -c1-150
timeout 120 python tools/trace_profile.py --workload blockout > $O/trace_blockout.json 2>/dev/null; tail -1 $O/trace_blockout.json | cut -c1-700
