#!/bin/bash
# Tooling: round-4 session 32: trace kernel's round records in one pass (a list slot reserved per round instead of all at once after a counting pass)
O=gpurun_out/r04_s32; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -q > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.txt; tail -4 $O/pytest_gpu.txt | cut -c1-200
bash tools/gpu_kernel_stats.sh r04_s32 blockout general 2>&1 | grep "irbpp_" | cut -c1-110
timeout 300 python tools/ab_matrix.py --repeat 2 blockout:4096:1:0 general:4096:1:0 blockout_k10:1024:1:0 > $O/ab_matrix.jsonl 2> $O/ab_matrix.err; cat $O/ab_matrix.jsonl | cut -c1-150
timeout 200 python tools/trace_profile.py --workload blockout --bins 4096 > $O/trace_blockout.json 2>/dev/null; tail -1 $O/trace_blockout.json | cut -c1-700
