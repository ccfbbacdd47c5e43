#!/bin/bash
# Tooling: round-4 session 36: bitonic sort of the > S selection: the workgroup meets only around the rounds whose stride crosses 128-element blocks
O=gpurun_out/r04_s36; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -q > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.txt; tail -4 $O/pytest_gpu.txt | cut -c1-200
bash tools/gpu_kernel_stats.sh r04_s36 general 2>&1 | grep "irbpp_emit" | cut -c1-110
timeout 300 python tools/ab_matrix.py --repeat 2 general:4096:1:0 abc_fine:2048:1:0 blockout:4096:1:0 > $O/ab_matrix.jsonl 2> $O/ab_matrix.err; cat $O/ab_matrix.jsonl | cut -c1-150
IRBPP_LIBRARY= timeout 10 true
