#!/bin/bash
# Tooling: round-4 session 7: polygon-kernel pipeline again + ticket-dealt generic tasks + speckled bins first in the emit kernel
O=gpurun_out/r04_s7; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -q > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.txt; tail -6 $O/pytest_gpu.txt | cut -c1-200
bash tools/gpu_kernel_stats.sh r04_s7 blockout general abc_fine 2>&1 | grep irbpp | cut -c1-110
timeout 400 python tools/ab_matrix.py --repeat 2 blockout:4096:1:0 general:4096:1:0 general:4096:1:512 abc_fine:2048:1:0 abc_fine:2048:1:512 blockout_k10:1024:1:0 blockout:8192:1:0 cube:4096:1:0 \
   > $O/ab_matrix.jsonl 2> $O/ab_matrix.err; cat $O/ab_matrix.jsonl | cut -c1-150
