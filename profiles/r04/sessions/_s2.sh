#!/bin/bash
# Tooling: round-4 session 2: parity suite on the short-iteration border walk, A/B against the round-3 library
O=gpurun_out/r04_s2; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.txt; tail -5 $O/pytest_gpu.txt
timeout 600 python tools/ab_matrix.py --repeat 2 blockout:4096:1:0 blockout:8192:1:0 blockout_k10:1024:1:0 blockout_k10:1024:1:256 blockout_k10:1024:1:384 \
  blockout:1024:1:0 blockout:1024:1:256 blockout:2048:1:256 blockout:4096:1:256 general:4096:1:0 abc_fine:2048:1:0 cube:4096:1:0 blockout_k10:4096:1:0 \
  > $O/ab_matrix.jsonl 2> $O/ab_matrix.err; cat $O/ab_matrix.jsonl | cut -c1-150
bash tools/gpu_kernel_stats.sh r04_s2 blockout general 2>&1 | grep irbpp | cut -c1-110
