#!/bin/bash
# round-5 session 1: GPU suite on the new goldens + specialised builds (SPEC_KEYS) against the run-time builds (tuning 1024)
O=gpurun_out/r05_s1; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.txt; tail -4 $O/pytest_gpu.txt
timeout 600 python tools/ab_matrix.py --repeat 2 blockout:8192:1:0 blockout:8192:1:1024 blockout:4096:1:0 blockout:4096:1:1024 \
  cube:4096:1:0 cube:4096:1:1024 general:4096:1:0 general:4096:1:1024 abc_fine:2048:1:0 abc_fine:2048:1:1024 \
  blockout_k10:1024:1:0 blockout_k10:1024:1:1024 blockout_r8:4096:1:0 blockout_r8:4096:1:1024 2>/dev/null | tee $O/ab.jsonl | cut -c1-200
timeout 300 bash tools/gpu_kernel_stats.sh r05_s1 blockout general 2>&1 | tail -16
