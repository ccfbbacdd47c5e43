#!/bin/bash
# round-5 session 7: emit kernel with a wave per bin (lattice / box data) against the workgroup-per-bin one (tuning 8192)
O=gpurun_out/r05_s7; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.txt; tail -6 $O/pytest_gpu.txt
timeout 600 python tools/ab_matrix.py --repeat 2 blockout:8192:1:0 blockout:8192:1:8192 blockout:4096:1:0 blockout:4096:1:8192 blockout:1024:1:0 blockout:1024:1:8192 \
  cube:4096:1:0 cube:4096:1:8192 blockout_k10:1024:1:0 blockout_k10:1024:1:8192 2>/dev/null | tee $O/ab.jsonl | cut -c1-160
timeout 300 bash tools/gpu_kernel_stats.sh r05_s7 blockout 2>&1 | tail -8
