#!/bin/bash
# round-5 session 5: step() = irbpp_apply_kernel (a wave per bin) + transition kernel in MODE_OBSERVE, against the fused form (tuning 2048)
O=gpurun_out/r05_s5; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -q > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.txt; tail -6 $O/pytest_gpu.txt
timeout 600 python tools/ab_matrix.py --repeat 2 blockout:8192:1:0 blockout:8192:1:2048 blockout:4096:1:0 cube:4096:1:0 general:4096:1:0 general:4096:1:2048 abc_fine:2048:1:0 \
  blockout_k10:1024:1:0 blockout_k10:1024:1:2048 blockout_r8:4096:1:0 blockout:1024:1:0 blockout:2048:1:0 2>/dev/null | tee $O/ab.jsonl | cut -c1-200
timeout 300 bash tools/gpu_kernel_stats.sh r05_s5 blockout general 2>&1 | tail -18
