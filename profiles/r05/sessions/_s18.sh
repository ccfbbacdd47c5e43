#!/bin/bash
# round-5 session 18: instruction diet of the transition kernel: np_floor_divide_int without the sign's second remainder where no lane of
# the wave is negative; block lists of all rotations through LDS (one broadcast ds_read_b128 per entry instead of three v_readlane)
O=gpurun_out/r05_s18; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.txt; tail -4 $O/pytest_gpu.txt
timeout 400 python tools/ab_matrix.py --repeat 2 --min-seconds 0.4 blockout:8192:1:0 blockout:8192:2:0 blockout:4096:2:0 cube:8192:2:0 general:8192:2:0 general:4096:2:0 abc_fine:2048:2:0 blockout_r8:8192:2:0 blockout_k10:1024:1:0 2>/dev/null | tee $O/ab.jsonl | cut -c1-150
timeout 200 bash tools/gpu_kernel_stats.sh r05_s18 blockout 2>&1 | tail -7
