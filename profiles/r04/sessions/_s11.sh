#!/bin/bash
# Tooling: round-4 session 11: one-pass candidate enumeration in the emit kernel; placed item's ShapeRot through LDS vs per thread
O=gpurun_out/r04_s11; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -q > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.txt; tail -4 $O/pytest_gpu.txt | cut -c1-200
bash tools/gpu_kernel_stats.sh r04_s11 blockout general cube 2>&1 | grep irbpp | cut -c1-110
IRBPP_LIBRARY=$PWD/irbpp_amd/libirbpp_var_srthread.so bash tools/gpu_kernel_stats.sh r04_s11/srthread blockout general cube 2>&1 | grep irbpp_env | cut -c1-110
timeout 300 python tools/ab_matrix.py --repeat 2 blockout:4096:1:0 general:4096:1:0 cube:4096:1:0 blockout_k10:1024:1:0 abc_fine:2048:1:0 > $O/ab_matrix.jsonl 2> $O/ab_matrix.err; cat $O/ab_matrix.jsonl | cut -c1-150
