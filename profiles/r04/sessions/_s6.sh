#!/bin/bash
O=gpurun_out/r04_s6; mkdir -p $O
bash tools/gpu_kernel_stats.sh r04_s6 blockout 2>&1 | grep irbpp | cut -c1-110
timeout 300 python tools/ab_matrix.py --repeat 2 blockout:4096:1:0 blockout_k10:1024:1:0 general:4096:1:0 > $O/ab_matrix.jsonl 2> $O/ab_matrix.err; cat $O/ab_matrix.jsonl | cut -c1-150
