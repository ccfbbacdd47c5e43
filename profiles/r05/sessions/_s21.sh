#!/bin/bash
# round-5 session 21: trace kernel with 32 candidates per wave (half the LDS per wave) when two groups overlap -- does the smaller footprint
# buy more overlap with the other group's transition kernel?
O=gpurun_out/r05_s21; mkdir -p $O
timeout 400 python tools/ab_matrix.py --repeat 2 --min-seconds 0.4 blockout:8192:2:0 blockout:8192:2:64 blockout:8192:1:64 blockout:4096:2:64 blockout:4096:2:0 general:4096:2:64 general:4096:2:0 cube:8192:2:64 2>/dev/null | tee $O/ab.jsonl | cut -c1-150
