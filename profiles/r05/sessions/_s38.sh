#!/bin/bash
# round-5 session 38: HIP path against the C oracle at the launch sizes that select the large-launch forms by themselves
O=gpurun_out/r05_s38; mkdir -p $O
timeout 400 python tools/soak_parity.py blockout:4096:1:110 blockout:4096:2:60 cube:4096:1:50 blockout_k10:2048:1:60 general:2048:2:30 2>$O/err.txt | tee $O/soak.jsonl | cut -c1-330
tail -3 $O/err.txt
