#!/bin/bash
# round-5 session 27: a buffered step with a WAVE per bin (tuning 4096) now that its fences are workgroup scope, against the workgroup per bin
O=gpurun_out/r05_s27; mkdir -p $O
timeout 500 python tools/ab_matrix.py --repeat 2 --min-seconds 0.4 blockout_k10:1024:1:0 blockout_k10:1024:1:4096 blockout_k10:2048:2:0 blockout_k10:2048:2:4096 blockout_k10:4096:2:0 blockout_k10:4096:2:4096 \
  blockout_k10:8192:2:0 blockout_k10:8192:2:4096 blockout_k10:8192:1:0 blockout_k10:8192:1:4096 2>/dev/null | tee $O/ab.jsonl | cut -c1-150
