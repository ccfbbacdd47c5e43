#!/bin/bash
# round-5 session 6: split (4096) vs fused (2048) apply on the specialised builds over sizes: where does the split start to pay?
O=gpurun_out/r05_s6; mkdir -p $O
timeout 600 python -m pytest tests -m gpu -q -k "split_apply or k10 or hierarchical" > $O/pytest_sel.txt 2>&1; tail -3 $O/pytest_sel.txt
timeout 900 python tools/ab_matrix.py --repeat 1 --min-seconds 0.3 \
  blockout:2048:1:2048 blockout:2048:1:4096 blockout:4096:1:2048 blockout:4096:1:4096 blockout:6144:1:2048 blockout:6144:1:4096 blockout:8192:1:2048 blockout:8192:1:4096 blockout:16384:1:2048 blockout:16384:1:4096 \
  cube:4096:1:2048 cube:4096:1:4096 cube:8192:1:2048 cube:8192:1:4096 general:4096:1:2048 general:4096:1:4096 general:8192:1:2048 general:8192:1:4096 \
  abc_fine:2048:1:2048 abc_fine:2048:1:4096 abc_fine:4096:1:2048 abc_fine:4096:1:4096 blockout_k10:1024:1:2048 blockout_k10:1024:1:4096 blockout_k10:8192:1:2048 blockout_k10:8192:1:4096 \
  blockout_r8:4096:1:2048 blockout_r8:4096:1:4096 blockout_r8:8192:1:2048 blockout_r8:8192:1:4096 2>/dev/null | tee $O/ab.jsonl | cut -c1-140
