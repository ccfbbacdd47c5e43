#!/bin/bash
# round-5 session 25: a buffered environment's step as irbpp_apply_wg_kernel (a workgroup per bin: wave 0 applies the action on the heightmap
# in HBM, all four waves write the order observation) instead of the transition kernel's fused apply (tuning 2048)
O=gpurun_out/r05_s25; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -x -k "k10 or hierarchical or buffered or specialised or item_streams or kat or grouped or boundary" > $O/pytest_sel.txt 2>&1; tail -3 $O/pytest_sel.txt
timeout 500 python tools/ab_matrix.py --repeat 2 --min-seconds 0.4 blockout_k10:1024:1:0 blockout_k10:1024:1:2048 blockout_k10:2048:2:0 blockout_k10:2048:2:2048 blockout_k10:4096:2:0 blockout_k10:4096:2:2048 \
  blockout_k10:8192:2:0 blockout_k10:8192:2:2048 blockout_k10:8192:1:0 blockout_k10:8192:1:2048 2>/dev/null | tee $O/ab.jsonl | cut -c1-150
