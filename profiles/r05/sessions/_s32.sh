#!/bin/bash
# round-5 session 32: polygon rounds: a kept point's rank in its border's polygon from bit counts filed per position (v_mbcnt + two LDS reads)
# instead of 64-bit mask arithmetic per ballot word; a point's distance from its slice's start carried from level to level
O=gpurun_out/r05_s32; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -x -k "golden or contour or vertex or polygon or trace or parity or many_bins" > $O/pytest_sel.txt 2>&1; tail -3 $O/pytest_sel.txt
timeout 400 python tools/ab_matrix.py --repeat 2 --min-seconds 0.4 blockout:8192:2:0 blockout:8192:1:0 blockout:4096:2:0 general:4096:2:0 cube:4096:2:0 2>/dev/null | tee $O/ab.jsonl | cut -c1-150
bash tools/gpu_kernel_stats.sh r05_s32 blockout 2>&1 | tail -7
