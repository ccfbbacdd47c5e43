#!/bin/bash
# round-5 session 31: generic path: the presence bits of a task's row groups reduced over the wave once per task instead of once per row group
O=gpurun_out/r05_s31; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -x -k "general or generic or fine or r8 or specialised or many_bins" > $O/pytest_sel.txt 2>&1; tail -3 $O/pytest_sel.txt
timeout 400 python tools/ab_matrix.py --repeat 2 --min-seconds 0.4 general:4096:2:0 general:4096:1:0 general:8192:2:0 abc_fine:2048:2:0 abc_fine:8192:2:0 blockout_r8:8192:2:0 2>/dev/null | tee $O/ab.jsonl | cut -c1-150
