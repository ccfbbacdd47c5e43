#!/bin/bash
# round-5 session 33: polygon rounds: the Douglas-Peucker level as straight-line code (state = seg / axy / bxy / tk per position,
# inactive positions bid 0 at their own word, selects instead of three `if (active)` regions and two flag registers)
O=gpurun_out/r05_s33; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -x -k "golden or contour or vertex or polygon or trace or parity or many_bins" > $O/pytest_sel.txt 2>&1; tail -3 $O/pytest_sel.txt
timeout 400 python tools/ab_matrix.py --repeat 2 --min-seconds 0.4 blockout:8192:2:0 blockout:8192:1:0 blockout:4096:2:0 general:4096:2:0 cube:4096:2:0 abc_fine:2048:2:0 2>/dev/null | tee $O/ab.jsonl | cut -c1-150
bash tools/gpu_kernel_stats.sh r05_s33 blockout 2>&1 | tail -7
