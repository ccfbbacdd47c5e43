#!/bin/bash
# round-5 session 17: MODE_OBSERVE keeps the item's ShapeRot dwords in a register across tile staging and the block-max grid
O=gpurun_out/r05_s17; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -x -k "golden or specialised or oracle or full_size" > $O/pytest_sel.txt 2>&1; tail -3 $O/pytest_sel.txt
timeout 400 python tools/ab_matrix.py --repeat 2 --min-seconds 0.4 blockout:8192:1:0 blockout:8192:2:0 blockout:16384:2:0 cube:8192:2:0 general:8192:2:0 blockout_r8:8192:2:0 2>/dev/null | tee $O/ab.jsonl | cut -c1-150
