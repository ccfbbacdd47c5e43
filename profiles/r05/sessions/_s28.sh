#!/bin/bash
# round-5 session 28: launch-order kernel (fine-heightmap data): the dependent reads of eight bins per thread in flight together
O=gpurun_out/r05_s28; mkdir -p $O
timeout 600 python -m pytest tests -m gpu -q -x -k "item_grouped or abc_fine or fine or full_size_properties" > $O/pytest_sel.txt 2>&1; tail -3 $O/pytest_sel.txt
timeout 400 python tools/ab_matrix.py --repeat 2 --min-seconds 0.4 abc_fine:2048:2:0 abc_fine:2048:1:0 abc_fine:8192:2:0 abc_fine:16384:2:0 2>/dev/null | tee $O/ab.jsonl | cut -c1-150
timeout 200 bash tools/gpu_kernel_stats.sh r05_s28 abc_fine 2>&1 | grep -E "order|env"
