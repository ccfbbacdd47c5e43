#!/bin/bash
# round-5 session 29: generic path with 512-thread workgroups (eight waves on one bin's tile: the second pass of irbpp_kernels.hip) for the
# 64 x 64 heightmap, against the 256-thread build (tuning 131072); forced onto the 32 x 32 free-form data too (tuning 65536)
O=gpurun_out/r05_s29; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -x -k "specialised or fine or abc_fine or generic or full_size_properties or many_bins" > $O/pytest_sel.txt 2>&1; tail -3 $O/pytest_sel.txt
timeout 500 python tools/ab_matrix.py --repeat 2 --min-seconds 0.4 abc_fine:2048:2:0 abc_fine:2048:2:131072 abc_fine:2048:1:0 abc_fine:2048:1:131072 abc_fine:8192:2:0 abc_fine:8192:2:131072 \
  general:4096:2:0 general:4096:2:65536 general:8192:1:0 general:8192:1:65536 2>/dev/null | tee $O/ab.jsonl | cut -c1-150
