#!/bin/bash
# round-5 session 37: polygon hop rounds as straight-line code (empty positions bid 0 at their own word) against the branchy hops of
# the 6th final pass (variant build of commit e83bed1), same box
O=gpurun_out/r05_s37; mkdir -p $O
for lib in default oldhops; do
  if [ $lib = oldhops ]; then export IRBPP_LIBRARY=$PWD/irbpp_amd/libirbpp_var_oldhops.so; else unset IRBPP_LIBRARY; fi
  timeout 300 python tools/ab_matrix.py --repeat 2 --min-seconds 0.4 blockout:8192:2:0 blockout:8192:1:0 general:4096:2:0 2>/dev/null | tee -a $O/ab_$lib.jsonl | cut -c1-140
  bash tools/gpu_kernel_stats.sh r05_s37/$lib blockout 2>&1 | grep "polygon\|trace_kernel" | cut -d, -f1,4
done
unset IRBPP_LIBRARY
timeout 600 python -m pytest tests -m gpu -q -x -k "golden or contour or vertex or polygon or trace or heuristic" 2>&1 | tail -2 | tee $O/pytest_sel.txt
