#!/bin/bash
# Tooling: round-4 session 28: clean-up pass of approxPolyDP without its loop (cleanup_convex_parallel): parity, wave account of the polygon kernel, kernel durations
O=gpurun_out/r04_s28; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -q > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.txt; tail -4 $O/pytest_gpu.txt | cut -c1-200
for spec in blockout:512 blockout:4096 general:4096; do
  wl=${spec%%:*}; bins=${spec##*:}
  IRBPP_LIBRARY=$PWD/irbpp_amd/libirbpp_var_polyacct.so timeout 300 python tools/polygon_profile.py --workload $wl --bins $bins > $O/polygon_$wl$bins.json 2> $O/err_$wl$bins.txt; tail -1 $O/polygon_$wl$bins.json | cut -c1-900
done
bash tools/gpu_kernel_stats.sh r04_s28 blockout general cube 2>&1 | grep "irbpp_trace\|irbpp_poly" | cut -c1-110
bash tools/_s26.sh 2>&1 | grep "irbpp_poly\|^[a-z_0-9]*:[0-9]*$"
timeout 300 python tools/ab_matrix.py --repeat 2 blockout:4096:1:0 general:4096:1:0 blockout_k10:1024:1:0 abc_fine:2048:1:0 blockout:1024:1:0 > $O/ab_matrix.jsonl 2> $O/ab_matrix.err; cat $O/ab_matrix.jsonl | cut -c1-150
