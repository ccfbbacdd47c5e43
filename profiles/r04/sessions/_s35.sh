#!/bin/bash
# Tooling: round-4 session 35: emit kernel lists its candidate rows with one wave (a lane per rotation and column) instead of one workgroup scan per rotation
O=gpurun_out/r04_s35; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -q > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.txt; tail -4 $O/pytest_gpu.txt | cut -c1-200
for spec in blockout:4096 general:4096; do
  wl=${spec%%:*}; bins=${spec##*:}
  IRBPP_LIBRARY=$PWD/irbpp_amd/libirbpp_var_emitacct.so timeout 300 python tools/emit_profile.py --workload $wl --bins $bins > $O/emit_$wl$bins.json 2> $O/err_$wl$bins.txt; tail -1 $O/emit_$wl$bins.json | cut -c1-800
done
bash tools/gpu_kernel_stats.sh r04_s35 blockout general cube 2>&1 | grep "irbpp_emit" | cut -c1-110
timeout 300 python tools/ab_matrix.py --repeat 2 blockout:4096:1:0 general:4096:1:0 cube:4096:1:0 blockout_k10:1024:1:0 abc_fine:2048:1:0 blockout:1024:1:0 > $O/ab_matrix.jsonl 2> $O/ab_matrix.err; cat $O/ab_matrix.jsonl | cut -c1-150
