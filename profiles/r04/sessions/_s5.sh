#!/bin/bash
# Tooling: round-4 session 5: approximating wave rotates with the bin; emit kernel at 8 / 7 / 6 waves per SIMD; heavy bins first
O=gpurun_out/r04_s5; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -x -k "many_bins or launch_shapes or adversarial or full_size_properties or grouped_stepping" > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.txt; tail -4 $O/pytest_gpu.txt | cut -c1-200
bash tools/gpu_kernel_stats.sh r04_s5 blockout general 2>&1 | grep irbpp | cut -c1-110
for v in emit7 emit6; do IRBPP_LIBRARY=$PWD/irbpp_amd/libirbpp_var_$v.so bash tools/gpu_kernel_stats.sh r04_s5/$v blockout 2>&1 | grep irbpp_emit | cut -c1-110; done
timeout 400 python tools/ab_matrix.py --repeat 2 blockout:4096:1:0 blockout_k10:1024:1:0 general:4096:1:0 general:4096:1:512 abc_fine:2048:1:0 blockout:8192:1:0 \
   > $O/ab_matrix.jsonl 2> $O/ab_matrix.err; cat $O/ab_matrix.jsonl | cut -c1-150
timeout 120 python tools/trace_profile.py --workload blockout > $O/trace_blockout.json 2>/dev/null; tail -1 $O/trace_blockout.json | cut -c1-700
