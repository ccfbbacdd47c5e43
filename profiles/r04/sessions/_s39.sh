#!/bin/bash
# Tooling: round-4 session 39: trace kernel with 12 288 B of LDS per wave (48 border points in LDS, round-packing scratch inside the frames): 13 waves per CU
O=gpurun_out/r04_s39; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_boundary.py tests/test_gpu_features.py -m gpu -q -x > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.txt; tail -3 $O/pytest_gpu.txt | cut -c1-200
bash tools/gpu_kernel_stats.sh r04_s39 blockout general cube 2>&1 | grep "irbpp_trace" | cut -c1-110
timeout 300 python tools/ab_matrix.py --repeat 2 blockout:4096:1:0 general:4096:1:0 blockout_k10:1024:1:0 > $O/ab_matrix.jsonl 2> $O/ab_matrix.err; cat $O/ab_matrix.jsonl | cut -c1-150
