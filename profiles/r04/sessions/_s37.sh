#!/bin/bash
# Tooling: round-4 session 37: emit kernel stores the candidate block a thread per row instead of a thread per float
O=gpurun_out/r04_s37; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_boundary.py tests/test_gpu_features.py -m gpu -q -x > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.txt; tail -3 $O/pytest_gpu.txt | cut -c1-200
IRBPP_LIBRARY=$PWD/irbpp_amd/libirbpp_var_emitacct.so timeout 300 python tools/emit_profile.py --workload blockout --bins 4096 > $O/emit_blockout4096.json 2> $O/err.txt; tail -1 $O/emit_blockout4096.json | cut -c1-800
bash tools/gpu_kernel_stats.sh r04_s37 blockout general cube 2>&1 | grep "irbpp_emit" | cut -c1-110
timeout 300 python tools/ab_matrix.py --repeat 2 blockout:4096:1:0 general:4096:1:0 cube:4096:1:0 blockout_k10:1024:1:0 > $O/ab_matrix.jsonl 2> $O/ab_matrix.err; cat $O/ab_matrix.jsonl | cut -c1-150
