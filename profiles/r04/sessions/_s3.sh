#!/bin/bash
# Tooling: round-4 session 3: parity suite; kernel stats (product / old block-max grid); phase instruction counts; trace wave account
O=gpurun_out/r04_s3; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.txt; tail -4 $O/pytest_gpu.txt
bash tools/gpu_kernel_stats.sh r04_s3 blockout 2>&1 | grep irbpp | cut -c1-110
IRBPP_LIBRARY=$PWD/irbpp_amd/libirbpp_var_oldgrid.so bash tools/gpu_kernel_stats.sh r04_s3/oldgrid blockout 2>&1 | grep irbpp | cut -c1-110
timeout 120 python tools/trace_profile.py --workload blockout > $O/trace_blockout.json 2>/dev/null; cat $O/trace_blockout.json | cut -c1-900
timeout 600 bash tools/env_phase_pmc.sh r04_s3/phase_pmc blockout 4096 2>&1 | grep -v "^$" | cut -c1-200
timeout 200 python tools/ab_matrix.py --repeat 2 blockout:4096:1:0 blockout_k10:1024:1:0 general:4096:1:0 > $O/ab_matrix.jsonl 2> $O/ab_matrix.err; cat $O/ab_matrix.jsonl | cut -c1-150
