#!/bin/bash
# Tooling: round-4 session 31: heavy-bin threshold 0.6 S with the run-level start filter (general's emit kernel)
O=gpurun_out/r04_s31; mkdir -p $O
bash tools/gpu_kernel_stats.sh r04_s31 general blockout 2>&1 | grep "irbpp_" | cut -c1-110
timeout 300 python tools/ab_matrix.py --repeat 2 general:4096:1:0 abc_fine:2048:1:0 blockout:4096:1:0 > $O/ab_matrix.jsonl 2> $O/ab_matrix.err; cat $O/ab_matrix.jsonl | cut -c1-150
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x > $O/pytest_parity.txt 2>&1; tail -2 $O/pytest_parity.txt
