#!/bin/bash
# round-5 session 26: trace kernel: workgroup-scope fence pair around the spill bytes instead of the agent-scope one
O=gpurun_out/r05_s26; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -x -k "adversarial or golden or full_size or trace_launch or many_bins" > $O/pytest_sel.txt 2>&1; tail -3 $O/pytest_sel.txt
timeout 400 python tools/ab_matrix.py --repeat 2 --min-seconds 0.4 blockout:8192:1:0 blockout:8192:2:0 general:4096:2:0 general:8192:1:0 abc_fine:2048:2:0 blockout_r8:8192:2:0 cube:8192:2:0 2>/dev/null | tee $O/ab.jsonl | cut -c1-150
timeout 200 bash tools/gpu_kernel_stats.sh r05_s26 blockout general 2>&1 | grep -E "trace|polygon"
