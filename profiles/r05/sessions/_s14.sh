#!/bin/bash
# round-5 session 14: small launches replayed as HIP graphs (default up to 2048 bins per launch) against direct launches (tuning 32768)
O=gpurun_out/r05_s14; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.txt; tail -5 $O/pytest_gpu.txt
timeout 600 python tools/ab_matrix.py --repeat 2 --min-seconds 0.3 blockout_k10:1024:1:0 blockout_k10:1024:1:32768 blockout_k10:1024:2:0 blockout_k10:1024:2:32768 blockout_k10:1024:4:0 \
  blockout:1024:1:0 blockout:1024:1:32768 blockout:1024:2:0 blockout:2048:1:0 blockout:2048:1:32768 blockout:2048:2:0 abc_fine:2048:1:0 abc_fine:2048:1:32768 abc_fine:2048:2:0 \
  blockout:8192:2:0 blockout:8192:2:65536 general:1024:1:0 general:1024:1:32768 2>/dev/null | tee $O/ab.jsonl | cut -c1-150
