#!/bin/bash
# round-5 session 30: the 512-thread build under the 64-VGPR cap (four workgroups = eight waves per SIMD instead of three = six)
O=gpurun_out/r05_s30; mkdir -p $O
timeout 400 python tools/ab_matrix.py --repeat 2 --min-seconds 0.4 abc_fine:2048:2:0 abc_fine:2048:2:65540 abc_fine:2048:1:0 abc_fine:2048:1:65540 abc_fine:8192:2:0 abc_fine:8192:2:65540 2>/dev/null | tee $O/ab.jsonl | cut -c1-150
