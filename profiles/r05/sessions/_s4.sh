#!/bin/bash
# round-5 session 4: apply phase with the placed item's ShapeRots staged in LDS in round 2 and the drop height read from the last
# observation's hand-over (w_posz / w_valid): whole GPU suite, then the A/B points of session 1
O=gpurun_out/r05_s4; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -q > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.txt; tail -6 $O/pytest_gpu.txt
timeout 600 python tools/ab_matrix.py --repeat 2 blockout:8192:1:0 blockout:4096:1:0 cube:4096:1:0 general:4096:1:0 abc_fine:2048:1:0 \
  blockout_k10:1024:1:0 blockout_r8:4096:1:0 2>/dev/null | tee $O/ab.jsonl | cut -c1-200
timeout 120 python tools/phase_profile.py --workload blockout --bins 8192 > $O/phase_blockout.json 2>/dev/null; head -c 400 $O/phase_blockout.json
