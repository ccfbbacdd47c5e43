#!/bin/bash
# round-6 session 4: PATH_MIXED (BlockOut at eight rotations: lattice rotations on the block path, 45-degree ones on cell lists) -- parity, then A/B
O=gpurun_out/r06_s4; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_features.py tests/test_gpu_large_forms.py -q -x -k "specialised or lattice_data or blockout_r8 or get_all_possible" > $O/pytest_a.txt 2>&1; echo "rc=$?" >> $O/pytest_a.txt; tail -4 $O/pytest_a.txt
timeout 1500 python -m pytest tests/test_gpu_parity.py -q -x -k "blockout_r8 or many_bins or possible_position or heuristic" > $O/pytest_b.txt 2>&1; echo "rc=$?" >> $O/pytest_b.txt; tail -4 $O/pytest_b.txt
timeout 600 python tools/ab_matrix.py --min-seconds 0.4 blockout_r8:8192:2:0 blockout_r8:8192:2:524288 blockout_r8:8192:1:0 blockout_r8:8192:1:524288 blockout_r8:4096:2:0 blockout_r8:4096:2:524288 blockout:8192:2:0 general:4096:2:0 2>/dev/null | tee $O/ab.jsonl | python -c "
import sys, json
for l in sys.stdin:
    j = json.loads(l); print(j['spec'], j['Msteps_per_s'])"
bash tools/gpu_kernel_stats.sh r06_s4 blockout_r8 2>&1 | tail -7
