#!/bin/bash
# round-6 session 5: PATH_MIXED with the wave-per-bin emit kernel (BlockOut at eight rotations has lattice level images) -- parity, A/B
O=gpurun_out/r06_s5; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_features.py tests/test_gpu_large_forms.py tests/test_gpu_parity.py -q -x -k "specialised or lattice_data or blockout_r8 or many_bins or fused_policy or registered" > $O/pytest_a.txt 2>&1; echo "rc=$?" >> $O/pytest_a.txt; tail -4 $O/pytest_a.txt
timeout 600 python tools/ab_matrix.py --min-seconds 0.4 blockout_r8:8192:2:0 blockout_r8:8192:2:8192 blockout_r8:8192:1:0 blockout_r8:8192:1:8192 blockout_r8:4096:2:0 blockout_r8:4096:2:8192 2>/dev/null | tee $O/ab.jsonl | python -c "
import sys, json
for l in sys.stdin:
    j = json.loads(l); print(j['spec'], j['Msteps_per_s'])"
bash tools/gpu_kernel_stats.sh r06_s5 blockout_r8 2>&1 | tail -7
