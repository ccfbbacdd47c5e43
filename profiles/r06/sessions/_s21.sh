#!/bin/bash
# round-6 session 21: isolated solid rectangles answered by the transition kernel (rect_component / rect_vertices) instead of the trace
# kernel: parity (boundary + features + large forms + parity files), then A/B against IRBPP_TUNE_NO_RECT (4194304), kernel stats
O=gpurun_out/r06_s21; rm -rf $O; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_boundary.py tests/test_gpu_features.py tests/test_gpu_large_forms.py tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -4 | tee $O/pytest.txt
SPECS="blockout:8192:2 blockout:8192:1 blockout:4096:1 cube:4096:2 blockout_k10:1024:1 blockout_r8:8192:2 general:4096:2 abc_fine:2048:2 blockout:1024:1"
for rep in 1 2; do
for tune in 0 4194304; do
  A=""; for s in $SPECS; do A="$A $s:$tune"; done
  timeout 600 python tools/ab_matrix.py --min-seconds 0.4 $A 2>/dev/null | python -c "
import sys, json
print('tune=$tune', ' '.join(str(json.loads(l)['Msteps_per_s'][0]) for l in sys.stdin))" | tee -a $O/variants.txt
done; done
cd /tmp && export TMPDIR=/tmp
for tune in 0 4194304; do
timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof_$tune -o k --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --bins 8192 --groups 1 --no-extra --no-cpu-baseline --tuning $tune > /dev/null 2>&1
f=$(ls $GRAFT_REPO_ROOT/$O/prof_$tune/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && head -7 $f | cut -c1-120 | tee $GRAFT_REPO_ROOT/$O/kernel_stats_$tune.txt
find $GRAFT_REPO_ROOT/$O/prof_$tune -type f ! -name "*kernel_stats.csv" -delete
done
