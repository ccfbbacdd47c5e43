#!/bin/bash
# round-6 session 22: the two list allocations (candidate list in the transition kernel's hand-over, round records in the trace kernel)
# asked for EARLY and looked at late, against the variant build that waits for the atomic on the spot (-DIRBPP_AB_LATE_RESERVE); parity first
O=gpurun_out/r06_s22; rm -rf $O; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_features.py tests/test_gpu_large_forms.py tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -3 | tee $O/pytest.txt
SPECS="blockout:8192:2:0 blockout:8192:1:0 blockout:4096:1:0 cube:4096:2:0 blockout_k10:1024:1:0 blockout_r8:8192:2:0 general:4096:2:0 abc_fine:2048:2:0 blockout:1024:1:0"
for v in base late base late; do
  if [ $v = base ]; then unset IRBPP_LIBRARY; else export IRBPP_LIBRARY=$PWD/irbpp_amd/libirbpp_var_$v.so; fi
  timeout 600 python tools/ab_matrix.py --min-seconds 0.4 $SPECS 2>/dev/null | python -c "
import sys, json
print('$v', ' '.join(str(json.loads(l)['Msteps_per_s'][0]) for l in sys.stdin))" | tee -a $O/variants.txt
done
cd /tmp && export TMPDIR=/tmp
for v in base late; do
if [ $v = base ]; then unset IRBPP_LIBRARY; else export IRBPP_LIBRARY=$GRAFT_REPO_ROOT/irbpp_amd/libirbpp_var_$v.so; fi
timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof_$v -o k --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --bins 8192 --groups 1 --no-extra --no-cpu-baseline > /dev/null 2>&1
f=$(ls $GRAFT_REPO_ROOT/$O/prof_$v/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && head -7 $f | cut -c1-120 | tee $GRAFT_REPO_ROOT/$O/kernel_stats_$v.txt
find $GRAFT_REPO_ROOT/$O/prof_$v -type f ! -name "*kernel_stats.csv" -delete
done
