#!/bin/bash
# round-6 session 25: 32 candidate starts per trace wave (IRBPP_TUNE_TRACE_CPW32 = 64) at the large launches, on this round's kernels: twice the
# waves, half the LDS each, the longest of 32 borders instead of 64 -- against the polygon rounds it leaves less full
O=gpurun_out/r06_s25; rm -rf $O; mkdir -p $O
for rep in 1 2; do
for tune in 0 64; do
  timeout 600 python tools/ab_matrix.py --min-seconds 0.4 blockout:8192:2:$tune blockout:8192:1:$tune blockout:4096:1:$tune cube:4096:2:$tune blockout_r8:8192:2:$tune general:4096:2:$tune blockout_k10:2048:1:$tune 2>/dev/null | python -c "
import sys, json
print('tune=$tune', ' '.join(str(json.loads(l)['Msteps_per_s'][0]) for l in sys.stdin))" | tee -a $O/variants.txt
done; done
cd /tmp && export TMPDIR=/tmp
for tune in 64; do
timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof_$tune -o k --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --bins 8192 --groups 1 --no-extra --no-cpu-baseline --tuning $tune > /dev/null 2>&1
f=$(ls $GRAFT_REPO_ROOT/$O/prof_$tune/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && head -6 $f | cut -c1-120 | tee $GRAFT_REPO_ROOT/$O/kernel_stats_$tune.txt
find $GRAFT_REPO_ROOT/$O/prof_$tune -type f ! -name "*kernel_stats.csv" -delete
done
