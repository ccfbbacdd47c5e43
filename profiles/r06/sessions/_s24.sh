#!/bin/bash
# round-6 session 24: grid sizes of the trace and polygon kernels.  A launch has ~0.4 chunks of 64 candidates and ~1.7 rounds per BlockOut bin, the
# grids are one trace wave and two polygon waves per bin: most trace waves find no chunk, but need an LDS slot (13.4 KB) before they can find that out.
# Variants (all built from the same sources): trace grid n / 2, 3 n / 8, polygon grid n, both
O=gpurun_out/r06_s24; rm -rf $O; mkdir -p $O
SPECS="blockout:8192:2:0 blockout:8192:1:0 blockout:4096:1:0 cube:4096:2:0 cube:8192:1:0 blockout_k10:2048:1:0 blockout_r8:8192:2:0 general:4096:2:0 abc_fine:2048:2:0"
for v in base0 tg2 tg38 pg2 tg2pg2 base0 tg2 tg38 pg2 tg2pg2; do
  export IRBPP_LIBRARY=$PWD/irbpp_amd/libirbpp_var_$v.so
  timeout 600 python tools/ab_matrix.py --min-seconds 0.4 $SPECS 2>/dev/null | python -c "
import sys, json
print('$v', ' '.join(str(json.loads(l)['Msteps_per_s'][0]) for l in sys.stdin))" | tee -a $O/variants.txt
done
cd /tmp && export TMPDIR=/tmp
for v in base0 tg2 tg38; do
export IRBPP_LIBRARY=$GRAFT_REPO_ROOT/irbpp_amd/libirbpp_var_$v.so
timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof_$v -o k --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --bins 8192 --groups 1 --no-extra --no-cpu-baseline > /dev/null 2>&1
f=$(ls $GRAFT_REPO_ROOT/$O/prof_$v/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && head -6 $f | cut -c1-120 | tee $GRAFT_REPO_ROOT/$O/kernel_stats_$v.txt
find $GRAFT_REPO_ROOT/$O/prof_$v -type f ! -name "*kernel_stats.csv" -delete
done
