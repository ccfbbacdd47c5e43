#!/bin/bash
# round-6 session 19: compact level-image frames in the trace kernel (-DIRBPP_COMPACT_FRAMES=1: 16-bit lines, 76 instead of 140 bytes of LDS
# per lane -> 9.3 instead of 13.4 KB per wave, 17 instead of 12 waves per CU) as a variant build against the shipped build, same box;
# parity of the variant first (goldens + features + large launch forms), then the A/B matrix and the kernel durations
O=gpurun_out/r06_s19; rm -rf $O; mkdir -p $O
export IRBPP_LIBRARY=$PWD/irbpp_amd/libirbpp_var_cf.so
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_large_forms.py tests/test_gpu_features.py -m gpu -x -q 2>&1 | tail -3 | tee $O/pytest_cf.txt
unset IRBPP_LIBRARY
SPECS="blockout:8192:2:0 blockout:8192:1:0 blockout:4096:1:0 general:4096:2:0 abc_fine:2048:2:0 blockout_k10:1024:1:0 cube:4096:2:0 blockout:1024:1:0"
for v in base cf base cf; do
  if [ $v = base ]; then unset IRBPP_LIBRARY; else export IRBPP_LIBRARY=$PWD/irbpp_amd/libirbpp_var_$v.so; fi
  timeout 600 python tools/ab_matrix.py --min-seconds 0.4 $SPECS 2>/dev/null | python -c "
import sys, json
print('$v', ' '.join(str(json.loads(l)['Msteps_per_s'][0]) for l in sys.stdin))" | tee -a $O/variants.txt
done
for v in base cf; do
  if [ $v = base ]; then unset IRBPP_LIBRARY; else export IRBPP_LIBRARY=$PWD/irbpp_amd/libirbpp_var_$v.so; fi
  timeout 600 python bench.py --no-extra --no-cpu-baseline > $O/bench_$v.json 2>/dev/null
  python -c "
import json; d=json.load(open('$O/bench_$v.json')); print('$v', d['value']/1e6, d['roofline'].get('kernels_us') or d['roofline'])"
done
