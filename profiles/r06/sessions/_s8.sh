#!/bin/bash
# round-6 session 8: compiler scheduling strategies as variant builds (tools/build_variant.sh): iterative-ilp, iterative-minreg, max-memory-clause,
# -fno-unroll-loops against the shipped build, same box
O=gpurun_out/r06_s8; mkdir -p $O
SPECS="blockout:8192:2:0 blockout:8192:1:0 general:4096:2:0 abc_fine:2048:2:0 blockout_k10:1024:1:0 cube:4096:2:0"
for v in base iterilp minreg memclause nounroll base; do
  if [ $v = base ]; then unset IRBPP_LIBRARY; else export IRBPP_LIBRARY=$PWD/irbpp_amd/libirbpp_var_$v.so; fi
  timeout 600 python tools/ab_matrix.py --min-seconds 0.4 $SPECS 2>/dev/null | python -c "
import sys, json
print('$v', ' '.join(str(json.loads(l)['Msteps_per_s'][0]) for l in sys.stdin))" | tee -a $O/variants.txt
done
