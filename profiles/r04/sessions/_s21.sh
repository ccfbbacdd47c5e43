#!/bin/bash
# Tooling: round-4 session 21: border slots of 64 / 48 points in the trace kernel (LDS 18.6 -> 14.5 / 13.4 KB per wave: 8 -> 11 waves per CU)
# (the variant builds of this session, -DIRBPP_TRACE_CAP=64 / 48, were hooks that were taken out again with the experiment: profiles/r04/LOG.md)
O=gpurun_out/r04_s21; mkdir -p $O
bash tools/gpu_kernel_stats.sh r04_s21 blockout general 2>&1 | grep irbpp_trace | cut -c1-110
for v in cap64 cap48; do
IRBPP_LIBRARY=$PWD/irbpp_amd/libirbpp_var_$v.so bash tools/gpu_kernel_stats.sh r04_s21/$v blockout general 2>&1 | grep irbpp_trace | cut -c1-110
done
