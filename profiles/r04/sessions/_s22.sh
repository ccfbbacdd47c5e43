#!/bin/bash
# Tooling: round-4 session 22: how the trace kernel's duration depends on its waves per CU (LDS padded: 8 -> 6 -> 4 waves per CU)
# (the variant builds of this session, -DIRBPP_AB_TRACE_LDS_PAD=8192 / 14336, were hooks that were taken out again with the experiment: profiles/r04/LOG.md)
O=gpurun_out/r04_s22; mkdir -p $O
bash tools/gpu_kernel_stats.sh r04_s22 blockout general 2>&1 | grep irbpp_trace | cut -c1-110
for v in pad8192 pad14336; do
IRBPP_LIBRARY=$PWD/irbpp_amd/libirbpp_var_$v.so bash tools/gpu_kernel_stats.sh r04_s22/$v blockout general 2>&1 | grep irbpp_trace | cut -c1-110
done
