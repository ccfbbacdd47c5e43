#!/bin/bash
# Tooling: round-4 session 23: the trace kernel's border slots in global scratch instead of LDS (10.1 KB of LDS per wave: 16 waves per CU, one pass)
# (the variant builds of this session, -DIRBPP_AB_TRACE_GSLOTS, were hooks that were taken out again with the experiment: profiles/r04/LOG.md)
O=gpurun_out/r04_s23; mkdir -p $O
IRBPP_LIBRARY=$PWD/irbpp_amd/libirbpp_var_gslots.so timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_boundary.py -m gpu -q -x > $O/pytest_gslots.txt 2>&1; tail -3 $O/pytest_gslots.txt | cut -c1-200
bash tools/gpu_kernel_stats.sh r04_s23 blockout general 2>&1 | grep "irbpp_trace\|irbpp_poly" | cut -c1-110
IRBPP_LIBRARY=$PWD/irbpp_amd/libirbpp_var_gslots.so bash tools/gpu_kernel_stats.sh r04_s23/gslots blockout general 2>&1 | grep "irbpp_trace\|irbpp_poly" | cut -c1-110
