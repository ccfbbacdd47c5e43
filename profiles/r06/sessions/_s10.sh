#!/bin/bash
# round-6 session 10: the wide path (action grids of 17 .. 32 cells a side, irbpp_wide.hip) against both oracles
O=gpurun_out/r06_s10; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_wide.py -q -x > $O/pytest_wide.txt 2>&1; echo "rc=$?" >> $O/pytest_wide.txt; tail -30 $O/pytest_wide.txt
