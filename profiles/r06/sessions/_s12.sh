#!/bin/bash
# round-6 session 12: the wide path with batched level images, 64 tracing lanes and the neighbour-mask walk: parity, then throughput
O=gpurun_out/r06_s12; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_wide.py -q -x > $O/pytest_wide.txt 2>&1; echo "rc=$?" >> $O/pytest_wide.txt; tail -5 $O/pytest_wide.txt
timeout 900 python tools/wide_throughput.py 2>&1 | tail -1 | tee $O/wide_throughput.json
