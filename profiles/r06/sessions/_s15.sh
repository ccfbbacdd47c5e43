#!/bin/bash
# round-6 session 15: the wide path with run jumps in its border walk: parity, throughput, phases
O=gpurun_out/r06_s15; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_wide.py -q -x > $O/pytest_wide.txt 2>&1; echo "rc=$?" >> $O/pytest_wide.txt; tail -4 $O/pytest_wide.txt
timeout 900 python tools/wide_throughput.py 2>&1 | tail -1 | tee $O/wide_throughput.json
timeout 600 python tools/wide_phase_profile.py 2>&1 | tail -1 | tee $O/wide_phases.json
