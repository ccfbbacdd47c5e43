#!/bin/bash
# round-6 session 7: the whole GPU suite on the tree with PATH_MIXED, the opt-in CHAIN / refill builds, get_all_possible_observation
O=gpurun_out/r06_s7; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q > $O/pytest_gpu.txt 2>&1; echo "rc=$?" >> $O/pytest_gpu.txt; tail -6 $O/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
