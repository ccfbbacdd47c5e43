#!/bin/bash
# round-5 session 2: whole GPU suite (no -x) on the specialised builds + default bench line
O=gpurun_out/r05_s2; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -q > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.txt; tail -8 $O/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
