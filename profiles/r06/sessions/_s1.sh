#!/bin/bash
# round-6 session 1: the new parity tests (bench launch forms, BAD_ACTION, RCCL single rank), then the whole GPU suite, smoke, quick bench
O=gpurun_out/r06_s1; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_large_forms.py tests/test_gpu_rccl.py tests/test_gpu_boundary.py -q -x --durations=15 > $O/pytest_new.txt 2>&1; echo "new rc=$?" >> $O/pytest_new.txt; tail -25 $O/pytest_new.txt
cp gpurun_out/rccl_single_rank.json $O/ 2>/dev/null
timeout 1700 python -m pytest tests -m gpu -q -x --deselect tests/test_gpu_large_forms.py --deselect tests/test_gpu_rccl.py --deselect tests/test_gpu_boundary.py > $O/pytest_rest.txt 2>&1; echo "rest rc=$?" >> $O/pytest_rest.txt; tail -5 $O/pytest_rest.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 600 python bench.py --no-cpu-baseline --no-extra --min-seconds 0.5 > $O/bench_quick.json 2>$O/bench_quick.err; tail -c 600 $O/bench_quick.json
nproc; free -g | head -2
