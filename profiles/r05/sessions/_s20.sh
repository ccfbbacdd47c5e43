#!/bin/bash
# round-5 session 20: the committed state once more, the driver's way: GPU suite, smoke, bench --gpus 1 --steps 20 --warmup 5
O=gpurun_out/r05_s20; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.txt; tail -3 $O/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2>$O/bench_driver.err; python -c "
import json; d=json.load(open('$O/bench_driver.json')); print(d['value'], d['config']['workload'], d['roofline']['frac'], d['roofline']['traffic'], d['steps'], d['steps_requested'])"
