#!/bin/bash
# round-6 session 13: the default bench once more after the PMC fold (the line then carries `traffic`), and called the driver's way
O=gpurun_out/final2; rm -rf $O; mkdir -p $O
timeout 900 python bench.py > $O/bench_default.json 2>$O/bench_default.err; tail -c 400 $O/bench_default.json; echo
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_style.json 2>/dev/null; python -c "
import json; d=json.load(open('$O/bench_driver_style.json')); print('driver style', d['value']/1e6, d['steps'], d['timed_blocks'], d['roofline']['traffic'], d['cpu_baseline']['value'])"
