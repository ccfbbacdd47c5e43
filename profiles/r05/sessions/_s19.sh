#!/bin/bash
# round-5 session 19: the default bench line again, after the PMC fold (traffic filled in from profiles/pmc_hbm.json)
O=gpurun_out/final; mkdir -p $O
timeout 900 python bench.py > $O/bench_default.json 2>$O/bench_default.err; tail -c 400 $O/bench_default.json; echo
timeout 200 python bench.py --steps 20 --warmup 5 --no-extra --no-cpu-baseline > $O/bench_driver_style.json 2>/dev/null
