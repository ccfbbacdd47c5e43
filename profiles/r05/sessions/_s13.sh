#!/bin/bash
# round-5 session 13: whole GPU suite + the default bench line (two groups on the checked pair of streams), driver-style call too
O=gpurun_out/r05_s13; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -q > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.txt; tail -5 $O/pytest_gpu.txt
timeout 900 python bench.py > $O/bench_default.json 2>$O/bench_default.err; python - <<'PY'
import json
d = json.load(open("gpurun_out/r05_s13/bench_default.json"))
print(d["value"], d["ms_per_step"], d["config"]["groups_per_gpu"], d["roofline"]["frac"], d["roofline"]["kernel"], d["steps"], d["cpu_baseline"]["value"])
for k, v in d["extra"].items():
    print(k, v.get("value"), v.get("groups"), v.get("roofline_frac"))
PY
timeout 300 python bench.py --steps 20 --warmup 5 --no-extra --no-cpu-baseline > $O/bench_driver_style.json 2>/dev/null; python -c "
import json; d=json.load(open('gpurun_out/r05_s13/bench_driver_style.json')); print('driver style', d['value'], d['steps'], d['steps_requested'])"
