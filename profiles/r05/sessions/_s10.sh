#!/bin/bash
# round-5 session 10: the default bench line (two groups by groups_for), buffered steps as two groups at 8192 bins, split apply inside groups
O=gpurun_out/r05_s10; mkdir -p $O
timeout 300 python tools/ab_matrix.py --repeat 2 --min-seconds 0.3 blockout_k10:8192:1:0 blockout_k10:8192:2:0 blockout_k10:4096:2:0 blockout_k10:4096:1:0 blockout:8192:2:0 blockout:8192:2:4096 \
  blockout:16384:2:0 blockout:16384:1:0 general:4096:2:0 general:8192:2:0 general:8192:1:0 cube:4096:2:0 blockout:2048:2:0 blockout:2048:1:0 2>/dev/null | tee $O/ab.jsonl | cut -c1-150
timeout 900 python bench.py > $O/bench_default.json 2>$O/bench_default.err; python - <<'PY'
import json
d = json.load(open("gpurun_out/r05_s10/bench_default.json"))
print(d["value"], d["ms_per_step"], d["config"], d["roofline"]["frac"], d["steps"], d["cpu_baseline"]["value"])
for k, v in d["extra"].items():
    print(k, v.get("value"), v.get("groups"), v.get("roofline_frac"))
PY
