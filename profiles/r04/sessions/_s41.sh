#!/bin/bash
# Tooling: round-4 session 41: 32 / 16 candidates per trace wave at small launches, again (tuning flags 64 / 128), now that the polygon and emit kernels are shorter
O=gpurun_out/r04_s41; mkdir -p $O
timeout 600 python tools/ab_matrix.py --repeat 2 blockout_k10:1024:1:0 blockout_k10:1024:1:64 blockout_k10:1024:1:128 blockout:1024:1:0 blockout:1024:1:64 blockout:1024:1:128 \
  blockout:512:1:0 blockout:512:1:64 blockout:2048:1:0 blockout:2048:1:64 blockout_k10:2048:1:0 blockout_k10:2048:1:64 abc_fine:2048:1:0 abc_fine:2048:1:64 blockout:4096:1:0 blockout:4096:1:64 > $O/ab_matrix.jsonl 2> $O/ab_matrix.err
python - <<'PY'
import json
for l in open('gpurun_out/r04_s41/ab_matrix.jsonl'):
    j=json.loads(l); print(j['spec'], j['Msteps_per_s'])
PY
