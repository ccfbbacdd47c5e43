#!/bin/bash
# Tooling: round-4 session 25: grids of the polygon kernel (2 or 4 waves per bin) and the trace kernel (1 or 2 per bin) at small and full launches
O=gpurun_out/r04_s25; mkdir -p $O
timeout 600 python tools/ab_matrix.py --repeat 2 blockout_k10:1024:1:0 blockout_k10:1024:1:2048 blockout_k10:1024:1:4096 blockout_k10:1024:1:6144 \
  blockout:1024:1:0 blockout:1024:1:2048 blockout:1024:1:6144 abc_fine:2048:1:0 abc_fine:2048:1:2048 abc_fine:2048:1:6144 \
  blockout:4096:1:0 blockout:4096:1:2048 blockout:4096:1:4096 general:4096:1:0 general:4096:1:2048 general:4096:1:4096 > $O/ab_matrix.jsonl 2> $O/ab_matrix.err
python - <<'PY'
import json
for l in open('gpurun_out/r04_s25/ab_matrix.jsonl'):
    j=json.loads(l); print(j['spec'], j['Msteps_per_s'], j.get('kernel_ms'))
PY
