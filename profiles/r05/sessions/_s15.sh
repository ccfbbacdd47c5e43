#!/bin/bash
# round-5 session 15: three / four groups on streams that were checked to overlap pairwise, against two
O=gpurun_out/r05_s15; mkdir -p $O
timeout 400 python tools/ab_matrix.py --repeat 2 --min-seconds 0.3 blockout:8192:2:0 blockout:8192:4:0 blockout:6144:3:0 blockout:16384:4:0 blockout:16384:2:0 blockout:4096:4:0 general:4096:4:0 general:4096:2:0 \
  abc_fine:2048:4:0 cube:8192:4:0 cube:8192:2:0 blockout_k10:8192:4:0 blockout:8192:8:0 2>/dev/null | tee $O/ab.jsonl | cut -c1-150
python - <<'PY'
from irbpp_amd import vec_env as V
for k in (2, 3, 4, 5):
    st, ok = V.group_streams("cuda:0", k); print(k, ok)
PY
