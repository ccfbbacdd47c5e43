#!/bin/bash
# round-5 session 12: two groups on the process's checked pair of streams (spin-kernel probe): repeatable within a process?
O=gpurun_out/r05_s12; mkdir -p $O
for i in 1 2; do
timeout 300 python tools/ab_matrix.py --repeat 4 --min-seconds 0.25 blockout:2048:2:0 blockout_k10:8192:2:0 blockout:4096:2:0 blockout:8192:2:0 general:4096:2:0 abc_fine:2048:2:0 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    j = json.loads(l); print('proc $i', j['spec'], j['Msteps_per_s'])" | tee -a $O/groups.txt
done
python - <<'PY'
import torch, time
from irbpp_amd import vec_env as V
t=time.perf_counter(); p, ok = V.group_stream_pair("cuda:0"); print("pair ok", ok, round(time.perf_counter()-t,3), "s")
pool=[torch.cuda.Stream() for _ in range(8)]
print([[int(V._run_side_by_side(a,b,torch.device("cuda:0"))) for b in pool] for a in pool])
PY
