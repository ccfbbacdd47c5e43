#!/bin/bash
# round-6 session 17: long soak -- the HIP path against the plain-C oracle on EVERY bin at the bench's launch sizes, hundreds of steps (many
# episodes per bin), every workload incl. BlockOut at R = 8 (PATH_MIXED) and the 64 x 64 heightmap at 4096 bins (_s4_w512c)
O=gpurun_out/r06_s17; mkdir -p $O
timeout 2400 python tools/soak_parity.py blockout:4096:1:1000 blockout:8192:2:400 cube:4096:1:400 general:2048:2:300 blockout_k10:2048:1:400 \
   blockout_r8:2048:1:400 abc_fine:4096:1:60 general:4096:1:150 2>&1 | grep '^{' | tee $O/soak.jsonl | python -c "
import sys, json
for l in sys.stdin:
    j = json.loads(l); print(j['spec'], j.get('identical'), j.get('observations_compared'), j.get('episodes_finished'), j.get('seconds'), j.get('kernels','')[:60])"
