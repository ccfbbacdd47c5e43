#!/bin/bash
# round-6 session 23: the long soak once more on the round's final sources (stamp 43beb598c1278397) -- every bin against the plain-C oracle at
# the bench's launch sizes -- and the opt-in forms of the round at full width: rectangles answered by the transition kernel (4194304), lane refill
# (262144), two waves per bin (2097152)
O=gpurun_out/r06_s23; rm -rf $O; mkdir -p $O
timeout 2700 python tools/soak_parity.py blockout:4096:1:600 blockout:8192:2:300 cube:4096:1:300 general:2048:2:200 blockout_k10:2048:1:300 \
   blockout_r8:2048:1:300 abc_fine:4096:1:40 blockout:4096:1:300:4194304 cube:4096:1:200:4194304 blockout_r8:2048:1:150:4194304 general:2048:1:100:4194304 \
   blockout:4096:1:200:262144 blockout:4096:1:200:2097152 2>&1 | grep '^{' | tee $O/soak.jsonl | python -c "
import sys, json
for l in sys.stdin:
    j = json.loads(l); print(j['spec'], j.get('identical'), j.get('observations_compared'), j.get('episodes_finished'), j.get('seconds'), j.get('kernels','')[:60])"
