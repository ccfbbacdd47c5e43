#!/bin/bash
# round-5 session 34: the synchronous VecEnv step (trainer.py's loop) as one group and as the groups make_vec_envs would pick; where the host's time goes
O=gpurun_out/r05_s34; mkdir -p $O
for spec in "4096 1" "4096 0" "8192 1" "8192 0" "1024 1"; do timeout 120 python tools/vecenv_throughput.py $spec 2>/dev/null | tee -a $O/vecenv.jsonl | cut -c1-420; done
timeout 300 python -m pytest tests -m gpu -q -x -k "vecenv or grouped or groups or boundary" 2>&1 | tail -2
