#!/bin/bash
# round-5 session 35: the synchronous VecEnv step as the groups make_vec_envs picks (num_groups = 0) against one group
O=gpurun_out/r05_s35; mkdir -p $O
for spec in "4096 0" "4096 1" "8192 0" "8192 1" "2048 0" "2048 1"; do timeout 120 python tools/vecenv_throughput.py $spec 2>$O/err.txt | tee -a $O/vecenv.jsonl | cut -c1-420; tail -2 $O/err.txt; done
