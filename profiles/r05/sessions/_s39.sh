#!/bin/bash
# round-5 session 39: the same soak on the fine heightmap (512-thread builds), BlockOut at R = 8 and 8192 BlockOut bins as two groups
O=gpurun_out/r05_s39; mkdir -p $O
timeout 500 python tools/soak_parity.py abc_fine:4096:2:10 abc_fine:2048:1:16 blockout_r8:2048:2:40 blockout:8192:2:40 2>$O/err.txt | tee $O/soak.jsonl | cut -c1-330
tail -3 $O/err.txt
