#!/bin/bash
# round-6 session 20: ReplayMemory.append for all envs as one launch (irbpp_replay_append) -- parity with the torch formulation
# and the memory.py fixtures, then the acting loop's rate at 4096 / 8192 bins
O=gpurun_out/r06_s20; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_replay.py tests/test_abi.py -x -q 2>&1 | tail -3 | tee $O/pytest_replay.txt
for n in 4096 8192; do
  timeout 600 python tools/actor_loop_throughput.py --bins $n --steps 200 --loop-bins 0 2>/dev/null | tail -1 | tee -a $O/actor_loop.jsonl
done
timeout 600 python tools/actor_loop_throughput.py --bins 4096 --steps 200 --capacity 1024 --loop-bins 0 2>/dev/null | tail -1 | tee -a $O/actor_loop.jsonl
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof -o actor --output-format csv -- python $GRAFT_REPO_ROOT/tools/actor_loop_throughput.py --bins 4096 --steps 200 --loop-bins 0 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT; f=$(ls $O/prof/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && head -14 $f | cut -c1-160 | tee $O/actor_kernel_stats.txt
find $O/prof -type f ! -name "*kernel_stats.csv" -delete
