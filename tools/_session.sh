O=gpurun_out/r2v; mkdir -p $O
timeout 1500 python -m pytest tests/test_replay.py -m gpu -q -x > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.txt
tail -3 $O/pytest_gpu.txt
timeout 600 python tools/actor_loop_throughput.py > $O/actor_loop.txt 2>&1; tail -1 $O/actor_loop.txt
R=$PWD
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/kt -o r02 -- python $R/tools/actor_loop_throughput.py --loop-bins 0 > /dev/null 2> $R/$O/kt.err)
find $O/kt -name '*kernel_stats.csv' | head -1 | xargs head -24 | cut -c1-130
find $O/kt -name '*kernel_trace.csv' -delete
