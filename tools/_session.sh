O=gpurun_out/r2q; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -q > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.txt
tail -25 $O/pytest_gpu.txt
timeout 600 python bench.py --no-cpu-baseline --no-extra > $O/bench.json 2> $O/bench.err; cut -c1-400 $O/bench.json
timeout 600 python tools/actor_loop_throughput.py --loop-bins 0 > $O/actor_loop.json 2> $O/actor.err; cat $O/actor_loop.json; tail -2 $O/actor.err
