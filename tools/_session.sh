set -x
O=gpurun_out/r2a; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.txt
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err
timeout 600 python tools/ablate.py > $O/ablate_blockout.json 2> $O/ablate_blockout.err
timeout 600 python tools/ablate.py --workload general --rounds 1 > $O/ablate_general.json 2> $O/ablate_general.err
tail -3 $O/pytest_gpu.txt; cat $O/bench_default.json | cut -c1-1500; cat $O/ablate_blockout.json; cat $O/ablate_general.json
