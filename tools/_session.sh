O=gpurun_out/r2s; mkdir -p $O
R=$PWD
timeout 1500 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.txt
tail -4 $O/pytest_gpu.txt
run() { # name workload env...
  n=$1; wl=$2; shift 2
  env "$@" timeout 600 python bench.py --no-cpu-baseline --no-extra --workload $wl > $O/bench_$n.json 2>$O/bench_$n.err || tail -3 $O/bench_$n.err
  python -c "
import json; d=json.load(open('$O/bench_$n.json')); print('$n value', round(d['value']), 'ms', round(d['ms_per_step'],4), 'kernel_ms', round(d['roofline']['kernel_ms'],4))"
}
run blockout blockout A=1
run blockout_e1 blockout IRBPP_BENCH_TIMING_EVERY=1
run blockout_e8 blockout IRBPP_BENCH_TIMING_EVERY=8
run blockout_b blockout A=1
run k10 blockout_k10 A=1
