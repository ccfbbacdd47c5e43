O=gpurun_out/r3f; mkdir -p $O
R=$PWD
timeout 1500 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.txt
tail -3 $O/pytest_gpu.txt
run() { # name workload env...
  n=$1; wl=$2; shift 2
  env "$@" timeout 600 python bench.py --no-cpu-baseline --no-extra --workload $wl > $O/bench_$n.json 2>$O/bench_$n.err || tail -3 $O/bench_$n.err
  python -c "
import json; d=json.load(open('$O/bench_$n.json')); print('$n value', round(d['value']), 'ms', round(d['ms_per_step'],4))"
}
run new blockout A=1
run prev blockout IRBPP_LIBRARY=$R/irbpp_amd/libirbpp_var_prev.so
run new2 blockout A=1
run prev2 blockout IRBPP_LIBRARY=$R/irbpp_amd/libirbpp_var_prev.so
run general_new general A=1
run general_prev general IRBPP_LIBRARY=$R/irbpp_amd/libirbpp_var_prev.so
run general_narrow general IRBPP_WIDE=0
run cube_new cube A=1
run cube_prev cube IRBPP_LIBRARY=$R/irbpp_amd/libirbpp_var_prev.so
