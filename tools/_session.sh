O=gpurun_out/r3a; mkdir -p $O
R=$PWD
run() { # name workload env...
  n=$1; wl=$2; shift 2
  env "$@" timeout 600 python bench.py --no-cpu-baseline --no-extra --workload $wl > $O/bench_$n.json 2>$O/bench_$n.err || tail -3 $O/bench_$n.err
  python -c "
import json; d=json.load(open('$O/bench_$n.json')); print('$n value', round(d['value']), 'ms', round(d['ms_per_step'],4))"
}
run base blockout A=1
run tg50 blockout IRBPP_TRACE_GRID_PCT=50
run tg35 blockout IRBPP_TRACE_GRID_PCT=35
run pg130 blockout IRBPP_POLY_GRID_PCT=130
run pg100 blockout IRBPP_POLY_GRID_PCT=100
run tg50pg130 blockout IRBPP_TRACE_GRID_PCT=50 IRBPP_POLY_GRID_PCT=130
run short0 blockout IRBPP_LIBRARY=$R/irbpp_amd/libirbpp_var_short0.so
run base2 blockout A=1
run general_base general A=1
run general_tg50pg130 general IRBPP_TRACE_GRID_PCT=50 IRBPP_POLY_GRID_PCT=130
