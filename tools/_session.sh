O=gpurun_out/r2m; mkdir -p $O
R=$PWD
timeout 1500 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.txt
tail -8 $O/pytest_gpu.txt
run() { # name workload env...
  n=$1; wl=$2; shift 2
  env "$@" timeout 600 python bench.py --no-cpu-baseline --no-extra --workload $wl > $O/bench_$n.json 2>$O/bench_$n.err || tail -3 $O/bench_$n.err
  python -c "
import json; d=json.load(open('$O/bench_$n.json')); print('$n value', round(d['value']), 'ms', round(d['ms_per_step'],4), 'kernel_ms', round(d['roofline']['kernel_ms'],4), 'lds', d['roofline']['lds_bytes_per_workgroup'])"
}
run blockout blockout A=1
run general general A=1
run abc_fine abc_fine A=1
run k10 blockout_k10 A=1
for wl in blockout general; do
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/kt_$wl -o r02 -- python $R/bench.py --workload $wl --no-cpu-baseline --no-extra --steps 100 --warmup 10 > $R/$O/bench_rocprof_$wl.json 2> $R/$O/kt_$wl.err)
find $O/kt_$wl -name '*kernel_stats.csv' | head -1 | xargs head -5 | cut -c1-150
find $O/kt_$wl -name '*kernel_trace.csv' -delete
done
