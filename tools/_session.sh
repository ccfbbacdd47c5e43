O=gpurun_out/r2d; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.txt
tail -6 $O/pytest_gpu.txt
for wl in blockout general abc_fine; do
timeout 600 python bench.py --no-cpu-baseline --workload $wl > $O/bench_$wl.json 2>$O/bench_$wl.err || tail -3 $O/bench_$wl.err
python -c "
import json; d=json.load(open('$O/bench_$wl.json')); print('$wl value', round(d['value']), 'ms', round(d['ms_per_step'],4), 'kernel_ms', round(d['roofline']['kernel_ms'],4), d.get('grouped_stepping',{}).get('value'), d.get('extra'))"
done
R=$PWD
for wl in blockout general; do
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/kt_$wl -o r02 -- python $R/bench.py --workload $wl --no-cpu-baseline --no-extra --steps 100 --warmup 10 > $R/$O/bench_rocprof_$wl.json 2> $R/$O/kt_$wl.err)
find $O/kt_$wl -name '*kernel_stats.csv' | head -1 | xargs head -5 | cut -c1-150
find $O/kt_$wl -name '*kernel_trace.csv' -delete
done
