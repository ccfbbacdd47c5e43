O=gpurun_out/r2a; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.txt
tail -4 $O/pytest_gpu.txt
timeout 600 python bench.py > $O/bench_default.json 2>$O/bench_default.err; tail -c 2500 $O/bench_default.json
for wl in general abc_fine blockout_r8 cube blockout_k10; do
timeout 600 python bench.py --no-cpu-baseline --no-extra --workload $wl > $O/bench_$wl.json 2>/dev/null
python -c "
import json; d=json.load(open('$O/bench_$wl.json')); print('$wl value', round(d['value']), 'ms', round(d['ms_per_step'],4), d.get('roofline'))"
done
R=$PWD
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/kt -o r02 -- python $R/bench.py --no-cpu-baseline --no-extra --steps 200 --warmup 20 > $R/$O/bench_under_rocprof.json 2> $R/$O/kt.err)
find $O/kt -name '*kernel_stats.csv' | head -1 | xargs head -12
find $O/kt -name '*kernel_trace.csv' -size +4M -delete
timeout 300 python tools/vecenv_throughput.py > $O/vecenv.txt 2>&1; tail -3 $O/vecenv.txt
