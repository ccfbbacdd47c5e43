O=gpurun_out/r3i; mkdir -p $O
R=$PWD
timeout 1500 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.txt
tail -3 $O/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 900 python bench.py > $O/bench_default.json 2>$O/bench_default.err; python -c "
import json; d=json.load(open('$O/bench_default.json')); print('default', round(d['value']), d['ms_per_step'], d['roofline']['frac'], d['grouped_stepping']['value'], d['extra']['bins8192_one_gpu']['value'], d['cpu_baseline']['value'])"
for wl in general abc_fine blockout_r8 cube blockout_k10; do
timeout 600 python bench.py --no-cpu-baseline --no-extra --workload $wl > $O/bench_$wl.json 2>/dev/null
python -c "
import json; d=json.load(open('$O/bench_$wl.json')); print('$wl value', round(d['value']), 'ms', round(d['ms_per_step'],4), 'frac', round(d['roofline']['frac'],4))"
done
bash tools/gpu_profile.sh r3i/prof blockout 16384 2>&1 | tail -3
python -c "
import json; d=json.load(open('$O/prof/kernel_trace_timed_region.json')); print({k:round(v['avg_us_last'],2) for k,v in d.items() if k.startswith('irbpp')})"
timeout 300 python tools/vecenv_throughput.py > $O/vecenv.txt 2>&1; tail -1 $O/vecenv.txt
timeout 300 python tools/phase_profile.py > $O/phase_blockout.json 2>/dev/null; python -c "
import json; d=json.load(open('$O/phase_blockout.json')); print(d['split_pipeline'])"
