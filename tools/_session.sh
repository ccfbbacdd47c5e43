O=gpurun_out/r3d; mkdir -p $O
R=$PWD
timeout 900 python bench.py > $O/bench_default.json 2>$O/bench_default.err; python -c "
import json; d=json.load(open('$O/bench_default.json')); print('default', round(d['value']), d['ms_per_step'], d['roofline']['frac'], d['grouped_stepping']['value'], d['extra']['bins8192_one_gpu']['value'], d['cpu_baseline']['value'])"
bash tools/gpu_profile.sh r3d/prof blockout 16384 2>&1 | tail -3
python -c "
import json; d=json.load(open('$O/prof/kernel_trace_timed_region.json')); print({k:round(v['avg_us_last'],2) for k,v in d.items() if k.startswith('irbpp')})"
