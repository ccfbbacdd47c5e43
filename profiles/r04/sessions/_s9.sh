#!/bin/bash
# Tooling: round-4 session 9: one emit build, polygon rounds of 256 points (variant), generic capped vs uncapped, bench line
O=gpurun_out/r04_s9; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -q > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.txt; tail -4 $O/pytest_gpu.txt | cut -c1-200
IRBPP_LIBRARY=$PWD/irbpp_amd/libirbpp_var_pp4.so timeout 600 python -m pytest tests -m gpu -q -x -k "many_bins or launch_shapes or adversarial or more_than_S" > $O/pytest_pp4.txt 2>&1; tail -2 $O/pytest_pp4.txt | cut -c1-200
bash tools/gpu_kernel_stats.sh r04_s9 blockout general 2>&1 | grep irbpp | cut -c1-110
IRBPP_LIBRARY=$PWD/irbpp_amd/libirbpp_var_pp4.so bash tools/gpu_kernel_stats.sh r04_s9/pp4 blockout general 2>&1 | grep irbpp | cut -c1-110
timeout 400 python tools/ab_matrix.py --repeat 2 blockout:4096:1:0 general:4096:1:0 general:4096:1:2 general:4096:2:0 abc_fine:2048:4:0 blockout_k10:1024:1:0 \
   > $O/ab_matrix.jsonl 2> $O/ab_matrix.err; cat $O/ab_matrix.jsonl | cut -c1-150
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; python - <<'PY'
import json
d=json.load(open('gpurun_out/r04_s9/bench_default.json'))
print('value',round(d['value']/1e6,2),'ms',round(d['ms_per_step'],4),'frac',round(d['roofline']['frac'],4))
for k,v in d.get('extra',{}).items():
    if 'value' in v: print(k, round(v['value']/1e6,2), round(v.get('roofline_frac',0),4))
print('grouped', d.get('grouped_stepping',{}).get('instances'))
print('vecenv', d['extra'].get('vecenv_step'))
print('cpu', d['cpu_baseline']['value'], d['cpu_baseline']['cores'])
PY
