O=gpurun_out/r2x; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.txt
tail -4 $O/pytest_gpu.txt
for wl in blockout general abc_fine blockout_r8 cube blockout_k10; do
timeout 600 python bench.py --no-cpu-baseline --no-extra --workload $wl > $O/bench_$wl.json 2>/dev/null
python -c "
import json; d=json.load(open('$O/bench_$wl.json')); print('$wl value', round(d['value']), 'ms', round(d['ms_per_step'],4))"
done
