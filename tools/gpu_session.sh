#!/bin/bash
# Tooling: one verification pass on the GPU box:  gpurun -- 'bash tools/gpu_session.sh <tag> [stages]'
# stages (default "test smoke bench"): test smoke bench workloads vecenv phase prof pmc_<workload>:<bins>
O=gpurun_out/${1:-session}; shift; STAGES=${*:-test smoke bench}
mkdir -p $O
for st in $STAGES; do
case $st in
test)   timeout 1700 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.txt; tail -4 $O/pytest_gpu.txt ;;
smoke)  python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 ;;
bench)  timeout 900 python bench.py > $O/bench_default.json 2>$O/bench_default.err; tail -c 1500 $O/bench_default.json ;;
quick)  timeout 600 python bench.py --no-cpu-baseline --no-extra --min-seconds 0.5 > $O/bench_quick.json 2>$O/bench_quick.err; tail -c 400 $O/bench_quick.json ;;
workloads) for wl in general abc_fine blockout_r8 cube blockout_k10; do
          timeout 600 python bench.py --no-cpu-baseline --no-extra --workload $wl > $O/bench_$wl.json 2>/dev/null; python -c "import json;d=json.load(open('$O/bench_$wl.json'));print('$wl',round(d['value']/1e6,2),'M',round(d['ms_per_step'],4),'ms')"; done ;;
vecenv) timeout 300 python tools/vecenv_throughput.py > $O/vecenv.txt 2>&1; tail -1 $O/vecenv.txt ;;
phase)  for wl in blockout general abc_fine cube; do timeout 300 python tools/phase_profile.py --workload $wl > $O/phase_$wl.json 2>/dev/null; head -c 600 $O/phase_$wl.json; echo; done ;;
prof)   bash tools/gpu_profile.sh ${O#gpurun_out/}/prof blockout 16384 2>&1 | tail -3 ;;
pmc_*)  a=${st#pmc_}; bash tools/gpu_profile.sh ${O#gpurun_out/}/prof_${a%%:*} ${a%%:*} ${a##*:} 2>&1 | tail -3 ;;
esac
done
