#!/bin/bash
# Tooling: the verification + profile pass behind profiles/r06/final (gpurun -- 'bash profiles/r06/sessions/_final.sh')
O=gpurun_out/final; rm -rf $O; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.txt; tail -3 $O/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 900 python bench.py > $O/bench_default.json 2>$O/bench_default.err; tail -c 300 $O/bench_default.json; echo
timeout 200 python bench.py --steps 20 --warmup 5 --no-extra --no-cpu-baseline > $O/bench_driver_style.json 2>/dev/null
timeout 400 bash tools/gpu_profile.sh final/prof_blockout blockout 16384 2>&1 | tail -2
timeout 400 bash tools/gpu_profile.sh final/prof_general general 8192 2>&1 | tail -2
timeout 500 bash tools/gpu_profile.sh final/prof_abc_fine abc_fine 8192 2>&1 | tail -2
timeout 200 bash tools/gpu_kernel_stats.sh final cube blockout_k10 blockout_r8 2>&1 | grep -c irbpp
for wl in blockout general abc_fine cube; do timeout 120 python tools/phase_profile.py --workload $wl > $O/phase_$wl.json 2>/dev/null; done
timeout 120 python tools/trace_profile.py --workload blockout > $O/trace_blockout.json 2>/dev/null
timeout 400 python tools/ab_matrix.py --repeat 1 --min-seconds 0.4 blockout_k10:8192:2:0 blockout_k10:4096:2:0 blockout_k10:2048:2:0 blockout_k10:1024:1:0 abc_fine:16384:2:0 abc_fine:8192:2:0 abc_fine:4096:2:0 abc_fine:2048:2:0 abc_fine:2048:1:0 general:4096:1:0 blockout_r8:8192:2:0 blockout_r8:4096:2:0 > $O/scaling_points.jsonl 2>/dev/null
ls $O
