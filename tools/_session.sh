# Tooling: the verification pass behind profiles/r02/final (run on the GPU box: gpurun -- 'bash tools/_session.sh').
O=gpurun_out/final; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.txt
tail -3 $O/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 900 python bench.py > $O/bench_default.json 2>$O/bench_default.err; tail -c 600 $O/bench_default.json
for wl in general abc_fine blockout_r8 cube blockout_k10; do
timeout 600 python bench.py --no-cpu-baseline --no-extra --workload $wl > $O/bench_$wl.json 2>/dev/null
done
bash tools/gpu_profile.sh final/prof blockout 16384 2>&1 | tail -3      # then: python tools/collect_pmc.py gpurun_out/final/prof blockout 16384 <rev>
timeout 300 python tools/vecenv_throughput.py > $O/vecenv.txt 2>&1
timeout 300 python tools/phase_profile.py > $O/phase_blockout.json 2>/dev/null
