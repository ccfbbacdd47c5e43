# Tooling: rocprofv3 kernel stats of every bench workload and the PMC passes of "general" (profiles/r02/final/other_workloads).
O=gpurun_out/final_workloads; mkdir -p $O
R=$PWD
for wl in general abc_fine cube blockout_r8 blockout_k10; do
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/kt_$wl -o r02 -- python $R/bench.py --workload $wl --no-cpu-baseline --no-extra --steps 100 --warmup 10 > $R/$O/bench_rocprof_$wl.json 2> $R/$O/kt_$wl.err)
find $O/kt_$wl -name '*kernel_stats.csv' | head -1 | xargs -I{} cp {} $O/kernel_stats_$wl.csv
find $O/kt_$wl -name '*kernel_trace.csv' -delete
done
bash tools/gpu_profile.sh final_workloads/prof_general general 8192 2>&1 | tail -2
