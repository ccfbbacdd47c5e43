O=gpurun_out/r03e; mkdir -p $O
timeout 1700 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.txt; tail -3 $O/pytest_gpu.txt
run() { name=$1; shift; timeout 600 python bench.py --no-cpu-baseline --no-extra --min-seconds 0.5 "$@" > $O/bench_$name.json 2>/dev/null; python -c "import json;d=json.load(open('$O/bench_$name.json'));print('$name',round(d['value']/1e6,2),'M',round(d['ms_per_step'],4),'ms',d['roofline']['kernel'].split(' ')[0],d['roofline']['lds_bytes_per_workgroup'])"; }
run blockout
run general --workload general
run general_wide --workload general --tuning 2
run cube --workload cube
run abc_fine --workload abc_fine
IRBPP_LIBRARY=irbpp_amd/libirbpp_var_ablate.so IRBPP_LDS_PAD=512 run abc_fine_3wg --workload abc_fine
IRBPP_LIBRARY=irbpp_amd/libirbpp_var_ablate.so IRBPP_LDS_PAD=4096 run general_lds22k --workload general --tuning 2
bash tools/gpu_kernel_stats.sh r03e blockout general
