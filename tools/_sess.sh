O=gpurun_out/r03g; mkdir -p $O
run() { name=$1; shift; timeout 600 python bench.py --no-cpu-baseline --no-extra --min-seconds 0.5 "$@" > $O/bench_$name.json 2>/dev/null; python -c "import json;d=json.load(open('$O/bench_$name.json'));print('$name',round(d['value']/1e6,2),'M',round(d['ms_per_step'],4),'ms',d['roofline']['kernel'].split(' ')[0],d['roofline']['lds_bytes_per_workgroup'])"; }
for v in ipt2 ipt2free ipt8free; do
IRBPP_LIBRARY=irbpp_amd/libirbpp_var_$v.so run ${v}_general --workload general
IRBPP_LIBRARY=irbpp_amd/libirbpp_var_$v.so run ${v}_abc_fine --workload abc_fine
done
run cur_abc_fine --workload abc_fine
IRBPP_LIBRARY=irbpp_amd/libirbpp_var_noblkall.so run noblkall_blockout
run cur_blockout
IRBPP_LIBRARY=irbpp_amd/libirbpp_var_noblkall.so run noblkall_blockout_b
run cur_blockout_b
