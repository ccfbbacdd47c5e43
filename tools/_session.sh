O=gpurun_out/r2p; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.txt
tail -3 $O/pytest_gpu.txt
cd /tmp; export TMPDIR=/tmp
for bpw in 1 2 4; do
IRBPP_TRACE_BPW=$bpw rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/kt$bpw -o r02 -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extra > $GRAFT_REPO_ROOT/$O/bench$bpw.json 2>/dev/null
python $GRAFT_REPO_ROOT/tools/kernel_trace_summary.py $GRAFT_REPO_ROOT/$O/kt$bpw 200 --rm | python -c "
import json,sys; d=json.load(sys.stdin); print($bpw, {k.replace('irbpp_','').replace('_kernel',''):(round(v['avg_us_last'],1),round(v['min_us_last'],1),round(v['max_us_last'],1)) for k,v in d.items() if 'irbpp' in k})"
python -c "
import json; d=json.load(open('$GRAFT_REPO_ROOT/$O/bench$bpw.json')); print('value', d['value'])"
done
