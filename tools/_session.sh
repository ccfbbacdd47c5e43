R=$PWD; O=$R/gpurun_out/r2e; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --no-cpu-baseline --pipeline-streams 0 --no-extra --steps 100 --warmup 10 --prefill 200"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INSTS_VALU --output-format csv -d $O/sq -o r02 -- $B > $O/b1.json 2> $O/sq.err
rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE SQ_INST_CYCLES_VMEM --output-format csv -d $O/sq2 -o r02 -- $B > $O/b2.json 2> $O/sq2.err
python - $O <<'PY'
import csv, collections, glob, json, os, sys
out = sys.argv[1]
for sub in ("sq", "sq2"):
    for path in glob.glob(os.path.join(out, sub, "**", "*counter_collection.csv"), recursive=True):
        d = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(path)):
            d[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        summ = {k: {c: sum(v[len(v)//2:]) / max(1, len(v[len(v)//2:])) for c, v in cs.items()} for k, cs in d.items() if "env_kernel" in k}
        print(json.dumps(summ))
        os.remove(path)
PY
tail -2 $O/sq.err $O/sq2.err
