#!/bin/bash
# round-5 session 3: instruction counters per kernel on the specialised builds (blockout, 8192 bins) + phase cycle stamps; retest the 4 fixed tests
O=gpurun_out/r05_s3; mkdir -p $O
timeout 300 python -m pytest tests -m gpu -q -k "reference_golden or both_overlap_paths or num_groups_zero or listed_reset" > $O/pytest_sel.txt 2>&1; tail -3 $O/pytest_sel.txt
R=$PWD; cd /tmp; export TMPDIR=/tmp
P="python $R/bench.py --workload blockout --no-cpu-baseline --no-extra --bins 8192 --steps 60 --warmup 10 --prefill 150 --min-seconds 0"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INSTS_VALU --output-format csv -d $R/$O/sq -o r05 -- $P > /dev/null 2> $R/$O/sq.err
rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES --output-format csv -d $R/$O/sq2 -o r05 -- $P > /dev/null 2> $R/$O/sq2.err
cd $R
python - $O <<'PY'
import csv, collections, glob, json, os, sys
out = sys.argv[1]
for sub in ("sq", "sq2"):
    for path in glob.glob(os.path.join(out, sub, "**", "*counter_collection.csv"), recursive=True):
        d = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(path)):
            d[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        summ = {k: {c: round(sum(v[len(v) // 2:]) / max(1, len(v[len(v) // 2:])), 1) for c, v in cs.items()} for k, cs in d.items() if k.startswith("irbpp")}
        json.dump(summ, open(os.path.join(out, sub + "_summary.json"), "w"), indent=1)
        os.remove(path)
        print(json.dumps(summ))
PY
timeout 120 python tools/phase_profile.py --workload blockout --bins 8192 > $O/phase_blockout.json 2>/dev/null; head -c 1500 $O/phase_blockout.json
