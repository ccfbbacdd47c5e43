#!/bin/bash
# round-5 session 24: dry run of the multi-rank path on the one GPU of the box (two and four ranks sharing the device, gloo): the launch,
# the sharding, the rank agreement on the timed blocks and the final all-reduce as the driver's scaling run will use them (RCCL there)
O=gpurun_out/r05_s24; mkdir -p $O
for n in 2 4; do
timeout 300 python bench.py --gpus $n --backend gloo --no-extra --no-cpu-baseline --bins 2048 --steps 20 --warmup 5 > $O/dry_run_${n}ranks.json 2>$O/dry_run_${n}ranks.err
python -c "
import json; d=json.load(open('$O/dry_run_${n}ranks.json')); print($n, 'ranks', d['n_gpus'], d['value'], d['config']['global_bins'], d['ranks']['devices'])"
done
timeout 300 python bench.py --gpus 2 --backend gloo --config cfg4 --no-extra --no-cpu-baseline --steps 20 --warmup 5 > $O/dry_run_cfg4_2ranks.json 2>/dev/null; python -c "
import json; d=json.load(open('$O/dry_run_cfg4_2ranks.json')); print('cfg4', d['n_gpus'], d['value'], d['scaling'], d['config'])"
