O=gpurun_out/r2g; mkdir -p $O
timeout 300 python tools/phase_profile.py --workload general > $O/phase_general.json 2>$O/phase.err; python -c "
import json; d=json.load(open('$O/phase_general.json')); print(d['mean_cycles'], d['p99_cycles'], d['max_cycles']); print(d['slowest_emit_bins']); print(d['emit_cycles_by_rows']); print(d['split_pipeline'])"
tail -3 $O/phase.err
timeout 300 python tools/phase_profile.py --workload blockout > $O/phase_blockout.json 2>$O/phase.err; python -c "
import json; d=json.load(open('$O/phase_blockout.json')); print(d['mean_cycles'], d['p99_cycles'], d['max_cycles']); print(d['slowest_emit_bins']); print(d['emit_cycles_by_rows']); print(d['split_pipeline'])"
