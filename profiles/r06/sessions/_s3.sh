#!/bin/bash
# round-6 session 3: irbpp_get_all_possible_observation, the bench's new extras (actor_loop, blockout_r8_8192, cfg2_4096_one_group), refill opt-in again
O=gpurun_out/r06_s3; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_features.py -q -x -k "get_all_possible or trace_launch" > $O/pytest.txt 2>&1; echo "rc=$?" >> $O/pytest.txt; tail -4 $O/pytest.txt
timeout 900 python bench.py --cpu-budget 5 > $O/bench_default.json 2>$O/bench_default.err; python - <<'PY'
import json
d = json.load(open("gpurun_out/r06_s3/bench_default.json"))
print("value", round(d["value"] / 1e6, 2), "M", d["value_definition"], "| cfg2 one group", round(d["value_cfg2_one_group"] / 1e6, 2))
for k, v in d["extra"].items():
    print(k, {kk: (round(vv / 1e6, 2) if isinstance(vv, float) and vv > 1e5 else vv) for kk, vv in v.items() if kk in ("value", "actor_steps_per_s", "with_trainer_per_env_loop", "without_per_env_loop", "device_action_tensor_no_loop", "groups", "bins", "ms_per_actor_step")})
PY
