#!/bin/bash
# Tooling: round-4 session 43: irbpp_amd.use_hardware_queues(8) called from Python before the first use of the GPU: does the runtime pick it up?
O=gpurun_out/r04_s43; mkdir -p $O
unset GPU_MAX_HW_QUEUES
for i in 1 2; do
timeout 120 python - <<'PY' 2>/dev/null | tee -a gpurun_out/r04_s43/queues.txt
import sys, runpy, json, io, contextlib
import torch
import irbpp_amd
irbpp_amd.use_hardware_queues(8)
sys.argv = ["ab_matrix.py", "--repeat", "1", "blockout:4096:4:0", "blockout:4096:1:0"]
buf = io.StringIO()
with contextlib.redirect_stdout(buf):
    runpy.run_path("tools/ab_matrix.py", run_name="__main__")
for l in buf.getvalue().splitlines():
    j = json.loads(l); print("helper(8)", j["spec"], j["Msteps_per_s"])
PY
done
