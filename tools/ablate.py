#!/usr/bin/env python
"""Tooling: what does each phase of the transition kernel cost at full chip load?

Uses the tooling build of the library (irbpp_amd.build.build(ablate=True), -DIRBPP_ABLATE), in which
IRBPP_DEBUG_REPEAT makes a phase run twice.  Every repeated phase is idempotent, so trajectories and
results are unchanged; the extra kernel time of a launch is the price of that phase under exactly the
contention the real launch has (unlike in-kernel cycle stamps, which measure a workgroup's wall time
while five other workgroups share its CU).

    python tools/ablate.py [--workload blockout] [--bins 4096]

Prints one JSON line: kernel ms per launch for each setting and the per-phase deltas."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from irbpp_amd import build as B  # noqa: E402

os.environ["IRBPP_LIBRARY"] = B.build(ablate=True)

import torch  # noqa: E402
from bench import make_workload  # noqa: E402
from irbpp_amd.vec_env import GpuPackingEnv  # noqa: E402

PHASES = {"trace": 1, "douglas_peucker": 2, "overlap_loops": 4, "emit_rows": 8, "contour_stage": 16}

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="blockout")
ap.add_argument("--bins", type=int, default=4096)
ap.add_argument("--prefill", type=int, default=150)
ap.add_argument("--steps", type=int, default=150)
ap.add_argument("--rounds", type=int, default=2)
a = ap.parse_args()

shapes, seqs, kw = make_workload(a.workload)
k = int(kw.get("bufferSize", 1))


def measure(bits):
    os.environ["IRBPP_DEBUG_REPEAT"] = str(bits)
    env = GpuPackingEnv(shapes, seqs, a.bins, device="cuda:0", **kw)
    obs = env.reset()
    slot0 = torch.zeros((a.bins,), dtype=torch.int32, device="cuda:0")

    def step(o):
        if k > 1:
            o = env.get_action_candidates(slot0)
        return env.step(env.policy_minz(o))[0]

    for _ in range(a.prefill):
        obs = step(obs)
    env.enable_kernel_timing(a.steps * (2 if k > 1 else 1))
    for _ in range(a.steps):
        obs = step(obs)
    torch.cuda.synchronize()
    ms = float(env.kernel_times_ms().mean()) * (2 if k > 1 else 1)
    env.check_device_error()
    chk = float(obs.double().sum().item())          # same trajectory in every setting
    env.close()
    return ms, chk


res = {}
for rnd in range(a.rounds):                         # interleaved rounds: drift shows up as disagreement
    for name, bits in [("base", 0)] + list(PHASES.items()):
        ms, chk = measure(bits)
        res.setdefault(name, []).append(ms)
        res.setdefault("_chk", []).append(chk)
assert len(set(res.pop("_chk"))) == 1, "a repeated phase changed the results"
base = min(res["base"])
out = {"workload": a.workload, "bins": a.bins, "kernel_ms": {n: min(v) for n, v in res.items()},
       "all_rounds_ms": res,
       "phase_cost_ms": {n: min(res[n]) - base for n in PHASES},
       "phase_share": {n: (min(res[n]) - base) / base for n in PHASES}}
print(json.dumps(out))
