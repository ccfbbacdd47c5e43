#!/usr/bin/env python
"""Tooling: account of the trace kernel's waves (split pipeline): cycles staging / tracing / approximating,
outer-loop iterations, Douglas-Peucker rounds and candidates per wave, from the rows the kernel writes when
irbpp_debug_phase_cycles is on."""
import argparse, json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import make_workload
from irbpp_amd.vec_env import GpuPackingEnv
ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="blockout"); ap.add_argument("--bins", type=int, default=4096)
ap.add_argument("--warm", type=int, default=300); ap.add_argument("--steps", type=int, default=6)
a = ap.parse_args()
shapes, seqs, kw = make_workload(a.workload)
env = GpuPackingEnv(shapes, seqs, a.bins, device="cuda:0", **kw)
obs = env.reset()
for _ in range(a.warm):
    obs, _, _ = env.step(env.policy_minz(obs))
cyc = env.enable_phase_cycles(True)
rows = []
for _ in range(a.steps):
    cyc.zero_()
    obs, _, _ = env.step(env.policy_minz(obs))
    torch.cuda.synchronize()
    c = cyc.cpu().numpy()
    c = c[c[:, 11] > 0]
    rows.append(c[:, 11:16].copy())
r = np.concatenate(rows)
outer, dp, cand = np.maximum(r[:, 4] & 0xFFFFF, 1), (r[:, 4] >> 20) & 0xFFFFF, r[:, 4] >> 40
q = lambda v: {"mean": float(v.mean()), "p50": float(np.percentile(v, 50)), "p99": float(np.percentile(v, 99)), "max": float(v.max())}
top = np.argsort(-r[:, 0])[:6]
print(json.dumps({"slowest_waves": [{"cycles": int(r[i, 0]), "staging": int(r[i, 1]), "trace": int(r[i, 2]), "dp": int(r[i, 3]),
                                     "outer": int(outer[i]), "dp_rounds": int(dp[i]), "candidates": int(cand[i])} for i in top]}))
print(json.dumps({"waves": int(len(r) / a.steps), "total_cycles": q(r[:, 0]), "staging_cycles": q(r[:, 1]), "trace_cycles": q(r[:, 2]),
                  "dp_cycles": q(r[:, 3]), "outer_iterations": q(outer), "dp_rounds": q(dp), "candidates": q(cand),
                  "trace_cycles_per_outer": float(r[:, 2].sum() / outer.sum()), "dp_cycles_per_round": float(r[:, 3].sum() / max(1, dp.sum()))}))
