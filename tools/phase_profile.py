#!/usr/bin/env python
"""Tooling: where do the cycles of the transition kernel go?  Runs a workload into steady state,
then switches on the per-bin phase stamps (irbpp_debug_phase_cycles) and prints, per phase, the
mean and the maximum over bins in shader-clock cycles."""
import argparse
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import irbpp_amd  # noqa: E402,F401
from bench import make_workload  # noqa: E402
from irbpp_amd.vec_env import GpuPackingEnv  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="blockout")
ap.add_argument("--bins", type=int, default=4096)
ap.add_argument("--warm", type=int, default=120)
ap.add_argument("--steps", type=int, default=8)
a = ap.parse_args()

shapes, seqs, kw = make_workload(a.workload)
env = GpuPackingEnv(shapes, seqs, a.bins, device="cuda:0", **kw)
obs = env.reset()
for _ in range(a.warm):
    obs, _, _ = env.step(env.policy_minz(obs))
cyc = env.enable_phase_cycles(True)
names = ["apply", "overlap", "contour", "emit"]
acc = []
extra = []
for _ in range(a.steps):
    obs, _, _ = env.step(env.policy_minz(obs))
    torch.cuda.synchronize()
    c = cyc.cpu().numpy()
    acc.append(np.diff(c[:, :5], axis=1))
    extra.append(np.concatenate([c[:, 5:8], c[:, 11:16]], axis=1))
d = np.concatenate(acc)
ncand = (obs[:, :2500].reshape(a.bins, 500, 5)[:, :, 4] == 1).sum(1).float()
out = {"workload": a.workload, "bins": a.bins, 
       "mean_cycles": {n: float(d[:, i].mean()) for i, n in enumerate(names)},
       "p99_cycles": {n: float(np.percentile(d[:, i], 99)) for i, n in enumerate(names)},
       "max_cycles": {n: float(d[:, i].max()) for i, n in enumerate(names)},
       "total_mean": float(d.sum(1).mean()),
       "total_pct": {str(q): float(np.percentile(d.sum(1), q)) for q in (50, 90, 99, 99.9, 100)},
       "mean_candidates": float(ncand.mean())}
ex = np.concatenate(extra)
out["split_pipeline"] = {"transition_kernel_apply": float(d[:, 0].mean()), "transition_kernel_overlap": float(d[:, 1].mean()),
                         "transition_kernel_handover": float(np.concatenate(extra)[:, 0].mean()),
                         "emit_kernel": float(d[:, 3].mean())}
out["contour_detail"] = {"extract_mean": float(ex[:, 0].mean()), "extract_max": float(ex[:, 0].max()),
                         "process_mean": float(ex[:, 1].mean()), "process_max": float(ex[:, 1].max()),
                         "borders_mean": float(ex[:, 2].mean()), "borders_max": float(ex[:, 2].max()),
                         "wave0_trace_mean": float(ex[:, 3].mean()), "wave0_dp_mean": float(ex[:, 4].mean()),
                         "wave0_barrier_wait_mean": float(ex[:, 5].mean()), "redo_mean": float(ex[:, 6].mean()),
                         "redo_borders_mean": float(ex[:, 7].mean())}
# the slowest bins of the emit kernel in the last step: how many candidate rows, which path (tooling)
last = np.diff(c[:, :5], axis=1)[:, 3]
rows = obs[:, :2500].reshape(a.bins, 500, 5)
nrow = (rows[:, :, 4] == 1).sum(1).cpu().numpy()
fb = (rows[:, :, 3] == rows[:, :1, 3]).all(1).cpu().numpy() & (rows[:, 0, 3] > 0.29).cpu().numpy()
top = np.argsort(-last)[:8]
out["slowest_emit_bins"] = [{"cycles": int(last[i]), "rows_V1": int(nrow[i]), "fallback_rows": bool(fb[i])} for i in top]
out["emit_cycles_by_rows"] = {"rows<500": float(last[nrow < 500].mean()), "rows==500": float(last[nrow == 500].mean()) if (nrow == 500).any() else None,
                              "share_rows==500": float((nrow == 500).mean()), "share_fallback": float(fb.mean())}
print(json.dumps(out))
