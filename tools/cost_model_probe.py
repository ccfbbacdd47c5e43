#!/usr/bin/env python
"""Tooling: what predicts a bin's cycle count in the next transition launch?  Dumps, for a run of
consecutive steps, per bin: phase cycles, the item observed, border and candidate counts."""
import argparse
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
ap.add_argument("--steps", type=int, default=12)
ap.add_argument("--out", default="gpurun_out/cost_probe.npz")
a = ap.parse_args()

shapes, seqs, kw = make_workload(a.workload)
env = GpuPackingEnv(shapes, seqs, a.bins, device="cuda:0", **kw)
obs = env.reset()
for _ in range(a.warm):
    obs, _, _ = env.step(env.policy_minz(obs))
cyc = env.enable_phase_cycles(True)
S = 500
rec = {k: [] for k in ("cyc", "item", "ncand", "hm_max", "hm_rough", "done")}
for _ in range(a.steps):
    obs, _, done = env.step(env.policy_minz(obs))
    torch.cuda.synchronize()
    rec["cyc"].append(cyc.cpu().numpy().copy())
    rec["item"].append(obs[:, 5 * S].cpu().numpy().astype(np.int32))
    rec["ncand"].append((obs[:, :5 * S].reshape(a.bins, S, 5)[:, :, 4] == 1).sum(1).cpu().numpy())
    hm = obs[:, 5 * S + 9:].reshape(a.bins, 32, 32) if obs.shape[1] - 5 * S - 9 == 1024 else None
    if hm is not None:
        rec["hm_max"].append(hm.amax((1, 2)).cpu().numpy())
        rough = (hm[:, 1:, :] != hm[:, :-1, :]).sum((1, 2)) + (hm[:, :, 1:] != hm[:, :, :-1]).sum((1, 2))
        rec["hm_rough"].append(rough.cpu().numpy())
    rec["done"].append(done.cpu().numpy())
os.makedirs(os.path.dirname(a.out), exist_ok=True)
np.savez_compressed(a.out, extents=shapes.extents, **{k: np.array(v) for k, v in rec.items() if v})
print("saved", a.out)
