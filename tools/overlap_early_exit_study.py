#!/usr/bin/env python
"""Tooling (CPU): what would an early exit buy the cell-list walk of the generic overlap test (irbpp_kernels.hip: gcell_walk)?
A lane's running max only grows, and once np.round(max + ext_z - bin_z, 6) > 0 the action cell is masked out for good
(space.py:120) -- its exact posZ is never read (posZValid = 1e3 there, space.py:124-127).  A wave task (one rotation, up to
three row groups of 64 action cells) could stop walking when every in-range lane is dead.  Steady-state bins of the plain-C
oracle under the bench's scripted policy -> per task: cells walked with a check every P cells / all cells, in list order and
with the list sorted by ascending bottom height (the cells that sit lowest collide first).

    python tools/overlap_early_exit_study.py [workload] [bins] [steps]"""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from bench import make_workload
from oracle.c_oracle import COracleVecEnv

wl = sys.argv[1] if len(sys.argv) > 1 else "general"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 256
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 60
sh, seqs, kw = make_workload(wl)
env = COracleVecEnv(n, sh, seqs[:2000], threads=8, **kw)
obs = env.reset()
S = 500
def minz(obs):
    c = obs[:, :5 * S].reshape(len(obs), S, 5).astype(np.float32); v = c[:, :, 4] == 1
    return np.where(v.any(1), np.argmin(np.where(v, c[:, :, 3], np.inf), 1), 0)
for t in range(steps):
    obs, _, _, _ = env.step(minz(obs))
res_a, res_h = kw["resolutionA"], kw["resolutionH"]
step = int(round(res_a / res_h))
MARGIN = 1e-5
periods = [4, 8, 16, 32]
tot = 0
walked = {("list", p): 0 for p in periods}; walked.update({("sorted", p): 0 for p in periods})
dead_share = []; valid_share = []
for b, e in enumerate(env.envs):
    H = e.heightmap(); item = int(obs[b, 5 * S]); Ax, Ay = e.Ax, e.Ay
    for r in range(sh.n_rot):
        T, B, mH, mB = sh.tables[item][r]
        fx, fy = B.shape
        ext = np.round(sh.extents[item, r], 6)
        ax, ay = int(np.ceil(np.round(ext[0], 6) / res_a)), int(np.ceil(np.round(ext[1], 6) / res_a))
        wx, wy = Ax - ax + 1, Ay - ay + 1
        if wx <= 0 or wy <= 0:
            continue
        ii, jj = np.nonzero(mB > 0)
        bv = B[ii, jj]
        nb = len(ii)
        thr = 0.30 - np.round(ext[2], 6) + MARGIN
        has_out = (mB == 0).any()
        # diff[X, Y, e] = H[X*step + i_e, Y*step + j_e] - B_e over the in-range action cells
        X = np.arange(wx)[:, None, None] * step + ii[None, None, :]
        Y = np.arange(wy)[None, :, None] * step + jj[None, None, :]
        D = H[X, Y] - bv[None, None, :]
        for name, order in (("list", np.arange(nb)), ("sorted", np.argsort(bv, kind="stable"))):
            run = np.maximum.accumulate(D[:, :, order], axis=2)
            if has_out:
                run = np.maximum(run, 0.0)
            dead_at = np.where(run > thr, np.arange(nb)[None, None, :], nb).min(axis=2)        # first list position a lane is dead at
            rpw = 64 // 16 if Ay == 16 else max(1, 64 // Ay)
            ngroups = (wx + rpw - 1) // rpw
            gmax = 3
            for g0 in range(0, ngroups, gmax):
                lanes = dead_at[g0 * rpw:(g0 + gmax) * rpw]
                G = min(gmax, ngroups - g0)
                all_dead = int(lanes.max())                                              # nb if some lane stays alive
                for p in periods:
                    stop = nb if all_dead >= nb else min(nb, ((all_dead + 1 + p - 1) // p) * p)
                    walked[(name, p)] += stop * G
                if name == "list":
                    tot += nb * G
        dead_share.append(float((run[:, :, -1] > thr).mean()))
print(json.dumps({"workload": wl, "bins": n, "steps": steps, "group_cells_walked_full": tot,
                  "walked_share": {f"{k[0]}_check_every_{k[1]}": round(v / tot, 4) for k, v in walked.items()},
                  "masked_out_share_of_in_range_action_cells": round(float(np.mean(dead_share)), 4)}, indent=1))
