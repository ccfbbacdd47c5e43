#!/usr/bin/env python
"""What is bit-parity of the shotInfo tables with trimesh worth?  (the tables are ray-cast
with trimesh at start-up, tools.py:98-135, and trimesh is not in the image, so oracle/shot.py and irbpp_shot_item are
unpinned against it: last-ulp differences of the table heights are plausible.)

Every masked-in entry of heightMapT / heightMapB of a bench workload is moved by +-1 ulp (seeded, independently), then the
plain-C oracle plays the same trajectories with the original and the perturbed tables under the ORIGINAL run's actions.
An episode "diverges" at the first step whose float32 observation, reward or done flag differs.  Lattice data sits on
exact multiples of its cube edge, so a one-ulp move flips `z // 0.01` (cvTools.py:78) and `round(z + e - 0.30, 6) <= 0`
(space.py:120) easily; irregular heights almost never do.

    python tests/analysis/table_ulp_sensitivity.py [bins] [steps]
"""
import copy
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from oracle.c_oracle import COracleVecEnv  # noqa: E402

S = 500


def minz(o):
    c = o[:5 * S].reshape(S, 5)
    v = c[:, 4] == 1
    return int(np.argmin(np.where(v, c[:, 3], np.inf))) if v.any() else 0


def perturbed(shapes, seed, ulps=1):
    rng = np.random.RandomState(seed)
    sh = copy.deepcopy(shapes)
    moved = 0
    for per_rot in sh.tables:
        for r, (T, B, mH, mB) in enumerate(per_rot):
            T, B = T.copy(), B.copy()
            for arr, mask in ((T, mH), (B, mB)):
                idx = np.nonzero(mask)
                step = rng.choice([-1.0, 1.0], size=len(idx[0]))
                for _ in range(ulps):
                    arr[idx] = np.nextafter(arr[idx], arr[idx] + step)
                moved += len(idx[0])
            per_rot[r] = (T, B, mH, mB)
    return sh, moved


def run(workload, bins, steps, seed):
    shapes, seqs, kw = bench.make_workload(workload)
    pert, moved = perturbed(shapes, seed)
    # the raw games (no auto-reset): both sides are reset together, exactly once per episode of the original, so that they
    # always play the same trajectory
    a = COracleVecEnv(bins, shapes, seqs, **kw).envs
    b = COracleVecEnv(bins, pert, seqs, **kw).envs
    oa = [g.reset() for g in a]
    ob = [g.reset() for g in b]
    episodes = diverged = same_steps = 0
    first = []
    for i in range(bins):
        alive, b_over, age = True, False, 0
        for t in range(steps):
            act = minz(oa[i])
            oa[i], ra, da, _ = a[i].step(act)
            if not b_over:
                ob[i], rb, db, _ = b[i].step(act)
                b_over = db
                eq = bool((oa[i].astype(np.float32) == ob[i].astype(np.float32)).all()) and ra == rb and da == db
            else:
                eq = False
            age += 1
            if alive and not eq:
                diverged += 1
                first.append(age)
            same_steps += int(alive and eq)
            alive = alive and eq
            if da:
                episodes += 1
                oa[i], ob[i] = a[i].reset(), b[i].reset()
                alive, b_over, age = True, False, 0
    return {"workload": workload, "bins": bins, "steps_per_bin": steps, "table_entries_moved": moved, "episodes": episodes,
            "episodes_diverged": diverged, "identical_bin_steps": same_steps, "bin_steps": bins * steps,
            "median_steps_to_divergence": float(np.median(first)) if first else None}


if __name__ == "__main__":
    bins = int(sys.argv[1]) if len(sys.argv) > 1 else 48
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 150
    for wl in ("blockout", "cube", "general"):
        print(json.dumps(run(wl, bins, steps, seed=7)), flush=True)
