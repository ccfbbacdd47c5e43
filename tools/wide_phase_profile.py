#!/usr/bin/env python
"""Tooling: where do the cycles of irbpp_wide_kernel go (32 x 32 action grid)?  Phase stamps per bin, mean over bins."""
import json, sys
import numpy as np, torch
sys.path.insert(0, '.')
import irbpp_amd  # noqa
from irbpp_amd import synthetic
from irbpp_amd.vec_env import GpuPackingEnv
out = {}
for name, sh in (("blockout_r4", synthetic.blockout_shapes(n_shapes=64, n_rot=4, seed=0)),
                 ("general_r4", synthetic.general_shapes(n_shapes=64, n_rot=4, fmin=4, fmax=14, seed=3))):
    n = 256
    env = GpuPackingEnv(sh, synthetic.make_sequences(sh.n_shapes, 2000, 160, seed=1), n, device="cuda:0", resolutionA=0.01, resolutionH=0.01)
    act = torch.zeros(n, dtype=torch.int32, device="cuda:0")
    env.set_auto_policy(act)
    obs = env.reset()
    for t in range(60):
        obs = env.step(act)[0]
    cyc = env.enable_phase_cycles(True)
    acc = []
    for t in range(6):
        obs = env.step(act)[0]
        torch.cuda.synchronize()
        c = cyc.cpu().numpy()
        acc.append(np.concatenate([np.diff(c[:, :5], axis=1), c[:, 5:8]], axis=1))
    d = np.concatenate(acc)
    ncand = (obs[:, :2500].reshape(n, 500, 5)[:, :, 4] == 1).sum(1).float().mean().item()
    out[name] = dict(zip(["bookkeeping+tile", "overlap", "contours", "emit", "of_contours_images", "of_contours_trace", "images"], [round(float(x)) for x in d.mean(0)]))
    out[name]["candidate_rows"] = ncand
    env.close()
print(json.dumps(out))
