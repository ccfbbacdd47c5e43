#!/usr/bin/env python
"""Tooling: account of the emit kernel's workgroups (cycles from entry to: bin index known, vertex bits in LDS, candidate rows
enumerated, row values loaded, observation stored, end) from the stamps of the -DIRBPP_AB_EMIT_ACCOUNT build
(tools/build_variant.sh emitacct -DIRBPP_AB_EMIT_ACCOUNT; IRBPP_LIBRARY=irbpp_amd/libirbpp_var_emitacct.so)."""
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
    rows.append(c[c[:, 14] > 0].copy())
r = np.concatenate(rows)
t0 = r[:, 6]
seg = {"bin_index_known": r[:, 7] - t0, "vertex_bits_in_lds": r[:, 3] - r[:, 7], "rows_enumerated": r[:, 11] - r[:, 3], "row_values_loaded": r[:, 12] - r[:, 11],
       "observation_stored": r[:, 13] - r[:, 12], "keys_policy_end": r[:, 14] - r[:, 13], "total": r[:, 14] - t0}
q = lambda v: {"mean": round(float(v.mean()), 1), "p50": float(np.percentile(v, 50)), "p99": float(np.percentile(v, 99)), "max": float(v.max())}
print(json.dumps({"workload": a.workload, "bins": a.bins, "workgroups_per_step": int(len(r) / a.steps), **{k: q(v) for k, v in seg.items()}}))
