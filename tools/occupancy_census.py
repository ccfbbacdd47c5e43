#!/usr/bin/env python
"""Tooling: how well does one transition launch fill the chip?  Uses the per-bin wall-clock stamps
and HW ids of irbpp_debug_phase_cycles to reconstruct, for one launch, which CU ran which bin and
when: resident workgroups per CU over time, the span first-entry..last-exit against the launch
duration seen by HIP events, and the slot utilisation inside that span."""
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
ap.add_argument("--steps", type=int, default=6)
a = ap.parse_args()

shapes, seqs, kw = make_workload(a.workload)
env = GpuPackingEnv(shapes, seqs, a.bins, device="cuda:0", **kw)
obs = env.reset()
for _ in range(a.warm):
    obs, _, _ = env.step(env.policy_minz(obs))
cyc = env.enable_phase_cycles(True)
rows = []
durs = []
for _ in range(a.steps):
    act = env.policy_minz(obs)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    obs, _, _ = env.step(act)
    e1.record()
    torch.cuda.synchronize()
    c = cyc.cpu().numpy()
    t0, t1 = c[:, 8].astype(np.float64) / 100.0, c[:, 9].astype(np.float64) / 100.0     # us
    hw = c[:, 10]
    cu = ((hw >> 32) & 0xF) * 1024 + ((hw >> 13) & 0x7) * 64 + ((hw >> 12) & 1) * 16 + ((hw >> 8) & 0xF)
    span = t1.max() - t0.min()
    busy = (t1 - t0).sum()
    cus = np.unique(cu)
    # resident workgroups per CU, sampled every 2 us inside the span
    ts = np.arange(t0.min(), t1.max(), 2.0)
    res = ((t0[None, :] <= ts[:, None]) & (t1[None, :] > ts[:, None]))
    per_cu_max = max(int(res[:, cu == k].sum(1).max()) for k in cus[:32])
    shader = (c[:, 4] - c[:, 0]).astype(np.float64)
    durs.append(t1 - t0)
    rows.append({
        "event_us": e0.elapsed_time(e1) * 1e3,
        "span_us": float(span),
        "first_entry_spread_us": float(np.percentile(t0, 1536 * 100.0 / a.bins) - t0.min()) if a.bins > 1536 else None,
        "cus_seen": int(len(cus)),
        "max_resident_per_cu": per_cu_max,
        "mean_resident_total": float(res.sum(1).mean()),
        "slot_utilisation_6_per_cu": float(busy / (span * 6 * len(cus))),
        "mean_bin_us": float((t1 - t0).mean()),
        "max_bin_us": float((t1 - t0).max()),
        "shader_ghz": float((shader / ((t1 - t0) * 1e3)).mean()),
        "bins_per_cu_min_max": [int(np.bincount(np.searchsorted(cus, cu)).min()), int(np.bincount(np.searchsorted(cus, cu)).max())],
        "busy_us_per_cu_min_mean_max": [float(x) for x in (lambda v: (v.min(), v.mean(), v.max()))(
            np.bincount(np.searchsorted(cus, cu), weights=(t1 - t0)))],
        "last_exit_minus_p90_exit_us": float(t1.max() - np.percentile(t1, 90)),
        "entry_us_of_nth_bin": {str(n): float(np.sort(t0)[n - 1] - t0.min())
                                for n in (256, 512, 1024, 1280, 1536, 2048, 3072, a.bins) if n <= a.bins},
        "mean_bin_us_by_entry_decile": [float(x.mean()) for x in np.array_split((t1 - t0)[np.argsort(t0)], 10)],
        "per_cu_resident_histogram_at_30pct": np.bincount(
            np.bincount(np.searchsorted(cus, cu), weights=res[int(len(ts) * 0.3)].astype(np.float64),
                        minlength=len(cus)).astype(np.int64)).tolist(),
        "resident_by_time_decile": [float(x.mean()) for x in np.array_split(res.sum(1), 10)],
    })
rows[-1]["corr_consecutive_step_durations"] = [float(np.corrcoef(durs[i], durs[i + 1])[0, 1]) for i in range(len(durs) - 1)]
print(json.dumps(rows[-1]))
print(json.dumps({k: float(np.mean([r[k] for r in rows])) for k in
                  ("event_us", "span_us", "slot_utilisation_6_per_cu", "mean_resident_total", "shader_ghz")}))
