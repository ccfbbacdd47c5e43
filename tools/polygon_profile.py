#!/usr/bin/env python
"""Tooling: account of the polygon kernel's waves (cycles loading the first round record, in the hop rounds, the
Douglas-Peucker levels, the rank / convexity part and the clean-up tail; rounds, levels and clean-ups per wave) from the rows
the kernel writes in the -DIRBPP_AB_POLY_ACCOUNT build (tools/build_variant.sh polyacct -DIRBPP_AB_POLY_ACCOUNT;
IRBPP_LIBRARY=irbpp_amd/libirbpp_var_polyacct.so) when irbpp_debug_phase_cycles is on."""
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
    sel = c[c[:, 11] > 0]
    rows.append(sel[:, 11:16].copy())
    w0, w1 = sel[:, 10], sel[:, 9]                       # 100 MHz wall clock (10 ns ticks), the same for all XCDs
    b = (w0 - w0.min()) * 10
    starts = {"first_start_to_last_end_ns": int((w1.max() - w0.min()) * 10), "start_ns": {"p50": int(np.percentile(b, 50)), "p90": int(np.percentile(b, 90)),
              "p99": int(np.percentile(b, 99)), "max": int(b.max())}, "end_ns_p50": int(np.percentile((w1 - w0.min()) * 10, 50)),
              "waves_running_at_ns": {str(t): int(((w0 - w0.min()) * 10 <= t).sum() - ((w1 - w0.min()) * 10 <= t).sum()) for t in (1000, 2000, 4000, 6000, 8000, 10000, 12000, 14000, 16000)}}
r = np.concatenate(rows)
M = (1 << 32) - 1
total, load = r[:, 0], r[:, 1] & M
hops, dp, rank, tail = r[:, 2] & M, r[:, 2] >> 32, r[:, 3] & M, r[:, 3] >> 32
rounds, levels, cleanups = r[:, 4] & 255, (r[:, 4] >> 8) & 0xFFFF, r[:, 4] >> 24
q = lambda v: {"mean": round(float(v.mean()), 1), "p50": float(np.percentile(v, 50)), "p99": float(np.percentile(v, 99)), "max": float(v.max())}
top = np.argsort(-total)[:8]
print(json.dumps({"slowest_waves": [{"cycles": int(total[i]), "first_load": int(load[i]), "hops": int(hops[i]), "dp": int(dp[i]), "rank": int(rank[i]),
                                     "tail": int(tail[i]), "rounds": int(rounds[i]), "levels": int(levels[i]), "cleanups": int(cleanups[i])} for i in top]}))
busy = rounds > 0
print(json.dumps({"starts_last_step": starts}))
print(json.dumps({"workload": a.workload, "bins": a.bins, "waves_sampled_per_step": int(len(r) / a.steps), "waves_with_rounds": int(busy.sum() / a.steps),
                  "total_cycles": q(total[busy]), "first_load": q(load[busy]), "hops": q(hops[busy]), "dp": q(dp[busy]), "rank": q(rank[busy]), "tail": q(tail[busy]),
                  "rounds": q(rounds[busy]), "levels_per_wave": q(levels[busy]), "cleanups": q(cleanups[busy]),
                  "dp_cycles_per_level": round(float(dp[busy].sum() / max(1, levels[busy].sum())), 1),
                  "cycles_per_round": round(float(total[busy].sum() / rounds[busy].sum()), 1)}))
