#!/usr/bin/env python
"""Tooling: placement-steps/s of several (workload, bins, groups, tuning) combinations measured back to back on one box
with bench.py's own timed_run (prefill, warm-up, timed blocks of steps).
    python tools/ab_matrix.py blockout_k10:1024:1:0 blockout_k10:1024:2:128 ... [--min-seconds 0.4] [--lib PATH]
Prints one JSON line per combination."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("specs", nargs="+")
    ap.add_argument("--min-seconds", type=float, default=0.4)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--repeat", type=int, default=1)
    a = ap.parse_args()
    import torch
    import bench
    from irbpp_amd.vec_env import GroupedPackingEnv
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    run_args = argparse.Namespace()
    for spec in a.specs:
        wl, bins, groups, tuning = (spec.split(":") + ["1", "0"])[:4]
        bins, groups, tuning = int(bins), int(groups), int(tuning)
        sh, sq, kw = bench.make_workload(wl)
        if tuning:
            kw["tuning"] = tuning
        k = int(kw.get("bufferSize", 1))
        vals = []
        for _ in range(a.repeat):
            env = GroupedPackingEnv(sh, sq, bins, groups, device=dev, **kw)
            t, n, kms, fin = bench.timed_run(env, run_args, dev, k, lambda: torch.cuda.synchronize(dev), a.steps, 300, 20, a.min_seconds)
            name = env.groups[0].kernel_info()[1]
            env.close()
            vals.append(bins * n / t)
        print(json.dumps({"spec": spec, "workload": wl, "bins": bins, "groups": groups, "tuning": tuning,
                          "Msteps_per_s": [round(v / 1e6, 3) for v in vals], "us_per_step": round(1e6 * bins / max(vals), 2),
                          "kernel_ms": round(kms, 4), "lib": os.environ.get("IRBPP_LIBRARY", "default"), "kernels": name}), flush=True)


if __name__ == "__main__":
    main()
