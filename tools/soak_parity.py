#!/usr/bin/env python
"""Tooling: the HIP path against the plain-C oracle at launch sizes where the library's large-launch forms are the ones
that run by themselves (apply kernel in front of the transition kernel, wave-per-bin emit, wave-per-bin buffered step, two
groups on two streams) -- every observation, reward and done flag of every bin and step, whole episodes with auto-resets.
(The -m gpu tests force those forms at small sizes through irbpp_config::tuning; this is the same comparison at the sizes
that select them.)     tools/soak_parity.py workload:bins:groups:steps [...]"""
import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import irbpp_amd  # noqa
from bench import make_workload
from irbpp_amd.vec_env import GpuVecEnv
from oracle.c_oracle import COracleVecEnv

for spec in sys.argv[1:]:
    wl, n, groups, steps, tuning = (spec.split(":") + ["0"])[:5]        # (optional fifth field: irbpp_config::tuning of the HIP side)
    n, groups, steps, tuning = int(n), int(groups), int(steps), int(tuning)
    shapes, seqs, kw = make_workload(wl)
    k = int(kw.get("bufferSize", 1))
    seqs = seqs[:2000]
    # (the C-oracle environments of one COracleVecEnv share ONE copy of the tables and trajectories since round 6 -- session 39 of
    # round 5 took 4096 private copies of the 64 x 64 data set's 54 MB, and with them the box -- and step on host threads;
    # tests/test_gpu_large_forms.py runs this comparison as -m gpu tests)
    genv = GpuVecEnv(shapes, seqs, n, device="cuda:0", num_groups=groups, **(dict(kw, tuning=tuning) if tuning else kw))
    genv.candidates_on_device = True
    cenv = COracleVecEnv(n, shapes, seqs, threads=min(16, os.cpu_count() or 1), **kw)
    t0 = time.time()
    gobs = genv.reset()
    ok = np.array_equal(gobs.cpu().numpy(), cenv.reset().astype(np.float32))
    episodes, compared = 0, 1
    for t in range(steps):
        if not ok:
            break
        if k > 1:
            order = (np.arange(n) * 3 + t) % k
            gloc = genv.get_action_candidates(order)
            if genv.num_groups > 1:
                genv.env.synchronize()
            ok = np.array_equal(gloc.cpu().numpy(), cenv.get_action_candidates(order).astype(np.float32))
            act = genv.env.policy_minz(gloc).cpu().numpy()
        else:
            act = genv.env.policy_minz(gobs).cpu().numpy()
        gobs, grew, gdone, _ = genv.step(act)
        cobs, crew, cdone, _ = cenv.step(act)
        ok = ok and np.array_equal(gobs.cpu().numpy(), cobs.astype(np.float32)) and np.array_equal(gdone, cdone) \
            and np.array_equal(grew.numpy()[:, 0], crew.astype(np.float32))
        episodes += int(cdone.sum())
        compared += 1
    genv.env.check_device_error()
    print(json.dumps({"spec": spec, "kernels": genv.env.groups[0].kernel_info()[1] if genv.num_groups > 1 else genv.env.kernel_info()[1],
                      "groups": genv.num_groups, "identical": bool(ok), "observations_compared": compared * n,
                      "episodes_finished": episodes, "seconds": round(time.time() - t0, 1)}), flush=True)
    genv.close()
