#!/usr/bin/env python
"""Tooling: throughput of the trainer's acting loop when environment, policy
stand-in and replay memory all stay on the device (irbpp_amd.replay.actor_step), next to the
reference's structure for the same work: per-env Python loop over numpy observations feeding one
replay object per env (trainer.py:167-186), here with the CPU restatement of memory.py."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import irbpp_amd  # noqa: E402,F401
from bench import make_workload  # noqa: E402
from irbpp_amd.replay import VectorReplayMemory, actor_step  # noqa: E402
from irbpp_amd.vec_env import GpuPackingEnv, GpuVecEnv  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="blockout")
ap.add_argument("--bins", type=int, default=4096)
ap.add_argument("--capacity", type=int, default=64, help="transitions per env")
ap.add_argument("--steps", type=int, default=100)
ap.add_argument("--segment", type=int, default=1, help="samples per env and learn() call")
ap.add_argument("--loop-bins", type=int, default=64, help="size of the per-env Python loop comparison (0 = skip)")
a = ap.parse_args()

shapes, seqs, kw = make_workload(a.workload)
env = GpuPackingEnv(shapes, seqs, a.bins, device="cuda:0", **kw)
mem = VectorReplayMemory(a.bins, a.capacity, env.obs_len, device="cuda:0")
policy = lambda s, m: env.policy_minz(s).to(torch.int64)      # noqa: E731  stands for Agent.act
state = env.reset()
for _ in range(a.capacity + 8):                                  # fill the ring, reach steady state
    state, _, _ = actor_step(env, policy, mem, state)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(a.steps):
    state, _, _ = actor_step(env, policy, mem, state)
torch.cuda.synchronize()
t_act = (time.perf_counter() - t0) / a.steps
t0 = time.perf_counter()
reps = 20
for _ in range(reps):
    batch = mem.sample(a.segment)
    mem.update_priorities(batch[0], torch.rand(a.bins * a.segment, device="cuda:0") + 0.1)
torch.cuda.synchronize()
t_learn_io = (time.perf_counter() - t0) / reps
env.check_device_error()
out = {"bins": a.bins, "capacity_per_env": a.capacity, "replay_bytes": int(mem.states.numel() * 4),
       "actor_steps_per_s": a.bins / t_act, "ms_per_actor_step": t_act * 1e3,
       "sample_plus_update_ms": t_learn_io * 1e3, "sampled_transitions": a.bins * a.segment}
env.close()

if a.loop_bins:
    class PerEnvReplay(object):
        """Stand-in with the reference's structure and cost profile (memory.py:55-70,117-121): numpy ring
        + a sum tree walked leaf-to-root in Python, one object per env.  Not a checker, just a clock."""

        def __init__(self, cap, obs_len):
            self.cap, self.i, self.t, self.max = cap, 0, 0, 1.0
            self.tree = np.zeros(2 * cap - 1, dtype=np.float32)
            self.states = np.zeros((cap, obs_len), dtype=np.float32)
            self.meta = np.zeros((cap, 4), dtype=np.float32)

        def append(self, state, action, reward, terminal):
            self.states[self.i] = state
            self.meta[self.i] = (self.t, action, reward, not terminal)
            k = self.i + self.cap - 1
            self.tree[k] = self.max
            while k:
                k = (k - 1) // 2
                self.tree[k] = self.tree[2 * k + 1] + self.tree[2 * k + 2]
            self.i = (self.i + 1) % self.cap
            self.t = 0 if terminal else self.t + 1

    n = a.loop_bins
    venv = GpuVecEnv(shapes, seqs, n, device="cuda:0", **kw)
    mems = [PerEnvReplay(a.capacity, venv.obs_len) for _ in range(n)]
    st = venv.reset()
    t0 = time.perf_counter()
    loops = 30
    for _ in range(loops):
        act = venv.env.policy_minz(st)
        nxt, reward, done, infos = venv.step(act.cpu().numpy())
        host = st.cpu().numpy()
        for i in range(n):                                       # trainer.py:184-186
            if infos[i]["Valid"]:
                mems[i].append(host[i], int(act[i]), float(reward[i]), bool(done[i]))
        st = nxt
    dt = (time.perf_counter() - t0) / loops
    out["per_env_python_loop"] = {"bins": n, "actor_steps_per_s": n / dt, "ms_per_actor_step": dt * 1e3}
print(json.dumps(out))
