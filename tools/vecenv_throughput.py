#!/usr/bin/env python
"""Tooling: placement-steps/s through the reference-compatible VecEnv surface, i.e. the way
trainer.py drives it (trainer.py:161-186): device policy -> action.cpu().numpy() -> envs.step() ->
(obs on device, reward CPU tensor, done numpy, infos) with every info dict touched.  This is the
PCIe/host-inclusive rate; bench.py's `value` keeps everything on the device.
    tools/vecenv_throughput.py [bins] [num_groups (0 = groups_for's choice)]
Also prints where the host's time goes per step (policy + action to the host, step_async, step_wait)."""
import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import irbpp_amd  # noqa
from bench import make_workload
from irbpp_amd.vec_env import GpuVecEnv

bins = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
groups = int(sys.argv[2]) if len(sys.argv) > 2 else 1
shapes, seqs, kw = make_workload("blockout")
envs = GpuVecEnv(shapes, seqs, bins, device="cuda:0", num_groups=groups, **kw)
state = envs.reset()
for _ in range(50):
    action = envs.env.policy_minz(state)
    state, reward, done, infos = envs.step(action.cpu().numpy())
def run(K, per_env_loop):
    global state
    finished = 0
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(K):
        mask = state[:, :2500].reshape(-1, 500, 5)[:, :, -1]              # get_mask_from_state (tools.py:298-299)
        action = envs.env.policy_minz(state)
        state, reward, done, infos = envs.step(action.cpu().numpy())
        if per_env_loop:
            for i in range(len(infos)):                                   # the trainer's per-env loop (trainer.py:167-178)
                if done[i] and infos[i]["Valid"]:
                    finished += 1
        else:
            finished += int(done.sum())                                   # the same bookkeeping without per-env Python
    torch.cuda.synchronize()
    return time.perf_counter() - t, finished


def account(K):
    """host seconds per step in the three parts of a synchronous step"""
    global state
    parts = np.zeros(3)
    for _ in range(K):
        t0 = time.perf_counter()
        a = envs.env.policy_minz(state).cpu().numpy()
        t1 = time.perf_counter()
        envs.step_async(a)
        t2 = time.perf_counter()
        state, reward, done, infos = envs.step_wait()
        t3 = time.perf_counter()
        parts += (t1 - t0, t2 - t1, t3 - t2)
    return (parts / K * 1e6).round(1).tolist()


K = 100
dt, finished = run(K, True)
dt2, _ = run(K, False)
print(json.dumps({"bins": bins, "groups": envs.num_groups, "vecenv_steps_per_s": bins * K / dt, "ms_per_step": dt / K * 1e3, "episodes": finished,
                  "without_the_trainers_per_env_python_loop": {"vecenv_steps_per_s": bins * K / dt2, "ms_per_step": dt2 / K * 1e3},
                  "host_us_per_step": dict(zip(("policy_and_action_to_host", "step_async", "step_wait"), account(K)))}))
