"""Batched evaluation in the shape of the reference's ``tools.test`` (tools.py:303-358) and, for buffered
environments, ``tools.test_hierachical`` (tools.py:361-431: order action -> get_action_candidates -> location action
per placement).

The reference evaluates one environment, episode after episode, over the trajectories of
``test_sequence.pt`` (LoadItemCreator, IRcreator.py:74-103: episode e reads trajectory e+1) and
saves ``env.packed`` of every episode to ``trajs.npy`` (tools.py:339-340).  Here the episodes run
side by side: bin g plays trajectory ``traj_start + g`` once, every finished bin is frozen, and the
result carries the same statistics (avg/var of reward sum, length, ratio) plus the placement
records in the reference's row format ``[item id, name, positionFLB, quaternion xyzw]``
(binPhy.py:296).
"""
from __future__ import annotations

from typing import Callable, Dict, Optional

import numpy as np
import torch

from .synthetic import ROT_DEGREES
from .vec_env import GpuPackingEnv


def rotation_quaternion_xyzw(rot_idx: int) -> np.ndarray:
    """The quaternion binPhy.py:77-78 stores for z-rotation ``rot_idx`` (mat2quat, saved as xyzw)."""
    half = np.deg2rad(ROT_DEGREES[rot_idx]) / 2.0
    w, z = np.cos(half), np.sin(half)
    if w < 0:
        w, z = -w, -z
    return np.array([0.0, 0.0, z, w])


def evaluate(shapes, sequences, n_episodes: int, *, policy: Optional[Callable] = None,
             order_policy: Optional[Callable] = None, device="cuda:0",
             names: Optional[Dict[int, str]] = None, traj_start: int = 1, max_steps: int = 4096,
             log_capacity: int = 256, save: Optional[str] = None, **env_kw):
    """Run ``n_episodes`` evaluation episodes, one per bin.

    ``policy(env, loc_obs) -> int32[N] device tensor`` picks the location actions; default = the scripted MINZ
    policy kernel.  With ``bufferSize`` > 1 every placement follows tools.py:379-388: ``order_policy(env, order_obs)
    -> int32[N]`` picks the buffer slot (orderDQN.act; default slot 0), ``get_action_candidates`` builds the location
    observation of that item (binPhy.py:161-169), ``policy`` acts on it, ``step`` places the item and refills the
    queue (update_item_queue + generate_item, binPhy.py:324-325).
    Returns a dict with the statistics ``tools.test`` prints and ``trajs``: a list
    over episodes of ``env.packed`` (binPhy.py:296), i.e. one row ``[item_id, name, positionFLB(3),
    quaternion_xyzw(4)]`` per placement INCLUDING the refused one that ended the episode (the reference
    appends before it looks at ``success``).  ``save``: also write them as ``tools.test`` does
    (``np.save(.../trajs.npy, all_episodes)``, tools.py:339-340,354): an object array, one entry per episode.
    """
    env = GpuPackingEnv(shapes, sequences, n_episodes, device=device, traj_start=traj_start,
                        global_bins=n_episodes, **env_kw)
    meta, logz = env.enable_placement_log(log_capacity)
    res_a = env_kw.get("resolutionA", 0.02)
    scale = np.array([100.0, 100.0, 100.0])
    bin_z = float(np.round(np.asarray(env_kw.get("bin_dimension", (0.32, 0.32, 0.30)), dtype=np.float64), 6)[2])
    obs = env.reset()
    n = n_episodes
    finished = np.zeros(n, dtype=bool)
    ratio = np.zeros(n)
    reward_sum = np.zeros(n)
    length = np.zeros(n, dtype=np.int64)
    trajs = [None] * n
    pick = policy if policy is not None else (lambda e, o: e.policy_minz(o))
    slot0 = torch.zeros((n,), dtype=torch.int32, device=env.device)
    pick_order = order_policy if order_policy is not None else (lambda e, o: slot0)
    for _ in range(max_steps):
        if env.K > 1:                                     # tools.py:379-388
            loc = env.get_action_candidates(pick_order(env, obs).to(torch.int32))
            obs, _, _ = env.step(pick(env, loc))
        else:
            obs, _, _ = env.step(pick(env, obs))
        h = env.step_info_host()
        newly = h["done"] & ~finished
        if newly.any():
            idx = np.nonzero(newly)[0]
            m = meta[idx].cpu().numpy().astype(np.uint32)
            z = logz[idx].cpu().numpy()
            for row, b in enumerate(idx):
                k = int(h["counter"][b])
                ratio[b], reward_sum[b], length[b] = h["ratio"][b], h["ep_reward"][b], h["ep_len"][b]
                ep = []
                for i in range(min(k + 1, log_capacity)):        # k accepted placements + the refused one
                    w = int(m[row, i])
                    item, rot, lx, ly = w & 0xFFFF, (w >> 16) & 15, (w >> 20) & 15, (w >> 24) & 15
                    if item == 0xFFFF:                            # trajectory exhausted: no item to place (None)
                        break
                    flb = np.round((lx * res_a, ly * res_a, bin_z), decimals=6) * scale     # addObject (Interface.py:201)
                    flb[2] = z[row, i] * scale[2]                                           # adjustHeight (Interface.py:185-187)
                    ep.append([item, names[item] if names else "%d.obj" % item, flb / scale, rotation_quaternion_xyzw(rot)])
                trajs[b] = ep
            finished |= newly
        if finished.all():
            break
    env.check_device_error()
    env.close()
    done = finished
    if save is not None:
        save_trajs(save, [trajs[b] for b in range(n) if done[b]])
    return {
        "episodes": int(done.sum()), "unfinished": int((~done).sum()),
        "avg_reward": float(reward_sum[done].mean()), "var_reward": float(reward_sum[done].var()),
        "avg_length": float(length[done].mean()), "var_length": float(length[done].var()),
        "mean_ratio": float(ratio[done].mean()), "var_ratio": float(ratio[done].var()),
        "ratio": ratio, "reward_sum": reward_sum, "length": length, "trajs": trajs,
    }


def save_trajs(path: str, all_episodes) -> None:
    """``np.save(path, all_episodes)`` as tools.test calls it (tools.py:339-340,354) under the reference's numpy
    1.21: a list of episodes of different lengths becomes a 1-D object array of lists; current numpy refuses that
    implicit conversion, so the container is built explicitly.  ``np.load(path, allow_pickle=True)`` reads it."""
    import os
    arr = np.empty(len(all_episodes), dtype=object)
    for i, ep in enumerate(all_episodes):
        arr[i] = ep
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    np.save(path, arr, allow_pickle=True)
