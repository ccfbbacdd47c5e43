#!/usr/bin/env python
"""Generates tests/golden/replay_*.npz by running the REFERENCE's own memory.py
(/root/reference/memory.py: SegmentTree + ReplayMemory, one object per environment, exactly as
main.py:61-63 and trainer.py:184-186 / agent.py:69-75,138-139 use them) on a scripted stream of
transitions.  Runs only in the build container (the reference tree is not on the GPU box); the
fixtures are committed.

    python tests/golden/make_replay_golden.py
"""
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REF)
import memory as ref_memory  # noqa: E402  (the reference's file, imported from where it lies)


def scenario(name, n_envs, capacity, obs_len, n_step, steps, checkpoints, segment, seed):
    rng = np.random.RandomState(seed)
    args = types.SimpleNamespace(distributed=False, device=torch.device("cpu"), discount=0.99, multi_step=n_step,
                                 priority_weight=0.4, priority_exponent=0.5)
    mems = [ref_memory.ReplayMemory(args, capacity, obs_len) for _ in range(n_envs)]

    draws = []
    real_uniform = np.random.uniform
    np.random.uniform = lambda lo, hi: draws.append(float(real_uniform(float(lo), float(hi)))) or draws[-1]

    rec = {"states": [], "actions": [], "rewards": [], "terminals": [], "valid": []}
    out = {}
    np.random.seed(seed + 1)
    for t in range(steps):
        state = rng.uniform(0, 0.3, size=(n_envs, obs_len)).astype(np.float32)
        action = rng.randint(0, 50, size=n_envs)
        reward = rng.uniform(0, 1, size=n_envs).astype(np.float32)
        terminal = rng.rand(n_envs) < 0.15
        valid = rng.rand(n_envs) < 0.9
        for k, v in zip(("states", "actions", "rewards", "terminals", "valid"), (state, action, reward, terminal, valid)):
            rec[k].append(v)
        for i in range(n_envs):                                      # trainer.py:184-186
            if valid[i]:
                mems[i].append(torch.from_numpy(state[i]), torch.tensor([action[i]]), torch.tensor([reward[i]]),
                               bool(terminal[i]))
        if t + 1 in checkpoints:
            c = f"c{t + 1}_"
            for m in mems:
                m.priority_weight = min(m.priority_weight + 0.1, 1)    # trainer.py:195-196
            accepted, batches = [], []
            for m in mems:                                            # agent.py:72-75
                acc = []
                orig = m._get_sample_from_segment

                def logged(seg, i, _orig=orig, _acc=acc):
                    r = _orig(seg, i)
                    _acc.append(draws[-1])                            # the draw that passed memory.py:175
                    return r
                m._get_sample_from_segment = logged
                batches.append(m.sample(segment))
                m._get_sample_from_segment = orig
                accepted.append(acc)
            out[c + "values"] = np.array(accepted, dtype=np.float64)
            out[c + "tree_idxs"] = np.array([b[0] for b in batches], dtype=np.int64)
            out[c + "states"] = torch.cat([b[1] for b in batches]).numpy()
            out[c + "actions"] = torch.cat([b[2] for b in batches]).numpy().reshape(-1)
            out[c + "returns"] = torch.cat([b[3] for b in batches]).numpy()
            out[c + "next_states"] = torch.cat([b[4] for b in batches]).numpy()
            out[c + "nonterminals"] = torch.cat([b[5] for b in batches]).numpy().reshape(-1)
            out[c + "weights"] = torch.cat([b[6] for b in batches]).numpy()
            out[c + "tree_before_update"] = np.stack([m.transitions.sum_tree.numpy().copy() for m in mems])
            loss = rng.uniform(0.01, 3.0, size=(n_envs, segment)).astype(np.float32)
            for i, m in enumerate(mems):                              # agent.py:138-139
                m.update_priorities(batches[i][0], torch.from_numpy(loss[i]))
            out[c + "loss"] = loss
            out[c + "tree_after_update"] = np.stack([m.transitions.sum_tree.numpy().copy() for m in mems])
            out[c + "max"] = np.array([float(m.transitions.max) for m in mems], dtype=np.float32)
            out[c + "index"] = np.array([m.transitions.index.value for m in mems])
            out[c + "full"] = np.array([bool(m.transitions.full.value) for m in mems])
    np.random.uniform = real_uniform
    out.update({k: np.array(v) for k, v in rec.items()})
    out["meta"] = np.array([n_envs, capacity, obs_len, n_step, steps, segment])
    out["checkpoints"] = np.array(sorted(checkpoints))
    path = os.path.join(HERE, f"replay_{name}.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, {k: v.shape for k, v in out.items() if k.startswith("c")and "tree_idxs" in k})


if __name__ == "__main__":
    scenario("pow2", n_envs=3, capacity=16, obs_len=6, n_step=3, steps=60, checkpoints={14, 30, 60}, segment=4, seed=11)
    scenario("odd", n_envs=4, capacity=11, obs_len=5, n_step=2, steps=45, checkpoints={10, 27, 45}, segment=3, seed=23)
