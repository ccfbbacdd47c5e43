"""Pin the CPU oracle against vectors produced by the reference's own Python
(tests/golden/make_golden.py, run in the build container where /root/reference exists)."""
import os

import numpy as np
import pytest

from oracle import cvtools
from oracle.packing import PackingGame, SequenceItemCreator
from helpers import (HIER_GOLDENS, ONLINE_GOLDENS, assert_fallback_rows_legal, golden_kwargs, golden_scenario,
                     minz_action)

S = 500


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name + ".npz"))


@pytest.mark.parametrize("name", ONLINE_GOLDENS)
def test_online_episode_matches_reference(golden_dir, name):
    """The small scenarios of round 1 and, from round 5, one recording per BASELINE.json config on the bench's own
    shape sets (bench_*: cfg 2 at R = 4 and R = 8, cfg 3 up to its first tied > S selection, cfg 5 at resolutionH
    0.005) -- all played by the reference's own PackingGame."""
    g = _load(golden_dir, name)
    sh = golden_scenario(name)
    env = PackingGame(sh, g["seq"], selectedAction=S, bufferSize=1, **golden_kwargs(name))
    obs = env.reset()
    np.testing.assert_array_equal(obs, g["obs"][0])
    rewards, fallbacks = [], 0
    for t in range(len(g["act"])):
        np.testing.assert_array_equal(env.space.naiveMask, g["mask"][t])
        np.testing.assert_array_equal(env.space.posZmap, g["posz"][t])
        a = minz_action(obs, S)
        assert a == g["act"][t]
        obs, r, d, info = env.step(a)
        rewards.append(r)
        assert r == g["rew"][t] and d == g["done"][t]
        if d:
            assert info["counter"] == g["counter"][t] and info["ratio"] == g["ratio"][t]
            assert round(sum(rewards), 6) == g["ep_r"][t]
            rewards = []
            obs = env.reset()
        # item vector + heightmap always; the candidate block whenever no unstable argsort was involved
        np.testing.assert_array_equal(obs[5 * S:], g["obs"][t + 1][5 * S:])
        if (obs[:5 * S].reshape(S, 5)[:, 4] == 1).any():
            np.testing.assert_array_equal(obs, g["obs"][t + 1])
        else:   # fallback rows come from np.argsort of an all-equal vector: tie order unspecified
            assert_fallback_rows_legal(obs[:5 * S].reshape(S, 5), sh.n_rot)
            assert_fallback_rows_legal(g["obs"][t + 1][:5 * S].reshape(S, 5), sh.n_rot)      # ... the reference's too
            fallbacks += 1
    assert g["done"].sum() >= 1 and fallbacks >= 1


@pytest.mark.parametrize("name,k", HIER_GOLDENS)
def test_hierarchical_episode_matches_reference(golden_dir, name, k):
    """k = 3 on the small BlockOut set and BASELINE config 4's k = 10 on the bench's BlockOut set."""
    g = _load(golden_dir, name)
    env = PackingGame(golden_scenario(name), g["seq"], selectedAction=S, bufferSize=k)
    order = env.reset()
    np.testing.assert_array_equal(order, g["order_obs"][0])
    for t in range(len(g["act"])):
        loc = env.get_action_candidates(int(g["order_act"][t]))
        if (loc[:5 * S].reshape(S, 5)[:, 4] == 1).any():
            np.testing.assert_array_equal(loc, g["loc_obs"][t])
        a = minz_action(loc, S)
        assert a == g["act"][t]
        order, r, d, info = env.step(a)
        assert r == g["rew"][t] and d == g["done"][t]
        if d:
            assert info["counter"] == g["counter"][t] and info["ratio"] == g["ratio"][t]
            order = env.reset()
        np.testing.assert_array_equal(order, g["order_obs"][t + 1])
    assert g["done"].sum() >= 1


def test_more_than_S_selection_matches_reference(golden_dir):
    """binPhy.py:209-212 with pairwise distinct placement heights (make_golden.py:more_than_s_cases): the S rows and
    their ORDER are the reference's own, not just the oracle's stable tie rule."""
    g = _load(golden_dir, "more_than_s")
    env = PackingGame(golden_scenario("more_than_s"), g["seq"], selectedAction=S, bufferSize=2)
    np.testing.assert_array_equal(env.reset(), g["order_obs0"])
    for t in range(len(g["act"])):
        env.space.heightmapC[:] = g["hm"][t]
        loc = env.get_action_candidates(int(g["order_act"][t]))
        np.testing.assert_array_equal(loc, g["loc_obs"][t])
        assert g["ncand"][t] > S and (np.diff(loc[:5 * S].reshape(S, 5)[:, 3]) > 0).all()
        a = minz_action(loc, S)
        assert a == g["act"][t]
        order, r, d, _ = env.step(a)
        assert r == g["rew"][t] and d == g["done"][t]
        np.testing.assert_array_equal(order, g["order_obs"][t])


def test_cvtools_glue_matches_reference(golden_dir):
    g = _load(golden_dir, "cvtools_cases")
    for i in range(len(g["n_rot"])):
        R = int(g["n_rot"][i])
        c = cvtools.getConvexHullActions(g["posz"][i][:R], g["mask"][i][:R], 0.01)
        n = int(g["cand_len"][i])
        if n == 0:
            assert c is None
        else:
            np.testing.assert_array_equal(c, g["cand"][i][:n])
    assert g["cand_len"].max() > 40


def test_item_queue_matches_reference(golden_dir):
    g = _load(golden_dir, "ircreator_trace")
    c = SequenceItemCreator(g["seqs"], first_traj=1, stride=1)
    it = iter(g["trace"])
    for ep in range(2):
        c.reset()
        for _ in range(8):
            row = next(it)
            assert c.preview(3) == list(row[:3])
            assert c.traj_index == row[4]
            c.update_item_queue(int(row[3]))
            c.generate_item()


def test_tools_test_trajectories_match_reference(golden_dir):
    """tools.test itself (tools.py:303-358, run by make_golden.py with a stub agent): per episode the rows of
    ``env.packed`` -- item, name, front-left-bottom position, quaternion -- including the refused placement that
    ends the episode, and the statistics it returns."""
    from irbpp_amd.evaluate import rotation_quaternion_xyzw
    from irbpp_amd import synthetic
    g = _load(golden_dir, "tools_test")
    sh = synthetic.blockout_shapes(n_shapes=20, n_rot=4, cube=0.06, seed=7)
    env = PackingGame(sh, g["seq"], selectedAction=S)
    row, rsums, lens = 0, [], []
    for ep_len in g["ep_len"]:
        obs = env.reset()
        rsum = steps = 0
        while True:
            had_candidate = bool((obs[:5 * S].reshape(S, 5)[:, 4] == 1).any())
            obs, r, d, info = env.step(minz_action(obs, S))
            rsum += r
            steps += 1
            if d:
                break
        assert len(env.packed) == ep_len == steps                       # the refused placement is recorded too
        for i, (item, rot, lx, ly, height) in enumerate(env.packed):
            assert item == g["ids"][row] and g["names"][row] == "%d.obj" % item
            # With no valid candidate the observation lists cells in the order of an UNSTABLE np.argsort over equal
            # keys (binPhy.py:219), so which cell "action 0" refuses depends on the numpy build: not comparable.
            if i < ep_len - 1 or had_candidate:
                pos = np.round((lx * 0.02, ly * 0.02, 0.30), decimals=6) * 100.0
                pos[2] = height * 100.0
                np.testing.assert_allclose(pos / 100.0, g["pos"][row], rtol=0, atol=1e-12)
                np.testing.assert_allclose(rotation_quaternion_xyzw(rot), g["quat"][row], rtol=0, atol=1e-12)
            row += 1
        rsums.append(rsum)
        lens.append(steps)
    assert row == len(g["ids"])
    assert abs(np.mean(rsums) - float(g["avg_reward"])) < 1e-12 and np.mean(lens) == float(g["avg_length"])


def test_tools_test_hierachical_trajectories_match_reference(golden_dir):
    """tools.test_hierachical itself (tools.py:361-431, run by make_golden.py with two stub agents: order = the slot
    with the smallest item id, location = scripted MINZ) on a k = 3 buffer: ``env.packed`` of every episode and the
    statistics, reproduced by the oracle's PackingGame driven through the same protocol."""
    from irbpp_amd.evaluate import rotation_quaternion_xyzw
    from irbpp_amd import synthetic
    g = _load(golden_dir, "tools_test_hier")
    k = int(g["k"])
    sh = synthetic.blockout_shapes(n_shapes=20, n_rot=4, cube=0.06, seed=7)
    env = PackingGame(sh, g["seq"], selectedAction=S, bufferSize=k)
    row, rsums, lens = 0, [], []
    for ep_len in g["ep_len"]:
        order_obs = env.reset()
        rsum = steps = 0
        while True:
            loc = env.get_action_candidates(int(np.argmin(order_obs[:k])))
            had_candidate = bool((loc[:5 * S].reshape(S, 5)[:, 4] == 1).any())
            order_obs, r, d, info = env.step(minz_action(loc, S))
            rsum += r
            steps += 1
            if d:
                break
        assert len(env.packed) == ep_len == steps
        for i, (item, rot, lx, ly, height) in enumerate(env.packed):
            assert item == g["ids"][row] and g["names"][row] == "%d.obj" % item
            if i < ep_len - 1 or had_candidate:
                pos = np.round((lx * 0.02, ly * 0.02, 0.30), decimals=6) * 100.0
                pos[2] = height * 100.0
                np.testing.assert_allclose(pos / 100.0, g["pos"][row], rtol=0, atol=1e-12)
                np.testing.assert_allclose(rotation_quaternion_xyzw(rot), g["quat"][row], rtol=0, atol=1e-12)
            row += 1
        rsums.append(rsum)
        lens.append(steps)
    assert row == len(g["ids"])
    assert abs(np.mean(rsums) - float(g["avg_reward"])) < 1e-12 and np.mean(lens) == float(g["avg_length"])


HEUR_METHODS = ("MINZ", "DBLF", "FIRSTFIT", "HM")


@pytest.mark.parametrize("tag", ["general", "blockout"])
def test_heuristic_actions_match_reference(golden_dir, tag):
    """Space.get_heuristic_action (space.py:162-218) as the reference's own Space computed it on the states of an
    episode its own PackingGame played (tests/golden/make_golden.py:heuristic_cases): 4 methods x 4 flips per state."""
    g = _load(golden_dir, "heuristic_cases")
    env = PackingGame(golden_scenario("heuristic_" + tag), g[tag + "_seq"], selectedAction=S, bufferSize=1)
    obs = env.reset()
    for t in range(len(g[tag + "_act"])):
        assert env.next_item_ID == g[tag + "_item"][t] and int(env.space.naiveMask.sum()) == g[tag + "_nvalid"][t]
        np.testing.assert_array_equal(env.space.heightmapC, g[tag + "_hm"][t])
        for mi, method in enumerate(HEUR_METHODS):
            for d in range(4):
                got = env.space.get_heuristic_action(method, env.next_item_ID, d)
                assert tuple(int(v) for v in got) == tuple(g[tag + "_heur"][t, mi, d]), (t, method, d)
        a = minz_action(obs, S)
        assert a == g[tag + "_act"][t]
        obs, _, done, _ = env.step(a)
        assert done == g[tag + "_done"][t]
        if done:
            obs = env.reset()
    assert bool(g[tag + "_random_raises"])       # the reference's RANDOM branch raises for every input: nothing to match


@pytest.mark.parametrize("which", ["numpy", "c"])
def test_wide_action_grid_matches_reference(golden_dir, which):
    """resolutionA = 0.01: a 32 x 32 action grid (space.py:19-24), as the reference's own PackingGame played it
    (tests/golden/make_golden.py: online_wide32 on free-form solids, hier_wide32_k3 on the small BlockOut set, S = 1000) --
    both oracles; the GPU's capacity path for such grids meets the same files in tests/test_gpu_wide.py."""
    from helpers import wide_scenario
    from oracle.c_oracle import CPackingGame
    SW = 1000
    g = _load(golden_dir, "online_wide32")
    sh = wide_scenario("online_wide32")
    kw = dict(selectedAction=SW, resolutionA=0.01)
    env = PackingGame(sh, g["seq"], bufferSize=1, **kw) if which == "numpy" else CPackingGame(sh, g["seq"], **kw)
    obs = env.reset()
    np.testing.assert_array_equal(obs, g["obs"][0])
    fallbacks = 0
    for t in range(len(g["act"])):
        if which == "numpy":
            np.testing.assert_array_equal(env.space.naiveMask, g["mask"][t])
            np.testing.assert_array_equal(env.space.posZmap, g["posz"][t])
        a = minz_action(obs, SW)
        assert a == g["act"][t]
        obs, r, d, info = env.step(a)
        assert r == g["rew"][t] and d == g["done"][t]
        if d:
            assert info["counter"] == g["counter"][t] and info["ratio"] == g["ratio"][t]
            obs = env.reset()
        np.testing.assert_array_equal(obs[5 * SW:], g["obs"][t + 1][5 * SW:])
        if (obs[:5 * SW].reshape(SW, 5)[:, 4] == 1).any():
            np.testing.assert_array_equal(obs, g["obs"][t + 1])
        else:
            assert_fallback_rows_legal(obs[:5 * SW].reshape(SW, 5), sh.n_rot, ax=32, ay=32)
            fallbacks += 1
    assert g["done"].sum() >= 2 and fallbacks >= 1 and g["ncand"].max() > 500
    g = _load(golden_dir, "hier_wide32_k3")
    kw = dict(selectedAction=SW, resolutionA=0.01, bufferSize=3)
    sh = wide_scenario("hier_wide32_k3")
    env = PackingGame(sh, g["seq"], **kw) if which == "numpy" else CPackingGame(sh, g["seq"], **kw)
    np.testing.assert_array_equal(env.reset(), g["order_obs"][0])
    for t in range(len(g["act"])):
        loc = env.get_action_candidates(int(g["order_act"][t]))
        np.testing.assert_array_equal(loc[5 * SW:], g["loc_obs"][t][5 * SW:])
        if (loc[:5 * SW].reshape(SW, 5)[:, 4] == 1).any():
            np.testing.assert_array_equal(loc, g["loc_obs"][t])
        a = minz_action(loc, SW)
        assert a == g["act"][t]
        order, r, d, info = env.step(a)
        assert r == g["rew"][t] and d == g["done"][t]
        if d:
            order = env.reset()
        np.testing.assert_array_equal(order, g["order_obs"][t + 1])
