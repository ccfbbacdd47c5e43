"""The plain-C restatement (oracle/c/) against the numpy oracle and the reference-generated goldens."""
import os
import time

import numpy as np
import pytest

import irbpp_amd  # noqa: F401
from irbpp_amd import synthetic
from oracle.c_oracle import COracleVecEnv, CPackingGame
from oracle.packing import OracleVecEnv
from helpers import HIER_GOLDENS, ONLINE_GOLDENS, assert_fallback_rows_legal, golden_kwargs, golden_scenario, minz_action

S = 500


@pytest.mark.parametrize("name,kw,steps", [
    ("cube", {}, 40), ("blockout", {}, 40), ("general", {}, 25), ("fine", {"resolutionH": 0.005}, 10)])
def test_c_oracle_equals_numpy_oracle_online(name, kw, steps):
    sh = {"cube": lambda: synthetic.cube_shapes(),
          "blockout": lambda: synthetic.blockout_shapes(n_shapes=24, n_rot=4, cube=0.06, seed=0),
          "general": lambda: synthetic.general_shapes(n_shapes=16, n_rot=8, seed=1),
          "fine": lambda: synthetic.general_shapes(n_shapes=8, n_rot=8, fmin=8, fmax=40, res_h=0.005, seed=4)}[name]()
    seqs = synthetic.make_sequences(sh.n_shapes, 32, 60, seed=9)
    n = 2
    penv, cenv = OracleVecEnv(n, sh, seqs, **kw), COracleVecEnv(n, sh, seqs, **kw)
    po, co = penv.reset(), cenv.reset()
    np.testing.assert_array_equal(co, po)
    for t in range(steps):
        act = [minz_action(o.astype(np.float32), S) for o in po]
        po, pr, pd, pi = penv.step(act)
        co, cr, cd, ci = cenv.step(act)
        np.testing.assert_array_equal(co, po)
        np.testing.assert_array_equal(cr, pr)
        np.testing.assert_array_equal(cd, pd)
        for a, b in zip(ci, pi):
            assert a == b
        for i in range(n):
            pz, mk = cenv.envs[i].grids()
            np.testing.assert_array_equal(pz, penv.envs[i].space.posZmap)
            np.testing.assert_array_equal(mk, penv.envs[i].space.naiveMask)


def test_c_oracle_equals_numpy_oracle_hierarchical():
    sh = synthetic.blockout_shapes(n_shapes=24, n_rot=4, cube=0.06, seed=0)
    seqs = synthetic.make_sequences(sh.n_shapes, 32, 150, seed=5)
    k = 4
    penv, cenv = OracleVecEnv(2, sh, seqs, bufferSize=k), COracleVecEnv(2, sh, seqs, bufferSize=k)
    np.testing.assert_array_equal(cenv.reset(), penv.reset())
    for t in range(50):
        oa = [(3 * t + i) % k for i in range(2)]
        pl, cl = penv.get_action_candidates(oa), cenv.get_action_candidates(oa)
        np.testing.assert_array_equal(cl, pl)
        act = [minz_action(o.astype(np.float32), S) for o in pl]
        po, pr, pd, _ = penv.step(act)
        co, cr, cd, _ = cenv.step(act)
        np.testing.assert_array_equal(co, po)
        np.testing.assert_array_equal(cr, pr)
        np.testing.assert_array_equal(cd, pd)


@pytest.mark.parametrize("name", ONLINE_GOLDENS)
def test_c_oracle_matches_reference_golden(golden_dir, name):
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    sh = golden_scenario(name)
    env = CPackingGame(sh, g["seq"], **golden_kwargs(name))
    obs = env.reset()
    np.testing.assert_array_equal(obs, g["obs"][0])
    for t in range(len(g["act"])):
        a = minz_action(obs, S)
        assert a == g["act"][t]
        obs, r, d, info = env.step(a)
        assert r == g["rew"][t] and d == g["done"][t]
        if d:
            assert info["counter"] == g["counter"][t] and info["ratio"] == g["ratio"][t]
            obs = env.reset()
        np.testing.assert_array_equal(obs[5 * S:], g["obs"][t + 1][5 * S:])
        if (obs[:5 * S].reshape(S, 5)[:, 4] == 1).any():
            np.testing.assert_array_equal(obs, g["obs"][t + 1])
        else:
            assert_fallback_rows_legal(obs[:5 * S].reshape(S, 5), sh.n_rot)
        pz, mk = env.grids()
        np.testing.assert_array_equal(pz, g["posz"][t + 1])
        np.testing.assert_array_equal(mk, g["mask"][t + 1])


@pytest.mark.parametrize("name,k", HIER_GOLDENS)
def test_c_oracle_matches_reference_hierarchical_golden(golden_dir, name, k):
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    env = CPackingGame(golden_scenario(name), g["seq"], bufferSize=k)
    np.testing.assert_array_equal(env.reset(), g["order_obs"][0])
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


def test_c_oracle_more_than_S_selection_matches_reference(golden_dir):
    g = np.load(os.path.join(golden_dir, "more_than_s.npz"))
    env = CPackingGame(golden_scenario("more_than_s"), g["seq"], bufferSize=2)
    np.testing.assert_array_equal(env.reset(), g["order_obs0"])
    for t in range(len(g["act"])):
        env.set_heightmap(g["hm"][t])
        loc = env.get_action_candidates(int(g["order_act"][t]))
        np.testing.assert_array_equal(loc, g["loc_obs"][t])
        order, r, d, _ = env.step(int(g["act"][t]))
        assert r == g["rew"][t] and d == g["done"][t]
        np.testing.assert_array_equal(order, g["order_obs"][t])


def test_c_oracle_more_than_S_candidates_and_speed():
    sh = synthetic.general_shapes(n_shapes=12, n_rot=8, fmin=4, fmax=8, seed=11)
    seqs = synthetic.make_sequences(sh.n_shapes, 16, 40, seed=3)
    penv, cenv = OracleVecEnv(1, sh, seqs, bufferSize=2), COracleVecEnv(1, sh, seqs, bufferSize=2)
    penv.reset(); cenv.reset()
    rng = np.random.RandomState(5)
    for t in range(3):
        hm = rng.uniform(0.0, 0.12, size=(32, 32))
        penv.envs[0].space.heightmapC[:] = hm
        cenv.envs[0].set_heightmap(hm)
        pl, cl = penv.get_action_candidates([t % 2]), cenv.get_action_candidates([t % 2])
        np.testing.assert_array_equal(cl, pl)
        act = [minz_action(pl[0].astype(np.float32), S)]
        np.testing.assert_array_equal(cenv.step(act)[0], penv.step(act)[0])
    sh = synthetic.blockout_shapes(n_shapes=64, n_rot=4, seed=0)
    seqs = synthetic.make_sequences(sh.n_shapes, 100, 160, seed=123)
    env = COracleVecEnv(4, sh, seqs)
    obs = env.reset()
    t0 = time.perf_counter()
    for _ in range(100):
        obs, _, _, _ = env.step([minz_action(o, S) for o in obs])
    rate = 400 / (time.perf_counter() - t0)
    assert rate > 500, rate


def test_c_oracle_threads_and_shared_tables_change_nothing():
    """COracleVecEnv deals its bins to host threads inside the C library (orc_*_many) and all bins borrow ONE copy of the
    tables (orc_create_borrowed): same observations, rewards, done flags and infos as a handle per bin that owns its copy,
    stepped one by one."""
    sh = synthetic.blockout_shapes(n_shapes=24, n_rot=4, cube=0.06, seed=0)
    seqs = synthetic.make_sequences(sh.n_shapes, 64, 150, seed=5)
    n, k = 12, 3
    many = COracleVecEnv(n, sh, seqs, threads=5, bufferSize=k, global_offset=7, global_num=40)
    ones = [CPackingGame(sh, seqs, bufferSize=k, first_traj=1 + 7 + g, traj_stride=40) for g in range(n)]
    np.testing.assert_array_equal(many.reset(), np.array([e.reset() for e in ones]))
    ndone = 0
    for t in range(80):
        oa = [(3 * i + t) % k for i in range(n)]
        loc = many.get_action_candidates(oa)
        np.testing.assert_array_equal(loc, np.array([e.get_action_candidates(a) for e, a in zip(ones, oa)]))
        act = [minz_action(o.astype(np.float32), S) for o in loc]
        obs, rew, done, infos = many.step(act)
        for i, e in enumerate(ones):
            o, r, d, info = e.step(act[i])
            assert r == rew[i] and d == done[i]
            if d:
                assert info["counter"] == infos[i]["counter"] and info["ratio"] == infos[i]["ratio"]
                o = e.reset()
                ndone += 1
            np.testing.assert_array_equal(obs[i], o)
    assert ndone >= 3
