"""GPU parity tests proper: the HIP path, called through the C ABI, against the CPU oracle and
the committed golden vectors.  Integer/index/heightmap/mask results bit-exact; float32
observations equal to the float64 oracle after the same final cast; reward within 1e-5."""
import os

import numpy as np
import pytest
import torch

import irbpp_amd  # noqa: F401
from irbpp_amd import synthetic
from irbpp_amd.vec_env import GpuPackingEnv, GpuVecEnv
from oracle import cvtools
from oracle.packing import OracleVecEnv
from oracle.space import Space
from helpers import (HIER_GOLDENS, ONLINE_GOLDENS, assert_fallback_rows_legal, golden_kwargs, golden_scenario,
                     minz_action)

pytestmark = pytest.mark.gpu
GpuVecEnv.candidates_on_device = True        # these tests feed the location observations back to device kernels
S = 500
DEV = "cuda:0"


def _f32(x):
    return np.asarray(x, dtype=np.float64).astype(np.float32)


def _run_online(shapes, seqs, n, steps, **kw):
    genv = GpuVecEnv(shapes, seqs, n, device=DEV, **kw)
    oenv = OracleVecEnv(n, shapes, seqs, **kw)
    gobs = genv.reset()
    oobs = _f32(oenv.reset())
    np.testing.assert_array_equal(gobs.cpu().numpy(), oobs)
    ndone = 0
    for t in range(steps):
        act = genv.env.policy_minz(gobs).cpu().numpy()
        ref_act = np.array([minz_action(o, kw.get("selectedAction", S)) for o in oobs])
        np.testing.assert_array_equal(act, ref_act)
        gobs, grew, gdone, ginfo = genv.step(act)
        oobs, orew, odone, oinfo = oenv.step(act)
        oobs = _f32(oobs)
        np.testing.assert_array_equal(gobs.cpu().numpy(), oobs, err_msg=f"obs step {t}")
        np.testing.assert_array_equal(gdone, odone)
        np.testing.assert_allclose(grew.numpy()[:, 0], orew, atol=1e-5, rtol=0)
        np.testing.assert_array_equal(grew.numpy()[:, 0], orew.astype(np.float32))
        for i in range(n):
            gi, oi = ginfo[i], oinfo[i]
            assert gi["Valid"] is True
            if odone[i]:
                ndone += 1
                assert gi["counter"] == oi["counter"] and gi["ratio"] == oi["ratio"]
                assert gi["episode"]["r"] == oi["episode"]["r"] and gi["episode"]["l"] == oi["episode"]["l"]
            else:
                assert "episode" not in gi
        hm = genv.env.get_heightmaps().cpu().numpy()
        for i in range(n):
            np.testing.assert_array_equal(hm[i], oenv.envs[i].space.heightmapC)
    genv.env.check_device_error()
    genv.close()
    return ndone


def test_online_cube_matches_oracle():
    sh = synthetic.cube_shapes()
    assert _run_online(sh, synthetic.make_sequences(sh.n_shapes, 64, 60, seed=123), 6, 45) >= 3


def test_online_blockout_matches_oracle():
    sh = synthetic.blockout_shapes(n_shapes=24, n_rot=4, cube=0.06, seed=0)
    assert _run_online(sh, synthetic.make_sequences(sh.n_shapes, 64, 150, seed=5), 6, 60) >= 2


def test_online_general_r8_matches_oracle():
    sh = synthetic.general_shapes(n_shapes=24, n_rot=8, seed=1)
    assert _run_online(sh, synthetic.make_sequences(sh.n_shapes, 64, 60, seed=9), 5, 30) >= 3


def test_online_coarse_action_grid_matches_oracle():
    """resolutionA = 0.04: an 8x8 action grid (stepSize 4), i.e. lane groups are not image rows."""
    sh = synthetic.cube_shapes()
    assert _run_online(sh, synthetic.make_sequences(sh.n_shapes, 64, 60, seed=21), 5, 30, resolutionA=0.04) >= 1


def test_online_odd_geometry_matches_oracle():
    """A 0.32 x 0.24 m bin (32x24 heightmap, 16x12 action grid: no power of two for the reciprocal-multiply
    index arithmetic to hide behind), S = 120 candidates, 2 cm height levels."""
    sh = synthetic.cube_shapes()
    seqs = synthetic.make_sequences(sh.n_shapes, 64, 60, seed=33)
    assert _run_online(sh, seqs, 6, 40, bin_dimension=(0.32, 0.24, 0.30), selectedAction=120, resolutionZ=0.02) >= 2


@pytest.mark.parametrize("shapes_kind", ["cube", "general"])
def test_online_odd_action_grid_matches_oracle(shapes_kind):
    """A 0.30 x 0.26 m bin: 15 x 13 action cells, i.e. the 2 x 2 blocks of the overlap test and the phase planes of the
    heightmap tile have padding entries (8 x 7 lanes, tile larger than the heightmap); block path impossible (30 and
    26 cells), so cubes take the generic path here too."""
    if shapes_kind == "cube":
        sh = synthetic.cube_shapes()
    else:
        sh = synthetic.general_shapes(n_shapes=20, n_rot=8, fmin=4, fmax=14, seed=6)
    seqs = synthetic.make_sequences(sh.n_shapes, 64, 60, seed=41)
    assert _run_online(sh, seqs, 5, 40, bin_dimension=(0.30, 0.26, 0.30), selectedAction=150) >= 1


@pytest.mark.parametrize("n_rot,fmin,fmax,min_done", [(1, 4, 8, 0), (1, 8, 16, 5), (2, 4, 8, 0), (2, 8, 16, 5), (3, 4, 14, 2)])
def test_online_generic_path_blocking_depths_match_oracle(n_rot, fmin, fmax, min_done):
    """The generic overlap path deals (rotation, row-group block) tasks to four waves and blocks up to three row
    groups on registers; with few rotations the blocks shrink so that no wave idles (R = 2: two groups, R = 1: one),
    and footprints of up to 8 cells have four row groups in range, which split 2 + 2.  Every depth against the oracle."""
    sh = synthetic.general_shapes(n_shapes=16, n_rot=n_rot, fmin=fmin, fmax=fmax, seed=50 + n_rot)
    seqs = synthetic.make_sequences(sh.n_shapes, 48, 80, seed=60 + fmin)
    assert _run_online(sh, seqs, 5, 36) >= min_done          # (the small items do not fill a bin in 36 steps)


def test_online_fine_heightmap_matches_oracle():
    sh = synthetic.general_shapes(n_shapes=12, n_rot=8, fmin=8, fmax=40, res_h=0.005, seed=4)
    assert _run_online(sh, synthetic.make_sequences(sh.n_shapes, 32, 60, seed=2), 3, 14, resolutionH=0.005) >= 1


def test_exhausted_trajectory_ends_episode():
    sh = synthetic.blockout_shapes(n_shapes=8, n_rot=4, cube=0.04, seed=2)
    assert _run_online(sh, synthetic.make_sequences(sh.n_shapes, 16, 5, seed=1), 3, 16) >= 3


def test_hierarchical_matches_oracle():
    sh = synthetic.blockout_shapes(n_shapes=24, n_rot=4, cube=0.06, seed=0)
    seqs = synthetic.make_sequences(sh.n_shapes, 64, 150, seed=5)
    n, k = 4, 5
    genv = GpuVecEnv(sh, seqs, n, device=DEV, bufferSize=k)
    oenv = OracleVecEnv(n, sh, seqs, bufferSize=k)
    gord = genv.reset()
    oord = _f32(oenv.reset())
    np.testing.assert_array_equal(gord.cpu().numpy(), oord)
    ndone = 0
    for t in range(70):
        oa = np.array([(t * 7 + 3 + i) % k for i in range(n)])
        gloc = genv.get_action_candidates(oa)
        oloc = _f32(oenv.get_action_candidates(oa))
        np.testing.assert_array_equal(gloc.cpu().numpy(), oloc)
        act = genv.env.policy_minz(gloc).cpu().numpy()
        gord, grew, gdone, ginfo = genv.step(act)
        oord, orew, odone, oinfo = oenv.step(act)
        np.testing.assert_array_equal(gord.cpu().numpy(), _f32(oord))
        np.testing.assert_array_equal(gdone, odone)
        np.testing.assert_array_equal(grew.numpy()[:, 0], orew.astype(np.float32))
        ndone += int(odone.sum())
    genv.close()
    assert ndone >= 2


@pytest.mark.parametrize("name", ONLINE_GOLDENS)
def test_online_matches_reference_golden(golden_dir, name):
    """The HIP path against episodes the REFERENCE'S OWN PackingGame played (tests/golden/make_golden.py): the small
    scenarios and one recording per BASELINE.json config on the bench's own shape sets (cfg 2 R = 4 / R = 8, cfg 3, cfg 5
    at resolutionH 0.005 on the bench's 256 solids and on a 12-solid set)."""
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    sh = golden_scenario(name)
    genv = GpuVecEnv(sh, g["seq"], 1, device=DEV, **golden_kwargs(name))
    obs = genv.reset().cpu().numpy()[0]
    np.testing.assert_array_equal(obs, _f32(g["obs"][0]))
    fallbacks = 0
    for t in range(len(g["act"])):
        # the recorded action: the reference's scripted policy saw float64 rows, and two heights that differ there
        # (0.12 and 0.12000000000000001 on the 4 cm lattice) are one float32 -- the policy is not what is under test
        a = int(g["act"][t])
        o, r, d, info = genv.step(np.array([a]))
        obs = o.cpu().numpy()[0]
        assert d[0] == g["done"][t] and r.numpy()[0, 0] == np.float32(g["rew"][t])
        if d[0]:
            assert info[0]["counter"] == g["counter"][t] and info[0]["ratio"] == g["ratio"][t]
            assert info[0]["episode"]["r"] == g["ep_r"][t]
        ref = _f32(g["obs"][t + 1])
        np.testing.assert_array_equal(obs[5 * S:], ref[5 * S:])
        if (ref[:5 * S].reshape(S, 5)[:, 4] == 1).any():
            np.testing.assert_array_equal(obs, ref)
        else:
            # binPhy.py:217-225: np.argsort over an all-equal vector -- the reference's row ORDER is its numpy build's; the
            # rows must be S distinct in-range cells with H = bin height, V = 0, and here they are the stable prefix
            rows = obs[:5 * S].reshape(S, 5)
            assert_fallback_rows_legal(rows, sh.n_rot)
            want = np.array([[c // 256, (c % 256) // 16, c % 16, 0.30, 0.0] for c in range(S)]).astype(np.float32)
            np.testing.assert_array_equal(rows, want)
            fallbacks += 1
    assert fallbacks >= 1
    genv.env.check_device_error()
    genv.close()


@pytest.mark.parametrize("name,k", HIER_GOLDENS)
def test_hierarchical_matches_reference_golden(golden_dir, name, k):
    """get_action_candidates + step against the reference's own hierarchical episodes: k = 3 on the small BlockOut set
    and BASELINE config 4 (k = 10) on the bench's BlockOut set, order actions cycling over every buffer slot."""
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    genv = GpuVecEnv(golden_scenario(name), g["seq"], 1, device=DEV, bufferSize=k)
    order = genv.reset().cpu().numpy()[0]
    np.testing.assert_array_equal(order, _f32(g["order_obs"][0]))
    for t in range(len(g["act"])):
        loc = genv.get_action_candidates(np.array([int(g["order_act"][t])])).cpu().numpy()[0]
        ref = _f32(g["loc_obs"][t])
        np.testing.assert_array_equal(loc[5 * S:], ref[5 * S:])
        if (ref[:5 * S].reshape(S, 5)[:, 4] == 1).any():
            np.testing.assert_array_equal(loc, ref)
        a = int(g["act"][t])                           # (recorded on float64 rows, see test_online_matches_reference_golden)
        o, r, d, info = genv.step(np.array([a]))
        assert d[0] == g["done"][t] and r.numpy()[0, 0] == np.float32(g["rew"][t])
        if d[0]:
            assert info[0]["counter"] == g["counter"][t] and info[0]["ratio"] == g["ratio"][t]
        np.testing.assert_array_equal(o.cpu().numpy()[0], _f32(g["order_obs"][t + 1]))
    assert g["done"].sum() >= 1
    genv.env.check_device_error()
    genv.close()


def test_more_than_S_selection_matches_reference_golden(golden_dir):
    """binPhy.py:209-212 where the answer does not depend on a tie rule: more than S candidates with pairwise distinct
    placement heights, rows and their order as the reference's own np.argsort left them."""
    g = np.load(os.path.join(golden_dir, "more_than_s.npz"))
    genv = GpuVecEnv(golden_scenario("more_than_s"), g["seq"], 1, device=DEV, bufferSize=2)
    np.testing.assert_array_equal(genv.reset().cpu().numpy()[0], _f32(g["order_obs0"]))
    for t in range(len(g["act"])):
        genv.env.set_heightmaps(torch.from_numpy(g["hm"][t][None]).to(DEV))
        loc = genv.get_action_candidates(np.array([int(g["order_act"][t])])).cpu().numpy()[0]
        np.testing.assert_array_equal(loc, _f32(g["loc_obs"][t]))
        o, r, d, _ = genv.step(np.array([int(g["act"][t])]))
        assert d[0] == g["done"][t] and r.numpy()[0, 0] == np.float32(g["rew"][t])
        np.testing.assert_array_equal(o.cpu().numpy()[0], _f32(g["order_obs"][t]))
    genv.env.check_device_error()
    genv.close()


def test_possible_position_random_heightmaps():
    sh = synthetic.general_shapes(n_shapes=20, n_rot=8, seed=6)
    n = 32
    env = GpuPackingEnv(sh, synthetic.make_sequences(sh.n_shapes, 8, 8), n, device=DEV)
    rng = np.random.RandomState(0)
    hm = np.zeros((n, 32, 32))
    for b in range(n):
        kind = b % 4
        if kind == 0:
            hm[b] = np.kron(rng.randint(0, 8, (8, 8)), np.ones((4, 4))) * 0.04
        elif kind == 1:
            hm[b] = rng.uniform(0, 0.3, (32, 32))
        elif kind == 2:
            hm[b] = np.round(rng.uniform(0, 0.29, (32, 32)), 2)
        # kind 3: empty bin
    ids = rng.randint(0, sh.n_shapes, n).astype(np.int32)
    env.set_heightmaps(torch.from_numpy(hm).to(DEV))
    posz, mask = env.possible_position(torch.from_numpy(ids).to(DEV))
    posz, mask = posz.cpu().numpy(), mask.cpu().numpy()
    sp = Space(np.round([0.32, 0.32, 0.30], 6), 0.02, 0.01, sh.n_rot, sh.shot_info(), sh.extents)
    for b in range(n):
        sp.heightmapC[:] = hm[b]
        m = sp.get_possible_position(int(ids[b]))
        np.testing.assert_array_equal(mask[b], m.astype(np.uint8))
        np.testing.assert_array_equal(posz[b], sp.posZmap)
    env.check_device_error()
    env.close()


def _vertex_rows_to_candidates(rows, posz, n_rot):
    out = []
    for r in range(n_rot):
        pts = [(x, y) for x in range(16) for y in range(16) if (int(rows[r, y]) >> x) & 1]
        for x, y in pts:                          # ordered by (col, row) like np.unique
            out.append([r, y, x, posz[r, y, x], 1.0])
    return np.array(out) if out else None


def test_convex_hull_actions_golden_and_random(golden_dir):
    g = np.load(os.path.join(golden_dir, "cvtools_cases.npz"))
    sh = synthetic.cube_shapes()
    rng = np.random.RandomState(3)
    for R in (2, 4, 8):
        env = GpuPackingEnv(synthetic.blockout_shapes(8, n_rot=R) if R > 2 else sh,
                            synthetic.make_sequences(8, 4, 4), 1, device=DEV)
        idx = [i for i in range(len(g["n_rot"])) if g["n_rot"][i] == R]
        posz = [g["posz"][i][:R] for i in idx]
        mask = [g["mask"][i][:R] for i in idx]
        for _ in range(40):                       # adversarial random images: dense, sparse, diagonal
            m = (rng.uniform(size=(R, 16, 16)) < rng.choice([0.3, 0.5, 0.7, 0.95])).astype(np.float64)
            z = np.where(m > 0, rng.randint(0, rng.choice([1, 2, 4, 30]), (R, 16, 16)) * 0.01, 1e3)
            posz.append(z)
            mask.append(m)
        posz, mask = np.array(posz), np.array(mask)
        rows = env.convex_hull_actions(torch.from_numpy(posz).to(DEV),
                                       torch.from_numpy(mask.astype(np.uint8)).to(DEV)).cpu().numpy()
        for i in range(len(posz)):
            ref = cvtools.getConvexHullActions(posz[i], mask[i], 0.01)
            got = _vertex_rows_to_candidates(rows[i], posz[i], R)
            if ref is None:
                assert got is None
            else:
                np.testing.assert_array_equal(got, ref)
            if i < len(idx):                      # and the reference-generated golden rows
                n = int(g["cand_len"][idx[i]])
                if n:
                    np.testing.assert_array_equal(got, g["cand"][idx[i]][:n])
        env.check_device_error()
        env.close()


@pytest.mark.parametrize("workload,n,parts,steps", [("blockout", 4096, 2, 110), ("blockout", 8192, 4, 110),
                                                    ("general", 4096, 2, 60), ("abc_fine", 2048, 2, 60)])
def test_full_size_properties_and_sharding_invariance(workload, n, parts, steps):
    """BASELINE config 2 at full width (4096 bins) and at north_star's 8192 bins on one GPU, config 3 at its 4096 bins
    and config 5 at its 2048 bins per GPU: size-independent properties instead of an oracle run, the C oracle on 64
    sampled bins, and the same bins as `parts` shards."""
    from bench import make_workload
    sh, seqs, kw = make_workload(workload)
    env = GpuPackingEnv(sh, seqs, n, device=DEV, **kw)
    half = [GpuPackingEnv(sh, seqs, n // parts, device=DEV, global_offset=o, global_bins=n, **kw)
            for o in range(0, n, n // parts)]
    obs = env.reset()
    hobs = [h.reset() for h in half]
    vol = torch.from_numpy(sh.volumes).to(DEV)
    prev_hm = env.get_heightmaps().clone()
    placed_vol = torch.zeros(n, dtype=torch.float64, device=DEV)
    # ... and the C oracle on a sample of the bins: groups of 16 global bins spread over the whole range
    from oracle.c_oracle import COracleVecEnv
    starts = [0, n // 3, n // 2 - 8, n - 16]
    sample = np.concatenate([np.arange(o, o + 16) for o in starts])
    cenvs = [COracleVecEnv(16, sh, seqs, global_offset=o, global_num=n, **kw) for o in starts]
    np.testing.assert_array_equal(obs[sample].cpu().numpy(), _f32(np.concatenate([c.reset() for c in cenvs])))
    ndone = 0
    for t in range(steps):
        act = env.policy_minz(obs)
        item = obs[:, 5 * S].to(torch.int64)
        obs, rew, done = env.step(act)
        sact = act[sample].cpu().numpy()
        cres = [c.step(sact[16 * j:16 * j + 16]) for j, c in enumerate(cenvs)]
        np.testing.assert_array_equal(obs[sample].cpu().numpy(), _f32(np.concatenate([r[0] for r in cres])),
                                      err_msg=f"sampled bins differ from the C oracle at step {t}")
        np.testing.assert_array_equal(done[sample].cpu().numpy().astype(bool), np.concatenate([r[2] for r in cres]))
        hacts = [h.policy_minz(o) for h, o in zip(half, hobs)]
        hobs = [h.step(a)[0] for h, a in zip(half, hacts)]
        assert torch.equal(torch.cat(hobs), obs), "result depends on how bins are sharded"
        hm = env.get_heightmaps()
        d = done.bool()
        assert bool((hm[d] == 0).all())                                   # auto-reset leaves an empty bin
        assert bool((hm[~d] >= prev_hm[~d]).all())                        # heights only grow inside an episode
        assert float(hm.max()) <= 0.30 + 1e-9
        assert torch.equal(obs[:, 5 * S + 9:], hm.reshape(n, -1).float())  # obs carries the heightmap
        exp_rew = torch.where(d, torch.zeros_like(rew), vol[item.clamp(min=0)] / 0.03072 * 10)
        assert torch.allclose(rew, exp_rew, atol=1e-12, rtol=0)
        placed_vol = torch.where(d, torch.zeros_like(placed_vol), placed_vol + vol[item.clamp(min=0)])
        assert bool((placed_vol <= 0.03072 + 1e-12).all())                # never more volume than the bin holds
        prev_hm = hm.clone()
        ndone += int(d.sum())
    tot = env.episode_totals().cpu().numpy()
    assert ndone > n // 4 and tot[0] == ndone                             # terminal steps and auto-resets were covered
    assert tot[0] >= 0 and tot[1] <= tot[0]                               # ratios are in [0,1]
    env.check_device_error()
    for e in [env] + half:
        e.close()


@pytest.mark.parametrize("n,parts,steps", [(1024, 2, 110), (8192, 4, 110)])
def test_full_size_hierarchical_properties_and_sharding_invariance(n, parts, steps):
    """BASELINE config 4 (BlockOut buffered k = 10) at its per-GPU width (1024 bins = 8192 / 8) and at its full 8192 bins:
    every placement through get_action_candidates + step (binPhy.py:161-169, 248-337) with order actions that exercise
    every buffer slot, the C oracle on 64 sampled bins for both observations, done and reward, the same bins as `parts`
    shards, and the size-independent properties of the online full-size test."""
    from bench import make_workload
    from oracle.c_oracle import COracleVecEnv
    sh, seqs, kw = make_workload("blockout_k10")
    k = kw["bufferSize"]
    env = GpuPackingEnv(sh, seqs, n, device=DEV, **kw)
    shards = [GpuPackingEnv(sh, seqs, n // parts, device=DEV, global_offset=o, global_bins=n, **kw)
              for o in range(0, n, n // parts)]
    per = n // parts
    obs = env.reset()
    sobs = [h.reset() for h in shards]
    assert torch.equal(torch.cat(sobs), obs)
    starts = [0, n // 3, n // 2 - 8, n - 16]
    sample = np.concatenate([np.arange(o, o + 16) for o in starts])
    cenvs = [COracleVecEnv(16, sh, seqs, global_offset=o, global_num=n, **kw) for o in starts]
    np.testing.assert_array_equal(obs[sample].cpu().numpy(), _f32(np.concatenate([c.reset() for c in cenvs])))
    vol = torch.from_numpy(sh.volumes).to(DEV)
    prev_hm = env.get_heightmaps().clone()
    placed_vol = torch.zeros(n, dtype=torch.float64, device=DEV)
    gbin = torch.arange(n, device=DEV)
    ndone = 0
    for t in range(steps):
        order = ((gbin * 3 + t) % k if t % 2 else torch.zeros_like(gbin)).to(torch.int32)
        assert torch.equal(obs[:, k:], env.get_heightmaps().reshape(n, -1).float())        # order obs = [k ids | heightmap]
        item = obs.gather(1, order.to(torch.int64)[:, None])[:, 0].to(torch.int64)         # the chosen buffer slot's item
        loc = env.get_action_candidates(order)
        assert torch.equal(loc[:, 5 * S], item.float())                                    # next_item_vec[0] (binPhy.py:191)
        sorder = order[sample].cpu().numpy()
        cloc = [c.get_action_candidates(sorder[16 * j:16 * j + 16]) for j, c in enumerate(cenvs)]
        np.testing.assert_array_equal(loc[sample].cpu().numpy(), _f32(np.concatenate(cloc)),
                                      err_msg=f"sampled location observations differ from the C oracle at step {t}")
        act = env.policy_minz(loc)
        obs, rew, done = env.step(act)
        sact = act[sample].cpu().numpy()
        cres = [c.step(sact[16 * j:16 * j + 16]) for j, c in enumerate(cenvs)]
        np.testing.assert_array_equal(obs[sample].cpu().numpy(), _f32(np.concatenate([r[0] for r in cres])),
                                      err_msg=f"sampled bins differ from the C oracle at step {t}")
        np.testing.assert_array_equal(done[sample].cpu().numpy().astype(bool), np.concatenate([r[2] for r in cres]))
        np.testing.assert_array_equal(rew[sample].cpu().numpy().astype(np.float32),
                                      np.concatenate([r[1] for r in cres]).astype(np.float32))
        slocs = [h.get_action_candidates(order[j * per:(j + 1) * per]) for j, h in enumerate(shards)]
        assert torch.equal(torch.cat(slocs), loc), "location observations depend on how bins are sharded"
        sobs = [h.step(h.policy_minz(lo))[0] for h, lo in zip(shards, slocs)]
        assert torch.equal(torch.cat(sobs), obs), "result depends on how bins are sharded"
        hm = env.get_heightmaps()
        d = done.bool()
        assert bool((hm[d] == 0).all())
        assert bool((hm[~d] >= prev_hm[~d]).all())
        assert float(hm.max()) <= 0.30 + 1e-9
        exp_rew = torch.where(d, torch.zeros_like(rew), vol[item.clamp(min=0)] / 0.03072 * 10)
        assert torch.allclose(rew, exp_rew, atol=1e-12, rtol=0)
        placed_vol = torch.where(d, torch.zeros_like(placed_vol), placed_vol + vol[item.clamp(min=0)])
        assert bool((placed_vol <= 0.03072 + 1e-12).all())
        prev_hm = hm.clone()
        ndone += int(d.sum())
    tot = env.episode_totals().cpu().numpy()
    assert ndone > n // 4 and tot[0] == ndone
    env.check_device_error()
    for e in [env] + shards:
        e.check_device_error()
        e.close()


def test_more_than_S_candidates_sorted_selection():
    """Random per-cell heights make almost every valid cell its own level component -> more than
    S=500 candidates: exercises the `np.argsort(candidates[:,3])[:S]` branch (binPhy.py:209-212).
    The heightmap is replaced on both sides and re-observed through get_action_candidates, which
    recomputes posZmap from the current heightmap (binPhy.py:161-169)."""
    sh = synthetic.general_shapes(n_shapes=12, n_rot=8, fmin=4, fmax=8, seed=11)
    seqs = synthetic.make_sequences(sh.n_shapes, 16, 40, seed=3)
    n, k = 3, 2
    genv = GpuVecEnv(sh, seqs, n, device=DEV, bufferSize=k)
    oenv = OracleVecEnv(n, sh, seqs, bufferSize=k)
    np.testing.assert_array_equal(genv.reset().cpu().numpy(), _f32(oenv.reset()))
    rng = np.random.RandomState(5)
    hits = 0
    for t in range(6):
        hm = rng.uniform(0.0, 0.12, size=(n, 32, 32))
        genv.env.set_heightmaps(torch.from_numpy(hm).to(DEV))
        for i in range(n):
            oenv.envs[i].space.heightmapC[:] = hm[i]
        oa = np.array([t % k] * n)
        gloc = genv.get_action_candidates(oa).cpu().numpy()
        oloc = _f32(oenv.get_action_candidates(oa))
        np.testing.assert_array_equal(gloc, oloc)
        hits += int(((oloc[:, :5 * S].reshape(n, S, 5)[:, :, 4] == 1).sum(1) == S).sum())
        act = np.array([minz_action(o, S) for o in oloc])
        gord, grew, gdone, _ = genv.step(act)
        oord, orew, odone, _ = oenv.step(act)
        np.testing.assert_array_equal(gord.cpu().numpy(), _f32(oord))
        np.testing.assert_array_equal(gdone, odone)
    genv.env.check_device_error()
    genv.close()
    assert hits >= 2, "the >S branch was not reached"


def test_cube_buffered_k10_matches_oracle():
    """BASELINE config 4 shape: hierarchical with a 10-item buffer (binPhy.py:161-169,228-230)."""
    sh = synthetic.cube_shapes()
    seqs = synthetic.make_sequences(sh.n_shapes, 64, 80, seed=8)
    n, k = 3, 10
    genv = GpuVecEnv(sh, seqs, n, device=DEV, bufferSize=k)
    oenv = OracleVecEnv(n, sh, seqs, bufferSize=k)
    np.testing.assert_array_equal(genv.reset().cpu().numpy(), _f32(oenv.reset()))
    rng = np.random.RandomState(0)
    ndone = 0
    for t in range(50):
        oa = rng.randint(0, k, n)
        gloc = genv.get_action_candidates(oa)
        np.testing.assert_array_equal(gloc.cpu().numpy(), _f32(oenv.get_action_candidates(oa)))
        act = genv.env.policy_minz(gloc).cpu().numpy()
        gord, grew, gdone, ginfo = genv.step(act)
        oord, orew, odone, oinfo = oenv.step(act)
        np.testing.assert_array_equal(gord.cpu().numpy(), _f32(oord))
        np.testing.assert_array_equal(gdone, odone)
        ndone += int(odone.sum())
    genv.close()
    assert ndone >= 2


def test_heuristic_actions_match_oracle():
    """Space.get_heuristic_action (space.py:162-218): MINZ / DBLF / FIRSTFIT / HM, all four flips."""
    sh = synthetic.general_shapes(n_shapes=16, n_rot=4, seed=21)
    seqs = synthetic.make_sequences(sh.n_shapes, 32, 60, seed=4)
    n = 4
    genv = GpuVecEnv(sh, seqs, n, device=DEV)
    oenv = OracleVecEnv(n, sh, seqs)
    gobs = genv.reset()
    oobs = _f32(oenv.reset())
    for t in range(7):
        for method in ("MINZ", "DBLF", "FIRSTFIT", "HM"):
            for d in ((0, 3) if t % 2 else (1, 2)):
                got = genv.env.heuristic_action(method, d).cpu().numpy()
                for i, e in enumerate(oenv.envs):
                    if e.space.naiveMask.sum() == 0:
                        continue
                    ref = e.space.get_heuristic_action(method, e.next_item_ID, d)
                    assert tuple(got[i]) == tuple(int(v) for v in ref), (t, method, d, i)
        act = genv.env.policy_minz(gobs).cpu().numpy()
        gobs, _, _, _ = genv.step(act)
        oobs, _, _, _ = oenv.step(act)
        np.testing.assert_array_equal(gobs.cpu().numpy(), _f32(oobs))
    genv.close()


def test_heuristic_hm_sums_large_windows_in_numpys_order():
    """HM on the 64 x 64 heightmap of BASELINE config 5 with its largest items first: windows of up to 56 x 56 = 3136 cells, i.e.
    numpy's pairwise sum five splits deep (the kernel walks that tree with an explicit stack and one cursor over the masked-in
    cells), on heightmaps that are no longer flat."""
    from bench import make_workload
    sh, _, kw = make_workload("abc_fine")
    size = [max(np.asarray(t[r][0]).size for r in range(len(t))) for t in sh.tables]
    big = np.argsort(size)[::-1][:12]
    assert size[big[0]] > 2048 and size[big[5]] > 1024
    seqs = np.stack([np.resize(np.roll(big, k), 40) for k in range(4)]).astype(np.int32)
    n = 2
    genv = GpuVecEnv(sh, seqs, n, device=DEV, **kw)
    oenv = OracleVecEnv(n, sh, seqs, **kw)
    gobs = genv.reset()
    oenv.reset()
    asked = 0
    for t in range(4):
        for d in (0, 3):
            got = genv.env.heuristic_action("HM", d).cpu().numpy()
            for i, e in enumerate(oenv.envs):
                if e.space.naiveMask.sum() == 0:
                    continue
                ref = e.space.get_heuristic_action("HM", e.next_item_ID, d)
                assert tuple(got[i]) == tuple(int(v) for v in ref), (t, d, i)
                asked += 1
        act = genv.env.policy_minz(gobs).cpu().numpy()
        gobs, _, _, _ = genv.step(act)
        oobs, _, _, _ = oenv.step(act)
        np.testing.assert_array_equal(gobs.cpu().numpy(), _f32(oobs))
    genv.close()
    assert asked >= 8


@pytest.mark.parametrize("tag", ["general", "blockout"])
def test_heuristic_actions_match_reference_golden(golden_dir, tag):
    """irbpp_heuristic_action against what the reference's own Space.get_heuristic_action (space.py:162-218) returned
    on the states of an episode the reference's PackingGame played (heuristic_cases.npz): the same episode is replayed
    here (trajectory 1, the recorded actions) and every method x flip is asked at every state, the states with no
    valid cell (all scores 1e6 -> index 0) included."""
    g = np.load(os.path.join(golden_dir, "heuristic_cases.npz"))
    sh = golden_scenario("heuristic_" + tag)
    genv = GpuVecEnv(sh, g[tag + "_seq"], 1, device=DEV)
    gobs = genv.reset()
    for t in range(len(g[tag + "_act"])):
        assert int(gobs[0, 5 * S].item()) == g[tag + "_item"][t]
        np.testing.assert_array_equal(genv.env.get_heightmaps()[0].cpu().numpy(), g[tag + "_hm"][t])
        for mi, method in enumerate(("MINZ", "DBLF", "FIRSTFIT", "HM")):
            for d in range(4):
                got = genv.env.heuristic_action(method, d).cpu().numpy()[0]
                assert tuple(int(v) for v in got) == tuple(g[tag + "_heur"][t, mi, d]), (t, method, d)
        act = genv.env.policy_minz(gobs).cpu().numpy()
        assert int(act[0]) == g[tag + "_act"][t]
        gobs, _, gdone, _ = genv.step(act)
        assert bool(gdone[0]) == bool(g[tag + "_done"][t])
    genv.close()


def test_batched_evaluation_matches_sequential_reference_protocol():
    """tools.test (tools.py:303-358): one env, episode e on trajectory e+1, env.packed per episode.
    The batched driver plays the same episodes side by side; statistics and placement records
    must equal the oracle run sequentially."""
    from irbpp_amd.evaluate import evaluate, rotation_quaternion_xyzw
    from oracle.packing import PackingGame
    sh = synthetic.blockout_shapes(n_shapes=20, n_rot=4, cube=0.06, seed=7)
    seqs = synthetic.make_sequences(sh.n_shapes, 16, 80, seed=1)
    E = 6
    out = evaluate(sh, seqs, E, device=DEV)
    assert out["episodes"] == E and out["unfinished"] == 0
    env = PackingGame(sh, seqs)                      # LoadItemCreator semantics: first reset -> trajectory 1
    for ep in range(E):
        obs = env.reset()
        rsum, steps = 0.0, 0
        while True:
            obs, r, d, info = env.step(minz_action(_f32(obs), S))
            rsum += r
            steps += 1
            if d:
                break
        assert out["ratio"][ep] == info["ratio"] and out["length"][ep] == steps
        assert out["reward_sum"][ep] == rsum
        placed = env.packed                          # the last entry is the refused placement (binPhy.py:296-311)
        assert len(out["trajs"][ep]) == len(placed) == info["counter"] + 1
        for got, (item, rot, lx, ly, height) in zip(out["trajs"][ep], placed):
            assert got[0] == item and got[1] == "%d.obj" % item
            np.testing.assert_allclose(got[2], [lx * 0.02, ly * 0.02, height], rtol=0, atol=1e-12)
            np.testing.assert_array_equal(got[3], rotation_quaternion_xyzw(rot))
    assert abs(out["mean_ratio"] - np.mean(out["ratio"])) < 1e-15


def test_evaluation_writes_the_reference_trajs_file(golden_dir, tmp_path):
    """evaluate(save=...) against the golden of the reference's own tools.test (tests/golden/make_golden.py):
    ``trajs.npy`` as an object array, one ``env.packed`` list per episode, rows [id, name, positionFLB, quaternion]."""
    from irbpp_amd.evaluate import evaluate
    g = np.load(os.path.join(golden_dir, "tools_test.npz"))
    sh = synthetic.blockout_shapes(n_shapes=20, n_rot=4, cube=0.06, seed=7)
    path = str(tmp_path / "logs" / "evaluation" / "run" / "trajs.npy")
    out = evaluate(sh, g["seq"], len(g["ep_len"]), device=DEV, save=path)
    trajs = np.load(path, allow_pickle=True)
    assert trajs.dtype == object and len(trajs) == len(g["ep_len"])
    row = 0
    for ep, want_len in zip(trajs, g["ep_len"]):
        assert len(ep) == want_len
        for i, (item, name, pos, quat) in enumerate(ep):
            assert item == g["ids"][row] and name == g["names"][row]
            # the cell a refused "action 0" points at when no candidate was valid depends on numpy's unstable argsort
            if i < want_len - 1 or g["pos"][row][2] < 1e3:
                np.testing.assert_allclose(pos, g["pos"][row], rtol=0, atol=1e-12)
                np.testing.assert_allclose(quat, g["quat"][row], rtol=0, atol=1e-12)
            row += 1
    assert abs(out["avg_reward"] - float(g["avg_reward"])) < 1e-9 and out["avg_length"] == float(g["avg_length"])


def test_dataset_directory_with_meshes_is_rasterised_and_cached(tmp_path):
    """load_reference_dataset on a directory without a shotInfo cache: the meshes are ray-cast on the GPU, the tables
    equal the analytic ones, and the cache the reference reads at start-up (tools.py:258-277) is written."""
    import torch as _t
    from irbpp_amd import dataset, meshes
    blk = synthetic.blockout_shapes(n_shapes=6, n_rot=4, seed=0)
    ms = {k: meshes.voxel_mesh(o, 0.04) for k, o in enumerate(synthetic.blockout_voxels(6, 0))}
    names = {k: "poly%d.obj" % k for k in range(6)}
    seqs = synthetic.make_sequences(6, 8, 40, seed=2)
    root = str(tmp_path)
    dataset.save_reference_dataset(root, "blk6", names, seqs, ms)
    shapes, seqs2, names2 = dataset.load_reference_dataset(root, "blk6", 0.01, n_rot=4, device=DEV)
    assert shapes.meta["tables_from"] == "rasteriser" and names2 == names
    np.testing.assert_allclose(shapes.extents, blk.extents, rtol=0, atol=1e-12)
    for k in range(6):
        for r in range(4):
            for got, ref in zip(shapes.tables[k][r], blk.tables[k][r]):
                np.testing.assert_allclose(got, ref, rtol=0, atol=1e-12)
    cache = dataset.shot_info_dir(root, "blk6", 0.01)
    T, B, mH, mB = _t.load(os.path.join(cache, "5_3.pt"), weights_only=False)        # as tools.py:271-272 reads it
    np.testing.assert_array_equal(T, shapes.tables[5][3][0])
    again, _, _ = dataset.load_reference_dataset(root, "blk6", 0.01, n_rot=4)        # now from the cache, no device
    assert again.meta["tables_from"] == "cache"
    genv = GpuVecEnv(again, seqs2, 3, device=DEV)                                    # and it drives the environment
    obs = genv.reset()
    for _ in range(5):
        obs, _, _, _ = genv.step(genv.env.policy_minz(obs).cpu().numpy())
    genv.close()


def test_shot_item_rasteriser_matches_oracle_and_analytic_tables():
    """tools.shot_item (tools.py:98-135) on the GPU: boxes and polycube meshes reproduce the
    analytic tables of the synthetic generators exactly; a slanted solid matches the numpy oracle."""
    from irbpp_amd import meshes
    from oracle.shot import shot_item
    # boxes (Cube dataset)
    cube = synthetic.cube_shapes()
    for k in (0, 37, 124):
        ex, ey, ez = cube.extents[k, 0]
        ss = meshes.shape_set_from_meshes([meshes.box_mesh(ex, ey, ez)], 2, 0.01, DEV)
        for r in range(2):
            np.testing.assert_array_equal(ss.extents[0, r], cube.extents[k, r])
            for got, ref in zip(ss.tables[0][r], cube.tables[k][r]):
                np.testing.assert_array_equal(got, ref)
        assert abs(ss.volumes[0] - cube.volumes[k]) < 1e-15
    # polycubes (BlockOut dataset), all four lattice rotations
    blk = synthetic.blockout_shapes(n_shapes=10, n_rot=4, seed=0)
    ms = [meshes.voxel_mesh(o, 0.04) for o in synthetic.blockout_voxels(10, 0)]
    ss = meshes.shape_set_from_meshes(ms, 4, 0.01, DEV)
    np.testing.assert_allclose(ss.extents, blk.extents, rtol=0, atol=1e-15)
    np.testing.assert_allclose(ss.volumes, blk.volumes, rtol=0, atol=1e-15)
    for k in range(10):
        for r in range(4):
            for got, ref in zip(ss.tables[k][r], blk.tables[k][r]):
                np.testing.assert_allclose(got, ref, rtol=0, atol=1e-15)
    # a slanted solid: skewed pyramid frustum, 45-degree pose included
    rng = np.random.RandomState(2)
    base = np.array([[0, 0, 0], [0.11, 0, 0.01], [0.12, 0.09, 0], [0.01, 0.1, 0.02]])
    apex = np.array([0.05, 0.04, 0.13])
    v = np.vstack([base, apex])
    f = np.array([[0, 2, 1], [0, 3, 2], [0, 1, 4], [1, 2, 4], [2, 3, 4], [3, 0, 4]], dtype=np.int32)
    for deg in (0.0, 45.0):
        vr = meshes.rotate_z(v, deg)
        ext, tab = meshes.shot_item_gpu(vr, f, 0.01, DEV)
        ref = shot_item(vr - vr.min(0), f, 0.01)
        for got, want in zip(tab, ref):
            np.testing.assert_allclose(got, want, rtol=0, atol=1e-12)
        assert tab[2].sum() > 20 and (tab[0] >= tab[1]).all()


def test_make_vec_envs_with_reference_style_args():
    """envs.make_vec_envs (envs.py:67-99): same call shape, same return triple, reference containers."""
    import types
    from irbpp_amd.vec_env import make_vec_envs
    sh = synthetic.cube_shapes()
    args = types.SimpleNamespace(
        num_processes=5, device=0, shotInfo=sh.shot_info(), infoDict=sh.info_dict(), resolutionA=0.02,
        resolutionH=0.01, resolutionZ=0.01, bin_dimension=np.round([0.32, 0.32, 0.30], 6), selectedAction=500,
        bufferSize=1, scale=[100, 100, 100], sequences=synthetic.make_sequences(sh.n_shapes, 32, 60, seed=123))
    envs, spaces, obs_len = make_vec_envs(args, "./logs/runinfo", True)
    assert obs_len == 5 * 500 + 9 + 1024 and envs.num_envs == 5
    assert spaces[0].shape == (obs_len,) and spaces[1].n == 500
    oenv = OracleVecEnv(5, sh, args.sequences)
    obs = envs.reset()
    assert obs.dtype == torch.float32 and obs.is_cuda and tuple(obs.shape) == (5, obs_len)
    np.testing.assert_array_equal(obs.cpu().numpy(), _f32(oenv.reset()))
    act = envs.env.policy_minz(obs).cpu().numpy()
    envs.step_async(act)
    with pytest.raises(RuntimeError):
        envs.step_async(act)                      # one outstanding step (shmem_vec_env.py:58-74)
    obs, rew, done, infos = envs.step_wait()
    o2, r2, d2, i2 = oenv.step(act)
    np.testing.assert_array_equal(obs.cpu().numpy(), _f32(o2))
    assert rew.dtype == torch.float32 and tuple(rew.shape) == (5, 1) and not rew.is_cuda
    assert done.dtype == bool and len(infos) == 5 and all(infos[i]["Valid"] for i in range(5))
    envs.close()


@pytest.mark.parametrize("k", [1, 3])
def test_reset_specific_matches_per_env_reset(k):
    """ShmemVecEnv.reset_specific (shmem_vec_env.py:113-117): the listed envs start their next
    trajectory with an empty bin, the others are untouched, no episode statistics are recorded."""
    shapes = synthetic.blockout_shapes(24, seed=5)
    seqs = synthetic.make_sequences(24, n_traj=64, length=50, seed=6)
    n = 7
    genv = GpuVecEnv(shapes, seqs, n, device=DEV, bufferSize=k)
    oenv = OracleVecEnv(n, shapes, seqs, bufferSize=k)
    gobs, oobs = genv.reset(), _f32(oenv.reset())
    np.testing.assert_array_equal(gobs.cpu().numpy(), oobs)

    def advance(steps, gobs, oobs):
        for _ in range(steps):
            if k > 1:
                order = np.arange(n) % k
                gloc = genv.get_action_candidates(order)
                oloc = _f32(oenv.get_action_candidates(order))
                np.testing.assert_array_equal(gloc.cpu().numpy(), oloc)
            else:
                gloc, oloc = gobs, oobs
            act = np.array([minz_action(o, S) for o in oloc])
            gobs, grew, gdone, ginfo = genv.step(act)
            oobs, orew, odone, oinfo = oenv.step(act)
            oobs = _f32(oobs)
            np.testing.assert_array_equal(gobs.cpu().numpy(), oobs)
            np.testing.assert_array_equal(gdone, odone)
            for i in range(n):
                if odone[i]:
                    assert ginfo[i]["episode"]["l"] == oinfo[i]["episode"]["l"]
                    assert abs(ginfo[i]["episode"]["r"] - oinfo[i]["episode"]["r"]) < 1e-5
        return gobs, oobs

    gobs, oobs = advance(9, gobs, oobs)
    before = genv.env.episode_totals().cpu().numpy().copy()
    for idxs in ([4, 1], [6], []):
        sub = genv.reset_specific(idxs)
        assert tuple(sub.shape) == (len(idxs), genv.obs_len)
        if not idxs:
            continue
        ref = _f32(oenv.reset_specific(idxs))
        np.testing.assert_array_equal(sub.cpu().numpy(), ref)
        for j, i in enumerate(idxs):
            gobs[i] = sub[j]
            oobs[i] = ref[j]
    np.testing.assert_array_equal(genv.env.episode_totals().cpu().numpy(), before)    # nothing recorded
    hm = genv.env.get_heightmaps().cpu().numpy()
    assert (hm[[1, 4, 6]] == 0).all() and (hm[[0, 2, 3, 5]] != 0).any()
    advance(40, gobs, oobs)                                # incl. whole episodes after the partial reset
    with pytest.raises(ValueError):
        genv.reset_specific([2, 2])
    with pytest.raises(ValueError):
        genv.reset_specific([n])
    strict = GpuVecEnv(shapes, seqs, 2, device=DEV, allow_early_resets=False)
    strict.reset()
    with pytest.raises(RuntimeError):
        strict.reset_specific([0])


def test_abi_rejects_calls_out_of_order():
    """Error behaviour of the boundary: status codes instead of crashes."""
    import ctypes as C
    from irbpp_amd import _lib
    lib = _lib.load()
    cfg = _lib.IrbppConfig(num_bins=2, n_rot=2, selected=500, buffer_size=1, resolution_a=0.02, resolution_h=0.01,
                           resolution_z=0.01, bin=(C.c_double * 3)(0.32, 0.32, 0.3), scale_z=100.0, traj_start=1,
                           global_offset=0, global_bins=2, device=0, stability=0)
    h = C.c_void_p()
    assert lib.irbpp_create(C.byref(cfg), C.byref(h)) == 0
    obs = torch.zeros((2, 3533), dtype=torch.float32, device=DEV)
    act = torch.zeros((2,), dtype=torch.int32, device=DEV)
    p = lambda t: C.c_void_p(t.data_ptr())   # noqa: E731
    assert lib.irbpp_reset(h, p(obs), None) == -3                 # IRBPP_ERR_STATE: nothing loaded yet
    assert lib.irbpp_step(h, p(act), p(obs), None, None) == -3    # step before reset
    assert lib.irbpp_get_action_candidates(h, p(act), p(obs), None) == -3
    assert lib.irbpp_reset(h, None, None) == -1                   # IRBPP_ERR_ARG
    assert lib.irbpp_reset_bins(h, p(act), 2, p(obs), None) == -3  # before the first reset
    assert lib.irbpp_reset_bins(h, p(act), 3, p(obs), None) == -1  # more bins than the env has
    bad = np.zeros(4, dtype=np.int32)
    assert lib.irbpp_load_sequences(h, bad.ctypes.data_as(_lib.c_i32_p), 0, 4) == -1
    assert lib.irbpp_destroy(h) == 0


def test_tooling_hooks_time_and_locate_every_bin():
    """irbpp_debug_kernel_timing / irbpp_debug_phase_cycles: the numbers bench.py and tools/ report."""
    shapes = synthetic.blockout_shapes(16, seed=3)
    seqs = synthetic.make_sequences(16, n_traj=40, length=60, seed=4)
    env = GpuPackingEnv(shapes, seqs, 64, device=DEV)
    obs = env.reset()
    env.enable_kernel_timing(4)
    cyc = env.enable_phase_cycles(True)
    for _ in range(3):
        obs, _, _ = env.step(env.policy_minz(obs))
    ms = env.kernel_times_ms()
    assert ms.shape == (3,) and (ms > 0).all() and (ms < 100).all()
    assert env.kernel_times_ms().shape == (0,)                   # the ring was drained
    for _ in range(6):                                            # more launches than pairs: the latest four
        obs, _, _ = env.step(env.policy_minz(obs))
    assert env.kernel_times_ms().shape == (4,)
    env.enable_kernel_timing(0)
    c = cyc.cpu().numpy()
    assert (np.diff(c[:, :3], axis=1) > 0).all()                  # stamps of every bin are ordered: transition kernel
    assert (c[:, 4] > c[:, 3]).all()                              # ... and emit kernel (its own launch, maybe another CU)
    assert (c[:, 9] > c[:, 8]).all()                              # wall clock exit after entry
    env.enable_phase_cycles(False)
    env.check_device_error()


@pytest.mark.parametrize("workload,n,steps", [("blockout", 192, 130), ("general", 96, 45), ("cube", 128, 60),
                                              ("blockout_k10", 128, 120), ("abc_fine", 256, 80),
                                              ("blockout_r8", 96, 150)])
def test_many_bins_full_episodes_vs_c_oracle(workload, n, steps):
    """Scale check made possible by the C oracle: the bench workloads themselves (every BASELINE config:
    cfg 2 blockout / blockout_r8, cfg 3 general, cfg 4 blockout_k10 through get_action_candidates + step,
    cfg 5 abc_fine on the 64x64 heightmap), hundreds of bins, whole episodes including auto-resets;
    every observation, reward, done and episode info equal."""
    from bench import make_workload
    from oracle.c_oracle import COracleVecEnv
    shapes, seqs, kw = make_workload(workload)
    seqs = seqs[:2000]
    k = int(kw.get("bufferSize", 1))
    genv = GpuVecEnv(shapes, seqs, n, device=DEV, **kw)
    cenv = COracleVecEnv(n, shapes, seqs, **kw)
    gobs = genv.reset()
    np.testing.assert_array_equal(gobs.cpu().numpy(), _f32(cenv.reset()))
    ndone = 0
    for t in range(steps):
        if k > 1:                                     # hierarchical placement (binPhy.py:161-169, trainer.py:266-281)
            order = (np.arange(n) * 3 + t) % k if t % 2 else np.zeros(n, dtype=np.int64)
            gloc = genv.get_action_candidates(order)
            np.testing.assert_array_equal(gloc.cpu().numpy(), _f32(cenv.get_action_candidates(order)),
                                          err_msg=f"location obs step {t}")
            act = genv.env.policy_minz(gloc).cpu().numpy()
        else:
            act = genv.env.policy_minz(gobs).cpu().numpy()
        gobs, grew, gdone, ginfo = genv.step(act)
        cobs, crew, cdone, cinfo = cenv.step(act)
        np.testing.assert_array_equal(gobs.cpu().numpy(), _f32(cobs), err_msg=f"step {t}")
        np.testing.assert_array_equal(gdone, cdone)
        np.testing.assert_array_equal(grew.numpy()[:, 0], crew.astype(np.float32))
        for i in np.nonzero(cdone)[0]:
            gi, ci = ginfo[int(i)], cinfo[int(i)]
            assert gi["counter"] == ci["counter"] and gi["ratio"] == ci["ratio"]
            assert gi["episode"]["r"] == ci["episode"]["r"] and gi["episode"]["l"] == ci["episode"]["l"]
            ndone += 1
    genv.env.check_device_error()
    genv.close()
    assert ndone >= n // 2


def test_bench_gpus2_spawns_two_real_ranks():
    """`python bench.py --gpus 2` launched as ONE plain process (how the driver launches it) re-executes itself
    under torch.distributed.run: two ranks, each with its own shard, totals reduced over the process group.
    One GPU here, so the ranks share the device and talk gloo (dry run of the RCCL path); the line must say
    n_gpus 2, world_size 2, two device entries and twice the bins."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    for var in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(var, None)
    def run(*flags):
        res = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--backend", "gloo",
                              "--warmup", "2", "--no-cpu-baseline", *flags],
                             capture_output=True, text=True, timeout=600, env=env)
        assert res.returncode == 0, res.stderr[-2000:]
        lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
        assert len(lines) == 1, res.stdout[-2000:]
        return json.loads(lines[0])

    # strong scaling of north_star's config 4: 8192 buffered (k = 10) bins divided over the two ranks
    strong = run("--config", "cfg4", "--steps", "4", "--min-seconds", "0", "--prefill", "30")
    assert strong["scaling"] == "strong" and strong["n_gpus"] == 2 and strong["config"]["baseline_config"] == "cfg4"
    assert strong["config"]["bins_per_gpu"] == 4096 and strong["config"]["global_bins"] == 8192 and strong["steps"] == 4
    assert strong["ranks"]["ms_per_step_min"] <= strong["ranks"]["ms_per_step_max"] == strong["ms_per_step"]
    assert abs(strong["value"] - 8192 * 4 / (strong["ms_per_step"] * 4e-3)) < 1e-3 * strong["value"]
    # the timed region repeats in blocks of EXACTLY --steps steps until --min-seconds: `steps` is the block (the K of the
    # contract), `steps_timed_total` what really ran
    timed = run("--bins", "256", "--steps", "5", "--min-seconds", "0.2", "--prefill", "20")
    assert timed["steps"] == 5 and timed["steps_per_block"] == 5 and timed["steps_timed_total"] == 5 * timed["timed_blocks"] >= 10
    assert timed["steps_timed_total"] * timed["ms_per_step"] >= 200.0

    out = run("--bins", "256", "--steps", "5", "--min-seconds", "0", "--prefill", "120")
    assert out["n_gpus"] == 2 and out["ranks"]["world_size"] == 2 and len(out["ranks"]["devices"]) == 2
    assert out["config"]["global_bins"] == 512 and out["scaling"] == "weak" and out["steps"] == 5
    assert abs(out["value"] - 512 * 5 / (out["ms_per_step"] * 5e-3)) < 1e-3 * out["value"]
    # the reduced totals cover both shards: with 120 prefill steps most of the 512 bins finished an episode
    assert out["episodes"]["finished_since_reset"] > 256
    # ... and the two shards really were global bins [0,256) and [256,512): same totals as one 512-bin env
    sh, seqs, kw = __import__("bench").make_workload("blockout")
    one = GpuPackingEnv(sh, seqs, 512, device=DEV, **kw)
    obs = one.reset()
    for _ in range(120 + 2 + 5):
        obs, _, _ = one.step(one.policy_minz(obs))
    assert float(one.episode_totals()[0].item()) == out["episodes"]["finished_since_reset"]
    one.close()


@pytest.mark.parametrize("mode", [1, 2])
def test_stability_proxy_matches_its_specification(mode):
    """irbpp_config::stability -- the static support test that stands in for the rigid-body settling the path
    leaves out (Interface.py:271-310).  Not reference behaviour: its specification is oracle/stability.py.  Mode 1
    only reports a verdict per accepted placement (observations identical to the plain path), mode 2 also refuses an
    unstable placement, which ends the episode."""
    checked = unstable = 0
    for sh, seqs in ((synthetic.general_shapes(n_shapes=24, n_rot=8, seed=1), synthetic.make_sequences(24, 64, 60, seed=9)),
                     (synthetic.blockout_shapes(n_shapes=24, n_rot=4, cube=0.06, seed=0), synthetic.make_sequences(24, 64, 150, seed=5))):
        n = 6
        genv = GpuVecEnv(sh, seqs, n, device=DEV, stability=mode)
        oenv = OracleVecEnv(n, sh, seqs, stability=mode)
        plain = OracleVecEnv(n, sh, seqs) if mode == 1 else None
        gobs = genv.reset()
        np.testing.assert_array_equal(gobs.cpu().numpy(), _f32(oenv.reset()))
        if plain:
            plain.reset()
        for t in range(40):
            act = genv.env.policy_minz(gobs).cpu().numpy()
            gobs, grew, gdone, ginfo = genv.step(act)
            oobs, orew, odone, _ = oenv.step(act)
            np.testing.assert_array_equal(gobs.cpu().numpy(), _f32(oobs), err_msg=f"step {t}")
            np.testing.assert_array_equal(gdone, odone)
            stable = genv.env.step_info_host()["stable"]
            for i in range(n):
                want = oenv.envs[i].last_stable and not odone[i]
                assert bool(stable[i]) == bool(want), (t, i)
                checked += 1
                unstable += int(not odone[i] and not want)
            if plain:                                         # mode 1 changes nothing but the extra output
                pobs, _, pdone, _ = plain.step(act)
                np.testing.assert_array_equal(_f32(oobs), _f32(pobs))
        genv.close()
    assert checked > 300 and (mode == 2 or unstable > 5)      # the irregular solids do produce unstable placements


def test_second_reset_moves_on_to_the_next_trajectories():
    """VecEnv.reset() mid-run (shmem_vec_env.py:61-68 -> PackingGame.reset -> LoadItemCreator.reset, IRcreator.py:86-92):
    every env drops its running episode and starts its NEXT trajectory, it does not rewind to the first one."""
    sh = synthetic.blockout_shapes(n_shapes=24, n_rot=4, cube=0.06, seed=0)
    seqs = synthetic.make_sequences(sh.n_shapes, 64, 150, seed=5)
    n = 5
    genv, oenv = GpuVecEnv(sh, seqs, n, device=DEV), OracleVecEnv(n, sh, seqs)
    gobs, oobs = genv.reset(), _f32(oenv.reset())
    for rnd in range(3):
        for t in range(7):
            act = genv.env.policy_minz(gobs).cpu().numpy()
            gobs, _, gdone, _ = genv.step(act)
            oobs, _, odone, _ = oenv.step(act)
            np.testing.assert_array_equal(gobs.cpu().numpy(), _f32(oobs))
        first_items = gobs.cpu().numpy()[:, 5 * S].copy()
        gobs, oobs = genv.reset(), _f32(oenv.reset())
        np.testing.assert_array_equal(gobs.cpu().numpy(), oobs, err_msg=f"reset number {rnd + 2}")
        assert not gobs.cpu().numpy()[:, 5 * S + 9:].any()                       # empty bins
    genv.close()


@pytest.mark.parametrize("workload,n,steps", [("blockout", 256, 140), ("general", 128, 60), ("blockout_k10", 64, 60)])
def test_fused_policy_matches_policy_kernel(workload, n, steps):
    """irbpp_set_auto_policy: the action written next to every emitted observation (reset, online step incl.
    auto-resets, get_action_candidates; rows selected from more than S candidates and the no-candidate fallback
    both occur in 'general') equals irbpp_policy_minz on that observation, and the error word published by the
    step's last workgroup is the device's."""
    from bench import make_workload
    shapes, seqs, kw = make_workload(workload)
    k = int(kw.get("bufferSize", 1))
    env = GpuPackingEnv(shapes, seqs[:500], n, device=DEV, **kw)
    auto = torch.full((n,), -7, dtype=torch.int32, device=DEV)
    env.set_auto_policy(auto)
    obs = env.reset()
    slot0 = torch.zeros((n,), dtype=torch.int32, device=DEV)
    seen_fallback = seen_sorted = 0
    for t in range(steps):
        if k > 1:
            loc = env.get_action_candidates(slot0)
        else:
            loc = obs
        ref = env.policy_minz(loc)
        assert torch.equal(auto, ref), f"step {t}: fused policy differs from the policy kernel"
        rows = loc[:, :5 * S].reshape(n, S, 5)
        seen_fallback += int(((rows[:, :, 4] == 0).any(1)).sum())
        seen_sorted += int((rows[:, S - 1, 4] == 1).sum())
        obs, _, _ = env.step(ref.clone())
        torch.cuda.synchronize()
        assert int(env._out_err.item()) == 0
    env.check_device_error()
    env.set_auto_policy(None)
    auto.fill_(-7)
    if k == 1:
        env.step(ref.clone())
        torch.cuda.synchronize()
        assert int(auto.min()) == -7 and int(auto.max()) == -7        # switched off: untouched
    if workload == "general":
        assert seen_fallback > 0 and seen_sorted > 0
    env.close()


@pytest.mark.parametrize("workload,n,steps", [("blockout", 128, 150), ("general", 64, 50)])
def test_registered_obs_buffers_deliver_the_same_observations(workload, n, steps):
    """irbpp_register_obs_buffer: two registered ping-pong buffers (poisoned before the hand-over) against an
    environment that writes fresh, unregistered buffers -- every observation identical, through auto-resets, the
    >S selection and the no-candidate fallback."""
    from bench import make_workload
    shapes, seqs, kw = make_workload(workload)
    a = GpuPackingEnv(shapes, seqs[:400], n, device=DEV, **kw)
    b = GpuPackingEnv(shapes, seqs[:400], n, device=DEV, **kw)
    bufs = [torch.full((n, a.loc_obs_len), 7.5, dtype=torch.float32, device=DEV) for _ in range(2)]
    for t in bufs:
        a.register_obs_buffer(t)
    oa = bufs[0]
    a_first = a.reset()                                # unregistered buffer: plain full write
    ob = b.reset()
    assert torch.equal(a_first, ob)
    act = b.policy_minz(ob)
    for t in range(steps):
        dst = bufs[t % 2]
        oa, _, _ = a.step(act, obs_out=dst)
        ob, _, _ = b.step(act)
        assert oa.data_ptr() == dst.data_ptr()
        assert torch.equal(oa, ob), f"step {t}: registered buffer differs"
        act = b.policy_minz(ob)
    a.check_device_error()
    a.close()
    b.close()


def test_possible_position_custom_matches_oracle_on_a_mesh_outside_the_dataset():
    """Space.get_possible_position_custom (space.py:131-160): ray-cast a mesh that is not in the data set and run
    the overlap test on the bins' current heightmaps; against the numpy Space fed with the same tables."""
    from irbpp_amd import meshes
    shapes = synthetic.blockout_shapes(16, seed=3)
    seqs = synthetic.make_sequences(16, n_traj=40, length=60, seed=4)
    n = 6
    env = GpuPackingEnv(shapes, seqs, n, device=DEV)
    obs = env.reset()
    for _ in range(12):
        obs, _, _ = env.step(env.policy_minz(obs))
    occ = np.zeros((2, 3, 2), dtype=bool)
    occ[0, :, 0] = True
    occ[1, 1, :] = True                                # an L/T-shaped polycube that blockout_shapes(seed=3) need not contain
    verts, faces = meshes.voxel_mesh(occ, 0.04)
    posz, mask = meshes.possible_position_custom(env, verts, faces, rot_idx=2)
    hm = env.get_heightmaps().cpu().numpy().reshape(n, env.Hx, env.Hy)
    ext, tab = meshes.shot_item_gpu(verts, faces, 0.01, DEV)
    for b in range(n):
        zmap = np.full((env.Ax, env.Ay), 1e3)
        nm = np.zeros((env.Ax, env.Ay))
        T, B, mH, mB = tab
        bs = np.round(ext, 6)
        fx, fy = np.ceil(bs[0:2] / 0.01).astype(int)
        ax, ay = np.ceil(bs[0:2] / 0.02).astype(int)
        for X in range(env.Ax - ax + 1):
            for Y in range(env.Ay - ay + 1):
                z = np.max((hm[b][2 * X:2 * X + fx, 2 * Y:2 * Y + fy] - B) * mB)
                if np.round(z + bs[2] - 0.30, 6) <= 0:
                    nm[X, Y] = 1
                zmap[X, Y] = z
        assert np.array_equal(posz[b, 2].cpu().numpy(), zmap)
        assert np.array_equal(mask[b, 2].cpu().numpy(), nm.astype(np.uint8))
        assert (posz[b, [0, 1, 3]].cpu().numpy() == 1e3).all() and not mask[b, [0, 1, 3]].any()
    env.close()
