"""Boundary behaviour of GpuVecEnv that the reference's VecEnv has and a GPU implementation can lose: step outputs that
belong to the caller (shmem_vec_env.py:76-81 builds fresh arrays every step), the error word of a buffered step (it has
no emit kernel behind it), an item ring that runs dry, the staging of order / location actions."""
import types

import numpy as np
import pytest
import torch

import irbpp_amd  # noqa: F401
from irbpp_amd import _lib, itemgen, synthetic
from irbpp_amd.vec_env import GpuVecEnv, make_vec_envs
from oracle.c_oracle import COracleVecEnv

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
S = 500


def _f32(x):
    return np.asarray(x, dtype=np.float64).astype(np.float32)


@pytest.mark.parametrize("groups", [1, 2])
def test_step_outputs_belong_to_the_caller(groups):
    """reward / done / infos of step t are kept as they were handed out while the environment takes further steps (a
    rollout or n-step storage holds them; the reference returns fresh arrays every step)."""
    sh = synthetic.blockout_shapes(n_shapes=24, n_rot=4, cube=0.06, seed=0)
    seqs = synthetic.make_sequences(sh.n_shapes, 64, 150, seed=5)
    n = 32
    genv = GpuVecEnv(sh, seqs, n, device=DEV, num_groups=groups)
    cenv = COracleVecEnv(n, sh, seqs)
    gobs = genv.reset()
    cenv.reset()
    kept = []
    for t in range(70):
        act = genv.env.policy_minz(gobs).cpu().numpy() if groups == 1 else \
            torch.cat([e.policy_minz(gobs[genv.env.rows(g)]) for g, e in enumerate(genv.env.groups)]).cpu().numpy()
        gobs, grew, gdone, ginfo = genv.step(act)
        _, crew, cdone, cinfo = cenv.step(act)
        kept.append((grew, gdone, ginfo, crew.astype(np.float32), cdone.copy(), cinfo))
    assert sum(int(k[4].sum()) for k in kept) >= 10                         # episodes ended along the way
    for grew, gdone, ginfo, crew, cdone, cinfo in kept:                     # ... and every step's outputs are still its own
        np.testing.assert_array_equal(grew.numpy()[:, 0], crew)
        np.testing.assert_array_equal(gdone, cdone)
        assert gdone is kept[-1][1] or not np.shares_memory(gdone, kept[-1][1])
        for i in range(n):
            gi, ci = ginfo[i], cinfo[i]
            assert gi["Valid"] is True and ("episode" in gi) == bool(cdone[i])
            if cdone[i]:
                assert gi["counter"] == ci["counter"] and gi["ratio"] == ci["ratio"]
                assert gi["episode"]["r"] == ci["episode"]["r"] and gi["episode"]["l"] == ci["episode"]["l"]
    genv.close()


@pytest.mark.parametrize("k", [1, 3])
def test_bad_item_id_reaches_the_error_word_of_the_step(k):
    """An id outside the shape table in the trajectories: IRBPP_DEVERR_BAD_ITEM is raised by the transition kernel --
    for a buffered environment (k > 1) the last kernel of step(), which has no emit kernel behind it to hand the word
    on -- and must surface in that very step's outputs."""
    sh = synthetic.blockout_shapes(n_shapes=8, n_rot=4, cube=0.06, seed=1)
    seqs = synthetic.make_sequences(sh.n_shapes, 16, 40, seed=2)
    seqs[:, 6] = sh.n_shapes + 3                                            # every trajectory meets a bad id at its 7th item
    genv = GpuVecEnv(sh, seqs, 4, device=DEV, bufferSize=k)
    genv.candidates_on_device = True
    gobs = genv.reset()
    with pytest.raises(_lib.IrbppError, match="flags=4"):
        for t in range(12):
            loc = genv.get_action_candidates(np.zeros(4, dtype=np.int64)) if k > 1 else gobs
            gobs, _, _, _ = genv.step(genv.env.policy_minz(loc).cpu().numpy())
    genv.close()


def test_error_raised_by_a_listed_reset_reaches_the_next_buffered_step():
    """irbpp_reset_bins with an index outside the environment raises IRBPP_DEVERR_BAD_BIN in the device error word only
    (a reset has no step outputs).  The buffered step that follows -- whose transition kernel is its last and which
    hands the library the SAME err_dev pointer as the step before -- must still report it: the library seeds the
    step's word again after every reset."""
    from irbpp_amd.vec_env import GpuPackingEnv
    sh = synthetic.blockout_shapes(n_shapes=8, n_rot=4, cube=0.06, seed=1)
    seqs = synthetic.make_sequences(sh.n_shapes, 16, 40, seed=2)
    env = GpuPackingEnv(sh, seqs, 4, device=DEV, bufferSize=3)
    env.reset()
    slot = torch.zeros(4, dtype=torch.int32, device=DEV)
    env.step(env.policy_minz(env.get_action_candidates(slot)))
    env.step_info_host()                                                    # clean so far, err_dev handed over twice
    env.reset_bins(torch.tensor([1, 9], dtype=torch.int32, device=DEV))     # bin 9 does not exist
    env.step(slot)                                                          # no get_action_candidates in between: nobody else copies the word
    with pytest.raises(_lib.IrbppError, match="flags=8"):
        env.step_info_host()
    env.close()


def test_item_ring_that_runs_dry_is_caught_at_the_fetch():
    """item_stream = 1 with a feeder that never refills: the bin that wraps around its 16-item ring reads a slot it has
    consumed already -> IRBPP_DEVERR_STREAM_DRY in that step's error word (not an episode that silently replays items)."""
    sh = synthetic.blockout_shapes(n_shapes=20, n_rot=4, cube=0.06, seed=7)
    dic = {i: "%d.obj" % i for i in range(20)}
    args = types.SimpleNamespace(
        num_processes=3, device=0, seed=11, shapes=sh, dicPath=dic, dataSample="pose", resolutionA=0.02,
        resolutionH=0.01, resolutionZ=0.01, bin_dimension=np.round([0.32, 0.32, 0.30], 6), selectedAction=S,
        bufferSize=1, scale=[100, 100, 100], evaluate=False, item_ring=16)
    envs, _, _ = make_vec_envs(args, "./logs/runinfo", True)
    envs.feeder.every = 10 ** 9                                             # the host never delivers again
    gobs = envs.reset()
    steps = 0
    with pytest.raises(_lib.IrbppError, match="flags=32"):
        for t in range(40):
            gobs, _, _, _ = envs.step(envs.env.policy_minz(gobs).cpu().numpy())
            steps += 1
    assert 10 <= steps <= 16                                                # 16 items: the reset took one, every step one more
    envs.close()


def test_order_and_location_actions_have_their_own_staging():
    """get_action_candidates (device tensor returned, nothing synchronised) followed at once by step with host actions:
    the order actions must not be overwritten under the candidate kernels.  Grouped environment, k = 3, against the C
    oracle."""
    sh = synthetic.blockout_shapes(n_shapes=24, n_rot=4, cube=0.06, seed=0)
    seqs = synthetic.make_sequences(sh.n_shapes, 64, 150, seed=5)
    n, k = 48, 3
    genv = GpuVecEnv(sh, seqs, n, device=DEV, bufferSize=k, num_groups=4)
    genv.candidates_on_device = True
    cenv = COracleVecEnv(n, sh, seqs, bufferSize=k)
    np.testing.assert_array_equal(genv.reset().cpu().numpy(), _f32(cenv.reset()))
    for t in range(40):
        order = (np.arange(n) * 5 + t) % k
        gloc = genv.get_action_candidates(order)                            # host int64 -> pinned staging -> device
        cloc = _f32(cenv.get_action_candidates(order))
        act = np.array([int(np.argmin(np.where(c.reshape(S + 0, 5)[:, 4] == 1, c.reshape(S, 5)[:, 3], np.inf)))
                        if (c.reshape(S, 5)[:, 4] == 1).any() else 0 for c in cloc[:, :5 * S]])
        gobs, _, gdone, _ = genv.step(act)                                  # stages the location actions right behind
        cobs, _, cdone, _ = cenv.step(act)
        np.testing.assert_array_equal(gloc.cpu().numpy(), cloc, err_msg=f"location obs step {t}")
        np.testing.assert_array_equal(gobs.cpu().numpy(), _f32(cobs), err_msg=f"step {t}")
        np.testing.assert_array_equal(gdone, cdone)
    genv.close()


def _unit_item_shapes(n_rot=2):
    """One 2 cm cube: its footprint is exactly one action cell (2 x 2 heightmap cells at resolutionH 0.01, step 2), so
    posZmap[r, X, Y] is the maximum of heightmap block (X, Y) -- any 16 x 16 level image can be dialled in."""
    from irbpp_amd.shapes import ShapeSet
    from irbpp_amd.synthetic import _box_tables
    ext = np.array([0.02, 0.02, 0.02])
    return ShapeSet(np.array([[ext] * n_rot]), np.array([8e-6]), [[_box_tables(ext, 0.01) for _ in range(n_rot)]], name="unit")


def _snake_image(zigzags=5):
    """One 8-connected component whose outer border has more than 128 CHAIN_APPROX_SIMPLE points: five zigzag rows
    joined at alternating ends (a border runs along both sides of a one-pixel line and turns at every pixel).  With 2, 3
    or 4 rows the border has 61, 92 or 123 points: more than the 56 the trace kernel keeps in LDS, within the 128 of its
    wave-parallel path."""
    img = np.zeros((16, 16), dtype=bool)
    for j, y0 in enumerate((0, 3, 6, 9, 12)[:zigzags]):
        for x in range(16):
            img[y0 + (x & 1), x] = True
        if j < zigzags - 1:
            xe = 15 if j % 2 == 0 else 0
            img[y0 + 2, xe] = True
            img[y0 + 1, xe] = True
            img[y0 + 3, xe] = True
    return img


def test_adversarial_level_images_through_transition_trace_polygon_emit():
    """Level images chosen at will -- random speckle at several densities, rings, a snake whose border outgrows the
    128-point slot of the trace kernel (its sequential redo), shorter snakes whose borders outgrow the 56 points kept in
    LDS (61, 92, 123: the rest lies in global scratch) -- fed through the whole split pipeline by way of the
    heightmap of a bin that observes a one-cell item; every location observation against the oracle."""
    from oracle import contours as OC
    from oracle.packing import OracleVecEnv
    sh = _unit_item_shapes()
    seqs = np.zeros((8, 40), dtype=np.int32)
    n, k = 6, 2
    genv = GpuVecEnv(sh, seqs, n, device=DEV, bufferSize=k)
    genv.candidates_on_device = True
    oenv = OracleVecEnv(n, sh, seqs, bufferSize=k)
    np.testing.assert_array_equal(genv.reset().cpu().numpy(), _f32(oenv.reset()))
    snake = _snake_image()
    outer = [c for c, hole in zip(*(lambda r: (r[0], r[2]))(OC.find_contours(snake.astype(np.uint8)))) if not hole]
    assert len(outer) == 1 and len(outer[0]) > 128, len(outer[0])           # the redo path is really taken
    mids = [_snake_image(z) for z in (2, 3, 4)]
    for m, want in zip(mids, (61, 92, 123)):
        got = [len(c) for c, hole in zip(*(lambda r: (r[0], r[2]))(OC.find_contours(m.astype(np.uint8)))) if not hole]
        assert got == [want], got
    rng = np.random.RandomState(12)
    big = 0
    for t in range(12):
        imgs = []
        for i in range(n):
            kind = (t + i) % 4
            if kind == 0 and t >= 8:
                m = mids[(t + i // 4) % 3]
                imgs.append(m if i % 2 == 0 else m.T.copy())
            elif kind == 0:
                imgs.append(snake if (t // 4) % 2 == 0 else snake.T.copy())
            elif kind == 1:
                imgs.append(rng.rand(16, 16) < rng.uniform(0.3, 0.7))
            elif kind == 2:
                g = np.maximum(np.abs(np.arange(16)[:, None] - 7.5), np.abs(np.arange(16)[None, :] - 7.5))
                imgs.append((np.floor(g) % 2 == 0) ^ (rng.rand(16, 16) < 0.03))
            else:
                imgs.append(np.kron(rng.rand(4, 4) < 0.5, np.ones((4, 4), dtype=bool)) ^ (rng.rand(16, 16) < 0.05))
        hm = np.zeros((n, 32, 32))
        for i, im in enumerate(imgs):                                       # image pixel (x, y) = action cell (row y, col x)
            lv = np.where(im, 0.05, 0.11) + rng.randint(0, 2, size=(16, 16)) * np.where(im, 0.0, 0.03)
            hm[i] = np.kron(lv, np.ones((2, 2)))
        big += sum(1 for im in imgs if im is snake or (im.shape == snake.shape and (im == snake.T).all()))
        genv.env.set_heightmaps(torch.from_numpy(hm).to(DEV))
        for i in range(n):
            oenv.envs[i].space.heightmapC[:] = hm[i]
        oa = np.array([t % k] * n)
        gloc = genv.get_action_candidates(oa).cpu().numpy()
        oloc = _f32(oenv.get_action_candidates(oa))
        np.testing.assert_array_equal(gloc, oloc, err_msg=f"round {t}")
        act = np.array([int(np.argmin(np.where(c.reshape(S, 5)[:, 4] == 1, c.reshape(S, 5)[:, 3], np.inf)))
                        if (c.reshape(S, 5)[:, 4] == 1).any() else 0 for c in oloc[:, :5 * S]])
        gord, _, gdone, _ = genv.step(act)
        oord, _, odone, _ = oenv.step(act)
        np.testing.assert_array_equal(gord.cpu().numpy(), _f32(oord))
        np.testing.assert_array_equal(gdone, odone)
    assert big >= 8
    genv.env.check_device_error()
    genv.close()


def test_make_vec_envs_hands_out_observations_that_stay():
    """The drop-in constructor returns a fresh observation tensor per call like the reference's VecPyTorch (envs.py:149-165):
    a caller may keep every state of a rollout.  (GpuVecEnv's own default, a ring of three registered buffers, recycles
    them: the opt-in `args.obs_ring`.)"""
    from oracle.c_oracle import COracleVecEnv as CO
    sh = synthetic.blockout_shapes(n_shapes=24, n_rot=4, cube=0.06, seed=0)
    seqs = synthetic.make_sequences(sh.n_shapes, 64, 150, seed=5)
    args = types.SimpleNamespace(
        num_processes=8, device=0, seed=1, shapes=sh, sequences=seqs, resolutionA=0.02, resolutionH=0.01, resolutionZ=0.01,
        bin_dimension=np.round([0.32, 0.32, 0.30], 6), selectedAction=S, bufferSize=1, scale=[100, 100, 100], evaluate=True)
    envs, _, _ = make_vec_envs(args, "./logs/runinfo", True)
    assert envs.num_groups == 1            # the trainer's synchronous step() is fastest as one group at every size (profiles/r05/s35)
    cenv = CO(8, sh, seqs)
    states = [envs.reset()]
    want = [_f32(cenv.reset())]
    for t in range(12):
        act = envs.env.policy_minz(states[-1]).cpu().numpy()
        obs, _, _, _ = envs.step(act)
        states.append(obs)
        want.append(_f32(cenv.step(act)[0]))
    for got, ref in zip(states, want):                                      # every state of the rollout, looked at afterwards
        np.testing.assert_array_equal(got.cpu().numpy(), ref)
    assert len({s.data_ptr() for s in states}) == len(states)
    envs.close()
    ring = GpuVecEnv(sh, seqs, 8, device=DEV)                                # the opt-in: three buffers in turn
    ptrs = [ring.reset().data_ptr()] + [ring.step(np.zeros(8, dtype=np.int64))[0].data_ptr() for _ in range(5)]
    assert len(set(ptrs)) == 3
    ring.close()


def test_num_groups_zero_lets_the_library_choose():
    """GpuVecEnv(num_groups=0): groups_for() by the overlap path the data set takes and the number of envs -- free-form tables
    at 1024 envs step as two groups, lattice data as one; observations equal an ungrouped environment's either way."""
    gen = synthetic.general_shapes(n_shapes=24, n_rot=8, seed=1)
    seqs = synthetic.make_sequences(gen.n_shapes, 256, 60, seed=9)
    auto = GpuVecEnv(gen, seqs, 1024, device=DEV, num_groups=0)
    one = GpuVecEnv(gen, seqs, 1024, device=DEV)
    assert auto.num_groups == 2 and one.num_groups == 1
    a, b = auto.reset(), one.reset()
    torch.cuda.synchronize()
    assert torch.equal(a, b)
    for t in range(6):
        act = one.env.policy_minz(b).cpu().numpy()
        a, ra, da, _ = auto.step(act)
        b, rb, db, _ = one.step(act)
        assert torch.equal(a, b) and np.array_equal(da, db) and torch.equal(ra, rb)
    auto.close()
    one.close()
    blk = synthetic.blockout_shapes(n_shapes=24, n_rot=4, cube=0.06, seed=0)
    lat = GpuVecEnv(blk, synthetic.make_sequences(blk.n_shapes, 64, 150, seed=5), 1024, device=DEV, num_groups=0)
    assert lat.num_groups == 1
    lat.close()
    from irbpp_amd.vec_env import group_stream_pair
    pair, seen = group_stream_pair(DEV)
    lat = GpuVecEnv(blk, synthetic.make_sequences(blk.n_shapes, 64, 150, seed=5), 4096, device=DEV, num_groups=0)
    assert lat.num_groups == (2 if seen else 1)             # two groups only on a pair of streams that was seen to overlap
    if seen:
        assert list(lat.env.streams) == list(pair)          # ... the process's one checked pair
    lat.close()


@pytest.mark.parametrize("tuning", [0, _lib.TUNE_SPLIT_APPLY])
def test_action_outside_the_candidate_rows_raises_bad_action(tuning):
    """candidates[action] (binPhy.py:235): a negative index counts from the end like any numpy index, anything else outside
    [0, S) is an IndexError in the reference -- IRBPP_DEVERR_BAD_ACTION in the step's error word here, from the fused apply
    phase and from irbpp_apply_kernel alike; never a silent clamp."""
    from irbpp_amd.vec_env import GpuPackingEnv
    sh = synthetic.blockout_shapes(n_shapes=8, n_rot=4, cube=0.06, seed=1)
    seqs = synthetic.make_sequences(sh.n_shapes, 16, 40, seed=2)
    a, b = (GpuPackingEnv(sh, seqs, 4, device=DEV, tuning=tuning) for _ in range(2))
    oa, ob = a.reset(), b.reset()
    act = a.policy_minz(oa)
    neg = act.clone(); neg[1] = -1                                         # row S - 1: a zero-padded row
    pos = act.clone(); pos[1] = S - 1
    ra, rb = a.step(neg), b.step(pos)
    for x, y in zip(ra, rb):
        assert torch.equal(x, y)
    a.step_info_host(); b.step_info_host()                                  # no error so far
    bad = a.policy_minz(ra[0]); bad[2] = S
    a.step(bad)
    with pytest.raises(_lib.IrbppError, match="BAD_ACTION"):
        a.step_info_host()
    bad = b.policy_minz(rb[0]); bad[0] = -S - 1
    b.step(bad)
    with pytest.raises(_lib.IrbppError, match="flags=64"):
        b.check_device_error()
    a.close(); b.close()


def test_order_action_outside_the_buffer_raises_bad_action():
    """next_k_item_ID[orderAction] (binPhy.py:163): a Python list index -- -1 is the last slot, k is an IndexError."""
    from irbpp_amd.vec_env import GpuPackingEnv
    sh = synthetic.blockout_shapes(n_shapes=8, n_rot=4, cube=0.06, seed=1)
    seqs = synthetic.make_sequences(sh.n_shapes, 16, 40, seed=2)
    k = 3
    env = GpuPackingEnv(sh, seqs, 4, device=DEV, bufferSize=k)
    env.reset()
    last = env.get_action_candidates(torch.full((4,), k - 1, dtype=torch.int32, device=DEV)).clone()
    wrap = env.get_action_candidates(torch.full((4,), -1, dtype=torch.int32, device=DEV))
    assert torch.equal(last, wrap)
    env.check_device_error()
    env.get_action_candidates(torch.tensor([0, k, 0, 0], dtype=torch.int32, device=DEV))
    with pytest.raises(_lib.IrbppError, match="BAD_ACTION"):
        env.check_device_error()
    env.close()


def test_step_right_after_set_heightmaps_drops_onto_the_new_map():
    """irbpp_set_heightmaps forgets the drop heights of the last observation (they belong to the old maps): a step that
    follows without a new observation recomputes posZmap[rot, lx, ly] on the new map (space.py:118-119)."""
    from irbpp_amd.vec_env import GpuPackingEnv
    sh = synthetic.cube_shapes()
    seqs = synthetic.make_sequences(sh.n_shapes, 16, 40, seed=2)
    for tuning in (0, _lib.TUNE_SPLIT_APPLY):
        env = GpuPackingEnv(sh, seqs, 3, device=DEV, tuning=tuning)
        obs = env.reset()
        item = obs[:, 5 * S].long().cpu().numpy()
        hm = torch.full((3, 32, 32), 0.05, dtype=torch.float64, device=DEV)
        hm[1, 0, 0] = 0.11                                                  # bin 1: one tall cell under the footprint
        env.set_heightmaps(hm)
        env.step(torch.zeros(3, dtype=torch.int32, device=DEV))            # row 0 of the empty-bin observation: rot 0, cell (0, 0)
        env.step_info_host()
        got = env.get_heightmaps().cpu().numpy()
        for b in range(3):
            T, _, mH, _ = sh.tables[item[b]][0]
            fx, fy = T.shape
            z = 0.11 if b == 1 else 0.05
            want = hm[b].cpu().numpy().copy()
            want[:fx, :fy] = np.maximum(want[:fx, :fy], (T + z) * mH)
            np.testing.assert_array_equal(got[b], want)
        env.close()


def test_graph_replay_follows_the_placement_log():
    """IRBPP_TUNE_GRAPH: captured kernel nodes carry the placement log's pointers by value, so irbpp_set_placement_log drops
    the cached graphs -- the log switched off, then moved to new buffers between replays: entries land where the direct
    launches of a second environment put them, and the old buffers stay as they were."""
    from irbpp_amd.vec_env import GpuPackingEnv
    sh = synthetic.blockout_shapes(n_shapes=8, n_rot=4, cube=0.06, seed=1)
    seqs = synthetic.make_sequences(sh.n_shapes, 16, 40, seed=2)
    n = 8
    g = GpuPackingEnv(sh, seqs, n, device=DEV, tuning=_lib.TUNE_GRAPH)
    d = GpuPackingEnv(sh, seqs, n, device=DEV)
    bufs = [[torch.empty((n, e.obs_len), dtype=torch.float32, device=DEV) for _ in range(2)] for e in (g, d)]
    og, od = g.reset(), d.reset()
    acts = [torch.empty((n,), dtype=torch.int32, device=DEV) for _ in range(2)]
    t = 0

    def steps(count):
        nonlocal og, od, t
        for _ in range(count):
            a = d.policy_minz(od)
            for ab in acts:
                ab.copy_(a)
            og = g.step(acts[0], obs_out=bufs[0][t & 1])[0]
            od = d.step(acts[1], obs_out=bufs[1][t & 1])[0]
            assert torch.equal(og, od)
            t += 1
    steps(6)                                                                # (replays from the third step on)
    lg, ld = g.enable_placement_log(64), d.enable_placement_log(64)
    steps(6)
    assert torch.equal(lg[0], ld[0]) and torch.equal(lg[1], ld[1]) and int((ld[0] != 0).sum()) > 0
    keep = (lg[0].clone(), lg[1].clone())
    for e in (g, d):
        _lib.check(e.lib.irbpp_set_placement_log(e._h, None, None, 0), "irbpp_set_placement_log")
    steps(6)
    assert torch.equal(lg[0], keep[0]) and torch.equal(lg[1], keep[1])     # switched off: a replay must not write the old buffers
    ng, nd = g.enable_placement_log(64), d.enable_placement_log(64)
    steps(6)
    assert torch.equal(ng[0], nd[0]) and torch.equal(ng[1], nd[1]) and int((nd[0] != 0).sum()) > 0
    assert torch.equal(lg[0], keep[0])
    for e in (g, d):
        e.check_device_error()
        e.close()


def test_isolated_rectangles_of_every_size_through_the_split_pipeline():
    """Round 6, IRBPP_TUNE_RECT: the transition kernel answers level-image components that are isolated solid rectangles itself
    (contours_device.h: rect_component / rect_vertices) instead of handing their borders to the trace kernel.  Level images
    packed with non-touching rectangles of every width and height up to 16 -- plus near-rectangles (a corner knocked out, a
    pixel stuck on diagonally) that must still be followed -- through the whole pipeline by way of the heightmap of a bin that
    observes a one-cell item, default build and IRBPP_TUNE_RECT side by side, every location observation against the oracle."""
    from irbpp_amd import _lib
    from oracle.packing import OracleVecEnv
    sh = _unit_item_shapes()
    seqs = np.zeros((8, 80), dtype=np.int32)
    n, k = 8, 2
    genvs = [GpuVecEnv(sh, seqs, n, device=DEV, bufferSize=k, tuning=f) for f in (0, _lib.TUNE_RECT)]
    for g in genvs:
        g.candidates_on_device = True
    oenv = OracleVecEnv(n, sh, seqs, bufferSize=k)
    ref = _f32(oenv.reset())
    for g in genvs:
        np.testing.assert_array_equal(g.reset().cpu().numpy(), ref)
    rng = np.random.RandomState(77)
    sizes = [(w, h) for w in range(1, 17) for h in range(1, 17)]
    rng.shuffle(sizes)
    seen = set()
    pos = 0
    for t in range(40):
        hm = np.zeros((n, 32, 32))
        for i in range(n):
            img = np.zeros((16, 16), dtype=bool)
            occ = np.zeros((18, 18), dtype=bool)                       # the rectangles dilated by one pixel: no two may touch
            for _ in range(24):
                w, h = sizes[pos % len(sizes)] if rng.rand() < 0.6 else (rng.randint(1, 5), rng.randint(1, 5))
                if w > 16 or h > 16:
                    continue
                x0, y0 = rng.randint(0, 17 - w), rng.randint(0, 17 - h)
                if occ[y0:y0 + h + 2, x0:x0 + w + 2].any():
                    continue
                img[y0:y0 + h, x0:x0 + w] = True
                occ[y0:y0 + h + 2, x0:x0 + w + 2] = True
                seen.add((w, h))
                pos += 1
            if (t + i) % 3 == 1:
                img ^= rng.rand(16, 16) < 0.015                        # near-rectangles
            lv = np.where(img, 0.05, 0.11) + rng.randint(0, 2, size=(16, 16)) * np.where(img, 0.0, 0.03)
            hm[i] = np.kron(lv, np.ones((2, 2)))
        for g in genvs:
            g.env.set_heightmaps(torch.from_numpy(hm).to(DEV))
        for i in range(n):
            oenv.envs[i].space.heightmapC[:] = hm[i]
        oa = np.array([t % k] * n)
        oloc = _f32(oenv.get_action_candidates(oa))
        for g in genvs:
            np.testing.assert_array_equal(g.get_action_candidates(oa).cpu().numpy(), oloc, err_msg=f"round {t}")
        act = np.array([int(np.argmin(np.where(c.reshape(S, 5)[:, 4] == 1, c.reshape(S, 5)[:, 3], np.inf)))
                        if (c.reshape(S, 5)[:, 4] == 1).any() else 0 for c in oloc[:, :5 * S]])
        oord, _, odone, _ = oenv.step(act)
        for g in genvs:
            gord, _, gdone, _ = g.step(act)
            np.testing.assert_array_equal(gord.cpu().numpy(), _f32(oord))
            np.testing.assert_array_equal(gdone, odone)
    assert len(seen) >= 200, len(seen)                                 # (big ones rarely find room beside others: they come alone)
    for g in genvs:
        g.env.check_device_error()
        g.close()
