"""GPU parity of the round-3 additions around the hot path: hierarchical evaluation against the golden of the
reference's own tools.test_hierachical, the training-time item streams behind make_vec_envs against the oracle's
restatement of the random item creators, lattice data through both overlap paths, the registered-buffer lifetime
calls."""
import os
import types

import numpy as np
import pytest
import torch

import irbpp_amd  # noqa: F401
from irbpp_amd import _lib, synthetic
from irbpp_amd.vec_env import GpuPackingEnv, GpuVecEnv, make_vec_envs
from oracle.packing import OracleVecEnv, RandomStreamItemCreator
from helpers import minz_action

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
S = 500


def _f32(x):
    return np.asarray(x, dtype=np.float64).astype(np.float32)


def test_hierarchical_evaluation_matches_the_reference_golden(golden_dir, tmp_path):
    """evaluate(bufferSize=k, order_policy=...) against tools.test_hierachical's own trajs.npy and statistics
    (tests/golden/make_golden.py: tools_test_hier.npz; tools.py:361-431)."""
    from irbpp_amd.evaluate import evaluate
    g = np.load(os.path.join(golden_dir, "tools_test_hier.npz"))
    k = int(g["k"])
    sh = synthetic.blockout_shapes(n_shapes=20, n_rot=4, cube=0.06, seed=7)
    path = str(tmp_path / "logs" / "evaluation" / "run" / "trajs.npy")
    out = evaluate(sh, g["seq"], len(g["ep_len"]), device=DEV, save=path, bufferSize=k,
                   order_policy=lambda env, order_obs: order_obs[:, :k].argmin(dim=1))
    assert out["episodes"] == len(g["ep_len"]) and out["unfinished"] == 0
    trajs = np.load(path, allow_pickle=True)
    assert len(trajs) == len(g["ep_len"])
    row = 0
    for ep, want_len in zip(trajs, g["ep_len"]):
        assert len(ep) == want_len
        for i, (item, name, pos, quat) in enumerate(ep):
            assert item == g["ids"][row] and name == g["names"][row]
            if i < want_len - 1 or g["pos"][row][2] < 1e3:     # (a refused "action 0" with no valid candidate: see test_gpu_parity)
                np.testing.assert_allclose(pos, g["pos"][row], rtol=0, atol=1e-12)
                np.testing.assert_allclose(quat, g["quat"][row], rtol=0, atol=1e-12)
            row += 1
    assert row == len(g["ids"])
    assert abs(out["avg_reward"] - float(g["avg_reward"])) < 1e-9 and out["avg_length"] == float(g["avg_length"])


@pytest.mark.parametrize("k,sample,groups,tuning", [(1, "instance", 1, 0), (3, "instance", 1, 0), (1, "category", 1, 0), (2, "pose", 1, 0),
                                                    (1, "instance", 3, 0), (3, "category", 2, 0),
                                                    (1, "instance", 1, _lib.TUNE_SPLIT_APPLY), (2, "pose", 2, _lib.TUNE_SPLIT_APPLY)])
def test_make_vec_envs_trains_on_the_reference_item_streams(k, sample, groups, tuning):
    """make_vec_envs(args) with nothing but the reference's namespace (no args.sequences): every environment draws its
    items like the reference's worker of that rank (RandomInstanceCreator / RandomCateCreator / RandomItemCreator on
    np.random seeded seed + rank; IRcreator.py:26-72, envs.py:41).  The oracle side restates the creators on numpy's
    RandomState itself; the product side is csrc/irbpp_itemgen.h + the per-bin item rings of stream mode, here with
    a ring of 64 items so that the feeder has to refill it many times."""
    sh = synthetic.blockout_shapes(n_shapes=20, n_rot=4, cube=0.06, seed=7)
    if sample == "instance":
        dic = {i: "%s_%d.obj" % (["tee", "ell", "bar", "zig"][i % 4], i // 4) for i in range(20)}
    elif sample == "category":
        dic = {i: "%s/%d.obj" % (["objects", "concave", "board"][(i * 5) % 3], i) for i in range(20)}
    else:
        dic = {i: "%d.obj" % i for i in range(20)}
    n, seed = 6, 321
    args = types.SimpleNamespace(
        num_processes=n, device=0, seed=seed, shapes=sh, dicPath=dic, dataSample=sample, resolutionA=0.02,
        resolutionH=0.01, resolutionZ=0.01, bin_dimension=np.round([0.32, 0.32, 0.30], 6), selectedAction=S,
        bufferSize=k, scale=[100, 100, 100], evaluate=False, item_ring=64, num_groups=groups, tuning=tuning)   # (tuning: the ring's
                                                               # consume-and-mark in irbpp_apply_kernel as well as in the transition kernel)
    envs, spaces, obs_len = make_vec_envs(args, "./logs/runinfo", True)
    assert envs.num_groups == groups              # grouped stepping: group g is fed rows [g*per, (g+1)*per) on its own stream
    envs.candidates_on_device = True
    creators = [RandomStreamItemCreator(seed + i, dic, sample, n_items=20) for i in range(n)]
    oenv = OracleVecEnv(n, sh, None, item_creators=creators, bufferSize=k)
    gobs, oobs = envs.reset(), _f32(oenv.reset())
    np.testing.assert_array_equal(gobs.cpu().numpy(), oobs)
    ndone, refills = 0, 0
    written0 = envs.feeder.written.copy()
    for t in range(170):
        if k > 1:
            order = (np.arange(n) + t) % k
            gloc = envs.get_action_candidates(order)
            oloc = _f32(oenv.get_action_candidates(order))
            np.testing.assert_array_equal(gloc.cpu().numpy(), oloc)
        else:
            oloc = oobs
        act = np.array([minz_action(o, S) for o in oloc])
        gobs, grew, gdone, ginfo = envs.step(act)
        oobs, orew, odone, oinfo = oenv.step(act)
        oobs = _f32(oobs)
        np.testing.assert_array_equal(gobs.cpu().numpy(), oobs, err_msg=f"step {t}")
        np.testing.assert_array_equal(gdone, odone)
        ndone += int(odone.sum())
        if t == 60:                                     # a mid-episode reset(): queues are dropped, the streams go on
            gobs, oobs = envs.reset(), _f32(oenv.reset())
            np.testing.assert_array_equal(gobs.cpu().numpy(), oobs)
        if t == 100:                                    # ... and a per-env reset
            sub = envs.reset_specific([4, 1])
            ref = _f32(oenv.reset_specific([4, 1]))
            np.testing.assert_array_equal(sub.cpu().numpy(), ref)
            for j, i in enumerate([4, 1]):
                gobs[i] = sub[j]
                oobs[i] = ref[j]
    refills = int(((envs.feeder.written - written0) > 0).sum())
    envs.env.check_device_error()
    envs.close()
    assert ndone >= 6 and refills == n and int((envs.feeder.written - written0).min()) > 64    # the rings went round


def test_lattice_data_agrees_on_both_overlap_paths():
    """irbpp_config::tuning = IRBPP_TUNE_NO_BLOCK_PATH plays BlockOut through the generic overlap test (and the
    unconstrained transition kernel): every observation equals the block path's."""
    from bench import make_workload
    shapes, seqs, kw = make_workload("blockout")
    n = 128
    a = GpuPackingEnv(shapes, seqs[:300], n, device=DEV, **kw)
    b = GpuPackingEnv(shapes, seqs[:300], n, device=DEV, tuning=_lib.TUNE_NO_BLOCK_PATH, **kw)
    assert a.lib.irbpp_overlap_path(a._h) == 1 and b.lib.irbpp_overlap_path(b._h) == 3
    assert a.kernel_info()[1].startswith("irbpp_env_kernel_s1 +") and "irbpp_env_kernel_generic" in b.kernel_info()[1]
    oa, ob = a.reset(), b.reset()
    assert torch.equal(oa, ob)
    for t in range(140):
        act = a.policy_minz(oa)
        oa, ra, da = a.step(act)
        ob, rb, db = b.step(act)
        assert torch.equal(oa, ob), f"step {t}"
        assert torch.equal(ra, rb) and torch.equal(da, db)
    ids = torch.arange(n, dtype=torch.int32, device=DEV) % shapes.n_shapes
    pa, ma = a.possible_position(ids)
    pb, mb = b.possible_position(ids)
    assert torch.equal(pa, pb) and torch.equal(ma, mb)
    a.check_device_error()
    b.check_device_error()
    a.close()
    b.close()


def test_registered_buffer_lifetime_calls():
    """irbpp_unregister_obs_buffer / irbpp_invalidate_obs_buffer / registering an address again: a buffer
    the caller scribbled over, or a new allocation at an old address, delivers complete observations again."""
    from bench import make_workload
    shapes, seqs, kw = make_workload("blockout")
    n = 64
    a = GpuPackingEnv(shapes, seqs[:300], n, device=DEV, **kw)
    b = GpuPackingEnv(shapes, seqs[:300], n, device=DEV, **kw)
    buf = torch.zeros((n, a.loc_obs_len), dtype=torch.float32, device=DEV)
    a.register_obs_buffer(buf)
    oa, ob = a.reset(), b.reset()
    act = b.policy_minz(ob)
    for t in range(40):
        if t % 10 == 5:
            buf.fill_(3.25)                              # the caller writes into the registered buffer ...
            a.invalidate_obs_buffers(buf)                # ... and says so
        if t == 20:
            a.unregister_obs_buffer(buf)
            buf.fill_(-1.0)                              # unregistered: plain full writes
        if t == 30:
            buf.fill_(9.0)
            a.register_obs_buffer(buf)                   # the same address registered again: contents unknown
        oa, _, _ = a.step(act, obs_out=buf)
        ob, _, _ = b.step(act)
        assert torch.equal(oa, ob), f"step {t}"
        act = b.policy_minz(ob)
    lib = a.lib
    import ctypes as C
    assert lib.irbpp_unregister_obs_buffer(a._h, C.c_void_p(ob.data_ptr())) == -1      # never registered
    a.close()
    b.close()


def test_item_grouped_launch_order_changes_nothing():
    """Online steps of a large generic data set launch the bins grouped by observed item per die
    (irbpp_item_order_kernel; irbpp_config::tuning = IRBPP_TUNE_NO_ITEM_ORDER keeps the index order): every
    observation, reward and done flag is the same either way, through auto-resets."""
    from bench import make_workload
    shapes, seqs, kw = make_workload("abc_fine")
    n = 512
    a = GpuPackingEnv(shapes, seqs[:700], n, device=DEV, **kw)
    b = GpuPackingEnv(shapes, seqs[:700], n, device=DEV, tuning=_lib.TUNE_NO_ITEM_ORDER, **kw)
    oa, ob = a.reset(), b.reset()
    assert torch.equal(oa, ob)
    done_total = 0
    for t in range(45):
        act = a.policy_minz(oa)
        oa, ra, da = a.step(act)
        ob, rb, db = b.step(act)
        assert torch.equal(oa, ob), f"step {t}"
        assert torch.equal(ra, rb) and torch.equal(da, db)
        done_total += int(da.sum())
    assert done_total > n
    a.check_device_error()
    b.check_device_error()
    a.close()
    b.close()


@pytest.mark.parametrize("workload,n,steps", [("blockout", 160, 120), ("general", 96, 45)])
def test_trace_launch_shapes_change_nothing(workload, n, steps):
    """The launch shapes the library can take -- 64 / 32 / 16 candidate starts per trace wave, borders approximated by the
    polygon kernel or inside the trace kernel, speckled bins first in the emit kernel or in launch order -- forced one by
    one through irbpp_config::tuning: every observation, reward and done flag equals the default's, through auto-resets."""
    from bench import make_workload
    shapes, seqs, kw = make_workload(workload)
    flags = [_lib.TUNE_TRACE_CPW64, _lib.TUNE_TRACE_CPW32, _lib.TUNE_TRACE_CPW16,
             _lib.TUNE_INLINE_POLYGON, _lib.TUNE_TRACE_CPW16 | _lib.TUNE_INLINE_POLYGON, _lib.TUNE_NO_HEAVY_FIRST,
             _lib.TUNE_TRACE_REFILL, _lib.TUNE_TRACE_REFILL | _lib.TUNE_INLINE_POLYGON,      # (lane refill: opt-in, measured slower)
             _lib.TUNE_CHAIN]                                                                # (one kernel per observation: opt-in, measured slower)
    envs = [GpuPackingEnv(shapes, seqs[:400], n, device=DEV, **kw)] + \
           [GpuPackingEnv(shapes, seqs[:400], n, device=DEV, tuning=f, **kw) for f in flags]
    obs = [e.reset() for e in envs]
    for o in obs[1:]:
        assert torch.equal(o, obs[0])
    done_total = 0
    for t in range(steps):
        act = envs[0].policy_minz(obs[0])
        res = [e.step(act) for e in envs]
        for j, (o, r, d) in enumerate(res[1:]):
            assert torch.equal(o, res[0][0]), f"step {t}, tuning {flags[j]}"
            assert torch.equal(r, res[0][1]) and torch.equal(d, res[0][2])
        obs = [r[0].clone() for r in res]
        done_total += int(res[0][2].sum())
    assert done_total > n // 2
    for e in envs:
        e.check_device_error()
        e.close()


@pytest.mark.parametrize("workload,n,steps,spec", [("blockout", 160, 130, "_s1"), ("cube", 128, 60, "_s2"), ("general", 96, 45, "_s3"),
                                                   ("abc_fine", 64, 40, "_s4_w512"), ("blockout_k10", 96, 120, "_s1"),
                                                   ("blockout_r8", 96, 100, "_s5")])
def test_specialised_builds_and_split_apply_change_nothing(workload, n, steps, spec):
    """BASELINE.json's geometries run builds of the transition and emit kernels that have the grid sizes, LDS offsets and
    division constants as compile-time literals (irbpp_device.h: SPEC_KEYS), and a step applies its actions in
    irbpp_apply_kernel (a wave per bin) in front of the transition kernel.  IRBPP_TUNE_NO_SPECIALISED forces the builds that
    read Params, IRBPP_TUNE_NO_WG512 / _WG512 the generic path's 256- / 512-thread workgroups, IRBPP_TUNE_FUSED_APPLY the round-4
    form (actions applied inside the transition kernel), IRBPP_TUNE_BLOCK_EMIT
    the emit kernel with a workgroup per bin where lattice / box data take the one with a wave per bin.  Same observations,
    rewards, done flags, step outputs and heightmaps through whole episodes -- with the scripted policy, and with actions
    drawn at random over all S rows for some bins (zero-padded rows: the drop height is then recomputed, not looked up)."""
    from bench import make_workload
    shapes, seqs, kw = make_workload(workload)
    k = int(kw.get("bufferSize", 1))
    flags = [0, _lib.TUNE_NO_SPECIALISED | _lib.TUNE_SPLIT_APPLY | _lib.TUNE_WAVE_EMIT, _lib.TUNE_FUSED_APPLY | _lib.TUNE_BLOCK_EMIT,
             _lib.TUNE_SPLIT_APPLY | _lib.TUNE_WAVE_EMIT | _lib.TUNE_GRAPH, _lib.TUNE_NO_SPECIALISED | _lib.TUNE_FUSED_APPLY | _lib.TUNE_BLOCK_EMIT,
             _lib.TUNE_NO_WG512, _lib.TUNE_WG512,          # (256- / 512-thread workgroups of the generic path whatever the data)
             _lib.TUNE_NARROW_KERNEL | _lib.TUNE_WG512,    # (the 512-thread build under the 64-VGPR cap: the default from 4096 bins on)
             _lib.TUNE_NO_MIXED_PATH,                      # (BlockOut at eight rotations through the cell lists entirely, as until round 5)
             _lib.TUNE_CHAIN,                              # (ONE kernel per observation: contour stage and candidate rows in the bin's workgroup)
             _lib.TUNE_WG128, _lib.TUNE_WG128 | _lib.TUNE_SPLIT_APPLY,     # (two waves per bin: BlockOut at R = 4 only)
             _lib.TUNE_RECT]                               # (isolated solid rectangles answered by the transition kernel, not followed)
    envs = [GpuPackingEnv(shapes, seqs[:400], n, device=DEV, tuning=f, **kw) for f in flags]
    names = [e.kernel_info()[1].split(" + ")[0] for e in envs]
    assert names[0].endswith(spec) and names[2] == names[3] == names[0], names
    assert names[1] == names[4] and "_s" not in names[1].replace("irbpp_env_kernel", ""), names
    assert "w512" not in names[5] and (("w512" in names[6]) == ("generic" in names[4] or "_s3" in names[0] or "_s4" in names[0])), names
    assert names[7] == ("irbpp_env_kernel_s4_w512c" if workload == "abc_fine" else names[7].replace("w512", "")), names
    assert names[8] == ("irbpp_env_kernel_s3" if workload == "blockout_r8" else names[0]), names
    chain = "irbpp_env_kernel_chain_s1 alone" if spec == "_s1" else "irbpp_env_kernel_chain alone"
    assert names[9].startswith(chain) == (workload != "abc_fine"), names      # (not for the 40 KB tile)
    assert names[10] == names[11] == ("irbpp_env_kernel_s1_w128" if spec == "_s1" else names[0]), names
    obs = [e.reset() for e in envs]
    assert all(torch.equal(obs[0], o) for o in obs[1:])
    gen = torch.Generator(device="cpu").manual_seed(5)
    bufs = [[torch.empty_like(obs[0]), torch.empty_like(obs[0])] for _ in envs]
    lbufs = [torch.empty((n, e.loc_obs_len), dtype=torch.float32, device=DEV) for e in envs]
    abufs = [torch.empty((n,), dtype=torch.int32, device=DEV) for _ in envs]
    slots = [torch.full((n,), j, dtype=torch.int32, device=DEV) for j in range(max(k, 1))]
    done_total = 0
    for t in range(steps):
        if k > 1:
            slot = slots[t % k]
            loc = [e.get_action_candidates(slot, obs_out=lbufs[j]) for j, e in enumerate(envs)]
            assert all(torch.equal(loc[0], x) for x in loc[1:]), f"location observation, step {t}"
            act = envs[0].policy_minz(loc[0])
        else:
            act = envs[0].policy_minz(obs[0])
        if t % 7 == 3:                                       # some bins act at random over all S rows, padding included
            rnd = torch.randint(0, S, (n,), generator=gen, dtype=torch.int32).to(DEV)
            pick = (torch.arange(n, device=DEV) % 4) == (t % 4)
            act = torch.where(pick, rnd, act)
        # (ping-pong observation buffers: the environment with IRBPP_TUNE_GRAPH sees every argument set again and replays
        # its steps as HIP graphs from the third step on)
        for ab in abufs:
            ab.copy_(act)
        res = [e.step(abufs[j], obs_out=bufs[j][t & 1]) for j, e in enumerate(envs)]
        info = [e.step_info_host() for e in envs]
        for j in range(1, len(envs)):
            for x, y in zip(res[0], res[j]):
                assert torch.equal(x, y), f"step {t}, env {j}"
            for key in info[0]:
                np.testing.assert_array_equal(info[0][key], info[j][key], err_msg=f"{key}, step {t}, env {j}")
        obs = [r[0].clone() for r in res]
        done_total += int(res[0][2].sum())
    hm = [e.get_heightmaps() for e in envs]
    assert all(torch.equal(hm[0], h) for h in hm[1:])
    assert done_total > 0
    for e in envs:
        e.check_device_error()
        e.close()


def test_wave_emit_serves_bins_that_need_the_workgroup():
    """The wave-per-bin emit kernel (lattice / box data) hands a bin with more than S candidates -- or with valid cells but
    no candidate -- to its workgroup (radix select + sort over LDS).  With S = 40 on the Cube set that happens at most
    steps: every observation against the C oracle and against the workgroup-per-bin emit kernel."""
    from oracle.c_oracle import COracleVecEnv
    sh = synthetic.cube_shapes()
    seqs = synthetic.make_sequences(sh.n_shapes, 64, 60, seed=77)
    n, s_sel = 10, 40
    wave = GpuPackingEnv(sh, seqs, n, device=DEV, selectedAction=s_sel, tuning=_lib.TUNE_WAVE_EMIT)
    block = GpuPackingEnv(sh, seqs, n, device=DEV, selectedAction=s_sel, tuning=_lib.TUNE_BLOCK_EMIT)
    cenv = COracleVecEnv(n, sh, seqs, selectedAction=s_sel)
    ow, ob, oc = wave.reset(), block.reset(), _f32(cenv.reset())
    full = 0
    for t in range(60):
        assert torch.equal(ow, ob), f"step {t}"
        np.testing.assert_array_equal(ow.cpu().numpy(), oc, err_msg=f"step {t}")
        full += int((ow[:, :5 * s_sel].reshape(n, s_sel, 5)[:, :, 4] == 1).all(dim=1).sum())
        act = torch.from_numpy(np.array([minz_action(o, s_sel) for o in oc], dtype=np.int32)).to(DEV)
        rw, rb = wave.step(act), block.step(act)
        for x, y in zip(rw, rb):
            assert torch.equal(x, y)
        oc = _f32(cenv.step(act.cpu().numpy())[0])
        ow, ob = rw[0].clone(), rb[0].clone()
    assert full > 20, full                       # observations whose S rows are all candidates: the selection ran
    for e in (wave, block):
        e.check_device_error()
        e.close()


@pytest.mark.parametrize("n,groups", [(5, 1), (16, 2)])
def test_get_all_possible_observation_matches_k_single_calls_and_the_oracle(n, groups):
    """PackingGame.get_all_possible_observation (binPhy.py:171-180): the location observation of EVERY buffer slot of every
    bin in one call ([N, k * (5S+9+Hc)]), against k calls of get_action_candidates and against the oracle's restatement;
    and its side effects as the reference has them: a step that follows indexes the LAST slot's candidate rows and pops the
    slot chosen before."""
    from oracle.packing import OracleVecEnv
    sh = synthetic.blockout_shapes(n_shapes=24, n_rot=4, cube=0.06, seed=0)
    seqs = synthetic.make_sequences(sh.n_shapes, 64, 150, seed=5)
    k = 4
    genv = GpuVecEnv(sh, seqs, n, device=DEV, bufferSize=k, num_groups=groups)
    genv.candidates_on_device = True
    oenv = OracleVecEnv(n, sh, seqs, bufferSize=k)
    np.testing.assert_array_equal(genv.reset().cpu().numpy(), _f32(oenv.reset()))
    L = 5 * S + 9 + 1024
    done_total = 0
    for t in range(50 if n < 8 else 24):
        every = genv.get_all_possible_observation()
        if groups > 1:
            genv.env.synchronize()
        every = every.cpu().numpy()
        assert every.shape == (n, k * L)
        np.testing.assert_array_equal(every, _f32(oenv.get_all_possible_observation()), err_msg=f"placement {t}")
        for j in range(k):
            single = genv.get_action_candidates(np.full(n, j))
            if groups > 1:
                genv.env.synchronize()
            np.testing.assert_array_equal(single.cpu().numpy(), every[:, j * L:(j + 1) * L], err_msg=f"slot {j}, placement {t}")
            oenv.get_action_candidates(np.full(n, j))
        # (the single calls left slot k - 1 chosen on both sides)
        if t % 3 == 2:
            # a step straight after get_all_possible_observation: the last slot's rows, the slot chosen before
            every = genv.get_all_possible_observation().reshape(n, k, L)
            oenv.get_all_possible_observation()
            loc = every[:, k - 1]
        else:
            oa = (np.arange(n) + t) % k
            loc = genv.get_action_candidates(oa)
            oenv.get_action_candidates(oa)
        act = genv.env.policy_minz(loc.contiguous()).cpu().numpy()
        gobs, grew, gdone, _ = genv.step(act)
        oobs, orew, odone, _ = oenv.step(act)
        np.testing.assert_array_equal(gobs.cpu().numpy(), _f32(oobs), err_msg=f"order observation, placement {t}")
        np.testing.assert_array_equal(gdone, odone)
        np.testing.assert_array_equal(grew.numpy()[:, 0], orew.astype(np.float32))
        done_total += int(odone.sum())
    assert done_total >= (1 if n < 8 else 0)
    genv.env.check_device_error()
    genv.close()
