"""Parity of the grouped-stepping product API (vec_env.GroupedPackingEnv, GpuVecEnv(num_groups=...),
step_async(group=g) / step_wait(group=g)): every observation, reward, done and info equal to the C oracle AND to one
ungrouped GpuVecEnv, through step(), interleaved per-group stepping, get_action_candidates and a reset_specific that
spans groups (shmem_vec_env.py:70-81,99-102,113-117; binPhy.py:161-169)."""
import numpy as np
import pytest
import torch

import irbpp_amd  # noqa: F401
from irbpp_amd.vec_env import GpuVecEnv, GroupedPackingEnv
from oracle.c_oracle import COracleVecEnv

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
S = 500


def _f32(x):
    return np.asarray(x, dtype=np.float64).astype(np.float32)


def _check_infos(ginfo, cinfo, cdone, lo=0):
    for i in np.nonzero(cdone)[0]:
        gi, ci = ginfo[int(i)], cinfo[lo + int(i)]
        assert gi["counter"] == ci["counter"] and gi["ratio"] == ci["ratio"]
        assert gi["episode"]["r"] == ci["episode"]["r"] and gi["episode"]["l"] == ci["episode"]["l"]


@pytest.mark.parametrize("workload,n,steps", [("blockout", 64, 150), ("general", 32, 50), ("blockout_k10", 32, 150)])
def test_grouped_stepping_matches_c_oracle(workload, n, steps):
    from bench import make_workload
    shapes, seqs, kw = make_workload(workload)
    seqs = seqs[:600]
    k = int(kw.get("bufferSize", 1))
    G = 4
    per = n // G
    grouped = GpuVecEnv(shapes, seqs, n, device=DEV, num_groups=G, **kw)
    single = GpuVecEnv(shapes, seqs, n, device=DEV, **kw)
    grouped.candidates_on_device = single.candidates_on_device = True
    cenv = COracleVecEnv(n, shapes, seqs, **kw)
    gobs, sobs, cobs = grouped.reset(), single.reset(), _f32(cenv.reset())
    np.testing.assert_array_equal(gobs.cpu().numpy(), cobs)
    np.testing.assert_array_equal(sobs.cpu().numpy(), cobs)
    ndone = 0
    for t in range(steps):
        if k > 1:
            order = (np.arange(n) * 5 + t) % k
            gloc = grouped.get_action_candidates(order)
            sloc = single.get_action_candidates(order)
            cloc = _f32(cenv.get_action_candidates(order))
            grouped.env.synchronize()
            np.testing.assert_array_equal(gloc.cpu().numpy(), cloc, err_msg=f"grouped location obs, step {t}")
            np.testing.assert_array_equal(sloc.cpu().numpy(), cloc)
            act = single.env.policy_minz(sloc).cpu().numpy()
            np.testing.assert_array_equal(grouped.env.policy_minz(gloc).cpu().numpy(), act)      # the grouped env's policy over all bins
        else:
            act = single.env.policy_minz(sobs).cpu().numpy()
            grouped.env.synchronize()
            np.testing.assert_array_equal(grouped.env.policy_minz(gobs).cpu().numpy(), act)
        cobs, crew, cdone, cinfo = cenv.step(act)
        cobs = _f32(cobs)
        sobs, srew, sdone, sinfo = single.step(act)
        if t % 3 == 2:
            # per-group stepping in an interleaved order: two groups in flight, the others started after the first waits
            order_g = [(t + j) % G for j in range(G)]
            grouped.step_async(act[order_g[0] * per:(order_g[0] + 1) * per], group=order_g[0])
            grouped.step_async(torch.from_numpy(act[order_g[1] * per:(order_g[1] + 1) * per]).to(DEV), group=order_g[1])
            with pytest.raises(RuntimeError):
                grouped.step_async(act[order_g[0] * per:(order_g[0] + 1) * per], group=order_g[0])
            parts = {}
            parts[order_g[1]] = grouped.step_wait(group=order_g[1])
            grouped.step_async(act[order_g[2] * per:(order_g[2] + 1) * per], group=order_g[2])
            parts[order_g[0]] = grouped.step_wait(group=order_g[0])
            grouped.step_async(act[order_g[3] * per:(order_g[3] + 1) * per], group=order_g[3])
            parts[order_g[3]] = grouped.step_wait(group=order_g[3])
            parts[order_g[2]] = grouped.step_wait(group=order_g[2])
            with pytest.raises(RuntimeError):
                grouped.step_wait(group=0)
            gobs = torch.cat([parts[g][0] for g in range(G)])
            grew = torch.cat([parts[g][1] for g in range(G)])
            gdone = np.concatenate([parts[g][2] for g in range(G)])
            for g in range(G):
                _check_infos(parts[g][3], cinfo, cdone[g * per:(g + 1) * per], lo=g * per)
        else:
            gobs, grew, gdone, ginfo = grouped.step(act)
            _check_infos(ginfo, cinfo, cdone)
        np.testing.assert_array_equal(gobs.cpu().numpy(), cobs, err_msg=f"grouped obs, step {t}")
        np.testing.assert_array_equal(sobs.cpu().numpy(), cobs, err_msg=f"ungrouped obs, step {t}")
        np.testing.assert_array_equal(gdone, cdone)
        np.testing.assert_array_equal(sdone, cdone)
        np.testing.assert_array_equal(grew.numpy()[:, 0], crew.astype(np.float32))
        np.testing.assert_array_equal(srew.numpy()[:, 0], crew.astype(np.float32))
        _check_infos(sinfo, cinfo, cdone)
        ndone += int(cdone.sum())
        if t == steps // 3:
            # reset_specific across group borders: last bin of group 0, first of group 1, one of group 3, in this order
            idxs = [per, per - 1, n - 2]
            gsub, ssub, csub = grouped.reset_specific(idxs), single.reset_specific(idxs), _f32(cenv.reset_specific(idxs))
            np.testing.assert_array_equal(gsub.cpu().numpy(), csub)
            np.testing.assert_array_equal(ssub.cpu().numpy(), csub)
            if k == 1:                               # the trainer-side pattern: patch the rows of the state it holds
                for j, i in enumerate(idxs):
                    gobs[i] = gsub[j]
                    sobs[i] = ssub[j]
                    cobs[i] = csub[j]
    hm_g = grouped.env.get_heightmaps().cpu().numpy()
    hm_s = single.env.get_heightmaps().cpu().numpy()
    np.testing.assert_array_equal(hm_g, hm_s)
    tg, ts = grouped.env.episode_totals().cpu().numpy(), single.env.episode_totals().cpu().numpy()
    assert tg[0] == ts[0] and tg[2] == ts[2]                 # episodes and items exactly; the float sums are added up in
    np.testing.assert_allclose(tg, ts, rtol=1e-13, atol=0)   # another order (per group, then over the groups)
    grouped.env.check_device_error()
    single.env.check_device_error()
    grouped.close()
    single.close()
    assert ndone >= n // 2


def test_grouped_env_survives_dropped_action_temporaries():
    """ADVICE r2 (cross-stream lifetime): the action tensor handed to GroupedPackingEnv.step is a temporary the caller
    drops at once, and the caching allocator is then asked for same-sized blocks that it fills with garbage on the
    current stream.  The groups' kernels run on their own streams: they must still read the actions that were passed
    (record_stream), and must not start before the policy output exists (wait_stream)."""
    from bench import make_workload
    shapes, seqs, kw = make_workload("blockout")
    n = 256
    env = GroupedPackingEnv(shapes, seqs[:400], n, 4, device=DEV, **kw)
    ref = GroupedPackingEnv(shapes, seqs[:400], n, 1, device=DEV, **kw)
    obs, robs = env.reset(), ref.reset()
    assert torch.equal(obs, robs)
    for t in range(60):
        act = ref.groups[0].policy_minz(robs)
        robs = ref.step(act.clone())
        # a temporary int32 copy, produced on the current stream right before the call and dropped right after it
        obs = env.step((act.to(torch.int64) + 0).to(torch.int32))
        junk = [torch.full((n,), 499, dtype=torch.int32, device=DEV) for _ in range(8)]     # reuse of the freed block
        del junk
        env.synchronize()
        assert torch.equal(obs, robs), f"step {t}"
    env.check_device_error()
    env.close()
    ref.close()
