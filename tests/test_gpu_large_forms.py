"""The launch forms the BENCH runs -- irbpp_apply_kernel in front of the transition kernel in MODE_OBSERVE, the
wave-per-bin emit kernel, 64 candidates per trace wave, the wave-per-bin buffered step, the 512-thread transition
builds, two groups on two streams -- at the launch sizes that select them BY THEMSELVES (no irbpp_config::tuning),
against (1) the episodes the reference's own PackingGame played (tests/golden/bench_*.npz), replayed by every one of
thousands of bins at once, and (2) the plain-C oracle on EVERY bin, whole episodes with auto-resets.
(binPhy.py:183-337, space.py:98-129; the N = 1 forms of the same comparisons are in test_gpu_parity.py.)"""
import os

import numpy as np
import pytest
import torch

import irbpp_amd  # noqa: F401
from irbpp_amd.vec_env import GpuPackingEnv, GpuVecEnv
from helpers import golden_kwargs, golden_scenario

pytestmark = pytest.mark.gpu
S = 500
DEV = "cuda:0"
THREADS = min(16, os.cpu_count() or 1)


def _f32(x):
    return np.asarray(x, dtype=np.float64).astype(np.float32)


def _replay_table(seq, n_bins, episodes):
    """A trajectory table in which every one of n_bins bins replays the episodes the reference's single environment
    played: the reference's episode e reads row (1 + e) % len(seq) (IRcreator.py:86-92); global bin g reads row
    (1 + g + e * n_bins) % n_traj in its episode e (include/irbpp.h: traj_start = 1)."""
    n_traj = 1 + n_bins * (episodes + 2)
    table = np.zeros((n_traj, seq.shape[1]), dtype=np.int32)
    for e in range(episodes + 2):
        table[1 + e * n_bins:1 + (e + 1) * n_bins] = seq[(1 + e) % len(seq)]
    return table


def _fallback_rows(n_rot):
    return np.array([[c // 256, (c % 256) // 16, c % 16, 0.30, 0.0] for c in range(S)]).astype(np.float32)


@pytest.mark.parametrize("name,n,kernels", [
    ("bench_blockout_r4", 8192, ("irbpp_env_kernel_s1", "irbpp_trace_kernel +", "irbpp_emit_wave_kernel_s1", "irbpp_apply_kernel in front")),
    ("bench_blockout_r8", 4096, ("irbpp_env_kernel_s5", "irbpp_trace_kernel +", "irbpp_emit_wave_kernel_s5")),
    ("bench_general", 4096, ("irbpp_env_kernel_s3", "irbpp_trace_kernel +", "irbpp_emit_kernel_s3")),
    ("bench_abc_fine", 4096, ("irbpp_env_kernel_s4_w512c", "irbpp_emit_kernel_s4"))])
def test_reference_goldens_replayed_by_every_bin_of_a_large_launch(golden_dir, name, n, kernels):
    """Every bin of a full-width launch replays the reference-recorded episode: each step's observation of EVERY bin must
    equal the recording (compared on the device), and reward / done / counter / ratio / episode reward too."""
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    sh = golden_scenario(name)
    env = GpuPackingEnv(sh, _replay_table(g["seq"], n, int(g["done"].sum())), n, device=DEV, **golden_kwargs(name))
    info = env.kernel_info()[1]
    for kname in kernels:
        assert kname in info, (kname, info)
    ref = torch.from_numpy(_f32(g["obs"])).to(DEV)
    obs = env.reset()
    assert torch.equal(obs, ref[0].expand(n, -1))
    fb = torch.from_numpy(_fallback_rows(sh.n_rot).reshape(-1)).to(DEV)
    act = torch.empty((n,), dtype=torch.int32, device=DEV)
    fallbacks = 0
    for t in range(len(g["act"])):
        act.fill_(int(g["act"][t]))
        obs, rew, done = env.step(act)
        h = env.step_info_host()
        assert (h["done"] == bool(g["done"][t])).all() and (h["reward"].astype(np.float32) == np.float32(g["rew"][t])).all(), t
        if g["done"][t]:
            assert (h["counter"] == g["counter"][t]).all() and (h["ratio"] == g["ratio"][t]).all()
            assert (h["ep_reward"] == h["ep_reward"][0]).all() and round(float(h["ep_reward"][0]), 6) == g["ep_r"][t]
        r = ref[t + 1]
        if bool((r[:5 * S].reshape(S, 5)[:, 4] == 1).any()):
            assert torch.equal(obs, r.expand(n, -1)), f"step {t}: {int((obs != r).any(dim=1).sum())} bins differ from the recording"
        else:                                       # no-candidate fallback: the reference's row order is its numpy build's (binPhy.py:217-225)
            assert torch.equal(obs[:, 5 * S:], r[5 * S:].expand(n, -1)) and torch.equal(obs[:, :5 * S], fb.expand(n, -1))
            fallbacks += 1
    assert fallbacks >= 1
    env.check_device_error()
    env.close()


@pytest.mark.parametrize("n", [1024, 4096])
def test_reference_hierarchical_golden_replayed_by_every_bin(golden_dir, n):
    """BASELINE config 4 (k = 10) as the reference played it, by every bin of a launch at its per-GPU width (1024 bins:
    irbpp_apply_wg_kernel, 32 candidates per trace wave) and at 4096 bins (a wave per bin: irbpp_apply_kernel alone)."""
    name, k = "bench_blockout_k10", 10
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    env = GpuPackingEnv(golden_scenario(name), _replay_table(g["seq"], n, int(g["done"].sum())), n, device=DEV, bufferSize=k)
    info = env.kernel_info()[1]
    assert ("irbpp_apply_wg_kernel alone" if n < 2048 else "irbpp_apply_kernel alone") in info, info
    assert ("irbpp_trace_kernel_c32" in info) == (n <= 1024), info
    order_ref = torch.from_numpy(_f32(g["order_obs"])).to(DEV)
    loc_ref = torch.from_numpy(_f32(g["loc_obs"])).to(DEV)
    order = env.reset()
    assert torch.equal(order, order_ref[0].expand(n, -1))
    oa = torch.empty((n,), dtype=torch.int32, device=DEV)
    act = torch.empty((n,), dtype=torch.int32, device=DEV)
    for t in range(len(g["act"])):
        oa.fill_(int(g["order_act"][t]))
        loc = env.get_action_candidates(oa)
        r = loc_ref[t]
        assert torch.equal(loc[:, 5 * S:], r[5 * S:].expand(n, -1)), t
        if bool((r[:5 * S].reshape(S, 5)[:, 4] == 1).any()):
            assert torch.equal(loc, r.expand(n, -1)), f"placement {t}: {int((loc != r).any(dim=1).sum())} bins differ from the recording"
        act.fill_(int(g["act"][t]))
        order, rew, done = env.step(act)
        h = env.step_info_host()
        assert (h["done"] == bool(g["done"][t])).all() and (h["reward"].astype(np.float32) == np.float32(g["rew"][t])).all(), t
        if g["done"][t]:
            assert (h["counter"] == g["counter"][t]).all() and (h["ratio"] == g["ratio"][t]).all()
        assert torch.equal(order, order_ref[t + 1].expand(n, -1)), t
    env.check_device_error()
    env.close()


@pytest.mark.parametrize("workload,n,groups,steps,min_episodes", [
    ("blockout", 4096, 1, 130, 2048), ("blockout", 8192, 2, 40, 0), ("cube", 4096, 1, 40, 2048), ("general", 2048, 2, 30, 2048),
    ("blockout_k10", 2048, 1, 130, 512), ("blockout_k10", 1024, 1, 40, 0), ("abc_fine", 4096, 1, 12, 1024),
    ("blockout_r8", 2048, 1, 130, 512)])
def test_every_bin_of_a_large_launch_vs_c_oracle(workload, n, groups, steps, min_episodes):
    """ALL bins of the bench's workloads against the plain-C oracle (tools/soak_parity.py as a test): every observation,
    reward and done flag of every bin at every step, through auto-resets, as one launch group and as two on two streams."""
    from bench import make_workload
    from oracle.c_oracle import COracleVecEnv
    shapes, seqs, kw = make_workload(workload)
    k = int(kw.get("bufferSize", 1))
    seqs = seqs[:2000]
    genv = GpuVecEnv(shapes, seqs, n, device=DEV, num_groups=groups, **kw)
    genv.candidates_on_device = True
    info = (genv.env.groups[0] if genv.num_groups > 1 else genv.env).kernel_info()[1]
    per_launch = n // genv.num_groups
    if workload in ("blockout", "cube"):
        assert "irbpp_emit_wave_kernel" in info and "irbpp_trace_kernel +" in info, info
        assert ("irbpp_apply_kernel in front" in info) == (per_launch >= 4096), info
    if workload == "abc_fine":
        assert "irbpp_env_kernel_s4_w512c" in info, info
    if workload == "blockout_r8":
        assert "irbpp_env_kernel_s5" in info, info             # rotations 0 .. 3 on the block path, 4 .. 7 on their cell lists
    cenv = COracleVecEnv(n, shapes, seqs, threads=THREADS, **kw)
    gobs = genv.reset()
    np.testing.assert_array_equal(gobs.cpu().numpy(), _f32(cenv.reset()))
    episodes = 0
    for t in range(steps):
        if k > 1:
            order = (np.arange(n) * 3 + t) % k
            gloc = genv.get_action_candidates(order)
            if genv.num_groups > 1:
                genv.env.synchronize()
            np.testing.assert_array_equal(gloc.cpu().numpy(), _f32(cenv.get_action_candidates(order)), err_msg=f"location obs, placement {t}")
            act = genv.env.policy_minz(gloc).cpu().numpy()
        else:
            act = genv.env.policy_minz(gobs).cpu().numpy()
        gobs, grew, gdone, _ = genv.step(act)
        cobs, crew, cdone, _ = cenv.step(act)
        bad = (gobs.cpu().numpy() != _f32(cobs)).any(axis=1)
        assert not bad.any(), f"step {t}: bins {np.nonzero(bad)[0][:8]} differ from the C oracle"
        np.testing.assert_array_equal(gdone, cdone)
        np.testing.assert_array_equal(grew.numpy()[:, 0], crew.astype(np.float32))
        episodes += int(cdone.sum())
    assert episodes >= min_episodes, episodes
    genv.env.check_device_error()
    genv.close()
