"""the vectorised replay memory against (1) fixtures produced by the reference's
own memory.py (tests/golden/make_replay_golden.py) and (2) the per-env restatement in
oracle/replay.py on random streams.  Integer/index results, stored states and tree sums
bit-exact (float32 sums of the same two operands); n-step returns and importance weights within
1e-6 (a 3-term float32 dot product and a pow whose evaluation order the reference leaves to
torch)."""
import os

import numpy as np
import pytest
import torch

import irbpp_amd  # noqa: F401
from irbpp_amd.replay import VectorReplayMemory, actor_step, mask_from_state
from oracle.replay import ReplayMemory as OracleReplay

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _replay_golden(name, device):
    g = np.load(os.path.join(GOLD, f"replay_{name}.npz"))
    n_envs, capacity, obs_len, n_step, steps, segment = (int(x) for x in g["meta"])
    mem = VectorReplayMemory(n_envs, capacity, obs_len, discount=0.99, multi_step=n_step, priority_weight=0.4,
                             priority_exponent=0.5, device=device)
    oracle = [OracleReplay(capacity, obs_len, 0.99, n_step, 0.4, 0.5) for _ in range(n_envs)]
    checkpoints = set(int(c) for c in g["checkpoints"])
    for t in range(steps):
        mem.append(torch.from_numpy(g["states"][t]), torch.from_numpy(g["actions"][t]), torch.from_numpy(g["rewards"][t]),
                   torch.from_numpy(g["terminals"][t]), torch.from_numpy(g["valid"][t]))
        for i in range(n_envs):
            if g["valid"][t][i]:
                oracle[i].append(g["states"][t][i], g["actions"][t][i], g["rewards"][t][i], bool(g["terminals"][t][i]))
        if t + 1 not in checkpoints:
            continue
        c = f"c{t + 1}_"
        mem.anneal(0.1)
        for o in oracle:
            o.priority_weight = min(o.priority_weight + 0.1, 1)
        np.testing.assert_array_equal(mem.sum_tree.cpu().numpy(), g[c + "tree_before_update"])
        tree_idxs, states, actions, returns, next_states, nonterminals, weights = \
            mem.sample(segment, values=torch.from_numpy(g[c + "values"]))
        np.testing.assert_array_equal(tree_idxs.cpu().numpy(), g[c + "tree_idxs"])
        np.testing.assert_array_equal(states.cpu().numpy(), g[c + "states"])
        np.testing.assert_array_equal(actions.cpu().numpy(), g[c + "actions"])
        np.testing.assert_allclose(returns.cpu().numpy(), g[c + "returns"], rtol=0, atol=1e-6)
        np.testing.assert_array_equal(next_states.cpu().numpy(), g[c + "next_states"])
        assert tuple(nonterminals.shape) == (n_envs * segment, 1)
        np.testing.assert_array_equal(nonterminals.cpu().numpy().reshape(-1), g[c + "nonterminals"])
        np.testing.assert_allclose(weights.cpu().numpy(), g[c + "weights"], rtol=0, atol=1e-6)
        # the restatement agrees with the reference too (it is what the random test below leans on)
        for i, o in enumerate(oracle):
            ti, st, ac, re, ns, nt, w = o.sample_at(g[c + "values"][i])
            sl = slice(i * segment, (i + 1) * segment)
            assert ti == list(g[c + "tree_idxs"][i])
            np.testing.assert_array_equal(st, g[c + "states"][sl])
            np.testing.assert_array_equal(ac, g[c + "actions"][sl])
            np.testing.assert_allclose(re, g[c + "returns"][sl], rtol=0, atol=1e-6)
            np.testing.assert_array_equal(ns, g[c + "next_states"][sl])
            np.testing.assert_array_equal(nt, g[c + "nonterminals"][sl])
            np.testing.assert_allclose(w, g[c + "weights"][sl], rtol=0, atol=1e-6)
            o.update_priorities(g[c + "tree_idxs"][i], g[c + "loss"][i])
        # numpy's float32 pow (what the reference calls) and torch.pow agree to the last bit or the one before
        powered = np.power(g[c + "loss"], np.float32(0.5))
        np.testing.assert_allclose(torch.pow(torch.from_numpy(g[c + "loss"]).to(device), 0.5).cpu().numpy(), powered,
                                   rtol=2e-7, atol=0)
        mem.update_priorities(tree_idxs, torch.from_numpy(powered).reshape(-1), powered=True)
        np.testing.assert_array_equal(mem.sum_tree.cpu().numpy(), g[c + "tree_after_update"])
        np.testing.assert_array_equal(np.stack([o.transitions.sum_tree for o in oracle]), g[c + "tree_after_update"])
        np.testing.assert_allclose(mem.max.cpu().numpy(), g[c + "max"], rtol=0, atol=0)
        np.testing.assert_array_equal(mem.index.cpu().numpy(), g[c + "index"])
        np.testing.assert_array_equal(mem.full.cpu().numpy(), g[c + "full"])


@pytest.mark.parametrize("name", ["pow2", "odd"])
def test_vector_replay_matches_reference_fixtures(name):
    _replay_golden(name, "cpu")


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["pow2", "odd"])
def test_vector_replay_matches_reference_fixtures_on_device(name):
    _replay_golden(name, "cuda:0")


@pytest.mark.parametrize("capacity,n_step,seed", [(8, 1, 0), (13, 2, 1), (32, 5, 2)])
def test_vector_replay_random_streams_vs_restatement(capacity, n_step, seed):
    rng = np.random.RandomState(seed)
    # two segments per env: a segment can then never lie wholly inside the n+1 leaves next to the write
    # index, where the reference's rejection loop (memory.py:170-176) would spin forever as well
    n_envs, obs_len, B = 6, 4, 2
    mem = VectorReplayMemory(n_envs, capacity, obs_len, multi_step=n_step, device="cpu")
    oracle = [OracleReplay(capacity, obs_len, 0.99, n_step) for _ in range(n_envs)]
    gen = torch.Generator().manual_seed(seed)
    for t in range(5 * capacity):
        state = rng.uniform(0, 0.3, size=(n_envs, obs_len)).astype(np.float32)
        action, reward = rng.randint(0, 500, size=n_envs), rng.uniform(0, 1, size=n_envs).astype(np.float32)
        terminal, valid = rng.rand(n_envs) < 0.2, rng.rand(n_envs) < 0.8
        mem.append(torch.from_numpy(state), torch.from_numpy(action), torch.from_numpy(reward)[:, None],
                   torch.from_numpy(terminal), torch.from_numpy(valid))
        for i in range(n_envs):
            if valid[i]:
                oracle[i].append(state[i], action[i], reward[i], bool(terminal[i]))
        np.testing.assert_array_equal(mem.sum_tree.numpy(), np.stack([o.transitions.sum_tree for o in oracle]))
        if t % 7 == 6 and all(o.transitions.full for o in oracle):
            # self-drawn sample: every draw must satisfy the reference's validity test, and looking the same
            # positions up in the restatement must give the same batch
            tree_idxs, states, actions, returns, next_states, nonterminals, weights = mem.sample(B, generator=gen)
            prob, data_idx, _ = (x.numpy() for x in (mem.sum_tree[torch.arange(n_envs)[:, None], tree_idxs],
                                                     tree_idxs - (capacity - 1), tree_idxs))
            for i, o in enumerate(oracle):
                assert all(o.valid(p, int(j)) for p, j in zip(prob[i], data_idx[i]))
                sl = slice(i * B, (i + 1) * B)
                rows = [o.transition(int(j)) for j in data_idx[i]]
                np.testing.assert_array_equal(states.numpy()[sl], np.stack([r[0] for r in rows]))
                np.testing.assert_array_equal(actions.numpy()[sl], np.array([r[1] for r in rows]))
                np.testing.assert_allclose(returns.numpy()[sl], np.array([r[2] for r in rows]), rtol=0, atol=1e-6)
                np.testing.assert_array_equal(next_states.numpy()[sl], np.stack([r[3] for r in rows]))
                np.testing.assert_array_equal(nonterminals.numpy()[sl, 0], np.array([r[4] for r in rows]))
            loss = rng.uniform(0.5, 1.5, size=(n_envs, B)).astype(np.float32)
            mem.update_priorities(tree_idxs, torch.from_numpy(np.power(loss, np.float32(0.5))), powered=True)
            for i, o in enumerate(oracle):
                o.update_priorities(tree_idxs.numpy()[i], loss[i])      # duplicates: last one wins in both
            np.testing.assert_array_equal(mem.sum_tree.numpy(), np.stack([o.transitions.sum_tree for o in oracle]))
            np.testing.assert_array_equal(mem.max.numpy(), np.array([o.transitions.max for o in oracle], dtype=np.float32))


def test_find_is_the_reference_descent():
    mem = VectorReplayMemory(2, 11, 3, device="cpu")
    rng = np.random.RandomState(5)
    pr = rng.uniform(0.1, 2.0, size=(2, 11)).astype(np.float32)
    ora = [OracleReplay(11, 3) for _ in range(2)]
    for j in range(11):
        mem._set_leaves(torch.arange(2), torch.full((2,), j + 10), torch.from_numpy(pr[:, j]))
        for i in range(2):
            ora[i].transitions.update(j + 10, pr[i, j])
    vals = rng.uniform(0, 1, size=(2, 200)) * mem.total().numpy()[:, None]
    p, d, t = mem.find(torch.from_numpy(vals))
    for i in range(2):
        ref = [ora[i].transitions.find(v) for v in vals[i]]
        np.testing.assert_array_equal(t.numpy()[i], [r[2] for r in ref])
        np.testing.assert_array_equal(d.numpy()[i], [r[1] for r in ref])
        np.testing.assert_array_equal(p.numpy()[i], [r[0] for r in ref])


def test_mask_and_sample_guard():
    s = torch.arange(2 * (5 * 4 + 3), dtype=torch.float32).reshape(2, 23)
    assert mask_from_state(s, 4).tolist() == [[4.0, 9.0, 14.0, 19.0], [27.0, 32.0, 37.0, 42.0]]   # tools.py:298-299
    mem = VectorReplayMemory(2, 8, 3, device="cpu")
    with pytest.raises(RuntimeError):
        mem.sample(2)                               # nothing appended yet: every draw has probability 0


@pytest.mark.gpu
def test_actor_step_fills_the_memory_from_the_device_env():
    """trainer.py:160-186 as tensor ops: act -> step -> clip -> append for 64 bins, no host loop."""
    from irbpp_amd import synthetic
    from irbpp_amd.vec_env import GpuPackingEnv
    shapes = synthetic.blockout_shapes(16, seed=3)
    seqs = synthetic.make_sequences(16, n_traj=80, length=60, seed=4)
    env = GpuPackingEnv(shapes, seqs, 64, device="cuda:0")
    mem = VectorReplayMemory(64, 32, env.obs_len, multi_step=3, device="cuda:0")
    state = env.reset()
    policy = lambda s, m: env.policy_minz(s).to(torch.int64)      # noqa: E731  stands for Agent.act
    seen = []
    for _ in range(40):
        prev = state
        state, reward, done = actor_step(env, policy, mem, state, reward_clip=0.5)
        seen.append((prev.clone(), reward.clone(), done.clone()))
    assert bool(mem.full.all()) and int(mem.index[0]) == 40 % 32
    # the last 32 transitions sit in the ring in order
    for back in range(1, 6):
        pos = (40 - back) % 32
        s, r, d = seen[-back]
        assert torch.equal(mem.states[:, pos], s)
        assert torch.equal(mem.rewards[:, pos], r.clamp(-0.5, 0.5))
        assert torch.equal(mem.nonterminals[:, pos], ~d.to(torch.bool))
    batch = mem.sample(2)
    assert tuple(batch[1].shape) == (128, env.obs_len) and bool(torch.isfinite(batch[6]).all())
    mem.update_priorities(batch[0], torch.rand(128, device="cuda:0") + 0.1)
    env.check_device_error()


@pytest.mark.gpu
@pytest.mark.parametrize("capacity,n_step,seed", [(8, 1, 0), (13, 2, 1), (64, 3, 2)])
def test_hip_sum_tree_kernels_equal_the_torch_formulation(capacity, n_step, seed):
    """csrc/irbpp_replay.hip (one launch per find / update) against the torch formulation of the same memory on the
    same device, random streams with partial appends, duplicate leaves in update_priorities and self-drawn samples:
    trees, maxima, sampled indices and batches bit-identical."""
    rng = np.random.RandomState(seed)
    n_envs, obs_len, B = 33, 5, 4
    hip = VectorReplayMemory(n_envs, capacity, obs_len, multi_step=n_step, device="cuda:0")
    ref = VectorReplayMemory(n_envs, capacity, obs_len, multi_step=n_step, device="cuda:0", use_hip=False)
    assert hip._lib is not None and ref._lib is None
    sampled = [0]
    for t in range(4 * capacity):
        state = torch.from_numpy(rng.uniform(0, 0.3, size=(n_envs, obs_len)).astype(np.float32)).cuda()
        action = torch.from_numpy(rng.randint(0, 500, size=n_envs)).cuda()
        reward = torch.from_numpy(rng.uniform(0, 1, size=n_envs).astype(np.float32)).cuda()
        terminal = torch.from_numpy(rng.rand(n_envs) < 0.2).cuda()
        valid = torch.from_numpy(rng.rand(n_envs) < (1.0 if t % 3 else 0.7)).cuda()
        ref.append(state, action, reward, terminal, valid)
        if t % 2:            # irbpp_replay_append takes what the environment hands out: int32 actions, float64 rewards, uint8 flags
            hip.append(state, action.to(torch.int32), reward.to(torch.float64), terminal.to(torch.uint8), valid.to(torch.uint8))
        else:
            hip.append(state, action, reward, terminal, None if bool(valid.all()) else valid)
        for name in ("sum_tree", "max", "states", "actions", "rewards", "nonterminals", "timesteps", "index", "full", "t"):
            assert torch.equal(getattr(hip, name), getattr(ref, name)), (name, t)
        if t % 5 == 4 and bool(ref.full.all()):
            vals = (torch.rand((n_envs, B), device="cuda:0") * ref.total()[:, None]).clamp(min=1e-6)
            for a, b in zip(hip.find(vals), ref.find(vals)):
                assert torch.equal(a, b)
            pr0, di0, _ = ref.find(vals)
            ok = ref._valid(pr0, di0)
            if bool(ok.any(dim=1).all()):                                # every env has a drawable position: use it for
                first = ok.float().argmax(dim=1)                         # the draws the reference would reject
                vals = torch.where(ok, vals, vals[torch.arange(n_envs, device="cuda:0"), first][:, None])
                pr0, di0, _ = ref.find(vals)
            if bool(ref._valid(pr0, di0).all()):                         # a drawable batch: the fused gather kernel too
                a, b = hip.sample(B, values=vals), ref.sample(B, values=vals)
                for k in (0, 1, 2, 4, 5):                                # tree idx, states, actions, next states, non-terminal
                    assert torch.equal(a[k], b[k]), k
                assert torch.allclose(a[3], b[3], rtol=0, atol=1e-6) and torch.allclose(a[6], b[6], rtol=0, atol=1e-6)
                sampled[0] += 1
            idx = ref.find(vals)[2]
            idx[:, 1] = idx[:, 0]                                        # a leaf listed twice: the last value wins
            pr = torch.from_numpy(rng.uniform(0.2, 2.0, size=(n_envs, B)).astype(np.float32)).cuda()
            for m in (hip, ref):
                m.update_priorities(idx, pr, powered=True)
            assert torch.equal(hip.sum_tree, ref.sum_tree) and torch.equal(hip.max, ref.max)
    assert sampled[0] >= 1 or capacity <= 8


@pytest.mark.gpu
def test_masked_greedy_action_is_agent_act():
    """agent.py:55-58 with get_mask_from_state (tools.py:298-299): -inf where the candidate flag is 0, argmax."""
    from irbpp_amd.replay import masked_greedy_action
    rng = np.random.RandomState(3)
    n, S = 257, 500
    state = np.zeros((n, 5 * S + 9 + 1024), dtype=np.float32)
    flags = rng.rand(n, S) < 0.3
    flags[5] = False                                                  # no valid candidate at all
    flags[6] = True
    state[:, :5 * S].reshape(n, S, 5)[:, :, 4] = flags
    q = rng.randn(n, S).astype(np.float32)
    q[7, 10] = q[7, 400] = 9.0                                        # a tie: the first maximum
    flags[7, [10, 400]] = True
    state[7, :5 * S].reshape(S, 5)[:, 4] = flags[7]
    got = masked_greedy_action(torch.from_numpy(q).cuda(), torch.from_numpy(state).cuda(), S).cpu().numpy()
    sum_q = torch.from_numpy(q).clone()
    sum_q[(1 - mask_from_state(torch.from_numpy(state), S)).bool()] = -float("inf")       # the reference's two lines
    want = sum_q.argmax(1).numpy()
    np.testing.assert_array_equal(got, want)
    assert got[5] == 0 and got[7] == 10
    cpu = masked_greedy_action(torch.from_numpy(q), torch.from_numpy(state), S).numpy()   # torch formulation (CPU)
    np.testing.assert_array_equal(cpu, want)


@pytest.mark.gpu
@pytest.mark.parametrize("capacity,n_step", [(16, 1), (64, 3)])
def test_fused_sampling_kernel_draws_valid_stratified_positions(capacity, n_step):
    """irbpp_sumtree_sample (draw + tree walk + the rejection loop of memory.py:170-176 in one launch): every
    position satisfies the reference's validity test, lies in its own segment of the priority mass, the same seed
    gives the same draws, and an empty memory reports failure instead of looping."""
    rng = np.random.RandomState(capacity)
    n_envs, obs_len, B = 37, 4, 8
    mem = VectorReplayMemory(n_envs, capacity, obs_len, multi_step=n_step, device="cuda:0")
    with pytest.raises(RuntimeError):
        mem.sample(2)                                   # nothing appended yet
    for t in range(3 * capacity):
        mem.append(torch.from_numpy(rng.uniform(0, 0.3, size=(n_envs, obs_len)).astype(np.float32)).cuda(),
                   torch.from_numpy(rng.randint(0, 500, size=n_envs)).cuda(),
                   torch.from_numpy(rng.uniform(0, 1, size=n_envs).astype(np.float32)).cuda(),
                   torch.from_numpy(rng.rand(n_envs) < 0.1).cuda())
        if t % 9 == 8 and bool(mem.full.all()):
            idx = mem.find((torch.rand((n_envs, B), device="cuda:0") * mem.total()[:, None]).clamp(min=1e-6))[2]
            mem.update_priorities(idx, torch.rand((n_envs, B), device="cuda:0") + 0.05)
    g1, g2 = torch.Generator().manual_seed(5), torch.Generator().manual_seed(5)
    a, b = mem.sample(B, generator=g1), mem.sample(B, generator=g2)
    assert torch.equal(a[0], b[0])                      # same seed, same positions
    tree_idx = a[0]
    prob = mem.sum_tree[torch.arange(n_envs, device="cuda:0")[:, None], tree_idx]
    data_idx = tree_idx - (capacity - 1)
    assert bool(mem._valid(prob, data_idx).all())
    # stratification: the prefix mass in front of leaf j's position lies in segment j (up to the leaf's own mass)
    leaves = mem.sum_tree[:, capacity - 1:]
    before = torch.cumsum(leaves, dim=1) - leaves
    seg = mem.total()[:, None] / B
    lo = torch.arange(B, device="cuda:0")[None, :] * seg
    start = torch.gather(before, 1, data_idx)
    assert bool((start <= lo + seg + 1e-3).all()) and bool((start + prob >= lo - 1e-3).all())


@pytest.mark.gpu
def test_append_kernel_serves_capacities_beyond_the_sum_tree_kernels():
    """A ring of more than 8192 transitions per env takes the sum-tree calls off the HIP kernels (their LDS row; warned once) --
    irbpp_replay_append walks the leaf's ancestors in global memory and still serves: every tensor equal to the torch formulation,
    through a wrap of the ring (capacity not a power of two)."""
    import warnings
    rng = np.random.RandomState(11)
    n_envs, obs_len, cap = 5, 7, 8200
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", RuntimeWarning)
        hip = VectorReplayMemory(n_envs, cap, obs_len, device="cuda:0")
    ref = VectorReplayMemory(n_envs, cap, obs_len, device="cuda:0", use_hip=False)
    assert hip._lib is None and hip._append_hip and not ref._append_hip
    for t in range(cap + 200):
        state = torch.from_numpy(rng.uniform(0, 0.3, size=(n_envs, obs_len)).astype(np.float32)).cuda()
        action = torch.from_numpy(rng.randint(0, 500, size=n_envs).astype(np.int32)).cuda()
        reward = torch.from_numpy(rng.uniform(0, 1, size=n_envs)).cuda()                  # float64, as the environment hands out
        terminal = torch.from_numpy((rng.rand(n_envs) < 0.1).astype(np.uint8)).cuda()
        valid = None if t % 41 else torch.from_numpy(rng.rand(n_envs) < 0.6).cuda()
        hip.append(state, action, reward, terminal, valid)
        ref.append(state, action, reward, terminal, valid)
        if t % 997 == 0 or t >= cap - 3:
            for name in ("sum_tree", "max", "states", "actions", "rewards", "nonterminals", "timesteps", "index", "full", "t"):
                assert torch.equal(getattr(hip, name), getattr(ref, name)), (name, t)
    assert bool(ref.full.all()) and int(ref.index.min()) > 0             # (every ring has wrapped)
