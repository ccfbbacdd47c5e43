"""Caller-side scaling: the reference keeps one ``ReplayMemory`` object per
environment process (main.py:61-63) and feeds/samples them in Python loops (trainer.py:184-186,
agent.py:69-75), each with its own recursive sum tree on the CPU (memory.py:15-100).  With
thousands of bins per GPU that loop is the bottleneck, so here the N memories are ONE set of
device tensors and every operation handles all environments at once:

    reference (per env i)                         here (all envs)
    mem[i].append(state[i], action[i], r[i], d[i])   append(state, action, reward, done, valid)
    mem[i].sample(segment_size)  (agent.py:72-75)    sample(segment_size)  -> env-major batch
    mem[i].update_priorities(idxs[i], loss[...])     update_priorities(tree_idxs, losses)

Semantics follow memory.py line by line (cited below): cyclic buffer + sum tree per env with
float32 node sums, new transitions enter with the env's maximum priority, stratified sampling
with the reference's validity test, n-step returns that blank everything after a terminal
transition, importance weights normalised per env.  On a HIP device the three sequential pieces -- the tree
descent of ``find``, the leaf-to-root update, and the masked greedy action of ``Agent.act`` -- are one kernel
launch each (csrc/irbpp_replay.hip through the C ABI); on the CPU (the CPU tests) the same results come from
the torch formulation kept below, which is also the fallback for capacities beyond the kernel's LDS row.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Tuple

import torch


def _hip_lib(device: torch.device):
    """The HIP library if ``device`` is a HIP device (raises like every product path if it is missing)."""
    if device.type != "cuda":
        return None
    from . import _lib
    return _lib.load()


def _p(t: Optional[torch.Tensor]):
    return C.c_void_p(0 if t is None else t.data_ptr())


_WARNED_TORCH_PATH = False


def _stream(device):
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


class VectorReplayMemory(object):
    def __init__(self, num_envs: int, capacity: int, obs_len: int, *, discount: float = 0.99, multi_step: int = 3,
                 priority_weight: float = 0.4, priority_exponent: float = 0.5, device="cpu",
                 state_dtype=torch.float32, use_hip: Optional[bool] = None):
        """``capacity`` is per environment (main.py:62: memory_capacity / num_processes)."""
        self.N, self.capacity, self.obs_len = int(num_envs), int(capacity), int(obs_len)
        self.device = torch.device(device)
        self.discount, self.n = float(discount), int(multi_step)
        self.priority_weight = float(priority_weight)        # beta, annealed by the trainer (trainer.py:196)
        self.priority_exponent = float(priority_exponent)    # omega
        d, N, cap = self.device, self.N, self.capacity
        # SegmentTree (memory.py:15-45): data arrays + sum tree, one row per env
        self.sum_tree = torch.zeros((N, 2 * cap - 1), dtype=torch.float32, device=d)
        self.timesteps = torch.zeros((N, cap), dtype=torch.int32, device=d)
        self.states = torch.zeros((N, cap, obs_len), dtype=state_dtype, device=d)
        self.actions = torch.zeros((N, cap), dtype=torch.int64, device=d)
        self.rewards = torch.zeros((N, cap), dtype=torch.float32, device=d)
        self.nonterminals = torch.zeros((N, cap), dtype=torch.bool, device=d)
        self.index = torch.zeros((N,), dtype=torch.int64, device=d)      # next write position
        self.full = torch.zeros((N,), dtype=torch.bool, device=d)
        self.max = torch.ones((N,), dtype=torch.float32, device=d)       # initial max priority 1 (memory.py:27)
        self.t = torch.zeros((N,), dtype=torch.int32, device=d)          # episode timestep counter (memory.py:111)
        self.n_step_scaling = torch.tensor([self.discount ** i for i in range(self.n)], dtype=torch.float32, device=d)
        self._rows = torch.arange(N, device=d)
        self._depth = max(1, (2 * cap - 1).bit_length())                 # >= height of the implicit tree
        # HIP kernels for find / update (one launch each) where the tree row fits their LDS buffer
        self._lib = _hip_lib(self.device) if use_hip in (None, True) else None
        self._append_hip = self._lib is not None              # (irbpp_replay_append walks one leaf's ancestors in global memory: any capacity)
        if use_hip and self._lib is None:
            raise RuntimeError("use_hip=True needs a HIP device")
        if self._lib is not None and 2 * cap - 1 > 16384:
            # said once per process, not silently: every sum-tree call of this memory is torch indexing from here on
            if use_hip:
                raise RuntimeError(f"use_hip=True: a sum tree of {2 * cap - 1} floats per env exceeds the HIP kernels' LDS row (16384)")
            global _WARNED_TORCH_PATH
            if not _WARNED_TORCH_PATH:
                import warnings
                warnings.warn(f"VectorReplayMemory: capacity {cap} per env (tree row of {2 * cap - 1} floats) exceeds the HIP sum-tree "
                              "kernels' LDS row of 16384 floats: using the torch formulation (several launches per call)", RuntimeWarning)
                _WARNED_TORCH_PATH = True
            self._lib = None

    # ------------------------------------------------------------------ sum tree ------------
    def _set_leaves(self, rows: torch.Tensor, tree_idx: torch.Tensor, value: torch.Tensor) -> None:
        """SegmentTree.update (memory.py:55-58) for one leaf per listed env: set, then recompute
        every ancestor as left + right in float32 (``_propagate``, :47-52)."""
        if self._lib is not None:
            from . import _lib
            N = self.N
            mask = None
            if rows.numel() != N:                                  # a subset of the envs: full-length arguments + mask
                mask = torch.zeros((N,), dtype=torch.uint8, device=self.device)
                mask[rows] = 1
                ti = torch.zeros((N,), dtype=torch.int64, device=self.device)
                ti[rows] = tree_idx
                va = torch.zeros((N,), dtype=torch.float32, device=self.device)
                va[rows] = value
                tree_idx, value = ti, va
            _lib.check(self._lib.irbpp_sumtree_update(_p(self.sum_tree), _p(self.max), N, self.capacity,
                                                      _p(tree_idx.contiguous()), _p(value.to(torch.float32).contiguous()), 1,
                                                      _p(mask), _stream(self.device)), "irbpp_sumtree_update")
            return
        self.sum_tree[rows, tree_idx] = value
        self.max[rows] = torch.maximum(self.max[rows], value)
        cap = self.capacity
        if cap & (cap - 1) == 0 and rows.numel() * 4 >= self.N:
            # power-of-two capacity: depth d of the implicit heap is the slice [2^d - 1, 2^(d+1) - 1), so every
            # level is one dense pairwise add over all envs -- the same left + right per node, far fewer launches
            width = cap // 2
            while width >= 1:
                lo = width - 1
                child = self.sum_tree[:, 2 * lo + 1: 4 * lo + 3].view(self.N, width, 2)
                self.sum_tree[:, lo: lo + width] = child[:, :, 0] + child[:, :, 1]
                width //= 2
            return
        idx = tree_idx
        for _ in range(self._depth):
            idx = torch.div(idx - 1, 2, rounding_mode="floor").clamp_(min=0)
            self.sum_tree[rows, idx] = self.sum_tree[rows, 2 * idx + 1] + self.sum_tree[rows, 2 * idx + 2]
        # (rows that reached the root early recompute it again: same two operands, same sum)

    def total(self) -> torch.Tensor:
        return self.sum_tree[:, 0]

    def find(self, values: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
        """SegmentTree.find / _retrieve (memory.py:72-86) for ``values[N, B]``:
        -> (priority, data index, tree index), each [N, B]."""
        N, B = values.shape
        if self._lib is not None:
            from . import _lib
            v = values.to(device=self.device, dtype=torch.float32).contiguous()
            prob = torch.empty((N, B), dtype=torch.float32, device=self.device)
            data_idx = torch.empty((N, B), dtype=torch.int64, device=self.device)
            tree_idx = torch.empty((N, B), dtype=torch.int64, device=self.device)
            _lib.check(self._lib.irbpp_sumtree_find(_p(self.sum_tree), N, self.capacity, _p(v), B, _p(prob), _p(data_idx),
                                                    _p(tree_idx), _stream(self.device)), "irbpp_sumtree_find")
            return prob, data_idx, tree_idx
        rows = self._rows[:, None].expand(N, B)
        idx = torch.zeros((N, B), dtype=torch.int64, device=self.device)
        v = values.to(torch.float32).clone()
        last = 2 * self.capacity - 2
        for _ in range(self._depth):
            left = 2 * idx + 1
            inner = left <= last                                   # `left >= len(sum_tree)` -> leaf
            lval = self.sum_tree[rows, left.clamp(max=last)]
            go_left = v <= lval
            nxt = torch.where(go_left, left, left + 1)
            v = torch.where(inner & ~go_left, v - lval, v)
            idx = torch.where(inner, nxt, idx)
        return self.sum_tree[rows, idx], idx - (self.capacity - 1), idx

    # ------------------------------------------------------------------ append --------------
    def _append_on_device(self, state, action, reward, terminal, valid) -> bool:
        """One launch of irbpp_replay_append for the whole call when the arguments are what an actor loop on the device
        hands over (float32 observation rows, int32 / int64 actions, float32 / float64 rewards, one-byte flags, all on
        this device) -- no conversion launches in front of it; False: the caller takes the torch formulation."""
        d = self.device
        if d.type != "cuda" or self.states.dtype != torch.float32:
            return False
        def on_dev(x, dtypes, n):        # noqa: E306
            return isinstance(x, torch.Tensor) and x.device == d and x.dtype in dtypes and x.numel() == n and x.is_contiguous()
        if not (isinstance(state, torch.Tensor) and state.device == d and state.dtype == torch.float32 and state.dim() == 2 and
                state.shape == (self.N, self.obs_len) and state.stride(1) == 1 and state.stride(0) >= self.obs_len):
            return False
        flags = (torch.bool, torch.uint8)
        if not (on_dev(action, (torch.int32, torch.int64), self.N) and on_dev(reward, (torch.float32, torch.float64), self.N) and
                on_dev(terminal, flags, self.N) and (valid is None or on_dev(valid, flags, self.N))):
            return False
        from . import _lib
        lib = self._lib if self._lib is not None else _lib.load()
        store = _lib.IrbppReplayStore(
            self.states.data_ptr(), self.actions.data_ptr(), self.rewards.data_ptr(), self.nonterminals.data_ptr(),
            self.timesteps.data_ptr(), self.sum_tree.data_ptr(), self.max.data_ptr(), self.index.data_ptr(), self.full.data_ptr(),
            self.t.data_ptr(), self.N, self.capacity, self.obs_len)
        _lib.check(lib.irbpp_replay_append(C.byref(store), _p(state), state.stride(0), _p(action), action.element_size(),
                                           _p(reward), reward.element_size(), _p(terminal), _p(valid), _stream(d)),
                   "irbpp_replay_append")
        return True

    def append(self, state: torch.Tensor, action: torch.Tensor, reward: torch.Tensor, terminal,
               valid: Optional[torch.Tensor] = None) -> None:
        """ReplayMemory.append (memory.py:117-121) for every env whose sample is valid
        (trainer.py:184-186): state/action at time t, reward/terminal at t+1; the new transition
        gets the env's maximum priority."""
        d = self.device
        if self._lib is not None or self._append_hip:
            if self._append_on_device(state, action, reward, terminal, valid):
                return
        terminal = torch.as_tensor(terminal, device=d).reshape(self.N).to(torch.bool)
        everyone = valid is None
        rows = self._rows if everyone else self._rows[torch.as_tensor(valid, device=d).reshape(self.N).to(torch.bool)]
        if rows.numel() == 0:
            return
        pick = (lambda x: x) if everyone else (lambda x: x[rows])
        pos = self.index[rows]
        self.timesteps[rows, pos] = self.t[rows]
        self.states[rows, pos] = pick(state.to(device=d, dtype=self.states.dtype))
        self.actions[rows, pos] = pick(action.to(d).reshape(self.N)).to(torch.int64)
        self.rewards[rows, pos] = pick(reward.to(d).reshape(self.N)).to(torch.float32)
        self.nonterminals[rows, pos] = ~pick(terminal)
        self._set_leaves(rows, pos + self.capacity - 1, self.max[rows])              # SegmentTree.append (:60-70)
        nxt = (pos + 1) % self.capacity
        self.index[rows] = nxt
        self.full[rows] = self.full[rows] | (nxt == 0)
        self.t[rows] = torch.where(terminal[rows], torch.zeros_like(self.t[rows]), self.t[rows] + 1)

    # ------------------------------------------------------------------ sample --------------
    def _valid(self, prob, data_idx):
        """memory.py:175: not straddling the write index, non-zero probability."""
        w = self.index[:, None]
        return ((w - data_idx) % self.capacity > self.n) & ((data_idx - w) % self.capacity >= 1) & (prob != 0)

    def _transitions(self, data_idx: torch.Tensor):
        """ReplayMemory._get_transition_new (memory.py:123-139) for data_idx[N, B]: the n+1
        consecutive transitions, blanked from the first one that follows a terminal transition."""
        N, B = data_idx.shape
        steps = torch.arange(self.n + 1, device=self.device)
        pos = (data_idx[:, :, None] + steps) % self.capacity                      # getBatch wraps (:89-91)
        rows = self._rows[:, None, None].expand(N, B, self.n + 1)
        nonterm = self.nonterminals[rows, pos]
        alive = torch.ones_like(nonterm)
        for t in range(1, self.n + 1):
            alive[:, :, t] = alive[:, :, t - 1] & nonterm[:, :, t - 1]
        state = self.states[self._rows[:, None].expand(N, B), pos[:, :, 0]]
        last = self.states[self._rows[:, None].expand(N, B), pos[:, :, self.n]]
        next_state = torch.where(alive[:, :, self.n, None], last, torch.zeros_like(last))
        rewards = torch.where(alive, self.rewards[rows, pos], torch.zeros((), device=self.device))
        action = self.actions[self._rows[:, None].expand(N, B), pos[:, :, 0]]
        returns = torch.matmul(rewards[:, :, :self.n], self.n_step_scaling)        # R^n (:186-188)
        nonterminal = (alive[:, :, self.n] & nonterm[:, :, self.n]).to(torch.float32)
        return state, action, returns, next_state, nonterminal

    def sample(self, segment_size: int, values: Optional[torch.Tensor] = None, generator=None, max_tries: int = 64):
        """ReplayMemory.sample (memory.py:194-204) on every env at once, concatenated env-major
        exactly as Agent.learn builds its batch (agent.py:69-84).

        ``values`` (optional, [N, segment_size]) are the tree positions to look up, as drawn by
        ``np.random.uniform(i*segment, (i+1)*segment)``; when omitted they are drawn here and
        invalid draws are redrawn like the reference's rejection loop (:170-176).
        Returns (tree_idxs [N,B], states, actions, returns, next_states, nonterminals [N*B,1], weights)."""
        N, B = self.N, int(segment_size)
        p_total = self.total()                                                    # [N]
        if values is not None or self._lib is None:
            segment = p_total / B
            lo = torch.arange(B, device=self.device, dtype=torch.float32)[None, :] * segment[:, None]
        if values is not None:
            prob, data_idx, tree_idx = self.find(values.to(self.device))
            if not bool(self._valid(prob, data_idx).all()):
                raise ValueError("a supplied sample position is invalid (memory.py:175)")
        elif self._lib is not None:
            # draw + tree walk + the rejection loop of memory.py:170-176 in ONE launch; the only host round trip is
            # the failure flag
            from . import _lib
            gdev = "cpu" if generator is None else generator.device          # a CPU generator costs no device round trip
            seed = int(torch.randint(0, 2 ** 62, (1,), generator=generator, device=gdev).item())
            prob = torch.empty((N, B), dtype=torch.float32, device=self.device)
            data_idx = torch.empty((N, B), dtype=torch.int64, device=self.device)
            tree_idx = torch.empty((N, B), dtype=torch.int64, device=self.device)
            failed = torch.zeros((1,), dtype=torch.int32, device=self.device)
            _lib.check(self._lib.irbpp_sumtree_sample(_p(self.sum_tree), _p(self.index), N, self.capacity, B, self.n, seed,
                                                      int(max_tries), _p(prob), _p(data_idx), _p(tree_idx), _p(failed),
                                                      _stream(self.device)), "irbpp_sumtree_sample")
            if int(failed.item()):
                raise RuntimeError("could not draw a valid sample from every segment; append more transitions first")
        else:
            draw = lambda: lo + torch.rand((N, B), device=self.device, generator=generator) * segment[:, None]  # noqa: E731
            prob, data_idx, tree_idx = self.find(draw())
            for _ in range(max_tries):
                bad = ~self._valid(prob, data_idx)
                if not bool(bad.any()):
                    break
                p2, d2, t2 = self.find(draw())
                prob, data_idx, tree_idx = torch.where(bad, p2, prob), torch.where(bad, d2, data_idx), torch.where(bad, t2, tree_idx)
            else:
                raise RuntimeError("could not draw a valid sample from every segment; append more transitions first")
        if self._lib is not None and self.states.dtype == torch.float32 and B <= 256:
            # one launch: transitions, n-step returns, next states, weights -- env-major rows as below
            from . import _lib
            dev, NB = self.device, N * B
            view = _lib.IrbppReplayView(
                states_dev=self.states.data_ptr(), actions_dev=self.actions.data_ptr(), rewards_dev=self.rewards.data_ptr(),
                nonterminals_dev=self.nonterminals.data_ptr(), tree_dev=self.sum_tree.data_ptr(), index_dev=self.index.data_ptr(),
                full_dev=self.full.data_ptr(), scaling_dev=self.n_step_scaling.data_ptr(), n_env=N, capacity=self.capacity,
                obs_len=self.obs_len, n_step=self.n)
            st = torch.empty((NB, self.obs_len), dtype=torch.float32, device=dev)
            nx = torch.empty((NB, self.obs_len), dtype=torch.float32, device=dev)
            ac = torch.empty((NB,), dtype=torch.int64, device=dev)
            re = torch.empty((NB,), dtype=torch.float32, device=dev)
            nt = torch.empty((NB, 1), dtype=torch.float32, device=dev)
            we = torch.empty((NB,), dtype=torch.float32, device=dev)
            _lib.check(self._lib.irbpp_replay_gather(C.byref(view), B, float(self.priority_weight), _p(data_idx.contiguous()),
                                                     _p(prob.contiguous()), _p(st), _p(ac), _p(re), _p(nx), _p(nt), _p(we),
                                                     _stream(dev)), "irbpp_replay_gather")
            return tree_idx, st, ac, re, nx, nt, we
        state, action, returns, next_state, nonterminal = self._transitions(data_idx)
        probs = prob / p_total[:, None]                                           # (:199)
        filled = torch.where(self.full, torch.full_like(self.index, self.capacity), self.index).to(torch.float32)
        weights = (filled[:, None] * probs) ** -self.priority_weight               # (:200-201)
        weights = weights / weights.max(dim=1, keepdim=True).values                # (:202) per memory
        flat = lambda x: x.reshape((N * B,) + tuple(x.shape[2:]))                  # noqa: E731
        return (tree_idx, flat(state).to(torch.float32), flat(action), flat(returns), flat(next_state).to(torch.float32),
                flat(nonterminal).reshape(N * B, 1), flat(weights))

    # ------------------------------------------------------------------ priorities ----------
    def update_priorities(self, tree_idxs: torch.Tensor, priorities: torch.Tensor, powered: bool = False) -> None:
        """ReplayMemory.update_priorities (memory.py:207-209): priority^omega into the listed
        leaves.  Position j of every env is applied before position j+1, so a leaf listed twice
        keeps its last value as in the reference's sequential loop.  The power is taken with
        torch.pow on the device; the reference's ``np.power`` (float32 powf on the host) can differ
        from it in the last bit, so ``powered=True`` accepts values that already are priority^omega
        (used by the parity tests to feed numpy's)."""
        N, B = tree_idxs.shape
        pr = priorities.to(self.device, torch.float32).reshape(N, B)
        if not powered:
            pr = torch.pow(pr, self.priority_exponent)
        if self._lib is not None:                                  # all B leaves of every env in one launch
            from . import _lib
            _lib.check(self._lib.irbpp_sumtree_update(_p(self.sum_tree), _p(self.max), N, self.capacity,
                                                      _p(tree_idxs.to(self.device, torch.int64).contiguous()), _p(pr.contiguous()), B, _p(None),
                                                      _stream(self.device)), "irbpp_sumtree_update")
            return
        for j in range(B):
            self._set_leaves(self._rows, tree_idxs[:, j].to(self.device), pr[:, j])

    def anneal(self, increase: float) -> None:
        """trainer.py:195-196."""
        self.priority_weight = min(self.priority_weight + increase, 1.0)

    def __len__(self) -> int:
        return self.N


def mask_from_state(state: torch.Tensor, selected_action: int) -> torch.Tensor:
    """get_mask_from_state (tools.py:283-300) for the candidate-selection layout: column 4 of the
    [S, 5] block is the validity flag of each candidate."""
    return state[:, :selected_action * 5].reshape(state.shape[0], selected_action, 5)[:, :, -1]


def masked_greedy_action(q: torch.Tensor, state: torch.Tensor, selected_action: int) -> torch.Tensor:
    """The tail of Agent.act (agent.py:55-58): ``q[(1 - mask).bool()] = -inf; q.argmax(1)`` with the mask taken from
    the observation (get_mask_from_state).  On a HIP device one kernel reads the flags straight from ``state``."""
    lib = _hip_lib(q.device)
    if lib is None:
        masked = q.masked_fill(mask_from_state(state, selected_action) == 0, float("-inf"))
        return masked.argmax(1)
    from . import _lib
    q = q.to(torch.float32).contiguous()
    state = state.to(torch.float32).contiguous()         # the kernel reads the validity flags as float32
    out = torch.empty((q.shape[0],), dtype=torch.int64, device=q.device)
    _lib.check(lib.irbpp_masked_argmax(_p(q), q.stride(0), _p(state), state.stride(0), int(selected_action), q.shape[0],
                                       _p(out), _stream(q.device)), "irbpp_masked_argmax")
    return out


def actor_step(envs, policy, memory: VectorReplayMemory, state: torch.Tensor, reward_clip: float = 0.0):
    """One iteration of the trainer's acting loop (trainer.py:160-186) without per-env Python:
    mask -> policy -> envs.step -> clip -> append.  ``envs`` is a GpuPackingEnv (device-tensor
    API); ``policy(state, mask) -> int64[N]`` stands for Agent.act.  Returns the next state and
    (reward, done) device tensors; episode statistics stay on the device (episode_totals)."""
    mask = mask_from_state(state, envs.S)
    action = policy(state, mask)
    next_state, reward, done = envs.step(action if action.dtype == torch.int32 else action.to(torch.int32))
    if reward_clip > 0:
        reward = reward.to(torch.float32).clamp(-reward_clip, reward_clip)           # trainer.py:181-182
    # every sample is Valid without physics.  (On a HIP device the append is one launch that takes the environment's own
    # float64 rewards and one-byte done flags: `reward` / `done` are then the step's views, overwritten by the next step.)
    memory.append(state, action, reward, done)
    return next_state, reward, done
