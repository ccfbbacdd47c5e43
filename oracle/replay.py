"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference's per-environment prioritised
n-step replay memory (memory.py), used to check ``irbpp_amd.replay.VectorReplayMemory``
(SURVEY.md 8f-3).  One object = one environment's memory, plain loops, float32 tree sums like
the reference's torch tensors.  Pinned against the reference's own memory.py by
``tests/golden/make_replay_golden.py`` -> ``tests/golden/replay_*.npz``.
"""
import numpy as np

f32 = np.float32


class SumTree(object):
    """SegmentTree (memory.py:15-100)."""

    def __init__(self, size, obs_len):
        self.size = size
        self.index = 0
        self.full = False
        self.max = f32(1.0)                                   # memory.py:27
        self.sum_tree = np.zeros(2 * size - 1, dtype=f32)
        self.timesteps = np.zeros(size, dtype=np.int64)
        self.states = np.zeros((size, obs_len), dtype=f32)
        self.actions = np.zeros(size, dtype=np.int64)
        self.rewards = np.zeros(size, dtype=f32)
        self.nonterminals = np.zeros(size, dtype=bool)

    def update(self, index, value):                          # memory.py:47-58
        value = f32(value)
        self.sum_tree[index] = value
        while index != 0:
            index = (index - 1) // 2
            self.sum_tree[index] = f32(self.sum_tree[2 * index + 1] + self.sum_tree[2 * index + 2])
        self.max = max(value, self.max)

    def append(self, data, value):                           # memory.py:60-70
        i = self.index
        self.timesteps[i], self.states[i], self.actions[i], self.rewards[i], self.nonterminals[i] = data
        self.update(i + self.size - 1, value)
        self.index = (i + 1) % self.size
        self.full = self.full or self.index == 0

    def find(self, value):                                   # memory.py:72-86; the comparison runs in float32
        value = f32(value)
        index = 0
        while 2 * index + 1 < len(self.sum_tree):
            left = 2 * index + 1
            if value <= self.sum_tree[left]:
                index = left
            else:
                value = f32(value - self.sum_tree[left])
                index = left + 1
        return self.sum_tree[index], index - self.size + 1, index

    def total(self):
        return self.sum_tree[0]


class ReplayMemory(object):
    """ReplayMemory (memory.py:100-209)."""

    def __init__(self, capacity, obs_len, discount=0.99, multi_step=3, priority_weight=0.4, priority_exponent=0.5):
        self.capacity, self.obs_len = capacity, obs_len
        self.discount, self.n = discount, multi_step
        self.priority_weight, self.priority_exponent = priority_weight, priority_exponent
        self.t = 0
        self.transitions = SumTree(capacity, obs_len)
        self.n_step_scaling = np.array([discount ** i for i in range(multi_step)], dtype=f32)

    def append(self, state, action, reward, terminal):       # memory.py:117-121
        self.transitions.append((self.t, np.asarray(state, dtype=f32), int(action), f32(reward), not terminal),
                                self.transitions.max)
        self.t = 0 if terminal else self.t + 1

    def valid(self, prob, idx):                              # memory.py:175
        w = self.transitions.index
        return (w - idx) % self.capacity > self.n and (idx - w) % self.capacity >= 1 and prob != 0

    def transition(self, idx):                               # memory.py:123-139 + 178-191
        tr = self.transitions
        states, rewards, nonterminals = [], [], []
        action = None
        for t in range(self.n + 1):
            if t == 0 or nonterminals[-1]:
                j = (idx + t) % tr.size
                s, a, r, nt = tr.states[j], tr.actions[j], tr.rewards[j], bool(tr.nonterminals[j])
            else:                                            # blank_trans (:114)
                s, a, r, nt = np.zeros(self.obs_len, dtype=f32), 0, f32(0), False
            if t == 0:
                action = a
            states.append(s)
            rewards.append(r)
            nonterminals.append(nt)
        ret = np.dot(np.array(rewards[:self.n], dtype=f32), self.n_step_scaling)
        return states[0], int(action), f32(ret), states[self.n], f32(nonterminals[self.n])

    def sample_at(self, values):
        """ReplayMemory.sample (memory.py:194-204) with the tree positions given instead of drawn."""
        p_total = self.transitions.total()
        found = [self.transitions.find(v) for v in values]
        assert all(self.valid(p, i) for p, i, _ in found)
        rows = [self.transition(i) for _, i, _ in found]
        probs = np.array([p for p, _, _ in found], dtype=f32) / p_total
        filled = self.capacity if self.transitions.full else self.transitions.index
        weights = (f32(filled) * probs) ** f32(-self.priority_weight)
        weights = (weights / weights.max()).astype(f32)
        cols = list(zip(*rows))
        return ([t for _, _, t in found], np.stack(cols[0]), np.array(cols[1]), np.array(cols[2], dtype=f32),
                np.stack(cols[3]), np.array(cols[4], dtype=f32), weights)

    def update_priorities(self, idxs, priorities):           # memory.py:207-209
        for i, p in zip(idxs, np.power(np.asarray(priorities, dtype=f32), f32(self.priority_exponent))):
            self.transitions.update(i, p)
