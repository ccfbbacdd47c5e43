"""The reference's random item creators as per-environment item streams (training-time item supply).

In training every environment process draws its items with ``np.random.choice`` on the process-global legacy
generator (RandomItemCreator / RandomInstanceCreator / RandomCateCreator, IRcreator.py:26-72), seeded with
``args.seed + rank`` (envs.py:41 -> PackingGame.seed, binPhy.py:118-123).  ``ItemStream`` reproduces that stream
bit for bit (host code of libirbpp_hip.so: csrc/irbpp_itemgen.h); ``StreamFeeder`` keeps the per-bin item rings of an
environment created with ``item_stream=1`` ahead of the bins, so that ``make_vec_envs(args)`` needs nothing but the
reference's own namespace to train on the same item sequence the reference's workers would have drawn.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Optional, Sequence

import numpy as np

from . import _lib


def instance_groups(dic_path: Dict[int, str]) -> List[List[int]]:
    """RandomInstanceCreator.__init__ (IRcreator.py:35-46): ids grouped by ``name[0:-6]`` in dict order."""
    inverse: Dict[str, List[int]] = {}
    for k in dic_path.keys():
        inverse.setdefault(dic_path[k][0:-6], []).append(k)
    return [inverse[name] for name in inverse.keys()]


def category_groups(dic_path: Dict[int, str]) -> List[List[int]]:
    """RandomCateCreator.__init__ (IRcreator.py:53-68): ids per category in the fixed order objects, concave, board
    (the probabilities in ``self.categories`` are never passed to np.random.choice: the draw is uniform)."""
    cates: Dict[str, List[int]] = {"objects": [], "concave": [], "board": []}
    for k, item in zip(dic_path.keys(), dic_path.values()):
        cate, _ = item.split("/")
        cates[cate].append(k)
    return [cates[c] for c in ("objects", "concave", "board")]


class ItemStream(object):
    """One environment's item stream.  ``groups``: list of id lists for the two-stage creators, or ``None`` with
    ``item_set`` for RandomItemCreator."""

    def __init__(self, seed: int, groups: Optional[Sequence[Sequence[int]]] = None, item_set: Optional[Sequence[int]] = None):
        self.lib = _lib.load()
        if groups is not None:
            members = np.ascontiguousarray(np.concatenate([np.asarray(g, dtype=np.int32) for g in groups]))
            offs = np.ascontiguousarray(np.concatenate([[0], np.cumsum([len(g) for g in groups])]).astype(np.int32))
            n_groups, offs_p = len(groups), offs.ctypes.data_as(_lib.c_i32_p)
        else:
            members = np.ascontiguousarray(np.asarray(item_set, dtype=np.int32))
            n_groups, offs_p = 0, None
        if not 0 <= int(seed) <= 2 ** 32 - 1:
            raise ValueError("Seed must be between 0 and 2**32 - 1")           # np.random.seed's own rule
        self._h = C.c_void_p()
        _lib.check(self.lib.irbpp_itemgen_create(int(seed), n_groups, offs_p, members.ctypes.data_as(_lib.c_i32_p),
                                                 len(members), C.byref(self._h)), "irbpp_itemgen_create")

    def draw(self, count: int) -> np.ndarray:
        out = np.empty((int(count),), dtype=np.int32)
        _lib.check(self.lib.irbpp_itemgen_draw(self._h, int(count), out.ctypes.data_as(_lib.c_i32_p)), "irbpp_itemgen_draw")
        return out

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self.lib.irbpp_itemgen_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def streams_for_args(args, num_envs: int, rank_offset: int = 0) -> List[ItemStream]:
    """The creators PackingGame.__init__ picks for training (binPhy.py:58-67), one per environment, seeded
    ``args.seed + rank`` like envs.py:41.  ``args.dicPath`` may be the dict itself or the path of ``id2shape.pt``."""
    dic = args.dicPath
    if isinstance(dic, str):
        import torch
        dic = torch.load(dic, weights_only=False)
    sample = getattr(args, "dataSample", "instance")
    if sample == "category":
        groups, item_set = category_groups(dic), None
    elif sample == "instance":
        groups, item_set = instance_groups(dic), None
    else:
        assert sample == "pose"
        groups, item_set = None, list(range(len(dic)))                      # np.arange(0, len(shapeDict))
    return [ItemStream(int(args.seed) + rank_offset + i, groups, item_set) for i in range(num_envs)]


class StreamFeeder(object):
    """Host side of the item rings of a ``GpuPackingEnv(item_stream=1)``: row b of the environment's sequence table
    is a ring of ``ring_len`` items that only bin b reads; every ``ring_len // (4 * bufferSize)`` ticks the feeder
    reads the bins' cursors and rewrites what they have consumed.  A tick is anything that can cost a bin up to
    ``bufferSize`` items: a step (one item, or a whole new queue when the step ends the episode), a reset, a
    reset_specific -- ``GpuVecEnv`` ticks from all three.  Should a ring run dry all the same, the bin notices at the
    fetch itself: consumed ring slots are poisoned by the kernel, reading one raises IRBPP_DEVERR_STREAM_DRY in the
    step's error word (and ``refill`` raises when it sees a cursor beyond what it has delivered).
    Grouped environments (``GroupedPackingEnv``): group g's sub-environment owns rows [g*per, (g+1)*per) of the table
    and is fed on its own stream."""

    def __init__(self, streams: Sequence[ItemStream], ring_len: int = 4096, buffer_size: int = 1):
        self.streams = list(streams)
        self.ring_len = int(ring_len)
        self.K = int(buffer_size)
        if self.ring_len < 16 * self.K:
            raise ValueError("ring_len must be at least 16 * bufferSize")
        self.n = len(self.streams)
        self.written = np.full((self.n,), self.ring_len, dtype=np.int64)     # items drawn per stream so far
        self.initial = np.stack([s.draw(self.ring_len) for s in self.streams])
        self.every = max(1, self.ring_len // (4 * self.K))
        self.steps = 0
        self.env = None
        self.parts = []

    def attach(self, env) -> None:
        """``env``: the GpuPackingEnv (or GroupedPackingEnv) that was created with ``sequences=self.initial,
        item_stream=1``."""
        import torch
        assert env.num_bins == self.n
        self.env = env
        if hasattr(env, "groups"):                                           # (sub-environment, its rows, its stream)
            self.parts = [(e, env.rows(g), env.streams[g]) for g, e in enumerate(env.groups)]
        else:
            self.parts = [(env, slice(0, self.n), None)]
        self._cur = [torch.zeros((rows.stop - rows.start,), dtype=torch.int32, device=env.device) for _, rows, _ in self.parts]

    def tick(self, steps: int = 1) -> None:
        """Call once per environment step (or get_action_candidates + step pair), reset or reset_specific."""
        self.steps += steps
        if self.steps >= self.every:
            self.refill()

    def refill(self) -> None:
        import contextlib
        import torch
        L = self.ring_len
        self.steps = 0
        for (env, rows, st), cur_dev in zip(self.parts, self._cur):
            with (torch.cuda.stream(st) if st is not None else contextlib.nullcontext()):
                _lib.check(env.lib.irbpp_stream_cursors(env._h, C.c_void_p(cur_dev.data_ptr()), 0, env._stream()), "irbpp_stream_cursors")
                cur = cur_dev.cpu().numpy().astype(np.int64)                 # synchronises this part's stream
                written = self.written[rows]
                if (cur > written).any():
                    raise RuntimeError("an item ring ran dry: a bin consumed items the stream had not delivered (ring_len too small)")
                count = (cur + L - written).astype(np.int32)                 # what each bin has consumed since the last refill
                width = int(count.max())
                if width <= 0:
                    continue
                ids = np.zeros((len(count), width), dtype=np.int32)
                for i, s in enumerate(self.streams[rows]):
                    if count[i] > 0:
                        ids[i, :count[i]] = s.draw(int(count[i]))
                first = (written % L).astype(np.int32)
                dev = env.device
                t_ids, t_first, t_count = (torch.from_numpy(a).to(dev) for a in (ids, first, count))
                _lib.check(env.lib.irbpp_stream_write(env._h, C.c_void_p(t_ids.data_ptr()), C.c_void_p(t_first.data_ptr()),
                                                      C.c_void_p(t_count.data_ptr()), width, env._stream()), "irbpp_stream_write")
                torch.cuda.current_stream(dev).synchronize()                 # the staging tensors may go now
                self.written[rows] += count
