"""The item streams of the reference's random item creators (IRcreator.py:26-72), reproduced by the host code of
libirbpp_hip.so (csrc/irbpp_itemgen.h) -- pinned here, without a GPU, against (1) goldens drawn by the reference's OWN
classes on numpy's global generator (tests/golden/make_golden.py: random_creators.npz), (2) numpy's legacy
RandomState itself on random group structures, (3) the oracle's restatement."""
import os

import numpy as np

import irbpp_amd  # noqa: F401
from irbpp_amd import itemgen
from oracle.packing import RandomStreamItemCreator


def _dicts(g):
    inst = dict(zip([int(k) for k in g["dic_inst_keys"]], [str(v) for v in g["dic_inst_vals"]]))
    cate = dict(zip([int(k) for k in g["dic_cate_keys"]], [str(v) for v in g["dic_cate_vals"]]))
    return inst, cate


def _episode_protocol(raw, n):
    """What the golden's loop makes of a raw stream: preview(1), update_item_queue(0), generate_item per step and an
    ItemCreator.reset -- which throws the queued item away (IRcreator.py:11-12) -- after every 37th item."""
    it = iter(raw)
    queue, items = [], []
    while len(items) < n:
        if not queue:
            queue.append(next(it))
        items.append(queue.pop(0))
        queue.append(next(it))
        if len(items) % 37 == 0:
            queue.clear()
    return items


def test_streams_equal_the_reference_creators_golden(golden_dir):
    g = np.load(os.path.join(golden_dir, "random_creators.npz"))
    inst, cate = _dicts(g)
    seed = int(g["seed"])
    n = g["stream_instance"].shape[1]
    for rank in range(4):
        s = itemgen.ItemStream(seed + rank, itemgen.instance_groups(inst))
        np.testing.assert_array_equal(_episode_protocol(s.draw(n + 40), n), g["stream_instance"][rank])
        s = itemgen.ItemStream(seed + rank, itemgen.category_groups(cate))
        np.testing.assert_array_equal(_episode_protocol(s.draw(n + 40), n), g["stream_category"][rank])
        s = itemgen.ItemStream(seed + rank, None, list(range(29)))
        raw = np.concatenate([s.draw(7), s.draw(n + 33)])                      # draws append to one stream
        np.testing.assert_array_equal(_episode_protocol(raw, n), g["stream_pose"][rank])


def test_streams_for_args_follow_binphy_and_envs_seeding(golden_dir):
    import types
    g = np.load(os.path.join(golden_dir, "random_creators.npz"))
    inst, cate = _dicts(g)
    for sample, dic, key in (("instance", inst, "stream_instance"), ("category", cate, "stream_category")):
        args = types.SimpleNamespace(seed=int(g["seed"]), dicPath=dic, dataSample=sample)
        streams = itemgen.streams_for_args(args, 4)
        for rank, s in enumerate(streams):
            np.testing.assert_array_equal(s.draw(37), g[key][rank][:37])         # up to the golden's first reset


def test_streams_equal_numpy_legacy_choice_on_random_group_structures():
    rng = np.random.RandomState(5)
    for case in range(12):
        n_groups = int(rng.randint(1, 40))
        sizes = rng.randint(1, 70, size=n_groups)            # one-element lists consume no random number (rng == 0)
        ids = rng.permutation(int(sizes.sum()))
        groups, at = [], 0
        for sz in sizes:
            groups.append([int(v) for v in ids[at:at + sz]])
            at += sz
        seed = int(rng.randint(0, 2 ** 31)) if case else 2 ** 32 - 1
        got = itemgen.ItemStream(seed, groups).draw(2000)
        rs = np.random.RandomState(seed)
        names = list(range(n_groups))
        ref = [rs.choice(groups[rs.choice(names)]) for _ in range(2000)]
        np.testing.assert_array_equal(got, ref)


def test_oracle_creator_restates_the_same_streams(golden_dir):
    g = np.load(os.path.join(golden_dir, "random_creators.npz"))
    inst, cate = _dicts(g)
    for kind, dic, key in (("instance", inst, "stream_instance"), ("category", cate, "stream_category"), ("pose", None, "stream_pose")):
        c = RandomStreamItemCreator(int(g["seed"]) + 2, dic, kind, n_items=29)
        items = []
        c.reset()
        while len(items) < 120:
            items.append(int(c.preview(1)[0]))
            c.update_item_queue(0)
            c.generate_item()
            if len(items) % 37 == 0:
                c.reset()
        np.testing.assert_array_equal(items, g[key][2][:120])
