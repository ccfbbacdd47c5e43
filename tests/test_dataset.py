"""The reference's on-disk dataset layout (id2shape.pt, test_sequence.pt, shape_vhacd/*.obj, the shotInfo cache of
tools.py:258-277) written and read back by irbpp_amd.dataset -- no GPU: tables come from the cache."""
import os

import numpy as np
import pytest
import torch

import irbpp_amd  # noqa: F401
from irbpp_amd import dataset, meshes, synthetic


def test_dataset_directory_round_trip(tmp_path):
    cube = synthetic.cube_shapes()
    keep = [0, 31, 62, 93, 124, 7]                                   # a few boxes of the Cube dataset
    edges = (0.03, 0.06, 0.09, 0.12, 0.15)
    boxes = {i: meshes.box_mesh(edges[k // 25], edges[(k // 5) % 5], edges[k % 5]) for i, k in enumerate(keep)}
    names = {i: "box%03d.obj" % k for i, k in enumerate(keep)}
    seqs = synthetic.make_sequences(len(keep), 12, 30, seed=4)
    root = str(tmp_path)
    dataset.save_reference_dataset(root, "cube6", names, seqs, boxes)
    sub = synthetic.ShapeSet(cube.extents[keep], cube.volumes[keep], [cube.tables[k] for k in keep], name="cube6")
    cache = dataset.shot_info_dir(root, "cube6", 0.01)
    assert cache.endswith(os.path.join("dataset", "shotInfo", "cube6_id2shape_0.01"))            # tools.py:258-262
    assert dataset.save_shot_info_cache(sub, cache) == len(keep) * 2
    assert dataset.save_shot_info_cache(sub, cache) == 0                                         # existing files are kept
    # what the reference does with a cache file (tools.py:271-272)
    heightMapT, heightMapB, maskH, maskB = torch.load(os.path.join(cache, "3_1.pt"), weights_only=False)
    np.testing.assert_array_equal(heightMapT, sub.tables[3][1][0])
    assert isinstance(maskB, np.ndarray) and maskB.dtype == np.float64
    # ... and what it does with the other two files (tools.py:232, IRcreator.py:81-95)
    assert torch.load(os.path.join(root, "dataset", "cube6", "id2shape.pt"), weights_only=False) == names
    trajs = torch.load(os.path.join(root, "dataset", "cube6", "test_sequence.pt"), weights_only=False)
    assert len(trajs) == 12 and trajs[5] == [int(v) for v in seqs[5]]

    shapes, seqs2, names2 = dataset.load_reference_dataset(root, "cube6", 0.01, n_rot=2)         # no device: cache is complete
    assert names2 == names and shapes.meta["tables_from"] == "cache"
    np.testing.assert_array_equal(seqs2, seqs)
    np.testing.assert_allclose(shapes.extents, sub.extents, rtol=0, atol=1e-15)
    np.testing.assert_allclose(shapes.volumes, sub.volumes, rtol=0, atol=1e-15)
    for k in range(len(keep)):
        for r in range(2):
            for got, want in zip(shapes.tables[k][r], sub.tables[k][r]):
                np.testing.assert_array_equal(got, want)
    shapes.validate(0.01, 0.02)
    os.remove(os.path.join(cache, "2_0.pt"))                                                     # incomplete cache, no device
    with pytest.raises(RuntimeError):
        dataset.load_reference_dataset(root, "cube6", 0.01, n_rot=2)


def test_sequences_with_sentinels_and_ragged_rows():
    m = dataset.sequences_matrix([[3, 1, None], [2], [0, 1, 2, 3]])
    np.testing.assert_array_equal(m, [[3, 1, -1, -1], [2, -1, -1, -1], [0, 1, 2, 3]])


def test_saved_trajs_have_the_reference_container_format(tmp_path):
    from irbpp_amd.evaluate import save_trajs
    eps = [[[3, "3.obj", np.zeros(3), np.array([0, 0, 0, 1.0])]], [[1, "1.obj", np.ones(3), np.array([0, 0, 1.0, 0])]] * 2]
    path = str(tmp_path / "logs" / "evaluation" / "t" / "trajs.npy")
    save_trajs(path, eps)
    back = np.load(path, allow_pickle=True)
    assert back.dtype == object and back.shape == (2,) and len(back[1]) == 2 and back[0][0][1] == "3.obj"
