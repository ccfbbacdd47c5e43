"""The reference's on-disk dataset layout (README.md:74-87, arguments.py:102-136), read and written without
trimesh / pybullet:

    dataset/<name>/id2shape.pt        torch-saved dict  {shape id: "<mesh>.obj"}          (tools.py:232)
    dataset/<name>/test_sequence.pt   torch-saved list of trajectories (lists of ids)     (IRcreator.py:81)
    dataset/<name>/shape_vhacd/*.obj  the meshes                                          (tools.py:236-237)
    dataset/shotInfo/<name>_id2shape_<resH>/<k>_<rot>.pt
                                      torch-saved [heightMapT, heightMapB, maskH, maskB]  (tools.py:258-277)

``load_reference_dataset`` turns such a directory into what the batched environment consumes (a ShapeSet, the
trajectories as an int32 matrix, the id -> name map); footprint tables come from the shotInfo cache when it is
there and from the GPU rasteriser (meshes.py) otherwise -- and are then written back in the reference's cache
format, so that an unmodified reference checkout picks them up at ``shotInfoPre`` instead of ray-casting.
Extents and volumes always come from the meshes (they are not in the cache; tools.py:240-242).
"""
from __future__ import annotations

import os
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch

from .shapes import ShapeSet
from .synthetic import ROT_DEGREES


def shot_info_dir(root: str, data_name: str, resolution_h: float, dic_name: str = "id2shape", mesh_scale=1) -> str:
    """tools.py:255-262: ``dataset/shotInfo/<data>_<dict>_<resH>[_<meshScale>]`` under ``root``."""
    leaf = "{}_{}_{}".format(data_name, dic_name, resolution_h) if mesh_scale == 1 else \
        "{}_{}_{}_{}".format(data_name, dic_name, resolution_h, mesh_scale)
    return os.path.join(root, "dataset", "shotInfo", leaf)


def save_shot_info_cache(shapes: ShapeSet, directory: str) -> int:
    """Write ``<k>_<rot>.pt`` = ``torch.save([heightMapT, heightMapB, maskH, maskB])`` (tools.py:271-277) for every
    (shape, rotation); existing files are left alone, as the reference does.  Returns the number written."""
    os.makedirs(directory, exist_ok=True)
    n = 0
    for k in range(shapes.n_shapes):
        for r in range(shapes.n_rot):
            path = os.path.join(directory, "{}_{}.pt".format(k, r))
            if not os.path.exists(path):
                torch.save([np.ascontiguousarray(a, dtype=np.float64) for a in shapes.tables[k][r]], path)
                n += 1
    return n


def load_shot_info_cache(directory: str, n_shapes: int, n_rot: int):
    """-> tables[k][rot] = (T, B, maskH, maskB), or None unless every file of the cache is present."""
    tables = []
    for k in range(n_shapes):
        per = []
        for r in range(n_rot):
            path = os.path.join(directory, "{}_{}.pt".format(k, r))
            if not os.path.exists(path):
                return None
            T, B, mH, mB = torch.load(path, weights_only=False)      # pickled numpy arrays, as the reference saves them
            per.append(tuple(np.asarray(a, dtype=np.float64) for a in (T, B, mH, mB)))
        tables.append(per)
    return tables


def sequences_matrix(trajs: List[List[Optional[int]]]) -> np.ndarray:
    """test_sequence.pt -> int32 [n_traj, L]; shorter trajectories and ``None`` entries become -1, which the
    kernels treat like the ``None`` sentinel LoadItemCreator appends (IRcreator.py:95)."""
    L = max(len(t) for t in trajs)
    out = -np.ones((len(trajs), L), dtype=np.int32)
    for i, t in enumerate(trajs):
        out[i, :len(t)] = [-1 if v is None else int(v) for v in t]
    return out


def save_reference_dataset(root: str, name: str, names: Dict[int, str], sequences: np.ndarray,
                           meshes: Optional[Dict[int, Tuple[np.ndarray, np.ndarray]]] = None) -> str:
    """Write ``id2shape.pt`` / ``test_sequence.pt`` (and OBJ files) in the reference's layout (tests, examples)."""
    d = os.path.join(root, "dataset", name)
    os.makedirs(os.path.join(d, "shape_vhacd"), exist_ok=True)
    torch.save({int(k): str(v) for k, v in names.items()}, os.path.join(d, "id2shape.pt"))
    torch.save([[int(v) for v in row if v >= 0] for row in np.asarray(sequences)], os.path.join(d, "test_sequence.pt"))
    for k, (verts, faces) in (meshes or {}).items():
        with open(os.path.join(d, "shape_vhacd", names[k]), "w") as fh:
            for v in verts:
                fh.write("v %.17g %.17g %.17g\n" % tuple(v))
            for f in faces:
                fh.write("f %d %d %d\n" % tuple(int(i) + 1 for i in f))
    return d


def load_reference_dataset(root: str, name: str, resolution_h: float = 0.01, n_rot: int = 8, device=None,
                           categories: Optional[int] = None, write_cache: bool = True):
    """-> (ShapeSet, sequences int32 [n_traj, L], names {id: "<mesh>.obj"}).

    ``root`` is the directory that holds ``dataset/`` (the reference's working directory).  Meshes are read with
    meshes.load_obj and posed like load_mesh_plain (tools.py:18-39: rotation about z by the reference's angle
    list; extents are those of the posed mesh, the volume that of pose 0).  Footprint tables: the shotInfo cache if
    complete, else ``irbpp_shot_item`` on ``device`` (a HIP device is then required) and, with ``write_cache``,
    stored in the cache for the reference to reuse."""
    from . import meshes as M
    d = os.path.join(root, "dataset", name)
    id2shape = torch.load(os.path.join(d, "id2shape.pt"), weights_only=False)
    names = {int(k): str(v) for k, v in id2shape.items()}
    n = len(names) if categories is None else min(categories, len(names))
    assert sorted(names)[:n] == list(range(n)), "shape ids must be 0..n-1"
    trajs = torch.load(os.path.join(d, "test_sequence.pt"), weights_only=False)
    seqs = sequences_matrix(trajs)
    cache_dir = shot_info_dir(root, name, resolution_h)
    tables = load_shot_info_cache(cache_dir, n, n_rot)
    extents, volumes, raster = [], [], []
    for k in range(n):
        verts, faces = M.load_obj(os.path.join(d, "shape_vhacd", names[k]))
        per_ext, per_tab = [], []
        for r in range(n_rot):
            vr = M.rotate_z(verts, ROT_DEGREES[r])
            if tables is None:
                if device is None:
                    raise RuntimeError(f"shotInfo cache {cache_dir} is incomplete: rasterising the meshes needs a HIP device")
                ext, tab = M.shot_item_gpu(vr, faces, resolution_h, device)
                per_tab.append(tab)
            else:
                ext = vr.max(0) - vr.min(0)
            per_ext.append(ext)
        extents.append(per_ext)
        raster.append(per_tab)
        volumes.append(M.mesh_volume(verts, faces))
    shapes = ShapeSet(np.array(extents), np.array(volumes), tables if tables is not None else raster, name=name,
                      meta={"res_h": resolution_h, "n_rot": n_rot, "tables_from": "cache" if tables is not None else "rasteriser"})
    if tables is None and write_cache:
        save_shot_info_cache(shapes, cache_dir)
    return shapes, seqs, names
