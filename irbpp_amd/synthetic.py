"""Synthetic datasets in ``shotInfo`` format.

The reference's datasets (``dataset/<name>/shape_vhacd`` meshes, ``id2shape.pt``,
``test_sequence.pt``; README.md:74-87) are Google-Drive downloads and the
footprint tables are ray-cast from them with trimesh at start-up
(tools.py:248-279).  Neither is available offline, so the packing environment is
driven with generated shapes that are emitted *directly* in the table format the
environment consumes.  Everything is seeded and pure numpy, so the CPU oracle and
the HIP library see identical inputs on any machine.

    cube_shapes       BASELINE config 1  (boxes, R=2)
    blockout_shapes   BASELINE config 2/4 (polycubes on a 4 cm lattice, R=4 or 8)
    general_shapes    BASELINE config 3/5 (irregular height-field solids, R=8)
"""
from __future__ import annotations

import itertools
from typing import List

import numpy as np

from .shapes import ShapeSet, Table, grid_extent

# z-rotation order of the reference (tools.py:76-79): 0,90,180,270,45,135,225,315 degrees
ROT_DEGREES = (0.0, 90.0, 180.0, 270.0, 45.0, 135.0, 225.0, 315.0)


def _box_tables(ext, res_h) -> Table:
    """A solid box as ``shot_item`` sees it (tools.py:98-135): bottom 0, top e_z, full masks.
    Cells of an extra ceil-fuzz row (e.g. ceil(0.07/0.01)=8) are ray misses: 0 / mask 0."""
    fx, fy = grid_extent(ext[0:2], res_h)
    # rays sit at i*res_h + 0.001 (tools.py:87-88); a cell is hit iff that point is inside the box
    hx = (np.arange(fx) * res_h + 0.001) < ext[0]
    hy = (np.arange(fy) * res_h + 0.001) < ext[1]
    hit = np.outer(hx, hy).astype(np.float64)
    T = hit * ext[2]
    B = np.zeros((fx, fy))
    return (T, B, hit.copy(), hit.copy())


def cube_shapes(res_h: float = 0.01, n_rot: int = 2) -> ShapeSet:
    """125 boxes with edges in {.03,.06,.09,.12,.15} m (README.md:39 'Cube' dataset)."""
    edges = (0.03, 0.06, 0.09, 0.12, 0.15)
    extents, volumes, tables = [], [], []
    for ex, ey, ez in itertools.product(edges, repeat=3):
        per_rot_ext, per_rot_tab = [], []
        for r in range(n_rot):
            e = np.array([ex, ey, ez]) if ROT_DEGREES[r] in (0.0, 180.0) else np.array([ey, ex, ez])
            per_rot_ext.append(e)
            per_rot_tab.append(_box_tables(e, res_h))
        extents.append(per_rot_ext)
        tables.append(per_rot_tab)
        volumes.append(ex * ey * ez)
    return ShapeSet(np.array(extents), np.array(volumes), tables, name="cube",
                    meta={"res_h": res_h, "n_rot": n_rot})


def _grow_polycube(rng: np.random.RandomState, n_cubes: int) -> np.ndarray:
    """Face-connected set of unit cubes grown inside a 3x3x3 lattice -> bool[nx,ny,nz]."""
    occ = np.zeros((3, 3, 3), dtype=bool)
    occ[tuple(rng.randint(0, 3, size=3))] = True
    nbrs = [(1, 0, 0), (-1, 0, 0), (0, 1, 0), (0, -1, 0), (0, 0, 1), (0, 0, -1)]
    while occ.sum() < n_cubes:
        cells = np.argwhere(occ)
        c = cells[rng.randint(len(cells))]
        d = nbrs[rng.randint(6)]
        p = c + d
        if np.all(p >= 0) and np.all(p < 3):
            occ[tuple(p)] = True
    idx = np.argwhere(occ)
    lo, hi = idx.min(0), idx.max(0) + 1
    return occ[lo[0]:hi[0], lo[1]:hi[1], lo[2]:hi[2]]


def _voxel_tables(occ: np.ndarray, cube: float, res_h: float):
    """Column envelope of a voxel solid: top = (highest+1)*cube, bottom = lowest*cube."""
    nx, ny, nz = occ.shape
    cells = int(round(cube / res_h))
    col = occ.any(axis=2)
    top = np.where(col, nz - np.argmax(occ[:, :, ::-1], axis=2), 0).astype(np.float64) * cube
    bot = np.where(col, np.argmax(occ, axis=2), 0).astype(np.float64) * cube
    up = np.ones((cells, cells))
    m = np.kron(col.astype(np.float64), up)
    T = np.kron(top, up) * m
    B = np.kron(bot, up) * m
    ext = np.array([nx * cube, ny * cube, nz * cube])
    return ext, (T, B, m.copy(), m.copy())


def _rotate_nearest(tab: Table, ext, deg: float, res_h: float):
    """Resample a height-field solid after a z-rotation by ``deg`` (nearest source cell).
    Used only for the synthetic 45-degree family; 90-degree multiples use exact rot90."""
    T, B, mH, mB = tab
    th = np.deg2rad(deg)
    c, s = np.cos(th), np.sin(th)
    corners = np.array([[0, 0], [ext[0], 0], [0, ext[1]], [ext[0], ext[1]]], dtype=np.float64)
    rc = corners @ np.array([[c, s], [-s, c]])        # rotate (x,y) by +deg
    lo, hi = rc.min(0), rc.max(0)
    new_ext = np.round(np.array([hi[0] - lo[0], hi[1] - lo[1], ext[2]]), 6)
    fx, fy = grid_extent(new_ext[0:2], res_h)
    gx = (np.arange(fx) + 0.5) * res_h + lo[0]
    gy = (np.arange(fy) + 0.5) * res_h + lo[1]
    X, Y = np.meshgrid(gx, gy, indexing="ij")
    sx = c * X + s * Y                                  # inverse rotation back to the source frame
    sy = -s * X + c * Y
    ix = np.floor(sx / res_h).astype(np.int64)
    iy = np.floor(sy / res_h).astype(np.int64)
    inside = (ix >= 0) & (ix < T.shape[0]) & (iy >= 0) & (iy < T.shape[1])
    ixc, iyc = np.clip(ix, 0, T.shape[0] - 1), np.clip(iy, 0, T.shape[1] - 1)
    out = []
    for arr in (T, B, mH, mB):
        out.append(np.where(inside, arr[ixc, iyc], 0.0))
    T2, B2, mH2, mB2 = out
    if mB2.sum() == 0:                                  # degenerate: behave like shot_item's no-hit branch
        B2[:] = 0.0
        mB2[:] = 1.0
        T2[:] = new_ext[2]
        mH2[:] = 1.0
    return new_ext, (T2 * mH2, B2 * mB2, mH2, mB2)


def _all_rotations(ext0, tab0: Table, n_rot: int, res_h: float):
    exts, tabs = [], []
    for r in range(n_rot):
        deg = ROT_DEGREES[r]
        if deg % 90.0 == 0.0:
            k = int(deg // 90)
            e = np.array([ext0[0], ext0[1], ext0[2]]) if k % 2 == 0 else np.array([ext0[1], ext0[0], ext0[2]])
            t = tuple(np.ascontiguousarray(np.rot90(a, k)) for a in tab0)
        else:
            e, t = _rotate_nearest(tab0, ext0, deg, res_h)
        exts.append(e)
        tabs.append(t)
    return exts, tabs


def blockout_voxels(n_shapes: int = 64, seed: int = 0) -> List[np.ndarray]:
    """The voxel occupancies behind ``blockout_shapes`` (same seed -> same polycubes)."""
    rng = np.random.RandomState(seed)
    return [_grow_polycube(rng, int(rng.randint(2, 6))) for _ in range(n_shapes)]


def blockout_shapes(n_shapes: int = 64, res_h: float = 0.01, n_rot: int = 4,
                    cube: float = 0.04, seed: int = 0) -> ShapeSet:
    """Polycubes of 2-5 face-connected 4 cm cubes ('BlockOut'-like, README.md:33)."""
    extents, volumes, tables = [], [], []
    for occ in blockout_voxels(n_shapes, seed):
        ext0, tab0 = _voxel_tables(occ, cube, res_h)
        e, t = _all_rotations(ext0, tab0, n_rot, res_h)
        extents.append(e)
        tables.append(t)
        volumes.append(float(occ.sum()) * cube ** 3)
    return ShapeSet(np.array(extents), np.array(volumes), tables, name="blockout",
                    meta={"res_h": res_h, "n_rot": n_rot, "seed": seed})


def general_shapes(n_shapes: int = 256, res_h: float = 0.01, n_rot: int = 8,
                   fmin: int = 4, fmax: int = 20, seed: int = 1) -> ShapeSet:
    """Irregular height-field solids: noisy elliptical outline, curved bottom, bumpy top.
    ``fmin..fmax`` are footprint sizes in heightmap cells (use
    8..40 at resolutionH=0.005 for config 5)."""
    rng = np.random.RandomState(seed)
    extents, volumes, tables = [], [], []
    for _ in range(n_shapes):
        fx, fy = int(rng.randint(fmin, fmax + 1)), int(rng.randint(fmin, fmax + 1))
        cx, cy = (fx - 1) / 2.0, (fy - 1) / 2.0
        X, Y = np.meshgrid(np.arange(fx), np.arange(fy), indexing="ij")
        rad = ((X - cx) / (fx / 2.0)) ** 2 + ((Y - cy) / (fy / 2.0)) ** 2
        m = (rad + rng.uniform(-0.15, 0.15, size=(fx, fy)) <= 1.0).astype(np.float64)
        m[int(round(cx)), :] = 1.0                       # the outline spans the whole bounding box
        m[:, int(round(cy))] = 1.0
        B = rng.uniform(0.0, 0.03, size=(fx, fy)) * m
        B = (B - B[m > 0].min()) * m                    # the lowest point rests on z=0
        T = (B + rng.uniform(0.02, 0.12, size=(fx, fy))) * m
        ez = float(T.max())
        ext0 = np.array([np.round((fx - rng.uniform(0.1, 0.9)) * res_h, 6),
                         np.round((fy - rng.uniform(0.1, 0.9)) * res_h, 6), ez])
        e, t = _all_rotations(ext0, (T, B, m.copy(), m.copy()), n_rot, res_h)
        extents.append(e)
        tables.append(t)
        volumes.append(float(((T - B) * m).sum()) * res_h * res_h)
    return ShapeSet(np.array(extents), np.array(volumes), tables, name="general",
                    meta={"res_h": res_h, "n_rot": n_rot, "seed": seed})


def make_sequences(n_shapes: int, n_traj: int = 10000, length: int = 100, seed: int = 123) -> np.ndarray:
    """Pre-drawn item-id trajectories, ``int32[n_traj, length]``: the shape of
    ``test_sequence.pt`` (README.md:81-82; IRcreator.py:81) and the stand-in for the
    training-time ``np.random.choice`` stream (IRcreator.py:33,49-51)."""
    rng = np.random.RandomState(seed)
    return rng.randint(0, n_shapes, size=(n_traj, length)).astype(np.int32)
