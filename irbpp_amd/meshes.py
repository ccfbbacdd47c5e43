"""Footprint tables from triangle meshes: the reference's start-up precompute without trimesh.

``load_shape_dict`` + ``shotInfoPre`` (tools.py:227-279) load every OBJ, rotate it about z into the
``ZRotNum`` poses (tools.py:18-39, 48-79), and ray-cast each pose once (``shot_item``,
tools.py:98-135).  ``shape_set_from_meshes`` does the same from plain vertex/face arrays: OBJ
parsing and the rigid transforms on the host (numpy), the ray casting on the GPU
(``irbpp_shot_item``).  Extents are ``bounds[1]-bounds[0]`` of the rotated pose, the volume is the
signed-tetrahedron sum of pose 0 (``mesh.volume``).

PARITY UNPINNED against trimesh itself (absent from the image): the rasteriser returns the exact
plane height of the first/last triangle a vertical ray crosses, trimesh's ray engine the same up to
its own floating-point path; rays grazing an edge may differ.
"""
from __future__ import annotations

import ctypes as C
from typing import List, Sequence, Tuple

import numpy as np
import torch

from . import _lib
from .shapes import ShapeSet, grid_extent
from .synthetic import ROT_DEGREES


def load_obj(path: str) -> Tuple[np.ndarray, np.ndarray]:
    """Minimal Wavefront OBJ reader: ``v`` and ``f`` records (polygons are fan-triangulated)."""
    verts, faces = [], []
    with open(path) as fh:
        for line in fh:
            p = line.split()
            if not p:
                continue
            if p[0] == "v":
                verts.append([float(p[1]), float(p[2]), float(p[3])])
            elif p[0] == "f":
                idx = [int(tok.split("/")[0]) for tok in p[1:]]
                idx = [i - 1 if i > 0 else len(verts) + i for i in idx]
                for k in range(1, len(idx) - 1):
                    faces.append([idx[0], idx[k], idx[k + 1]])
    return np.asarray(verts, dtype=np.float64), np.asarray(faces, dtype=np.int32)


def mesh_volume(verts: np.ndarray, faces: np.ndarray) -> float:
    a, b, c = verts[faces[:, 0]], verts[faces[:, 1]], verts[faces[:, 2]]
    return float(np.abs(np.einsum("ij,ij->i", a, np.cross(b, c)).sum()) / 6.0)


def rotate_z(verts: np.ndarray, deg: float) -> np.ndarray:
    """Pose of tools.getRotationMatrix (tools.py:61-68): rotation about z, exact for multiples of 90."""
    k = int(round(deg / 90.0))
    if abs(deg - 90.0 * k) < 1e-12:
        c, s = [(1.0, 0.0), (0.0, 1.0), (-1.0, 0.0), (0.0, -1.0)][k % 4]
    else:
        c, s = np.cos(np.deg2rad(deg)), np.sin(np.deg2rad(deg))
    out = verts.copy()
    out[:, 0] = c * verts[:, 0] - s * verts[:, 1]
    out[:, 1] = s * verts[:, 0] + c * verts[:, 1]
    return out


def shot_item_gpu(verts: np.ndarray, faces: np.ndarray, res_h: float, device="cuda:0", shift: float = 0.001):
    """One pose -> (extents, (heightMapT, heightMapB, maskH, maskB)) as float64 numpy arrays."""
    lib = _lib.load()
    dev = torch.device(device)
    lo, hi = verts.min(0), verts.max(0)
    ext = hi - lo
    v = verts - lo                                            # bbox minimum at the origin (tools.py:100)
    fx, fy = (int(t) for t in grid_extent(ext[0:2], res_h))
    vd = torch.from_numpy(np.ascontiguousarray(v)).to(dev)
    fd = torch.from_numpy(np.ascontiguousarray(faces, dtype=np.int32)).to(dev)
    outs = [torch.empty((fx, fy), dtype=torch.float64, device=dev) for _ in range(4)]
    scratch = torch.zeros((1,), dtype=torch.int32, device=dev)
    stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    _lib.check(lib.irbpp_shot_item(C.c_void_p(vd.data_ptr()), C.c_void_p(fd.data_ptr()), len(faces), fx, fy,
                                   float(res_h), float(shift), float(ext[2]), *[C.c_void_p(o.data_ptr()) for o in outs],
                                   C.c_void_p(scratch.data_ptr()), stream), "irbpp_shot_item")
    T, B, mH, mB = (o.cpu().numpy() for o in outs)
    return ext, (T, B, mH, mB)


def shape_set_from_meshes(meshes: Sequence[Tuple[np.ndarray, np.ndarray]], n_rot: int, res_h: float,
                          device="cuda:0", name: str = "meshes") -> ShapeSet:
    extents, volumes, tables = [], [], []
    for verts, faces in meshes:
        per_ext, per_tab = [], []
        for r in range(n_rot):
            ext, tab = shot_item_gpu(rotate_z(np.asarray(verts, dtype=np.float64), ROT_DEGREES[r]), faces, res_h, device)
            per_ext.append(ext)
            per_tab.append(tab)
        extents.append(per_ext)
        tables.append(per_tab)
        volumes.append(mesh_volume(np.asarray(verts, dtype=np.float64), np.asarray(faces)))
    return ShapeSet(np.array(extents), np.array(volumes), tables, name=name, meta={"res_h": res_h, "n_rot": n_rot})


def box_mesh(ex: float, ey: float, ez: float):
    """Axis-aligned box [0,ex]x[0,ey]x[0,ez] as 12 triangles (tests, examples)."""
    v = np.array([[x, y, z] for x in (0, ex) for y in (0, ey) for z in (0, ez)], dtype=np.float64)
    f = np.array([[0, 1, 3], [0, 3, 2], [4, 6, 7], [4, 7, 5], [0, 4, 5], [0, 5, 1],
                  [2, 3, 7], [2, 7, 6], [0, 2, 6], [0, 6, 4], [1, 5, 7], [1, 7, 3]], dtype=np.int32)
    return v, f


def voxel_mesh(occ: np.ndarray, cube: float):
    """Surface mesh of a voxel solid (every exposed voxel face as two triangles)."""
    verts, faces = [], []
    nx, ny, nz = occ.shape
    quads = {  # outward faces: axis, sign -> corner offsets
        (0, 1): [(1, 0, 0), (1, 1, 0), (1, 1, 1), (1, 0, 1)], (0, -1): [(0, 0, 0), (0, 0, 1), (0, 1, 1), (0, 1, 0)],
        (1, 1): [(0, 1, 0), (0, 1, 1), (1, 1, 1), (1, 1, 0)], (1, -1): [(0, 0, 0), (1, 0, 0), (1, 0, 1), (0, 0, 1)],
        (2, 1): [(0, 0, 1), (1, 0, 1), (1, 1, 1), (0, 1, 1)], (2, -1): [(0, 0, 0), (0, 1, 0), (1, 1, 0), (1, 0, 0)]}
    for i in range(nx):
        for j in range(ny):
            for k in range(nz):
                if not occ[i, j, k]:
                    continue
                for (ax, sg), corners in quads.items():
                    n = [i, j, k]
                    n[ax] += sg
                    if 0 <= n[0] < nx and 0 <= n[1] < ny and 0 <= n[2] < nz and occ[tuple(n)]:
                        continue
                    base = len(verts)
                    for c in corners:
                        verts.append([(i + c[0]) * cube, (j + c[1]) * cube, (k + c[2]) * cube])
                    faces += [[base, base + 1, base + 2], [base, base + 2, base + 3]]
    return np.asarray(verts, dtype=np.float64), np.asarray(faces, dtype=np.int32)


def possible_position_custom(env, verts: np.ndarray, faces: np.ndarray, rot_idx: int = 0):
    """``Space.get_possible_position_custom`` (space.py:131-160): the overlap test of a mesh that is NOT in the data
    set, in the pose it is given (one "rotation"), on the current heightmaps of every bin of ``env``
    (a GpuPackingEnv).  The reference ray-casts the mesh (``shot_item``) and runs the same window loop as
    ``get_possible_position``; here: ``irbpp_shot_item``, then ``irbpp_possible_position`` of a scratch environment
    that shares the geometry and the heightmaps and holds just this shape.  Returns (posZmap, naiveMask) as
    float64 / uint8 device tensors [N, R, Ax, Ay] with only row ``rot_idx`` filled (posZmap 1e3, mask 0 elsewhere),
    which is what the reference leaves in ``self.posZmap`` / returns."""
    from .vec_env import GpuPackingEnv
    res_h = env.resolutionH
    ext, tab = shot_item_gpu(np.asarray(verts, dtype=np.float64), np.asarray(faces, dtype=np.int32), res_h, env.device)
    one = ShapeSet(np.array([[ext]]), np.array([mesh_volume(np.asarray(verts, dtype=np.float64), np.asarray(faces))]),
                   [[tab]], name="custom", meta={"res_h": res_h, "n_rot": 1})
    tmp = GpuPackingEnv(one, np.zeros((1, 1), dtype=np.int32), env.num_bins, device=env.device, resolutionA=env.resolutionA,
                        resolutionH=res_h, resolutionZ=env.resolutionZ, bin_dimension=env.bin_dimension,
                        selectedAction=env.S, bufferSize=1)
    try:
        tmp.reset()
        tmp.set_heightmaps(env.get_heightmaps())
        z1, m1 = tmp.possible_position(torch.zeros((env.num_bins,), dtype=torch.int32, device=env.device))
    finally:
        tmp.close()
    posz = torch.full((env.num_bins, env.n_rot, env.Ax, env.Ay), 1e3, dtype=torch.float64, device=env.device)
    mask = torch.zeros((env.num_bins, env.n_rot, env.Ax, env.Ay), dtype=torch.uint8, device=env.device)
    posz[:, rot_idx] = z1.reshape(env.num_bins, env.Ax, env.Ay)
    mask[:, rot_idx] = m1.reshape(env.num_bins, env.Ax, env.Ay)
    return posz, mask
