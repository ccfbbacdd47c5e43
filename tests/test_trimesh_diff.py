"""Differential switch for the third-party ray caster behind tools.shot_item (tools.py:98-135).

The reference builds every footprint table with trimesh (``mesh.ray.intersects_id(..., multiple_hits=False)`` from
below and from above, tools.py:109,124).  trimesh is not in this image, so ``oracle/shot.py`` states the geometry
directly and ``irbpp_shot_item`` (the z-ray/triangle rasteriser) is gated on that restatement: "parity unpinned"
(DESIGN.md 4).  Wherever a real ``trimesh`` IS importable this file diffs both against it on random convex and
concave meshes, upright and rotated; without trimesh it skips and says so.

The trimesh calls below are the reference's own lines (tools.py:81-95 ray grid, :98-135 the two casts and the
no-hit fall-backs), restated so that the test does not need /root/reference at run time.

Compared per mesh: maskB / maskH exactly wherever a ray passes at least 1e-7 m from every projected edge (a ray
through an edge is a tie between two triangles that any two ray engines may break differently), heights within 1e-9.
"""
import numpy as np
import pytest

trimesh = pytest.importorskip("trimesh", reason="trimesh is not installed: shot_item parity stays 'unpinned' (DESIGN.md 4)")

import irbpp_amd  # noqa: E402,F401
from irbpp_amd import meshes  # noqa: E402
from oracle.shot import shot_item as oracle_shot_item  # noqa: E402


def _reference_shot_item(verts, faces, res_h, shift=0.001):
    """tools.gen_ray_origin_direction + tools.shot_item with the real trimesh."""
    mesh = trimesh.Trimesh(vertices=np.asarray(verts, dtype=np.float64), faces=np.asarray(faces), process=False)
    mesh.apply_translation(-mesh.bounding_box.vertices[0])                          # tools.py:100
    x_range, y_range = np.ceil(np.round(mesh.extents[0:2], decimals=6) / res_h).astype(np.int32)
    bottom = np.arange(0, x_range * y_range).reshape((x_range, y_range))
    origin = np.zeros((x_range, y_range, 3))
    origin[:, :, 0] = bottom // y_range * res_h + shift                            # tools.py:87-88
    origin[:, :, 1] = bottom % y_range * res_h + shift
    origin[:, :, 2] = -10e2
    direction = np.zeros_like(origin)
    direction[:, :, 2] = 1
    ray_origins, ray_directions = origin.reshape(-1, 3).copy(), direction.reshape(-1, 3).copy()
    n = x_range * y_range
    B, mB, T, mH = np.zeros(n), np.zeros(n), np.zeros(n), np.zeros(n)
    _, idx, loc = mesh.ray.intersects_id(ray_origins=ray_origins, ray_directions=ray_directions,
                                         return_locations=True, multiple_hits=False)
    if len(idx) != 0:
        B[idx] = loc[:, 2]
        mB[idx] = 1
    else:
        mB[:] = 1
    ray_origins[:, 2] *= -1
    ray_directions[:, 2] *= -1
    _, idx, loc = mesh.ray.intersects_id(ray_origins=ray_origins, ray_directions=ray_directions,
                                         return_locations=True, multiple_hits=False)
    if len(idx) != 0:
        T[idx] = loc[:, 2]
        mH[idx] = 1
    else:
        T[:] = mesh.extents[2]
        mH[:] = 1
    shp = (x_range, y_range)
    return T.reshape(shp), B.reshape(shp), mH.reshape(shp), mB.reshape(shp)


def _random_meshes(seed, n):
    from scipy.spatial import ConvexHull
    rng = np.random.RandomState(seed)
    out = []
    for i in range(n):
        if i % 3 == 0:                                   # convex: hull of a random point cloud
            pts = rng.uniform(0, 1, (rng.randint(6, 30), 3)) * rng.uniform(0.04, 0.18, 3)
            hull = ConvexHull(pts)
            faces = hull.simplices.copy()
            centre = pts[hull.vertices].mean(0)          # orient outward (the rasteriser does not care, trimesh's volume does)
            for f in faces:
                a, b, c = pts[f]
                if np.dot(np.cross(b - a, c - a), a - centre) < 0:
                    f[1], f[2] = f[2], f[1]
            out.append((pts, faces.astype(np.int32)))
        elif i % 3 == 1:                                 # concave: polycubes (L / T / U shapes, overhangs)
            occ = rng.uniform(size=(3, 3, 2)) < 0.6
            occ[1, 1, 0] = True
            idx = np.argwhere(occ)
            lo, hi = idx.min(0), idx.max(0) + 1
            v, f = meshes.voxel_mesh(occ[lo[0]:hi[0], lo[1]:hi[1], lo[2]:hi[2]], float(rng.choice([0.03, 0.04, 0.05])))
            out.append((v, f))
        else:                                            # concave: a star-shaped prism with a slanted top
            k = rng.randint(5, 9)
            ang = np.linspace(0, 2 * np.pi, 2 * k, endpoint=False)
            rad = np.where(np.arange(2 * k) % 2 == 0, 0.08, 0.035) * rng.uniform(0.8, 1.2)
            ring = np.stack([rad * np.cos(ang), rad * np.sin(ang)], 1)
            h = 0.05 + 0.3 * ring[:, 0] + rng.uniform(0.02, 0.06)
            v = np.vstack([np.c_[ring, np.zeros(2 * k)], np.c_[ring, h], [[0, 0, 0]], [[0, 0, h.mean()]]])
            f = []
            for j in range(2 * k):
                j2 = (j + 1) % (2 * k)
                f += [[j, j2, 2 * k + j2], [j, 2 * k + j2, 2 * k + j], [4 * k, j2, j], [4 * k + 1, 2 * k + j, 2 * k + j2]]
            out.append((v, np.array(f, dtype=np.int32)))
    return out


def _edge_clear(verts, faces, res_h, shift, tol=1e-7):
    """True where the ray through a cell keeps `tol` away from every projected triangle edge."""
    v = np.asarray(verts, dtype=np.float64)
    v = v - v.min(0)
    fx, fy = np.ceil(np.round(v.max(0)[0:2], decimals=6) / res_h).astype(np.int32)
    px = (np.arange(fx) * res_h + shift)[:, None, None]
    py = (np.arange(fy) * res_h + shift)[None, :, None]
    ok = np.ones((fx, fy), bool)
    e = np.concatenate([faces[:, [0, 1]], faces[:, [1, 2]], faces[:, [2, 0]]])
    a, b = v[e[:, 0], 0:2], v[e[:, 1], 0:2]
    d = b - a
    ln = np.maximum(np.hypot(d[:, 0], d[:, 1]), 1e-30)
    t = np.clip(((px - a[:, 0]) * d[:, 0] + (py - a[:, 1]) * d[:, 1]) / ln ** 2, 0, 1)
    dist = np.hypot(px - (a[:, 0] + t * d[:, 0]), py - (a[:, 1] + t * d[:, 1]))
    ok &= (dist > tol).all(-1)
    return ok


def _compare(got, ref, ok):
    for name, g, r in zip(("heightMapT", "heightMapB", "maskH", "maskB"), got, ref):
        assert g.shape == r.shape, name
        if name.startswith("mask"):
            np.testing.assert_array_equal(g[ok], r[ok], err_msg=name)
        else:
            np.testing.assert_allclose(g[ok], r[ok], rtol=0, atol=1e-9, err_msg=name)


@pytest.mark.parametrize("res_h", [0.01, 0.005])
def test_oracle_shot_item_matches_trimesh(res_h):
    for verts, faces in _random_meshes(11, 12):
        for deg in (0.0, 45.0, 90.0):
            vr = meshes.rotate_z(np.asarray(verts, dtype=np.float64), deg)
            ref = _reference_shot_item(vr, faces, res_h)
            got = oracle_shot_item(vr - vr.min(0), faces, res_h)
            _compare(got, ref, _edge_clear(vr, faces, res_h, 0.001))


@pytest.mark.gpu
@pytest.mark.parametrize("res_h", [0.01, 0.005])
def test_gpu_shot_item_matches_trimesh(res_h):
    for verts, faces in _random_meshes(12, 24):
        for deg in (0.0, 45.0, 135.0, 270.0):
            vr = meshes.rotate_z(np.asarray(verts, dtype=np.float64), deg)
            ref = _reference_shot_item(vr, faces, res_h)
            ext, got = meshes.shot_item_gpu(vr, faces, res_h, "cuda:0")
            _compare(got, ref, _edge_clear(vr, faces, res_h, 0.001))
