"""ORACLE (test infrastructure) -- tools.shot_item (tools.py:98-135) restated with numpy.

The reference ray-casts with trimesh (``mesh.ray.intersects_id(..., multiple_hits=False)``): a ray
from below gives the lowest surface point over a grid cell, a ray from above the highest.  trimesh
is absent, so this states the geometry directly: a vertical line through (px, py) crosses a
triangle iff (px, py) lies in its xy-projection; the crossing height is the plane's z there.
PARITY UNPINNED against trimesh's own ray engine.
"""
import numpy as np


def shot_item(verts, faces, res_h, shift=0.001):
    """verts: [n,3] with the bounding-box minimum at the origin -> (T, B, maskH, maskB)."""
    verts = np.asarray(verts, dtype=np.float64)
    ext = verts.max(0)
    fx, fy = np.ceil(np.round(ext[0:2], decimals=6) / res_h).astype(np.int32)
    T = np.zeros((fx, fy)); B = np.zeros((fx, fy)); mH = np.zeros((fx, fy)); mB = np.zeros((fx, fy))
    for i in range(fx):
        for j in range(fy):
            px, py = i * res_h + shift, j * res_h + shift
            zs = []
            for f in faces:
                a, b, d = verts[f[0]], verts[f[1]], verts[f[2]]
                area = (b[0] - a[0]) * (d[1] - a[1]) - (b[1] - a[1]) * (d[0] - a[0])
                if area == 0.0:
                    continue
                w0 = (b[0] - px) * (d[1] - py) - (b[1] - py) * (d[0] - px)
                w1 = (d[0] - px) * (a[1] - py) - (d[1] - py) * (a[0] - px)
                w2 = (a[0] - px) * (b[1] - py) - (a[1] - py) * (b[0] - px)
                inside = (w0 >= 0 and w1 >= 0 and w2 >= 0) if area > 0 else (w0 <= 0 and w1 <= 0 and w2 <= 0)
                if inside:
                    zs.append(a[2] if (a[2] == b[2] == d[2]) else (w0 * a[2] + w1 * b[2] + w2 * d[2]) / area)
            if zs:
                T[i, j], B[i, j], mH[i, j], mB[i, j] = max(zs), min(zs), 1.0, 1.0
    if mB.sum() == 0:                      # tools.py:112-117,126-131
        B[:] = 0; mB[:] = 1; T[:] = ext[2]; mH[:] = 1
    return T, B, mH, mB
