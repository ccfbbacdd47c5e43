"""ORACLE (test infrastructure, not product code) -- restated OpenCV contour routines.

The reference calls two functions of ``opencv-contrib-python==4.4.0.46``
(requirements.txt:12), which is neither vendored under /root/reference nor
installed in this image:

    cv2.findContours(img, cv2.RETR_TREE, cv2.CHAIN_APPROX_SIMPLE)   cvTools.py:86
    cv2.approxPolyDP(contour, 1, True)                              cvTools.py:91

This module restates the published algorithms those calls execute in OpenCV 4.4
(modules/imgproc/src/contours.cpp: cvFindNextContour + icvFetchContourEx, i.e.
Suzuki & Abe 1985 border following on an 8-connected foreground; and
modules/imgproc/src/approx.cpp: approxPolyDP_<int>, Douglas-Peucker with the
three-pass farthest-point start and the final collinear clean-up).

PARITY UNPINNED for these two calls: the reference has no test or golden vector
for them and cv2 cannot be imported here, so this restatement is anchored only on
the reference's call sites and on the algorithm as published.  Everything that
*surrounds* the two calls (cvTools.py:7-102) is pinned against the reference's
own Python, see tests/golden/make_golden.py.
"""
from __future__ import annotations

from typing import List, Tuple

import numpy as np

# 8-neighbourhood direction codes of OpenCV (icvCodeDeltas): 0=E,1=NE,2=N,3=NW,4=W,5=SW,6=S,7=SE (y down)
_DX = (1, 1, 0, -1, -1, -1, 0, 1)
_DY = (0, -1, -1, -1, 0, 1, 1, 1)


def _fetch_contour(img: np.ndarray, x0: int, y0: int, is_hole: bool, nbd: int) -> List[Tuple[int, int]]:
    """icvFetchContourEx with CHAIN_APPROX_SIMPLE on the padded label image ``img``.

    ``img`` holds 0 (background), 1 (untouched foreground), ``nbd`` (visited) or
    ``-nbd`` (visited, and the pixel to its east is background: 'right bound').
    Returns the points in padded coordinates.
    """
    pts: List[Tuple[int, int]] = []
    s_end = s = 0 if is_hole else 4
    while True:                                   # find the first non-zero neighbour, clockwise
        s = (s - 1) & 7
        x1, y1 = x0 + _DX[s], y0 + _DY[s]
        if img[y1, x1] != 0 or s == s_end:
            break
    if s == s_end:                                # isolated pixel (the pixel at s_end is background
        img[y0, x0] = -nbd                        # by construction of the start conditions)
        pts.append((x0, y0))
        return pts

    x3, y3 = x0, y0
    prev_s = s ^ 4
    px, py = x0, y0
    while True:
        s_end = s
        while s < 15:                             # counter-clockwise search, at most 15 probes
            s += 1
            x4, y4 = x3 + _DX[s & 7], y3 + _DY[s & 7]
            if img[y4, x4] != 0:
                break
        s &= 7
        if ((s - 1) & 0xFFFFFFFF) < s_end:        # (unsigned)(s-1) < (unsigned)s_end: east was probed empty
            img[y3, x3] = -nbd
        elif img[y3, x3] == 1:
            img[y3, x3] = nbd
        if s != prev_s:                           # CHAIN_APPROX_SIMPLE: keep direction changes only
            pts.append((px, py))
        prev_s = s
        px += _DX[s]
        py += _DY[s]
        if (x4, y4) == (x0, y0) and (x3, y3) == (x1, y1):
            break
        x3, y3 = x4, y4
        s = (s + 4) & 7
    return pts


def find_contours(image: np.ndarray):
    """``cv2.findContours(image, RETR_TREE, CHAIN_APPROX_SIMPLE)`` -> (contours, hierarchy).

    contours: list of int32 arrays [n,1,2] holding (x, y) = (column, row);
    hierarchy: int32 [1, m, 4] rows ``[next, previous, first_child, parent]``.
    Non-zero pixels are foreground; the image is padded by one background pixel
    (OpenCV >= 3.2 does the same internally) so shapes touching the edge are traced.
    Contours are reported in raster discovery order (OpenCV's own output order differs;
    every consumer in the reference is order-independent, cvTools.py:7-38,100-101).
    """
    image = np.asarray(image)
    h, w = image.shape
    img = np.zeros((h + 2, w + 2), dtype=np.int32)
    img[1:-1, 1:-1] = (image != 0)
    contours: List[np.ndarray] = []
    is_hole_l: List[bool] = []
    parent_l: List[int] = []
    nbd = 1                                       # label 1 is the frame
    for y in range(1, h + 1):                     # OpenCV scans rows 1..H, columns 1..W of the padded image
        lnbd_x = 0                                # position of the last border pixel met on this row
        prev = 0
        x = 1
        while x <= w:
            p = int(img[y, x])
            if p == prev:
                x += 1
                continue
            is_hole = False
            start = False
            if prev == 0 and p == 1:              # outer border starts at (x, y)
                start = True
            elif p == 0 and prev >= 1:            # hole border starts at (x-1, y)
                is_hole = True
                start = True
                if prev > 1:
                    lnbd_x = x - 1
            if start:
                nbd += 1
                # parent from the last border met (Suzuki & Abe 1985, Table 1)
                if lnbd_x <= 0:
                    par = -1
                else:
                    b = abs(int(img[y, lnbd_x])) - 2
                    par = parent_l[b] if is_hole_l[b] == is_hole else b
                ox = x - 1 if is_hole else x
                lnbd_x = ox
                pts = _fetch_contour(img, ox, y, is_hole, nbd)
                contours.append(np.array([[px - 1, py - 1] for px, py in pts],
                                         dtype=np.int32).reshape(-1, 1, 2))
                is_hole_l.append(is_hole)
                parent_l.append(par)
                p = int(img[y, x])                # scanning resumes behind the relabelled pixel
            elif abs(p) > 1:
                lnbd_x = x
            prev = p
            x += 1
    m = len(contours)
    hier = -np.ones((m, 4), dtype=np.int32)
    last_child = {}
    for i in range(m):
        par = parent_l[i]
        hier[i, 3] = par
        if par in last_child:
            j = last_child[par]
            hier[j, 0] = i
            hier[i, 1] = j
        elif par >= 0:
            hier[par, 2] = i
        last_child[par] = i
    return contours, hier.reshape(1, m, 4), is_hole_l


def approx_poly_dp(curve: np.ndarray, epsilon: float, closed: bool = True) -> np.ndarray:
    """``cv2.approxPolyDP(curve, epsilon, closed)`` for integer points -> int32 [k,1,2].

    Only the closed form is exercised by the reference (cvTools.py:91).
    """
    src = np.asarray(curve, dtype=np.int64).reshape(-1, 2)
    count = len(src)
    if count == 0:
        return np.zeros((0, 1, 2), dtype=np.int32)
    assert closed, "the reference only approximates closed contours"
    eps = float(epsilon) * float(epsilon)
    dst: List[Tuple[int, int]] = []
    stack: List[Tuple[int, int]] = []

    # 1. approximately the two farthest points: three farthest-point hops
    pos = 0
    right_start = 0
    le_eps = False
    start_pt = (0, 0)
    for _ in range(3):
        max_dist = 0.0
        pos = (pos + right_start) % count
        start_pt = (int(src[pos, 0]), int(src[pos, 1]))
        pos = (pos + 1) % count
        for j in range(1, count):
            dx = float(src[pos, 0] - start_pt[0])
            dy = float(src[pos, 1] - start_pt[1])
            pos = (pos + 1) % count
            dist = dx * dx + dy * dy
            if dist > max_dist:
                max_dist = dist
                right_start = j
        le_eps = max_dist <= eps
    # 2. the two half-curves
    if not le_eps:
        s0 = pos % count
        far = (right_start + s0) % count
        stack.append((far, s0))                   # right slice
        stack.append((s0, far))                   # slice (processed first)
    else:
        dst.append(start_pt)

    # 3. Douglas-Peucker
    while stack:
        s_start, s_end = stack.pop()
        end_pt = (int(src[s_end, 0]), int(src[s_end, 1]))
        pos = s_start
        start_pt = (int(src[pos, 0]), int(src[pos, 1]))
        pos = (pos + 1) % count
        if pos != s_end:
            dx = float(end_pt[0] - start_pt[0])
            dy = float(end_pt[1] - start_pt[1])
            assert dx != 0 or dy != 0
            max_dist = 0.0
            split = 0
            while pos != s_end:
                ptx, pty = int(src[pos, 0]), int(src[pos, 1])
                pos = (pos + 1) % count
                dist = abs(float(pty - start_pt[1]) * dx - float(ptx - start_pt[0]) * dy)
                if dist > max_dist:
                    max_dist = dist
                    split = (pos + count - 1) % count
            le = max_dist * max_dist <= eps * (dx * dx + dy * dy)
        else:
            le = True
        if le:
            dst.append(start_pt)
        else:
            stack.append((split, s_end))
            stack.append((s_start, split))

    return np.array(cleanup_pass(dst, eps), dtype=np.int32).reshape(-1, 1, 2)


def cleanup_pass(dst, eps: float = 1.0):
    """Step 4 of approxPolyDP_ on its own (tests drive the device's clean-up routines with polygons of their choosing):
    `dst` is the polygon after the Douglas-Peucker recursion, a list of (x, y), `eps` the squared tolerance; returns what the
    pass leaves of it."""
    dst = list(dst)
    # 4. clean-up: drop points on [almost] straight lines (in place, as OpenCV does)
    count = len(dst)
    new_count = count
    pos = count - 1
    start_pt = dst[pos]
    pos = (pos + 1) % count
    wpos = pos
    pt = dst[pos]
    pos = (pos + 1) % count
    i = 0
    while i < count and new_count > 2:
        end_pt = dst[pos]
        pos = (pos + 1) % count
        dx = float(end_pt[0] - start_pt[0])
        dy = float(end_pt[1] - start_pt[1])
        dist = abs(float(pt[0] - start_pt[0]) * dy - float(pt[1] - start_pt[1]) * dx)
        inner = float(pt[0] - start_pt[0]) * float(end_pt[0] - pt[0]) + \
            float(pt[1] - start_pt[1]) * float(end_pt[1] - pt[1])
        if dist * dist <= 0.5 * eps * (dx * dx + dy * dy) and dx != 0 and dy != 0 and inner >= 0:
            new_count -= 1
            dst[wpos] = start_pt = end_pt
            wpos = (wpos + 1) % count
            pt = dst[pos]
            pos = (pos + 1) % count
            i += 2
            continue
        dst[wpos] = start_pt = pt
        wpos = (wpos + 1) % count
        pt = end_pt
        i += 1
    return dst[:new_count]
