"""ORACLE (test infrastructure) -- candidate actions from iso-height contours.

Restates environment/physics0/cvTools.py of the reference:
    find_out_contour      cvTools.py:7-38
    find_convex_vetex     cvTools.py:40-59
    getConvexHullActions  cvTools.py:61-75
    convexHulls           cvTools.py:77-102
with the two cv2 calls served by oracle/contours.py.
"""
from __future__ import annotations

import numpy as np

from .contours import approx_poly_dp, find_contours


def find_out_contour(contour, hierarchy):
    """Keep contours at even nesting depth (outer borders), cvTools.py:7-38.
    ``hierarchy`` rows are [next, previous, first_child, parent]."""
    valid, invalid = [], []
    level_counter = 0
    this_level = list(np.where(hierarchy[:, -1] == -1)[0])
    valid.extend(this_level)
    while len(valid) + len(invalid) != len(hierarchy):
        next_level = []
        for i in this_level:
            child = hierarchy[i, 2]
            if child != -1:
                next_level.append(child)
                pointer = child
                while hierarchy[pointer][0] != -1:           # following siblings
                    next_level.append(hierarchy[pointer][0])
                    pointer = hierarchy[pointer][0]
                pointer = child
                while hierarchy[pointer][1] != -1:           # preceding siblings
                    next_level.append(hierarchy[pointer][1])
                    pointer = hierarchy[pointer][1]
        if level_counter % 2 != 0:
            valid.extend(next_level)
        else:
            invalid.extend(next_level)
        level_counter += 1
        this_level = next_level
    return [contour[i] for i in valid], valid


def find_convex_vetex(approx):
    """Indices of polygon vertices B with cross(B-A, C-A) < 0 (A previous, C next);
    all of them when the polygon has <= 3 vertices.  cvTools.py:40-59."""
    length = len(approx)
    if length <= 3:
        return np.arange(length)
    vertex = np.array(approx)[:, 0, :]
    last_vertex = np.roll(vertex, 1, axis=0)
    next_vertex = np.roll(vertex, -1, axis=0)
    AB = vertex - last_vertex
    AC = next_vertex - last_vertex
    cross = AB[:, 0] * AC[:, 1] - AB[:, 1] * AC[:, 0]
    return np.where(cross < 0)[0]


def convexHulls(posZMap, mask, heightResolution=0.01):
    """Candidate (x=col, y=row) points of one rotation, cvTools.py:77-102."""
    mapInt = (posZMap // heightResolution).astype(np.int32)   # numpy floor_divide on float64 (:78)
    mapInt[mask == 0] = -1
    allCandidates = []
    for h in np.unique(mapInt):
        if h == -1:
            continue
        check = np.where(mapInt == h, 255, 0).astype(np.uint8)
        contours, hierarchy, _ = find_contours(check)
        newContour, _ = find_out_contour(contours, hierarchy[0])
        for c in newContour:
            approx = approx_poly_dp(c, 1, True)
            convexIndex = find_convex_vetex(approx)
            allCandidates.append(approx[convexIndex].reshape((-1, 2)))
    V = None
    if len(allCandidates) != 0:
        allCandidates = np.concatenate(allCandidates, axis=0)
        allCandidates = np.unique(allCandidates, axis=0)
        V = mask[(allCandidates[:, 1], allCandidates[:, 0])]
    return allCandidates, V


def getConvexHullActions(posZValid, mask, heightResolution):
    """Rows ``[rot, row(lx), col(ly), H, V]`` over all rotations or None, cvTools.py:61-75."""
    allCandidates = []
    for rotIdx in range(len(posZValid)):
        allHulls, V = convexHulls(posZValid[rotIdx], mask[rotIdx], heightResolution)
        if len(allHulls) != 0:
            H = posZValid[rotIdx][allHulls[:, 1], allHulls[:, 0]]
            ROT = np.ones((len(allHulls))) * rotIdx
            allCandidates.append(np.concatenate(
                [ROT.reshape(-1, 1), allHulls[:, 1].reshape(-1, 1), allHulls[:, 0].reshape(-1, 1),
                 H.reshape(-1, 1), V.reshape(-1, 1)], axis=1))
    if len(allCandidates) != 0:
        return np.concatenate(allCandidates, axis=0)
    return None
