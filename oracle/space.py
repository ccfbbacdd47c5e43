"""ORACLE (test infrastructure) -- bin geometry state, restating
environment/physics0/space.py of the reference (float64 numpy, same op order).

    Space.__init__                 space.py:15-47
    Space.reset                    space.py:49-52
    Space.get_possible_position    space.py:98-129
    Space.place_item               space.py:75-94 reduced to the closed form the
                                   reference itself states at space.py:213
    Space.get_heuristic_action     space.py:162-227 (MINZ/DBLF/FIRSTFIT/HM)
"""
from __future__ import annotations

import numpy as np


class Space(object):
    def __init__(self, bin_dimension, resolutionAct, resolutionH, ZRotNum, shotInfo, extents):
        self.bin_dimension = np.asarray(bin_dimension, dtype=np.float64)
        self.resolutionH = resolutionH
        self.resolutionAct = resolutionAct
        self.stepSize = int(self.resolutionAct / self.resolutionH)
        assert self.stepSize == self.resolutionAct / self.resolutionH          # space.py:20
        self.rotNum = ZRotNum
        self.rangeX_C, self.rangeY_C = np.ceil(self.bin_dimension[0:2] / resolutionH).astype(np.int32)
        self.rangeX_A, self.rangeY_A = np.ceil(self.bin_dimension[0:2] / resolutionAct).astype(np.int32)
        self.heightmapC = np.zeros((self.rangeX_C, self.rangeY_C))
        self.shotInfo = shotInfo          # id -> [rot] -> (heightMapT, heightMapB, maskH, maskB)
        self.extents = extents            # [n, R, 3] raw mesh extents (next_item[rotIdx].extents)
        self.posZmap = np.zeros((self.rotNum, self.rangeX_A, self.rangeY_A))
        self.posZValid = np.zeros((self.rotNum, self.rangeX_A, self.rangeY_A))
        self.naiveMask = np.zeros((self.rotNum, self.rangeX_A, self.rangeY_A))
        bottom = np.arange(0, self.rangeX_A * self.rangeY_A).reshape((self.rangeX_A, self.rangeY_A))
        self.coors = np.zeros((self.rangeX_A, self.rangeY_A, 2))
        self.coors[:, :, 0] = bottom // self.rangeY_A
        self.coors[:, :, 1] = bottom % self.rangeY_A

    def reset(self):
        self.heightmapC[:] = 0

    def get_possible_position(self, next_item_ID):
        """space.py:98-129 -- brute-force overlap test over (rot, X, Y)."""
        rotNum = self.rotNum
        naiveMask = np.zeros((rotNum, self.rangeX_A, self.rangeY_A))
        self.posZmap[:] = 1e3
        if next_item_ID is not None and next_item_ID >= 0:
            for rotIdx in range(rotNum):
                boundingSize = np.round(self.extents[next_item_ID][rotIdx], decimals=6)
                rangeX_OH, rangeY_OH = np.ceil(boundingSize[0:2] / self.resolutionH).astype(np.int32)
                rangeX_OA, rangeY_OA = np.ceil(boundingSize[0:2] / self.resolutionAct).astype(np.int32)
                heightMapT, heightMapB, maskH, maskB = self.shotInfo[next_item_ID][rotIdx]
                for X in range(self.rangeX_A - rangeX_OA + 1):
                    for Y in range(self.rangeY_A - rangeY_OA + 1):
                        coorX, coorY = X * self.stepSize, Y * self.stepSize
                        posZ = np.max((self.heightmapC[coorX: coorX + rangeX_OH, coorY: coorY + rangeY_OH]
                                       - heightMapB) * maskB)
                        if np.round(posZ + boundingSize[2] - self.bin_dimension[2], decimals=6) <= 0:
                            naiveMask[rotIdx, X, Y] = 1
                        self.posZmap[rotIdx, X, Y] = posZ
        self.naiveMask = naiveMask.copy()
        invalidIndex = np.where(naiveMask == 0)
        self.posZValid[:] = self.posZmap[:]
        self.posZValid[invalidIndex] = 1e3
        return naiveMask

    def place_item(self, item_ID, rotIdx, lx, ly, posZ):
        """Heightmap update after a placement without physics motion.

        The reference ray-casts the placed mesh (place_item_trimesh, space.py:75-94 +
        tools.py:137-161) and takes ``np.maximum`` with the window.  For an item that
        stays where it was dropped the ray heights are ``heightMapT + posZ`` on
        ``maskH`` and 0 elsewhere, which is the closed form the reference writes at
        space.py:213.  trimesh is absent here: PARITY UNPINNED for the ray cast itself.
        """
        heightMapT, heightMapB, maskH, maskB = self.shotInfo[item_ID][rotIdx]
        fx, fy = heightMapT.shape
        X, Y = lx * self.stepSize, ly * self.stepSize
        win = self.heightmapC[X:X + fx, Y:Y + fy]
        self.heightmapC[X:X + fx, Y:Y + fy] = np.maximum(win, (heightMapT + posZ) * maskH)

    def get_heuristic_action(self, method, next_item_ID, dirIdx=0):
        """space.py:162-227 (without RANDOM).  Returns (rotIdx, lx, ly)."""
        Xflip, Yflip = [(False, False), (False, True), (True, False), (True, True)][dirIdx]
        invalidIndex = np.where(self.naiveMask == 0)
        coorsX = self.coors[:, :, 0] if not Xflip else self.rangeX_A - self.coors[:, :, 0]
        coorsY = self.coors[:, :, 1] if not Yflip else self.rangeY_A - self.coors[:, :, 1]
        if method == 'MINZ':
            score = self.posZmap.copy()
        elif method == 'DBLF':
            score = (coorsX + coorsY).reshape((1, -1)).repeat(self.rotNum, axis=0).reshape(self.naiveMask.shape)
            score = score * self.resolutionAct + 100 * self.posZmap
        elif method == 'FIRSTFIT':
            score = (coorsX + coorsY).reshape((1, -1)).repeat(self.rotNum, axis=0).reshape(self.naiveMask.shape)
        else:
            assert method == 'HM'
            score = ((coorsX + coorsY) * self.resolutionAct)
            score = score.reshape((1, -1)).repeat(self.rotNum, axis=0).reshape(self.naiveMask.shape)
        score = score.astype(np.float64)
        score[invalidIndex] = 1e6
        if method == 'HM':
            for rotIdx in range(self.rotNum):
                heightMapT, heightMapB, maskH, maskB = self.shotInfo[next_item_ID][rotIdx]
                fx, fy = heightMapT.shape
                for coorX in range(self.rangeX_A):
                    for coorY in range(self.rangeY_A):
                        if self.naiveMask[rotIdx, coorX, coorY] == 0:
                            continue
                        posZ = self.posZmap[rotIdx, coorX, coorY]
                        X, Y = coorX * self.stepSize, coorY * self.stepSize
                        prime = np.max(((heightMapT + posZ) * maskH, self.heightmapC[X:X + fx, Y:Y + fy]), axis=0)
                        score[rotIdx, coorX, coorY] += np.sum(prime) * 100
        score = np.round(score, decimals=6)
        index = np.argmin(score)
        return np.unravel_index(index, score.shape)
