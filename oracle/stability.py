"""ORACLE (test infrastructure) -- numpy statement of the stability proxy of ``irbpp_config::stability``.

NOT a restatement of reference code: the reference settles every placed item with pybullet
(``simulateToQuasistatic``, Interface.py:271-310), which the hot path leaves out.  The proxy rates a placement
by a static support test; this file is its specification, the HIP kernel its implementation:

  * column solid: where both rays hit (maskH and maskB), the item occupies [heightMapB, heightMapT]; its centre
    of mass is the thickness-weighted mean of the cell centres (i + 0.5, j + 0.5), in heightmap cells;
  * contact cells: masked-in bottom cells whose gap ``H[window] - heightMapB`` is within half a height level
    (0.5 * resolutionZ) of the drop height posZ;
  * stable iff the centre of mass lies inside the octagonal hull of the contact cells, i.e. for the eight
    directions (+-1,0), (0,+-1), (+-1,+-1): d . com <= max over contact-cell corners of d . corner.
"""
import numpy as np

DIRS = ((1, 0), (-1, 0), (0, 1), (0, -1), (1, 1), (-1, -1), (1, -1), (-1, 1))


def centre_of_mass(T, B, maskH, maskB):
    w = np.where((maskH != 0) & (maskB != 0), np.maximum(T - B, 0.0), 0.0)
    fx, fy = T.shape
    if w.sum() <= 0:
        return 0.5 * fx, 0.5 * fy
    ii, jj = np.meshgrid(np.arange(fx) + 0.5, np.arange(fy) + 0.5, indexing="ij")
    # same accumulation order as the library (row-major running sums), so the comparison below is exact
    mass = mx = my = 0.0
    for i in range(fx):
        for j in range(fy):
            if w[i, j] > 0 or ((maskH[i, j] != 0) and (maskB[i, j] != 0)):
                mass += w[i, j]; mx += w[i, j] * (i + 0.5); my += w[i, j] * (j + 0.5)
    return mx / mass, my / mass


def placement_is_stable(window, T, B, maskH, maskB, posz, resolution_z):
    gap = window - B
    contact = (maskB != 0) & (gap >= posz - 0.5 * resolution_z)
    if not contact.any():
        return False
    ci, cj = np.nonzero(contact)
    cx, cy = centre_of_mass(T, B, maskH, maskB)
    for dx, dy in DIRS:
        support = (dx * ci + dy * cj + max(dx, 0) + max(dy, 0)).max()
        if dx * cx + dy * cy > support:
            return False
    return True
