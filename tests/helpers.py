"""Shared scenario builders for the tests (same seeds as tests/golden/make_golden.py)."""
import numpy as np

import irbpp_amd  # noqa: F401
from irbpp_amd import synthetic


def minz_action(obs, S=500):
    """Scripted policy: lowest-H valid candidate, first on ties; 0 when nothing is valid."""
    c = np.asarray(obs[:5 * S]).reshape(S, 5)
    v = c[:, 4] == 1
    if not v.any():
        return 0
    return int(np.argmin(np.where(v, c[:, 3], np.inf)))


def golden_scenario(name):
    if name == "online_cube":
        sh = synthetic.cube_shapes()
    elif name in ("online_blockout", "hier_blockout_k3"):
        sh = synthetic.blockout_shapes(n_shapes=24, n_rot=4, cube=0.06, seed=0)
    elif name == "online_general":
        sh = synthetic.general_shapes(n_shapes=16, n_rot=8, seed=1)
    elif name == "heuristic_general":
        sh = synthetic.general_shapes(n_shapes=16, n_rot=4, seed=21)
    elif name == "heuristic_blockout":
        sh = synthetic.blockout_shapes(n_shapes=24, n_rot=4, cube=0.06, seed=0)
    else:
        raise KeyError(name)
    return sh
