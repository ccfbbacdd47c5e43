"""Per-shape footprint tables in the reference's ``shotInfo`` format.

The packing environment never touches meshes on its hot path; it consumes, per
(shape id, z-rotation), four ``[fx, fy]`` float64 arrays produced once at
start-up by ``shotInfoPre`` / ``shot_item`` (reference tools.py:248-279,
tools.py:98-135):

    heightMapT  top-surface z of the shape (bbox-min at the origin)
    heightMapB  bottom-surface z
    maskH       1 where the top-down ray hit, else 0
    maskB       1 where the bottom-up ray hit, else 0

plus ``extents`` (``mesh.extents``, tools.py:242) and ``volume``
(``infoDict[id][0]['volume']``, binPhy.py:152,156).  ``ShapeSet`` is that data
for a whole dataset, held as plain numpy so it can be handed to the CPU oracle
and packed for the HIP library alike.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, List, Tuple

import numpy as np

Table = Tuple[np.ndarray, np.ndarray, np.ndarray, np.ndarray]  # (T, B, maskH, maskB)


def np_round6(x):
    """``np.round(x, decimals=6)`` as the reference applies it (space.py:104)."""
    return np.round(np.asarray(x, dtype=np.float64), decimals=6)


def grid_extent(extent_xy, resolution) -> np.ndarray:
    """``np.ceil(round(extents,6)[0:2] / res).astype(int32)`` (space.py:104-106)."""
    return np.ceil(np_round6(extent_xy) / resolution).astype(np.int32)


@dataclass
class ShapeSet:
    """A dataset of shapes: ``tables[id][rot] = (T, B, maskH, maskB)``."""

    extents: np.ndarray                 # [n, R, 3] float64, raw mesh.extents per rotation
    volumes: np.ndarray                 # [n] float64, volume of rotation 0
    tables: List[List[Table]]           # [n][R] of four [fx, fy] float64 arrays
    name: str = "synthetic"
    meta: Dict[str, object] = field(default_factory=dict)

    def __post_init__(self):
        self.extents = np.ascontiguousarray(self.extents, dtype=np.float64)
        self.volumes = np.ascontiguousarray(self.volumes, dtype=np.float64)
        assert self.extents.ndim == 3 and self.extents.shape[2] == 3
        assert len(self.tables) == self.n_shapes
        for per_rot in self.tables:
            assert len(per_rot) == self.n_rot

    @property
    def n_shapes(self) -> int:
        return int(self.extents.shape[0])

    @property
    def n_rot(self) -> int:
        return int(self.extents.shape[1])

    # -- the reference's own containers -------------------------------------------------
    def shot_info(self) -> Dict[int, List[Table]]:
        """``args.shotInfo`` (tools.py:248-279): id -> list over rotations of 4-tuples."""
        return {k: list(self.tables[k]) for k in range(self.n_shapes)}

    def info_dict(self) -> Dict[int, List[dict]]:
        """``args.infoDict`` (tools.py:240-242)."""
        return {k: [{"volume": float(self.volumes[k]), "extents": self.extents[k, r].copy()}
                    for r in range(self.n_rot)] for k in range(self.n_shapes)}

    def validate(self, resolution_h: float, resolution_a: float) -> None:
        """Check the invariants ``get_possible_position`` relies on (space.py:104-119):
        table shape == ceil(round(extents)/resH) and the window never clips."""
        step = int(resolution_a / resolution_h)
        assert step == resolution_a / resolution_h, "resolutionA must be a multiple of resolutionH"
        for k in range(self.n_shapes):
            for r in range(self.n_rot):
                fx, fy = grid_extent(self.extents[k, r, 0:2], resolution_h)
                ax, ay = grid_extent(self.extents[k, r, 0:2], resolution_a)
                for arr in self.tables[k][r]:
                    if arr.shape != (fx, fy):
                        raise ValueError(f"shape {k} rot {r}: table {arr.shape} != ({fx},{fy})")
                if fx > ax * step or fy > ay * step:
                    raise ValueError(f"shape {k} rot {r}: footprint exceeds its action window")
