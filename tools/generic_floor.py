#!/usr/bin/env python
"""Tooling: the issue floors of the generic overlap loop for a workload, from its shape tables and the issue costs
measured by tools/microbench_issue.hip on the MI355X (profiles/r03/sessions/s*):

    VALU wave-instruction 4.4 cycles of its SIMD; ds_read_b64 2.2 cycles of the CU's LDS; one 64-byte scalar load
    (four cells) per 14-18 cycles per CU from L2.

A (footprint cell, row group) pair costs one LDS read and 28/12 VALU instructions (four cells x three row groups per
trip: 4 address adds + 12 subtracts + 12 maxima), a cell of a task one quarter of a scalar load.  Prints pairs per
bin-step (mean over the data set's items) and the three floors of one step of `bins` bins on 256 CUs / 1024 SIMDs.

    python tools/generic_floor.py --workload abc_fine --bins 4096"""
import argparse
import math
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import make_workload  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="general")
ap.add_argument("--bins", type=int, default=4096)
ap.add_argument("--bin-width", type=float, default=0.32)
a = ap.parse_args()

shapes, _, kw = make_workload(a.workload)
res_a, res_h = kw["resolutionA"], kw["resolutionH"]
step = int(round(res_a / res_h))
hn = int(round(a.bin_width / res_h))
an = int(math.ceil(hn / step))
ysh = max(0, (an - 1).bit_length())
rpw = 64 >> ysh                                           # rows of the action grid per wave
pairs, cells, tasks = [], [], []
for s in range(shapes.n_shapes):
    p = c = t = 0
    for r in range(shapes.n_rot):
        tab = shapes.tables[s][r]
        fx = np.asarray(tab[0]).shape[0]
        nb = int((np.asarray(tab[-1]) > 0).sum())          # masked-in bottom cells
        wx = an - int(math.ceil(fx / step)) + 1            # rows in range
        ng = max(0, math.ceil(wx / rpw))                   # row groups in range
        nt = 0 if ng == 0 else (1 if ng <= 3 else 2)       # tasks: up to three groups each, four split 2 + 2
        p += nb * ng
        c += nb * nt
        t += nt
    pairs.append(p)
    cells.append(c)
    tasks.append(t)
P, C = float(np.mean(pairs)), float(np.mean(cells))
GHZ, SIMDS, CUS = 2.4, 1024, 256
valu = a.bins * P * (28 / 12) * 4.4 / SIMDS / GHZ * 1e-3   # us
lds = a.bins * P * 2.2 / CUS / GHZ * 1e-3
smem = a.bins * C / 4 * 16 / CUS / GHZ * 1e-3
print(f"{a.workload}: action grid {an}x{an}, step {step}, R = {shapes.n_rot}; per bin-step {P:.0f} (cell, row group) pairs, "
      f"{C:.0f} scalar cells, {np.mean(tasks):.1f} tasks")
print(f"floors of one step of {a.bins} bins: VALU {valu:.1f} us, LDS {lds:.1f} us, scalar loads {smem:.1f} us "
      f"(one row group per task would need {a.bins * P / 4 * 16 / CUS / GHZ * 1e-3:.1f} us of scalar loads)")
