#!/usr/bin/env python
"""Analysis (not a test; lives under tests/ because it drives the CPU oracle, which only tests may import):
what sets the trace kernel's duration, and would a different treatment of long borders shorten it?

    python tests/analysis/trace_walk_sim.py [workload]          # ~2 minutes of the numpy oracle

1. runs the numpy oracle for 48 bins (140 warm-up + 6 recorded steps, scripted MINZ policy) and records every level
   image cv2.findContours is called on (cvTools.py:77-102);
2. walks every candidate start pixel (W, NW, N, NE background: the trace kernel's candidates) with the per-pixel
   icvFetchContourEx walk and counts the iterations the device's run-jumping walk needs (a same-direction axis-aligned
   move continues the previous iteration's run) and the pixel steps;
3. groups the candidates into waves of 64 in launch order and prices (a) the lockstep walk, calibrated on the measured
   mean of 26.8 k trace cycles per wave (profiles/r03/final/trace_blockout.json), and (b) a hybrid that hands the last
   K walking lanes of a wave to a whole-wave successor-table walk (3.3 k cycles to build a table, 110 cycles per pixel
   step).

Result for blockout (round 3): 34 candidates per bin-step, 8.7 iterations on average, 2.26 pixel steps per iteration;
a wave's longest walk is 33 iterations on average (p99 51) = 806 cycles per iteration = the walk's ~140 instructions at
the 5.7 cycles a lone wave needs per VALU instruction (tools/microbench_issue.hip, one wave per SIMD) -- the trace
kernel has 1.3 waves per SIMD.  The hybrid gains 2 % on the mean and nothing on the slowest waves: a wave's long borders
come several at a time.  What shortens the kernel is fewer instructions per iteration or fewer iterations."""
import os
import pickle
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from bench import make_workload  # noqa: E402
from helpers import minz_action  # noqa: E402
from oracle import cvtools  # noqa: E402
from oracle.packing import OracleVecEnv  # noqa: E402

DX = (1, 1, 0, -1, -1, -1, 0, 1)
DY = (0, -1, -1, -1, 0, 1, 1, 1)
NB, WARM, REC = 48, 140, 6


def collect(workload):
    shapes, seqs, kw = make_workload(workload)
    env = OracleVecEnv(NB, shapes, seqs, **kw)
    obs = np.asarray(env.reset(), dtype=np.float64).astype(np.float32)
    images, rec = [], [False]
    orig = cvtools.find_contours

    def spy(check):
        if rec[0]:
            images.append((np.asarray(check) != 0).astype(np.uint8))
        return orig(check)

    cvtools.find_contours = spy
    sel = kw.get("selectedAction", 500)
    try:
        for t in range(WARM + REC):
            rec[0] = t >= WARM
            obs, _, _, _ = env.step(np.array([minz_action(o, sel) for o in obs]))
            obs = np.asarray(obs, dtype=np.float64).astype(np.float32)
    finally:
        cvtools.find_contours = orig
    return images


def walk(img, x0, y0):
    """(iterations of the run-jumping walk, pixel steps) of the border from (x0, y0) on the padded image; the walk stops
    early at a pixel that precedes the start in raster order (not the first pixel of its component / a hole border)."""
    s_end = s = 4
    while True:
        s = (s - 1) & 7
        x1, y1 = x0 + DX[s], y0 + DY[s]
        if img[y1, x1] or s == s_end:
            break
    if s == s_end:
        return 0, 0                                           # isolated pixel: marked by the transition kernel itself
    x3, y3, iters, pix, last = x0, y0, 0, 0, -1
    while True:
        k = s
        for _ in range(8):
            k = (k + 1) & 7
            x4, y4 = x3 + DX[k], y3 + DY[k]
            if img[y4, x4]:
                break
        pix += 1
        if not (k == last and (k & 1) == 0):
            iters += 1
        last = k
        if (x4, y4) == (x0, y0) and (x3, y3) == (x1, y1):
            return iters, pix
        if y4 * 64 + x4 < y0 * 64 + x0:
            return iters, pix
        x3, y3, s = x4, y4, (k + 4) & 7


def main():
    workload = sys.argv[1] if len(sys.argv) > 1 else "blockout"
    cache = f"/tmp/irbpp_trace_sim_{workload}.pkl"
    if os.path.exists(cache):
        images = pickle.load(open(cache, "rb"))
    else:
        t0 = time.time()
        images = collect(workload)
        pickle.dump(images, open(cache, "wb"))
        print(f"{len(images)} level images from {NB} bins x {REC} steps in {time.time() - t0:.0f} s")
    it, px = [], []
    for im in images:
        h, w = im.shape
        p = np.zeros((h + 2, w + 2), np.uint8)
        p[1:-1, 1:-1] = im
        for y, x in zip(*np.nonzero(p)):
            if p[y, x - 1] or p[y - 1, x - 1] or p[y - 1, x] or p[y - 1, x + 1]:
                continue
            a, b = walk(p, x, y)
            if a:
                it.append(a)
                px.append(b)
    it, px = np.array(it), np.array(px)
    print(f"{workload}: {len(it) / (NB * REC):.1f} candidates per bin-step, {it.mean():.1f} iterations each (max {it.max()}), "
          f"{px.sum() / it.sum():.2f} pixel steps per iteration")
    n = len(it) // 64 * 64
    g_it, g_px = it[:n].reshape(-1, 64), px[:n].reshape(-1, 64)
    mx = g_it.max(1)
    c_step = 26766.0 / mx.mean()
    print(f"{len(mx)} waves: longest walk {mx.mean():.1f} iterations on average, p99 {np.percentile(mx, 99):.0f}, max {mx.max()}; "
          f"{c_step:.0f} cycles per iteration at the measured 26.8 k cycles per wave")
    base = c_step * mx
    print(f"lockstep: mean {base.mean():.0f}, p99 {np.percentile(base, 99):.0f}, max {base.max():.0f} cycles")
    for keep in (1, 2, 4):
        t = []
        for gi, gp in zip(g_it, g_px):
            top = int(gi.max())
            sw = next((k for k in range(8, top + 1) if (gi > k).sum() <= keep), None)
            if sw is None or not (gi > sw).any():
                t.append(c_step * top)
                continue
            rem = gi > sw
            left = gp[rem] * (gi[rem] - sw) / gi[rem]
            t.append(min(c_step * top, c_step * sw + rem.sum() * 3300.0 + 110.0 * left.max()))
        t = np.array(t)
        print(f"hybrid, last {keep} lane(s) by successor table: mean {t.mean():.0f}, p99 {np.percentile(t, 99):.0f}, max {t.max():.0f} cycles")


if __name__ == "__main__":
    main()
