#!/usr/bin/env python
"""Tooling (CPU): how many of the borders the trace kernel follows belong to ISOLATED SOLID RECTANGLES -- components whose
contour is their four corners (two end points for a line), which the transition kernel could recognise from the rows of the
level image and answer from a (w, h) table instead of handing them to the trace kernel?  Steady-state bins of the plain-C
oracle -> level images -> candidate starts -> per candidate: is its component an isolated solid rectangle; iterations and
points of its border (host build of the device routine).  Reports the share of candidates, of wave time (a wave of 64
candidates costs the MAX of its lanes' iterations + a fixed part) and of contour points that would leave the trace / polygon
kernels.

    python tools/rect_component_study.py [workload] [bins] [steps]"""
import ctypes as C, json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from bench import make_workload
from oracle.c_oracle import COracleVecEnv
import subprocess
HOST = os.path.join(ROOT, "tests", "host")

out = os.path.join(HOST, "_build", "libcontours_host.so")
os.makedirs(os.path.dirname(out), exist_ok=True)
subprocess.run(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-Wno-unused-value", "-I", os.path.join(HOST, "stub"),
                os.path.join(HOST, "contours_host.cpp"), "-o", out], check=True)
lib = C.CDLL(out)
wl = sys.argv[1] if len(sys.argv) > 1 else "blockout"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 512
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 60
lib.host_trace_border_fast.argtypes = [C.POINTER(C.c_uint16), C.c_int, C.c_int, C.POINTER(C.c_uint8), C.c_int]
lib.host_trace_iters.restype = C.c_long
lib.host_start_candidates.argtypes = [C.POINTER(C.c_uint16), C.POINTER(C.c_uint32)]
sh, seqs, kw = make_workload(wl)
env = COracleVecEnv(n, sh, seqs[:2000], threads=8, **kw)
obs = env.reset()
S = 500
def minz(obs):
    c = obs[:, :5 * S].reshape(len(obs), S, 5).astype(np.float32); v = c[:, :, 4] == 1
    return np.where(v.any(1), np.argmin(np.where(v, c[:, :, 3], np.inf), 1), 0)
for t in range(steps):
    obs, _, _, _ = env.step(minz(obs))
rows_t = (C.c_uint16 * 16)
pts = (C.c_uint8 * 256)()
cands = []          # (bin, iters, points, is_rect, w, h)
def rect_at(words, x0, y0):
    row = words[y0] >> x0
    w = 0
    while x0 + w < 16 and (row >> w) & 1:
        w += 1
    m = ((1 << w) - 1) << x0
    mb = (m | (m << 1) | (m >> 1)) & 0xFFFF
    if y0 > 0 and words[y0 - 1] & mb:
        return 0, 0
    h = 0
    while y0 + h < 16 and (words[y0 + h] & mb) == m:
        h += 1
    if y0 + h < 16 and words[y0 + h] & mb:
        return 0, 0
    return w, h
for b, e in enumerate(env.envs):
    pz, mk = e.grids()
    for r in range(pz.shape[0]):
        lev = np.where(mk[r] > 0, np.floor_divide(pz[r], 0.01), -1).astype(np.int32)
        for hh in np.unique(lev[lev >= 0]):
            img = (lev == hh)
            words = [int(sum(1 << x for x in range(16) if img[y, x])) for y in range(16)]
            cw = rows_t(*words)
            o = (C.c_uint32 * 16)()
            lib.host_start_candidates(cw, o)
            for y in range(16):
                for x in range(16):
                    if not (o[y] >> x) & 1:
                        continue
                    row = words[y]
                    iso = not ((row >> (x + 1)) & 1) and (y == 15 or not ((words[y + 1] >> max(x - 1, 0)) & (7 if x else 3)))
                    if iso:
                        continue
                    lib.host_trace_iters()
                    npts = lib.host_trace_border_fast(cw, x, y, pts, 255)
                    it = lib.host_trace_iters()
                    w, h = rect_at(words, x, y)
                    cands.append((b, it, npts, 1 if w else 0, w, h))
c = np.array(cands)
it = c[:, 1].astype(float); rect = c[:, 3] == 1
FIXED = 21.7        # staging + flush of a wave in walk-iteration equivalents (trace_length_study's refill model)
def wave_cost(v):
    pad = (-len(v)) % 64
    v = np.concatenate([v, np.zeros(pad)]).reshape(-1, 64)
    return float((v.max(1) + FIXED).sum())
now, then = wave_cost(it), wave_cost(it[~rect])
print(json.dumps({"workload": wl, "bins": n, "candidates_per_bin": len(it) / n, "rect_share_of_candidates": float(rect.mean()),
                  "false_start_share": float((c[:, 2] == 0).mean()),
                  "lane_iters_mean": {"all": it.mean(), "rect": it[rect].mean() if rect.any() else None, "other": it[~rect].mean()},
                  "wave_time_after_over_now": then / now,
                  "points_share_in_rects": float(c[rect, 2].sum() / max(1, c[:, 2].sum())),
                  "rect_sizes_top": [[int(k[0]), int(k[1]), int(v)] for k, v in sorted(
                      __import__("collections").Counter(map(tuple, c[rect][:, 4:6])).items(), key=lambda kv: -kv[1])[:12]]}, indent=1))
