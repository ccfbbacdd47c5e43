#!/usr/bin/env python
"""Tooling (CPU): how long are the border walks of the trace kernel, lane by lane, and what would sorting the candidate
starts by a cheap length proxy buy?  Steady-state bins from the plain-C oracle -> level images (cvTools.py:77-85) ->
candidate starts (contours_device.h: start_candidates) -> iterations of trace_border_fast per candidate (the host build of
the device routine, tests/host/) -> waves of 64 candidates in arrival order / sorted by each proxy / sorted by the true
length (the bound).  A wave walks until its longest border is done: its cost is the MAX over its lanes.

    python tools/trace_length_study.py [workload] [bins] [steps]"""
import ctypes as C, json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from bench import make_workload
from oracle.c_oracle import COracleVecEnv
import subprocess
HOST = os.path.join(ROOT, "tests", "host")


def _host_lib():
    out = os.path.join(HOST, "_build", "libcontours_host.so")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    subprocess.run(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-Wno-unused-value", "-I", os.path.join(HOST, "stub"),
                    os.path.join(HOST, "contours_host.cpp"), "-o", out], check=True)
    return C.CDLL(out)



wl = sys.argv[1] if len(sys.argv) > 1 else "blockout"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 512
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 60
lib = _host_lib()
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
rng = np.random.RandomState(0)
for t in range(steps):                       # bins deep in their own episodes, out of phase
    obs, _, _, _ = env.step(minz(obs))
rows_t = (C.c_uint16 * 16)
cands = []                                   # (bin, iters, points, popcount, ncand_image, runs_image, first_run_len, rows_below)
pts = (C.c_uint8 * 256)()
for b, e in enumerate(env.envs):
    pz, mk = e.grids()
    R = pz.shape[0]
    for r in range(R):
        lev = np.where(mk[r] > 0, np.floor_divide(pz[r], 0.01), -1).astype(np.int32)      # cvTools.py:78-79
        for h in np.unique(lev[lev >= 0]):
            img = (lev == h)                 # img[row = lx, col = ly]; OpenCV pixel (x = col, y = row)
            words = rows_t(*[int(sum(1 << x for x in range(16) if img[y, x])) for y in range(16)])
            out = (C.c_uint32 * 16)()
            lib.host_start_candidates(words, out)
            pop = int(img.sum())
            runs = int(sum(bin(words[y] & ~(words[y] << 1) & 0xFFFF).count("1") for y in range(16)))
            starts = [(x, y) for y in range(16) for x in range(16) if (out[y] >> x) & 1]
            for x, y in starts:
                row = words[y]
                iso = not ((row >> (x + 1)) & 1) and (y == 15 or not ((words[y + 1] >> max(x - 1, 0)) & (7 if x else 3)))
                if iso:
                    continue                 # isolated pixels never reach the trace kernel
                lib.host_trace_iters()
                npts = lib.host_trace_border_fast(words, x, y, pts, 255)
                it = lib.host_trace_iters()
                run_len = 0
                while x + run_len < 16 and (row >> (x + run_len)) & 1:
                    run_len += 1
                below = 0
                while y + below < 16 and (words[y + below] >> x) & 1:
                    below += 1
                cands.append((b, it, npts, pop, len(starts), runs, run_len, below))
c = np.array(cands)
it = c[:, 1].astype(np.float64)
def waves(order):
    v = it[order]
    pad = (-len(v)) % 64
    v = np.concatenate([v, np.zeros(pad)]).reshape(-1, 64)
    return v.max(1)
arr = waves(np.arange(len(it)))
res = {"workload": wl, "bins": n, "candidates": len(it), "per_bin": len(it) / n, "false_starts": float((c[:, 2] == 0).mean()),
       "lane_iters": {"mean": it.mean(), "p50": np.percentile(it, 50), "p90": np.percentile(it, 90), "p99": np.percentile(it, 99), "max": it.max()},
       "hist_iters": np.bincount(np.minimum(it.astype(int) // 8, 15)).tolist(),
       "wave_max_arrival": {"mean": arr.mean(), "p99": np.percentile(arr, 99), "max": arr.max(), "max_over_lane_mean": arr.mean() / it.mean()}}
proxies = {"true_length": it, "popcount": c[:, 3], "ncand_image": -c[:, 4], "runs_image": c[:, 5], "first_run": c[:, 6], "column_below": c[:, 7],
           "run_x_below": c[:, 6] * c[:, 7], "pop_over_ncand": c[:, 3] / c[:, 4], "run_plus_below": c[:, 6] + c[:, 7]}
for name, p in proxies.items():
    w = waves(np.argsort(-p, kind="stable"))
    res["sorted_by_" + name] = {"wave_mean": w.mean(), "vs_arrival": w.mean() / arr.mean(), "corr": float(np.corrcoef(p, it)[0, 1])}
    for k in (2, 3, 4):                      # k classes by quantile of the proxy, arrival order inside a class
        q = np.quantile(p, np.linspace(0, 1, k + 1)[1:-1])
        cls = np.searchsorted(q, p, side="right")
        w = waves(np.argsort(-cls, kind="stable"))
        res["sorted_by_" + name][f"classes{k}"] = w.mean() / arr.mean()
print(json.dumps(res, indent=1, default=float))

# ---- lane refill: a wave owns a batch of `batch` candidates and hands a lane the next one when enough lanes idle ----
def simulate_refill(iters, batch, threshold, refill_cost, tail_cost):
    """cost in walk-iteration equivalents of all waves: every loop iteration costs 1 whatever the number of walking lanes; a
    refill round (flush the finished borders, stage the new images, start the walks) costs `refill_cost`; per batch `tail_cost`
    (first staging + last flush)"""
    total = 0.0
    for s in range(0, len(iters), batch):
        q = list(iters[s:s + batch])
        lanes = [0] * 64
        nxt = 0
        cost = tail_cost
        for l in range(64):
            if nxt < len(q):
                lanes[l] = q[nxt]; nxt += 1
        while True:
            act = [v for v in lanes if v > 0]
            if not act:
                if nxt >= len(q):
                    break
            idle = 64 - len(act)
            if nxt < len(q) and (idle >= threshold or not act):
                cost += refill_cost
                for l in range(64):
                    if lanes[l] <= 0 and nxt < len(q):
                        lanes[l] = q[nxt]; nxt += 1
                continue
            # walk until the refill condition can change: the smallest remaining count among walking lanes
            step = min(act)
            cost += step
            lanes = [v - step if v > 0 else 0 for v in lanes]
        total += cost
    return total

if os.environ.get("REFILL", "1") != "0":
    ii = np.maximum(it, 1).astype(int)          # (a false start still costs its iterations; isolated pixels are not listed)
    base = simulate_refill(ii, 64, 65, 0.0, 7.7 + 14.0)
    out = {"now_per_candidate": base / len(ii)}
    for batch in (128, 256, 512):
        for thr in (16, 24, 32, 48):
            for rc in (4.0, 6.0):
                c = simulate_refill(ii, batch, thr, rc, 7.7 + 14.0)
                out[f"batch{batch}_thr{thr}_refill{rc}"] = round(c / base, 3)
    print(json.dumps(out, indent=1))
