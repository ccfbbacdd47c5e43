#!/usr/bin/env python
"""Analysis (not a test; lives under tests/ because it drives the CPU oracle, which only tests may import): three counts on
the level images cv2.findContours is called on (cvTools.py:77-102), behind decisions of round 4 (profiles/r04/LOG.md).

    python tests/analysis/level_image_stats.py [workload] [what ...]      # what: starts depth heavy (default: all)

starts  candidate start pixels under the local rule (W, NW, N, NE background) and under the per-run rule of
        contours_device.h::start_candidates (no pixel of the row above touches the pixel's horizontal run anywhere), against
        the true starts (first pixel of every outer border of more than one point); checks that the run rule loses none.
        Session 30: BlockOut 3623 -> 2803 candidates for 2442 borders, general 7068 -> 6345 for 5168.
depth   recursion depth of the level-synchronous Douglas-Peucker per border and per round of 128 points, greedy packing in
        candidate order against packing by decreasing length (what a sort of a wave's borders would buy the polygon
        kernel).  BlockOut: 5.6 -> 4.5 levels per round, general 4.7 -> 3.7.
heavy   starts + isolated pixels of a bin against its number of convex vertices (the emit kernel's candidate rows): which
        threshold separates the bins with more than S rows.  general, session 31 (48 bins over 60 steps): a > S bin has >= 350
        (median 404), 99 % of the others < 325; this script's sample (32 bins, 3 states): >= 402, 99 % of the others < 343.
"""
import sys
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from oracle import contours as OC, cvtools  # noqa: E402
from oracle.c_oracle import COracleVecEnv  # noqa: E402


def minz(o, S=500):
    c = o[:5 * S].reshape(S, 5)
    v = c[:, 4] == 1
    return int(np.argmin(np.where(v, c[:, 3], np.inf))) if v.any() else 0


def level_images(env):
    """(bin, rotation, image [16, 16] bool) for every level of every rotation of every bin's current observation"""
    for bi, e in enumerate(env.envs):
        pz, mk = e.grids()
        for r in range(pz.shape[0]):
            valid = mk[r] == 1
            levels = np.where(valid, (pz[r] // 0.01).astype(np.int64), -1)
            for h in np.unique(levels):
                if h >= 0:
                    img = levels == h
                    if img.shape != (16, 16):
                        img = np.pad(img, ((0, 16 - img.shape[0]), (0, 16 - img.shape[1])))
                    yield bi, r, img


def row_words(img):
    return [int(sum(1 << x for x in range(16) if img[y, x])) for y in range(16)]


def candidates(rw):
    """per image: (local-rule candidates, run-rule candidates, isolated pixels) as sets of (x, y)"""
    loc, run, iso = set(), set(), set()
    for y in range(16):
        row, up, dn = rw[y], (rw[y - 1] if y > 0 else 0), (rw[y + 1] if y < 15 else 0)
        upm = up | (up << 1) | (up >> 1)
        first = row & ~(row << 1) & ~upm & 0xFFFF
        isolated = first & ~(row >> 1) & ~dn & ~(dn << 1) & ~(dn >> 1)
        t = upm & row
        for _ in range(16):
            t |= (t >> 1) & row
        for x in range(16):
            if (isolated >> x) & 1:
                iso.add((x, y))
            elif (first >> x) & 1:
                loc.add((x, y))
                if not (t >> x) & 1:
                    run.add((x, y))
    return loc, run, iso


def dp_depth(pts):
    """levels of approx_convex_segmented's Douglas-Peucker loop for one closed border (after its three hops)"""
    p = np.array(pts)
    cnt = len(p)
    if cnt < 3:
        return 0
    pos = rs = md = 0
    for _ in range(3):
        pos = (pos + rs) % cnt
        md, rs = 0, 0
        for j in range(1, cnt):
            d = int(((p[(pos + j) % cnt] - p[pos]) ** 2).sum())
            if d > md:
                md, rs = d, j
    if md <= 1:
        return 0
    far = (pos + rs) % cnt
    slices, depth = [(pos, far), (far, pos)], 0
    while slices:
        depth += 1
        nxt = []
        for a, b in slices:
            idx, k = [], (a + 1) % cnt
            while k != b:
                idx.append(k)
                k = (k + 1) % cnt
            if not idx:
                continue
            dx, dy = (p[b] - p[a]).tolist()
            dist = [abs(int((p[i][1] - p[a][1]) * dx - (p[i][0] - p[a][0]) * dy)) for i in idx]
            m = max(dist)
            if m * m > dx * dx + dy * dy:
                sp = idx[dist.index(m)]
                nxt += [(a, sp), (sp, b)]
        slices = [s for s in nxt if (s[0] + 1) % cnt != s[1]]
    return depth


def main():
    wl = sys.argv[1] if len(sys.argv) > 1 else "blockout"
    what = set(sys.argv[2:]) or {"starts", "depth", "heavy"}
    shapes, seqs, kw = bench.make_workload(wl)
    env = COracleVecEnv(32, shapes, seqs, **kw)
    obs = env.reset()
    for _ in range(120):
        obs, _, _, _ = env.step([minz(o) for o in obs])
    n_loc = n_run = n_true = n_iso = lost = 0
    borders, per_bin = [], {}
    for _rep in range(3):
        for _ in range(5):
            obs, _, _, _ = env.step([minz(o) for o in obs])
        for bi, r, img in level_images(env):
            cs, _, hole = OC.find_contours(img.astype(np.uint8))
            outer = [c.reshape(-1, 2) for c, ho in zip(cs, hole) if not ho]
            starts = {tuple(int(v) for v in c[0]) for c in outer if len(c) > 1}
            loc, run, iso = candidates(row_words(img))
            n_loc, n_run, n_true, n_iso = n_loc + len(loc), n_run + len(run), n_true + len(starts), n_iso + len(iso)
            lost += len(starts - run)
            acc = per_bin.setdefault((_rep, bi), [0, 0, 0])
            acc[0] += len(loc) + len(iso)
            acc[1] += len(run) + len(iso)
            for c in outer:
                if "depth" in what and len(c) > 1:
                    borders.append((len(c), dp_depth([tuple(q) for q in c])))
                if "heavy" in what:
                    acc[2] += len(cvtools.find_convex_vetex(OC.approx_poly_dp(c.reshape(-1, 1, 2), 1, True)))
    if "starts" in what:
        print(f"{wl}: candidates local rule {n_loc}, run rule {n_run}, true starts {n_true}, isolated {n_iso}; "
              f"false share {1 - n_true / max(1, n_loc):.3f} -> {1 - n_true / max(1, n_run):.3f}; true starts lost {lost}")
        assert lost == 0
    if "depth" in what and borders:
        a = np.array(borders)

        def rounds(seq):
            out, fill, d = [], 0, 0
            for n, dd in seq:
                if fill + n > 128:
                    out.append(d)
                    fill, d = 0, 0
                fill, d = fill + n, max(d, dd)
            return out + ([d] if fill else [])
        g, s = [], []
        for i in range(0, len(a), 40):
            ch = [tuple(x) for x in a[i:i + 40]]
            g += rounds(ch)
            s += rounds(sorted(ch, key=lambda x: -x[0]))
        print(f"{wl}: {len(a)} borders, {a[:, 0].sum()} points, depth mean {a[:, 1].mean():.2f} max {a[:, 1].max()}; levels per round "
              f"{np.mean(g):.2f} (candidate order, {len(g)} rounds) vs {np.mean(s):.2f} (longest first, {len(s)} rounds)")
    if "heavy" in what:
        b = np.array(list(per_bin.values()))
        S = 500
        hv, lt = b[b[:, 2] > S], b[b[:, 2] <= S]
        for name, col in (("local rule", 0), ("run rule", 1)):
            print(f"{wl}: starts + isolated, {name}: bins with > {S} rows {len(hv)} (min {hv[:, col].min() if len(hv) else '-'}, "
                  f"median {np.median(hv[:, col]) if len(hv) else '-'}); others median {np.median(lt[:, col]):.0f}, p99 {np.percentile(lt[:, col], 99):.0f}, "
                  f"max {lt[:, col].max()}; r = {np.corrcoef(b[:, col], b[:, 2])[0, 1]:.4f}")


if __name__ == "__main__":
    main()
