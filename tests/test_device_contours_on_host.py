"""The lane-serial device routines of irbpp_amd/csrc/contours_device.h (candidate starts, border
tracing with run jumps, flattened Douglas-Peucker + convexity) compiled for the HOST by
tests/host/contours_host.cpp and run against the oracle on thousands of 16x16 images: the code the
GPU executes, checked without a GPU.  (The wave-cooperative Douglas-Peucker needs real lanes and is
covered by the -m gpu tests and, here, by tests/host/wave_host.cpp: 64 host threads in lockstep with
every cross-lane operation emulated as a barrier-bracketed exchange.)"""
import ctypes as C
import os
import shutil
import subprocess

import numpy as np
import pytest
from scipy import ndimage

from oracle import contours as OC
from oracle import cvtools

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "host", "contours_host.cpp")
OUT = os.path.join(HERE, "host", "_build", "libcontours_host.so")
u16p, u32p, u8p = C.POINTER(C.c_uint16), C.POINTER(C.c_uint32), C.POINTER(C.c_uint8)


WAVE_SRC = os.path.join(HERE, "host", "wave_host.cpp")
WAVE_OUT = os.path.join(HERE, "host", "_build", "libwave_host.so")


@pytest.fixture(scope="module")
def wave():
    if shutil.which("g++") is None:
        pytest.skip("no host compiler")
    os.makedirs(os.path.dirname(WAVE_OUT), exist_ok=True)
    subprocess.run(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-pthread", "-Wno-unused-value",
                    "-I", os.path.join(HERE, "host", "stub"), WAVE_SRC, "-o", WAVE_OUT], check=True)
    lib = C.CDLL(WAVE_OUT)
    lib.host_approx_convex_segmented.argtypes = [u8p, C.POINTER(C.c_int), C.c_int, u32p, C.c_int]
    lib.host_approx_convex_segmented.restype = C.c_int
    return lib


@pytest.fixture(scope="module")
def host():
    if shutil.which("g++") is None:
        pytest.skip("no host compiler")
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    subprocess.run(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-Wno-unused-value"]
                   + os.environ.get("IRBPP_HOST_TEST_DEFINES", "").split()        # e.g. -DIRBPP_COMPACT_FRAMES=1: the A/B forms
                   + ["-I", os.path.join(HERE, "host", "stub"), SRC, "-o", OUT], check=True)
    lib = C.CDLL(OUT)
    lib.host_start_candidates.argtypes = [u16p, u32p]
    lib.host_trace_border.argtypes = [u16p, C.c_int, C.c_int, u8p, C.c_int]
    lib.host_trace_border.restype = C.c_int
    lib.host_trace_border_fast.argtypes = [u16p, C.c_int, C.c_int, u8p, C.c_int]
    lib.host_trace_border_fast.restype = C.c_int
    lib.host_trace_border_fast_spill.argtypes = [u16p, C.c_int, C.c_int, u8p, C.c_int, C.c_int]
    lib.host_trace_border_fast_spill.restype = C.c_int
    lib.host_wide_image_vertices.argtypes = [u32p, C.c_int, C.c_int, C.c_int, C.c_int, u32p, u32p, C.POINTER(C.c_int)]
    lib.host_wide_image_vertices.restype = C.c_int
    lib.host_wide_runs_equal_plain.argtypes = [u32p, C.c_int, C.c_int]
    lib.host_wide_runs_equal_plain.restype = C.c_int
    lib.host_trace_border_walk.argtypes = [u16p, C.c_int, C.c_int, u8p, C.c_int]
    lib.host_trace_border_walk.restype = C.c_int
    lib.host_trace_border_walk_spill.argtypes = [u16p, C.c_int, C.c_int, u8p, C.c_int, C.c_int]
    lib.host_trace_border_walk_spill.restype = C.c_int
    lib.host_approx_and_convex.argtypes = [u8p, C.c_int, C.c_int, u32p]
    lib.host_approx_and_convex.restype = C.c_int
    lib.host_contour_vertices.argtypes = [u16p, C.c_int, C.c_int, C.c_int, C.c_int, u32p]
    lib.host_contour_vertices.restype = C.c_int
    lib.host_rect_component.argtypes = [u16p, C.c_int, C.c_int, u32p]
    lib.host_rect_component.restype = C.c_int
    return lib


def _images(seed, n):
    rng = np.random.RandomState(seed)
    for k in range(n):
        kind = k % 5
        if kind == 0:
            img = rng.rand(16, 16) < rng.uniform(0.15, 0.85)
        elif kind == 1:
            img = ndimage.binary_dilation(rng.rand(16, 16) < 0.12, structure=np.ones((3, 3))) & ~(rng.rand(16, 16) < 0.1)
        elif kind == 2:                                   # unions of rectangles: level sets of box stacks
            img = np.zeros((16, 16), dtype=bool)
            for _ in range(rng.randint(1, 6)):
                y, x = rng.randint(0, 16), rng.randint(0, 16)
                img[y:y + rng.randint(1, 9), x:x + rng.randint(1, 9)] ^= True
        elif kind == 3:                                   # rings, full frame included
            img = np.zeros((16, 16), dtype=bool)
            for m in range(rng.randint(0, 2), 8, rng.randint(2, 4)):
                img[m:16 - m, m:16 - m] = True
                img[m + 1:15 - m, m + 1:15 - m] = False
        else:                                             # diagonals and staircases
            img = np.zeros((16, 16), dtype=bool)
            for _ in range(rng.randint(1, 5)):
                y, x, d = rng.randint(0, 16), rng.randint(0, 16), rng.choice([-1, 1])
                for s in range(rng.randint(2, 12)):
                    yy, xx = y + s, x + d * s
                    if 0 <= yy < 16 and 0 <= xx < 16:
                        img[yy, xx:xx + rng.randint(1, 3)] = True
        yield img.astype(np.uint8)


def _rows(img):
    return (C.c_uint16 * 16)(*[int(sum(int(img[y, x]) << x for x in range(16))) for y in range(16)])


def _oracle_outer(img):
    contours, hierarchy, is_hole = OC.find_contours(img)
    return [[tuple(int(v) for v in p) for p in c.reshape(-1, 2)] for c, h in zip(contours, is_hole) if not h], contours, is_hole


@pytest.mark.parametrize("seed", range(8))
def test_candidates_and_trace_equal_the_oracle_borders(host, seed):
    pts = (C.c_uint8 * 512)()
    cand = (C.c_uint32 * 16)()
    for img in _images(seed, 150):
        rows = _rows(img)
        outer, _, _ = _oracle_outer(img)
        starts = {c[0]: c for c in outer}
        host.host_start_candidates(rows, cand)
        listed = {(x, y) for y in range(16) for x in range(16) if (cand[y] >> x) & 1}
        assert set(starts) <= listed                                   # every true start is a candidate
        for trace in (host.host_trace_border, host.host_trace_border_fast, host.host_trace_border_walk):     # the plain walk, the trace kernel's, and its start / iteration form
            for (x, y) in listed:
                n = trace(rows, x, y, pts, 511)
                if (x, y) in starts:
                    got = [(pts[i] & 15, pts[i] >> 4) for i in range(n)]
                    assert got == starts[(x, y)]                           # same points, same order
                else:
                    assert n == 0                                          # a later pixel of some component
            if outer:                                                      # a tight slot reports the true length
                longest = max(outer, key=len)
                assert trace(rows, longest[0][0], longest[0][1], pts, 3) == len(longest)


@pytest.mark.parametrize("seed", range(4))
def test_trace_with_split_slot_equals_the_oracle_borders(host, seed):
    """The trace kernel keeps the first 56 points of a border in LDS and the rest in global scratch
    (irbpp_kernels.hip: TRACE_LDS_CAP / TRACE_SPILL): the walk with a split slot, at splits that every border crosses."""
    pts = (C.c_uint8 * 512)()
    for img in _images(300 + seed, 120):
        rows = _rows(img)
        outer, _, _ = _oracle_outer(img)
        for c in outer:
            for cap_lds, spill_cap in ((1, 255), (3, 200), (7, 3), (56, 72)):
                for trace in (host.host_trace_border_fast_spill, host.host_trace_border_walk_spill):
                    n = trace(rows, c[0][0], c[0][1], pts, cap_lds, spill_cap)
                    assert n == len(c)                                     # the true length, whatever fits
                    m = min(n, cap_lds + spill_cap)
                    assert [(pts[i] & 15, pts[i] >> 4) for i in range(m)] == c[:m]


def _oracle_vertices(contour_xy):
    c = np.array(contour_xy, dtype=np.int32).reshape(-1, 1, 2)
    approx = OC.approx_poly_dp(c, 1, True)
    keep = cvtools.find_convex_vetex(approx)
    return {tuple(int(v) for v in approx[i, 0]) for i in keep}


@pytest.mark.parametrize("seed", range(8))
def test_vertices_equal_approxpolydp_plus_convexity(host, seed):
    vrows = (C.c_uint32 * 16)()
    overflowed = 0
    for img in _images(100 + seed, 150):
        rows = _rows(img)
        outer, _, _ = _oracle_outer(img)
        for c in outer:
            for i in range(16):
                vrows[i] = 0
            st = host.host_contour_vertices(rows, c[0][0], c[0][1], 64, 12, vrows)     # the kernel's slot sizes
            if st == 1:                                                # outgrew the slot: the kernel redoes it big
                overflowed += 1
                for i in range(16):
                    vrows[i] = 0
                st = host.host_contour_vertices(rows, c[0][0], c[0][1], 1360, 1360, vrows)
            assert st == 0
            got = {(x, y) for y in range(16) for x in range(16) if (vrows[y] >> x) & 1}
            assert got == _oracle_vertices(c)
    assert overflowed < 150 * 4                                        # the redo path exists but is the exception


def test_flattened_douglas_peucker_on_synthetic_polygons(host):
    """Point lists that do not come from a trace: long staircases, spikes, repeated points."""
    rng = np.random.RandomState(7)
    vrows = (C.c_uint32 * 16)()
    for _ in range(400):
        n = rng.randint(1, 60)
        x, y = rng.randint(0, 16), rng.randint(0, 16)
        poly = [(x, y)]
        for _ in range(n - 1):                                         # 8-connected random walk inside the grid
            x = int(np.clip(x + rng.randint(-1, 2), 0, 15))
            y = int(np.clip(y + rng.randint(-1, 2), 0, 15))
            if (x, y) != poly[-1]:
                poly.append((x, y))
        arr = (C.c_uint8 * len(poly))(*[px | (py << 4) for px, py in poly])
        assert host.host_approx_and_convex(arr, len(poly), 1360, vrows) == 1
        got = {(xx, yy) for yy in range(16) for xx in range(16) if (vrows[yy] >> xx) & 1}
        assert got == _oracle_vertices(poly)


def _cleanup_changes(poly):
    """Does the final clean-up pass of approxPolyDP change this contour's Douglas-Peucker polygon?  Checked with
    an independent recursion that stops before the clean-up (hops, then split-or-accept per slice)."""
    P, n = np.array(poly, dtype=np.int64), len(poly)
    if n == 1:
        return False
    pos = rs = md = 0
    for _ in range(3):
        pos, md = (pos + rs) % n, 0
        for j in range(1, n):
            d = int(((P[(pos + j) % n] - P[pos]) ** 2).sum())
            if d > md:
                md, rs = d, j
    if md <= 1:
        return False
    s0, far = pos, (pos + rs) % n
    keep, work = {s0, far}, [(s0, far), (far, s0)]
    while work:
        a, b = work.pop()
        ln = (b - a) % n or n
        dx, dy = (P[b] - P[a]).tolist()
        best, sp = 0, -1
        for t in range(1, ln):
            q = P[(a + t) % n]
            d = abs(int((q[1] - P[a][1]) * dx - (q[0] - P[a][0]) * dy))
            if d > best:
                best, sp = d, (a + t) % n
        if best * best > dx * dx + dy * dy:
            keep.add(sp)
            work += [(a, sp), (sp, b)]
    before = [tuple(int(v) for v in P[j]) for j in sorted(keep, key=lambda j: (j - s0) % n)]
    c = np.array(poly, dtype=np.int32).reshape(-1, 1, 2)
    after = [tuple(int(v) for v in q) for q in OC.approx_poly_dp(c, 1, True).reshape(-1, 2)]
    return before != after


@pytest.mark.parametrize("ppl", [1, 2])
def test_segmented_douglas_peucker_in_lockstep_emulation(wave, ppl):
    """approx_convex_segmented<P>: approxPolyDP + convexity for several borders at once, P contour points per
    lane (64 * P per wave and round), as the kernels run it.  Borders are packed back to back exactly like the
    kernels pack them, including borders whose polygon the sequential clean-up pass of approxPolyDP changes
    (handled wave-uniformly inside the routine) and, for P = 2, borders of more than 64 points."""
    room0 = 64 * ppl

    def run(polys):
        counts = (C.c_int * len(polys))(*[len(q) for q in polys])
        flat = [x | (y << 4) for q in polys for x, y in q]
        arr = (C.c_uint8 * max(1, len(flat)))(*flat)
        vrows = (C.c_uint32 * (16 * len(polys)))()
        assert wave.host_approx_convex_segmented(arr, counts, len(polys), vrows, ppl) == 0
        for b, q in enumerate(polys):
            got = {(x, y) for y in range(16) for x in range(16) if (vrows[b * 16 + y] >> x) & 1}
            assert got == _oracle_vertices(q), q
        return sum(1 for q in polys if _cleanup_changes(q))

    def walk(rng, n):
        x, y = rng.randint(0, 16), rng.randint(0, 16)
        poly = [(x, y)]
        while len(poly) < n:
            x = int(np.clip(x + rng.randint(-1, 2), 0, 15))
            y = int(np.clip(y + rng.randint(-1, 2), 0, 15))
            if (x, y) != poly[-1]:
                poly.append((x, y))
        return poly

    done = flagged = 0
    batch, room = [], room0
    for img in _images(300, 70):
        outer, _, _ = _oracle_outer(img)
        for c in outer:
            if not 1 <= len(c) <= room0:
                continue
            if len(c) > room:
                flagged += run(batch)
                done += len(batch)
                batch, room = [], room0
            batch.append(c)
            room -= len(c)
    flagged += run(batch)
    done += len(batch)
    rng = np.random.RandomState(11)
    for _ in range(60 // ppl):                                        # arbitrary closed walks, packed three to a wave
        polys = [walk(rng, rng.randint(1, 30 * ppl)), walk(rng, rng.randint(1, 20 * ppl)), walk(rng, rng.randint(1, 15))]
        flagged += run(polys)
        done += 3
    full = [(i % 16, (i // 16) * 2 + (i % 2)) for i in range(64)]     # one border filling 64 lanes
    flagged += run([full])
    if ppl == 2:                                                      # borders of more than 64 points, alone and straddling the words
        for seed in range(12):
            r2 = np.random.RandomState(100 + seed)
            flagged += run([walk(r2, r2.randint(65, 129))])
            flagged += run([walk(r2, r2.randint(20, 50)), walk(r2, r2.randint(40, 78))])
            done += 3
    assert done > 300 and flagged >= 3                                # the clean-up branch was exercised


def test_loop_free_cleanup_pass_equals_the_sequential_ones(wave):
    """cleanup_convex_parallel (removed(i) = T(i) and not removed(i-1), by bit operations on the ballot of T) against the
    two sequential statements of approx.cpp's clean-up pass, on polygons built to make the pass fire: vertices on
    near-straight diagonals, runs of several removable vertices, removable first and last vertices (the in-place
    wrap-around), and polygons the pass cuts down to two vertices (which the routine must hand over)."""
    for f in (wave.host_cleanup_parallel, wave.host_cleanup_wave, wave.host_cleanup_serial):
        f.argtypes = [u8p, C.c_int, u32p]
    wave.host_cleanup_parallel.restype = C.c_int
    rng = np.random.RandomState(5)
    vp, vw, vs = ((C.c_uint32 * 16)() for _ in range(3))
    settled = handed = changed = first_goes = 0
    for case in range(1500):
        cnt = int(rng.randint(3, 65)) if case % 3 else int(rng.randint(3, 9))
        kind = case % 4
        pts = []
        x, y = int(rng.randint(0, 16)), int(rng.randint(0, 16))
        dx, dy = 1, 1
        while len(pts) < cnt:
            pts.append((x, y))
            if kind == 0:                                              # anything
                x, y = int(rng.randint(0, 16)), int(rng.randint(0, 16))
            else:                                                      # drifting diagonals with small kinks: many removable triples
                if rng.rand() < (0.15 if kind == 1 else 0.4):
                    dx, dy = int(rng.choice([-1, 1])), int(rng.choice([-1, 1]))
                sx, sy = dx * int(rng.randint(1, 3)), dy * int(rng.randint(1, 3))
                if rng.rand() < 0.3:
                    sx += int(rng.choice([-1, 0, 1]))
                x, y = int(np.clip(x + sx, 0, 15)), int(np.clip(y + sy, 0, 15))
        arr = (C.c_uint8 * cnt)(*[px | (py << 4) for px, py in pts])
        after = OC.cleanup_pass(pts, 1.0)                               # the oracle's statement of the pass ...
        poly = np.array(after, dtype=np.int32).reshape(-1, 1, 2)
        want = {tuple(int(v) for v in poly[i, 0]) for i in cvtools.find_convex_vetex(poly)}     # ... and of the convexity test
        bits = lambda v: {(xx, yy) for yy in range(16) for xx in range(16) if (v[yy] >> xx) & 1}
        wave.host_cleanup_serial(arr, cnt, vs)
        assert bits(vs) == want, pts
        if case % 10 == 0:                                             # (slow: 64 threads through ~cnt barriers)
            wave.host_cleanup_wave(arr, cnt, vw)
            assert bits(vw) == want, pts
        verdict = wave.host_cleanup_parallel(arr, cnt, vp)
        assert verdict in (0, 1)
        if verdict == 1:
            assert bits(vp) == want, pts
            settled += 1
        else:                                                          # handed over: the pass left fewer than three vertices
            assert len(after) < 3, pts
            handed += 1
        changed += int(len(after) != cnt)
        first_goes += int(_removable(pts[-1], pts[0], pts[1]))
    assert settled > 1200 and handed >= 1 and changed > 600 and first_goes > 100


def _removable(s, p, e):
    dx, dy, ux, uy = e[0] - s[0], e[1] - s[1], p[0] - s[0], p[1] - s[1]
    dist = abs(ux * dy - uy * dx)
    inner = ux * (e[0] - p[0]) + uy * (e[1] - p[1])
    return 2 * dist * dist <= dx * dx + dy * dy and dx != 0 and dy != 0 and inner >= 0


def _removable_any(pts):
    n = len(pts)
    return any(_removable(pts[i - 1], pts[i], pts[(i + 1) % n]) for i in range(n))


def test_register_transpose_of_a_level_image(host):
    """transpose16 (the trace kernel builds the column words of its level image with it): bit y of column word x
    == bit x of row word y, on random and extreme images."""
    host.host_transpose16.argtypes = [u16p, u16p]
    rng = np.random.RandomState(11)
    imgs = [rng.rand(16, 16) < p for p in (0.05, 0.3, 0.5, 0.8) for _ in range(20)]
    imgs += [np.zeros((16, 16), bool), np.ones((16, 16), bool), np.eye(16, dtype=bool), np.triu(np.ones((16, 16), bool))]
    for img in imgs:                                   # img[y, x]
        rows = np.array([sum(int(img[y, x]) << x for x in range(16)) for y in range(16)], dtype=np.uint16)
        cols = np.zeros(16, dtype=np.uint16)
        host.host_transpose16(rows.ctypes.data_as(u16p), cols.ctypes.data_as(u16p))
        want = np.array([sum(int(img[y, x]) << y for y in range(16)) for x in range(16)], dtype=np.uint16)
        assert np.array_equal(cols, want)


def _wide_images(seed, n, W, H):
    rng = np.random.RandomState(seed)
    for k in range(n):
        kind = k % 5
        if kind == 0:
            img = rng.rand(H, W) < rng.uniform(0.15, 0.85)
        elif kind == 1:
            img = ndimage.binary_dilation(rng.rand(H, W) < 0.08, structure=np.ones((3, 3))) & ~(rng.rand(H, W) < 0.1)
        elif kind == 2:                                   # unions of rectangles: level sets of box stacks
            img = np.zeros((H, W), dtype=bool)
            for _ in range(rng.randint(1, 9)):
                y, x = rng.randint(0, H), rng.randint(0, W)
                img[y:y + rng.randint(1, 17), x:x + rng.randint(1, 17)] ^= True
        elif kind == 3:                                   # rings, full frame included
            img = np.zeros((H, W), dtype=bool)
            for m in range(rng.randint(0, 2), min(H, W) // 2, rng.randint(2, 4)):
                img[m:H - m, m:W - m] = True
                img[m + 1:H - 1 - m, m + 1:W - 1 - m] = False
        else:                                             # diagonals, staircases, combs
            img = np.zeros((H, W), dtype=bool)
            for _ in range(rng.randint(1, 8)):
                y, x, d = rng.randint(0, H), rng.randint(0, W), rng.choice([-1, 1])
                for s in range(rng.randint(2, 24)):
                    yy, xx = y + s, x + d * s
                    if 0 <= yy < H and 0 <= xx < W:
                        img[yy, max(xx, 0):xx + rng.randint(1, 4)] = True
            if k % 10 == 9:
                img[::2, :] = True                        # a comb: the longest borders a grid can hold
                img[:, 0] = True
        yield img.astype(np.uint8)


@pytest.mark.parametrize("W,H,seed", [(32, 32, 0), (32, 32, 1), (32, 32, 2), (24, 32, 3), (32, 17, 4), (20, 27, 5)])
def test_wide_grid_routines_equal_the_oracle(host, W, H, seed):
    """Action grids of 17 .. 32 cells a side (resolutionA = 0.01): candidate starts, the plain border walk with 16-bit points
    and approxPolyDP + convexity (contours_device.h: start_candidates_wide, trace_border_wide, approx_and_convex_t<uint16_t, 5>)
    on whole level images against the oracle's findContours / approxPolyDP / find_convex_vetex."""
    cand = (C.c_uint32 * H)()
    vrows = (C.c_uint32 * H)()
    longest = C.c_int(0)
    seen_long = 0
    for img in _wide_images(seed, 120, W, H):
        rows = (C.c_uint32 * H)(*[int(sum(int(img[y, x]) << x for x in range(W))) for y in range(H)])
        outer, _, _ = _oracle_outer(img)
        n = host.host_wide_image_vertices(rows, W, H, 4096, 4096, cand, vrows, C.byref(longest))
        assert n == len(outer)                                             # one border per component
        assert host.host_wide_runs_equal_plain(rows, W, H) >= n            # the walk with run jumps: same points from every start
        starts = {(x, y) for y in range(H) for x in range(W) if (cand[y] >> x) & 1}
        assert {c[0] for c in outer} <= starts                              # every first pixel is listed
        want = set()
        for c in outer:
            want |= _oracle_vertices(c)
        got = {(x, y) for y in range(H) for x in range(W) if (vrows[y] >> x) & 1}
        assert got == want
        assert longest.value == max([len(c) for c in outer], default=0)
        seen_long = max(seen_long, longest.value)
    assert seen_long > 64                                                   # borders beyond the 16 x 16 grid's slot were covered


def _rect_truth(img, x0, y0):
    """(w, h) if the 8-connected component of pixel (x0, y0) is a solid rectangle whose first pixel that is, else None."""
    lab, _ = ndimage.label(img, structure=np.ones((3, 3)))
    comp = lab == lab[y0, x0]
    ys, xs = np.nonzero(comp)
    if (xs.min(), ys.min()) != (x0, y0) or not comp[ys.min():ys.max() + 1, xs.min():xs.max() + 1].all():
        return None
    return int(xs.max() - xs.min() + 1), int(ys.max() - ys.min() + 1)


def test_isolated_rectangles_every_size_and_position(host):
    """rect_component / rect_vertices (contours_device.h): every w x h rectangle at every position is recognised, and its
    vertex bits are what trace + approxPolyDP + convexity (the device routines, and the oracle) give for it."""
    vr, vt = (C.c_uint32 * 16)(), (C.c_uint32 * 16)()
    for w in range(1, 17):
        for h in range(1, 17):
            if w * h == 1:
                continue                                              # an isolated pixel: never a listed candidate
            for x0 in sorted({0, 1, (16 - w) // 2, 16 - w}):
                for y0 in sorted({0, 1, (16 - h) // 2, 16 - h}):
                    if x0 + w > 16 or y0 + h > 16:
                        continue
                    img = np.zeros((16, 16), dtype=np.uint8)
                    img[y0:y0 + h, x0:x0 + w] = 1
                    rows = _rows(img)
                    assert host.host_rect_component(rows, x0, y0, vr) == (w | (h << 8))
                    vt[:] = [0] * 16
                    assert host.host_contour_vertices(rows, x0, y0, 1360, 1360, vt) == 0
                    assert list(vr) == list(vt), (w, h, x0, y0)
                    outer, _, _ = _oracle_outer(img)
                    got = {(x, y) for y in range(16) for x in range(16) if (vr[y] >> x) & 1}
                    assert len(outer) == 1 and got == _oracle_vertices(outer[0]), (w, h, x0, y0)


@pytest.mark.parametrize("seed", range(4))
def test_rect_component_says_yes_exactly_for_isolated_solid_rectangles(host, seed):
    """On unions of rectangles, speckle and near-rectangles (a corner pixel missing, a pixel touching diagonally): for every
    listed candidate start, rect_component == the definition (8-connected component is its solid bounding box), and where
    it says yes the vertex bits equal those of the followed border."""
    rng = np.random.RandomState(500 + seed)
    cand, vr, vt = (C.c_uint32 * 16)(), (C.c_uint32 * 16)(), (C.c_uint32 * 16)()
    yes = no = 0
    for k in range(400):
        img = np.zeros((16, 16), dtype=bool)
        for _ in range(rng.randint(1, 9)):
            y, x = rng.randint(0, 16), rng.randint(0, 16)
            img[y:y + rng.randint(1, 7), x:x + rng.randint(1, 7)] = True
        if k % 3 == 1:
            img ^= rng.rand(16, 16) < 0.02                              # a pixel knocked out of / stuck onto a rectangle
        if k % 3 == 2:
            img = rng.rand(16, 16) < rng.uniform(0.05, 0.4)
        img = img.astype(np.uint8)
        rows = _rows(img)
        host.host_start_candidates(rows, cand)
        for y in range(16):
            for x in range(16):
                if not (cand[y] >> x) & 1:
                    continue
                truth = _rect_truth(img, x, y)
                got = host.host_rect_component(rows, x, y, vr)
                if truth is None or truth == (1, 1):
                    assert got == 0 or truth == (1, 1), (k, x, y)
                    no += 1
                    continue
                assert got == (truth[0] | (truth[1] << 8)), (k, x, y, truth, got)
                vt[:] = [0] * 16
                assert host.host_contour_vertices(rows, x, y, 1360, 1360, vt) == 0
                assert list(vr) == list(vt), (k, x, y, truth)
                yes += 1
    assert yes > 300 and no > 300
