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
    lib.host_approx_and_convex_wave.argtypes = [u8p, C.c_int, u32p]
    lib.host_approx_and_convex_wave.restype = C.c_int
    return lib


@pytest.fixture(scope="module")
def host():
    if shutil.which("g++") is None:
        pytest.skip("no host compiler")
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    subprocess.run(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-Wno-unused-value",
                    "-I", os.path.join(HERE, "host", "stub"), SRC, "-o", OUT], check=True)
    lib = C.CDLL(OUT)
    lib.host_start_candidates.argtypes = [u16p, u32p]
    lib.host_trace_border.argtypes = [u16p, C.c_int, C.c_int, u8p, C.c_int]
    lib.host_trace_border.restype = C.c_int
    lib.host_approx_and_convex.argtypes = [u8p, C.c_int, C.c_int, u32p]
    lib.host_approx_and_convex.restype = C.c_int
    lib.host_contour_vertices.argtypes = [u16p, C.c_int, C.c_int, C.c_int, C.c_int, u32p]
    lib.host_contour_vertices.restype = C.c_int
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
        for (x, y) in listed:
            n = host.host_trace_border(rows, x, y, pts, 512)
            if (x, y) in starts:
                got = [(pts[i] & 15, pts[i] >> 4) for i in range(n)]
                assert got == starts[(x, y)]                           # same points, same order
            else:
                assert n == 0                                          # a later pixel of some component
        if outer:                                                      # a tight slot reports the true length
            longest = max(outer, key=len)
            assert host.host_trace_border(rows, longest[0][0], longest[0][1], pts, 3) == len(longest)


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


def test_wave_cooperative_douglas_peucker_in_lockstep_emulation(wave):
    """approx_and_convex_wave: the routine the kernel uses for borders of more than 12 points."""
    vrows = (C.c_uint32 * 16)()
    done = 0
    for img in _images(300, 60):
        outer, _, _ = _oracle_outer(img)
        for c in outer:
            if not 1 <= len(c) <= 64 or (len(c) <= 12 and done % 4):      # mostly the long ones
                continue
            arr = (C.c_uint8 * len(c))(*[x | (y << 4) for x, y in c])
            assert wave.host_approx_and_convex_wave(arr, len(c), vrows) == 1
            got = {(x, y) for y in range(16) for x in range(16) if (vrows[y] >> x) & 1}
            assert got == _oracle_vertices(c), c
            done += 1
    rng = np.random.RandomState(11)
    for _ in range(40):                                               # arbitrary closed walks, up to the 64-point limit
        n = rng.randint(1, 65)
        x, y = rng.randint(0, 16), rng.randint(0, 16)
        poly = [(x, y)]
        while len(poly) < n:
            x = int(np.clip(x + rng.randint(-1, 2), 0, 15))
            y = int(np.clip(y + rng.randint(-1, 2), 0, 15))
            if (x, y) != poly[-1]:
                poly.append((x, y))
        arr = (C.c_uint8 * len(poly))(*[px | (py << 4) for px, py in poly])
        assert wave.host_approx_and_convex_wave(arr, len(poly), vrows) == 1
        got = {(xx, yy) for yy in range(16) for xx in range(16) if (vrows[yy] >> xx) & 1}
        assert got == _oracle_vertices(poly), poly
        done += 1
    assert done > 60                                                  # (each call is 64 threads and ~10^3 barriers)
