"""The two OpenCV calls on the path (cv2.findContours RETR_TREE/CHAIN_APPROX_SIMPLE and
cv2.approxPolyDP) are restated in oracle/contours.py with parity unpinned: cv2 is not in the image
and the reference holds no vectors for them.  These tests anchor the restatement on facts that do
not depend on it: connected components and border pixels computed with scipy.ndimage, the
hierarchy's parity structure, and the Douglas-Peucker error bound."""
import numpy as np
import pytest
from scipy import ndimage

from oracle import contours as C
from oracle import cvtools

EIGHT = np.ones((3, 3), dtype=int)
FOUR = ndimage.generate_binary_structure(2, 1)


def _images(seed, n):
    rng = np.random.RandomState(seed)
    out = []
    for k in range(n):
        h, w = rng.randint(3, 17), rng.randint(3, 17)
        kind = k % 4
        if kind == 0:                                   # speckle
            img = rng.rand(h, w) < rng.uniform(0.2, 0.8)
        elif kind == 1:                                 # blobs with holes
            img = ndimage.binary_dilation(rng.rand(h, w) < 0.15, structure=EIGHT)
            img &= ~(rng.rand(h, w) < 0.1)
        elif kind == 2:                                 # rectangles (what level sets of BlockOut look like)
            img = np.zeros((h, w), dtype=bool)
            for _ in range(rng.randint(1, 5)):
                y, x = rng.randint(0, h), rng.randint(0, w)
                img[y:y + rng.randint(1, 6), x:x + rng.randint(1, 6)] ^= True
        else:                                           # nested rings
            img = np.zeros((h, w), dtype=bool)
            for m in range(0, min(h, w) // 2, 2):
                img[m:h - m, m:w - m] = True
                img[m + 1:h - m - 1, m + 1:w - m - 1] = False
        out.append(img.astype(np.uint8))
    return out


def _segment_pixels(p, q):
    """Pixels of the straight (axis or diagonal) run from p to q inclusive; None if not such a run."""
    dx, dy = q[0] - p[0], q[1] - p[1]
    if not (dx == 0 or dy == 0 or abs(dx) == abs(dy)):
        return None
    n = max(abs(dx), abs(dy))
    sx, sy = int(np.sign(dx)), int(np.sign(dy))
    return [(p[0] + sx * k, p[1] + sy * k) for k in range(n + 1)]


def _outer_border_pixels(comp):
    """Pixels of the component that touch (4-neighbourhood) the background region surrounding it:
    exactly the pixels an outer border of an 8-connected component consists of (Suzuki & Abe 1985)."""
    pad = np.pad(comp, 1)
    bg, _ = ndimage.label(~pad, structure=FOUR)
    outside = bg == bg[0, 0]
    touch = ndimage.binary_dilation(outside, structure=FOUR) & pad
    return set(zip(*np.nonzero(touch[1:-1, 1:-1])[::-1]))          # (x, y)


@pytest.mark.parametrize("seed", range(6))
def test_outer_contours_are_the_components_and_their_border_pixels(seed):
    for img in _images(seed, 60):
        contours, hierarchy, is_hole = C.find_contours(img)
        lab, ncomp = ndimage.label(img, structure=EIGHT)
        if ncomp == 0:
            assert len(contours) == 0
            continue
        kept, kept_idx = cvtools.find_out_contour(contours, hierarchy[0])
        # find_out_contour (cvTools.py:7-38) keeps the even nesting depths = the outer borders, one per component
        assert sorted(kept_idx) == [i for i, hole in enumerate(is_hole) if not hole]
        assert len(kept) == ncomp
        seen = set()
        for c in kept:
            pts = [tuple(int(v) for v in p) for p in c.reshape(-1, 2)]
            comp_id = lab[pts[0][1], pts[0][0]]
            assert comp_id > 0 and comp_id not in seen
            seen.add(comp_id)
            comp = lab == comp_id
            ys, xs = np.nonzero(comp)
            first = (xs[ys == ys.min()].min(), ys.min())
            assert pts[0] == first                                   # raster-first pixel starts the border
            walked = set()
            for a, b in zip(pts, pts[1:] + pts[:1]):
                seg = _segment_pixels(a, b)
                assert seg is not None, "CHAIN_APPROX_SIMPLE keeps direction changes: consecutive points are collinear runs"
                walked.update(seg)
            assert walked == _outer_border_pixels(comp)
            if len(pts) > 2:                                         # no point in the middle of a straight run
                for a, b, c3 in zip(pts, pts[1:] + pts[:1], pts[2:] + pts[:2]):
                    d1 = (np.sign(b[0] - a[0]), np.sign(b[1] - a[1]))
                    d2 = (np.sign(c3[0] - b[0]), np.sign(c3[1] - b[1]))
                    assert d1 != d2 or len(pts) <= 2


@pytest.mark.parametrize("seed", range(4))
def test_hierarchy_alternates_outer_and_hole_borders(seed):
    for img in _images(100 + seed, 50):
        contours, hierarchy, is_hole = C.find_contours(img)
        h = hierarchy[0] if len(contours) else np.zeros((0, 4), dtype=int)
        holes_expected = 0
        lab, ncomp = ndimage.label(img, structure=EIGHT)
        for k in range(1, ncomp + 1):                                # holes of a component = background regions it encloses
            comp = np.pad(lab == k, 1)
            bg, nbg = ndimage.label(~comp, structure=FOUR)
            holes_expected += nbg - 1
        assert sum(is_hole) == holes_expected
        for i, (nxt, prv, child, parent) in enumerate(h):
            if parent == -1:
                assert not is_hole[i]
            else:
                assert is_hole[i] != is_hole[parent]                # a hole's parent is an outer border and vice versa
            if nxt != -1:
                assert h[nxt][1] == i and h[nxt][3] == parent


def _dist_point_segment(p, a, b):
    p, a, b = (np.asarray(v, dtype=np.float64) for v in (p, a, b))
    ab = b - a
    t = 0.0 if not ab.any() else np.clip(np.dot(p - a, ab) / np.dot(ab, ab), 0.0, 1.0)
    return float(np.linalg.norm(p - (a + t * ab)))


@pytest.mark.parametrize("seed", range(4))
def test_approx_poly_dp_is_an_ordered_subset_within_epsilon(seed):
    eps = 1.0
    for img in _images(200 + seed, 50):
        contours, hierarchy, _ = C.find_contours(img)
        for c in contours:
            src = [tuple(int(v) for v in p) for p in c.reshape(-1, 2)]
            out = [tuple(int(v) for v in p) for p in C.approx_poly_dp(c, eps, True).reshape(-1, 2)]
            assert 1 <= len(out) <= len(src)
            # subset, in the cyclic order of the source (thin parts are walked both ways, so points repeat:
            # try every occurrence of the first output point as the rotation origin; greedy matching is exact
            # for a fixed origin)
            def embeds(start):
                rot, j = src[start:] + src[:start], 0
                for p in out:
                    while j < len(rot) and rot[j] != p:
                        j += 1
                    if j == len(rot):
                        return False
                    j += 1
                return True
            assert any(embeds(k) for k, p in enumerate(src) if p == out[0]), "not an ordered subset of the contour"
            # Douglas-Peucker keeps every source point within eps of its slice's chord; the clean-up pass then
            # drops vertices closer than eps/sqrt(2) to the chord of their neighbours (approx.cpp), so the
            # closed result polygon stays within eps * (1 + 1/sqrt(2)) of every source point
            edges = list(zip(out, out[1:] + out[:1]))
            for p in src:
                assert min(_dist_point_segment(p, a, b) for a, b in edges) <= eps * (1 + 0.5 ** 0.5) + 1e-9
