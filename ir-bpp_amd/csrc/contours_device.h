// contours_device.h -- candidate-vertex extraction from one binary level image, per lane.
//
// One lane owns one (rotation, height-level) image of the <=16x16 action grid and runs,
// serially, what the reference does per level in convexHulls (cvTools.py:83-96):
//   cv2.findContours(RETR_TREE, CHAIN_APPROX_SIMPLE)  -> Suzuki-Abe border following
//   find_out_contour (cvTools.py:7-38)                -> keep outer borders, drop hole borders
//   cv2.approxPolyDP(c, 1, True)                      -> Douglas-Peucker, eps = 1
//   find_convex_vetex (cvTools.py:40-59)              -> vertices with cross(B-A, C-A) < 0
// and ORs the surviving vertices into a per-rotation 16x16 bit grid (np.unique at
// cvTools.py:101 makes the result a set, so a bit grid is its exact representation).
//
// Representation: the image is 16 rows of 16 bits (bit x of row y = pixel (x,y)), read-only.
// OpenCV's label image only ever distinguishes 0 / 1 (untouched) / visited / visited with the
// "right bound" sign bit (nbd|0x80) in its raster scan (contours.cpp, cvFindNextContour), so the
// labels are two more bit planes per slot (visited, right-bound), packed into one word per row.  A border-following step looks at
// the 3x3 neighbourhood as one 8-bit mask and finds the next direction with a rotate + bit
// scan instead of up to eight probes.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace irbpp {

struct SlotMem {
    uint32_t* lab;      // [16] label planes per row: bits 0-15 visited (label != 1), bits 16-31 right-bound (label < 0)
    uint8_t*  pts;      // [cap] contour points, x | y<<4
    uint8_t*  dst;      // [cap] approximated polygon
    uint32_t* stk;      // [cap_stk] Douglas-Peucker slices, start | end<<16
    int cap;            // point capacity
    int cap_stk;
};

// direction codes 0..7 = E,NE,N,NW,W,SW,S,SE with y down (OpenCV icvCodeDeltas)
__device__ __forceinline__ int dir_dx(int s) { return (int)((0x901Au >> (2 * s)) & 3u) - 1; }
__device__ __forceinline__ int dir_dy(int s) { return (int)((0xA901u >> (2 * s)) & 3u) - 1; }

// icvFetchContourEx with CHAIN_APPROX_SIMPLE.  img rows hold the 16-bit foreground rows.
// The 3-row window around the current pixel stays in registers and slides with the walk (one
// LDS row read per vertical move, off the critical path); label bits are ORed into LDS with
// non-returning atomics.  Returns the number of points produced (stored only while they fit
// in cap), or -1 if the iteration guard tripped.
__device__ __forceinline__ uint32_t nb_mask(uint32_t a, uint32_t b, uint32_t c, int x) {
    const uint32_t ta = ((a << 1) >> x) & 7u;      // bit0 = x-1, bit1 = x, bit2 = x+1
    const uint32_t tb = ((b << 1) >> x) & 7u;
    const uint32_t tc = ((c << 1) >> x) & 7u;
    return (tb >> 2) | ((ta >> 2) << 1) | (((ta >> 1) & 1u) << 2) | ((ta & 1u) << 3) |
           ((tb & 1u) << 4) | ((tc & 1u) << 5) | (((tc >> 1) & 1u) << 6) | ((tc >> 2) << 7);
}

__device__ inline int trace_border(const uint32_t* img, uint32_t* lab, int x0, int y0,
                                   bool is_hole, bool store, uint8_t* pts, int cap) {
    const int s_first = is_hole ? 0 : 4;
    uint32_t ra = y0 > 0 ? img[y0 - 1] : 0u, rb = img[y0], rc = y0 < 15 ? img[y0 + 1] : 0u;
    uint32_t nb = nb_mask(ra, rb, rc, x0);
    // clockwise search s_first-1, s_first-2, ... for the first foreground neighbour
    const int k = (s_first - 1) & 7;                              // first direction probed
    const uint32_t rot = ((nb << (7 - k)) | (nb >> (k + 1))) & 0xFFu;   // direction k -> bit 7
    if (rot == 0u) {                                              // isolated pixel
        atomicOr(&lab[y0], 0x10001u << x0);
        if (store && cap > 0) pts[0] = (uint8_t)(x0 | (y0 << 4));
        return 1;
    }
    const int p = 31 - __clz((int)rot);                           // highest set bit, 7 = direction k
    const int s = (k - (7 - p)) & 7;
    int x3 = x0, y3 = y0;
    const int x1 = x0 + dir_dx(s), y1 = y0 + dir_dy(s);
    int prev_s = s ^ 4;
    int cur_s = s;
    int n = 0;
    for (int guard = 0; guard < 4096; ++guard) {
        const int s_end = cur_s;
        // counter-clockwise search s_end+1, s_end+2, ... for the next border pixel
        const int k2 = (s_end + 1) & 7;
        const uint32_t r2 = ((nb >> k2) | (nb << (8 - k2))) & 0xFFu;   // direction k2 -> bit 0
        const int s2 = (k2 + __ffs((int)r2) - 1) & 7;
        const int dx = dir_dx(s2), dy = dir_dy(s2);
        const int x4 = x3 + dx, y4 = y3 + dy;
        // east neighbour probed empty -> right bound (sign bit); else 1 -> nbd, other labels unchanged
        atomicOr(&lab[y3], ((unsigned)(s2 - 1) < (unsigned)s_end ? 0x10001u : 0x1u) << x3);
        if (s2 != prev_s) {                                       // CHAIN_APPROX_SIMPLE
            if (store && n < cap) pts[n] = (uint8_t)(x3 | (y3 << 4));
            ++n;
        }
        prev_s = s2;
        if (x4 == x0 && y4 == y0 && x3 == x1 && y3 == y1) return n;
        if (dy > 0) { ra = rb; rb = rc; rc = y4 < 15 ? img[y4 + 1] : 0u; }
        else if (dy < 0) { rc = rb; rb = ra; ra = y4 > 0 ? img[y4 - 1] : 0u; }
        x3 = x4;
        y3 = y4;
        cur_s = (s2 + 4) & 7;
        nb = nb_mask(ra, rb, rc, x3);
    }
    return -1;
}

#define IRBPP_PX(p) ((int)((p) & 15))
#define IRBPP_PY(p) ((int)((p) >> 4))

// approxPolyDP_<int>(closed, eps=1) (OpenCV approx.cpp) followed by find_convex_vetex.
// pts[0..count) -> vertex bits ORed into vrows[y] (bit x).  Returns false on stack overflow.
__device__ inline bool approx_and_convex(const uint8_t* pts, int count, uint8_t* dst, uint32_t* stk,
                                         int cap_stk, uint32_t* vrows) {
    int new_count = 0;
    int top = 0;
    // 1. three farthest-point hops
    int pos = 0, right_start = 0;
    bool le_eps = false;
    uint8_t start_pt = 0;
    for (int it = 0; it < 3; ++it) {
        int max_dist = 0;
        pos += right_start;
        if (pos >= count) pos -= count;
        start_pt = pts[pos];
        if (++pos >= count) pos = 0;
        const int sx = IRBPP_PX(start_pt), sy = IRBPP_PY(start_pt);
        for (int j = 1; j < count; ++j) {
            const uint8_t pt = pts[pos];
            if (++pos >= count) pos = 0;
            const int dx = IRBPP_PX(pt) - sx, dy = IRBPP_PY(pt) - sy;
            const int dist = dx * dx + dy * dy;
            if (dist > max_dist) { max_dist = dist; right_start = j; }
        }
        le_eps = max_dist <= 1;
    }
    if (!le_eps) {
        const int s0 = pos;                          // pos < count always
        int far = right_start + s0;
        if (far >= count) far -= count;
        if (cap_stk < 2) return false;
        stk[top++] = (uint32_t)far | ((uint32_t)s0 << 16);       // right slice
        stk[top++] = (uint32_t)s0 | ((uint32_t)far << 16);       // slice, processed first
    } else {
        dst[new_count++] = start_pt;
    }
    // 3. Douglas-Peucker
    while (top > 0) {
        const uint32_t sl = stk[--top];
        const int s_start = (int)(sl & 0xFFFFu), s_end = (int)(sl >> 16);
        const uint8_t end_pt = pts[s_end];
        pos = s_start;
        start_pt = pts[pos];
        if (++pos >= count) pos = 0;
        bool le;
        int split = 0;
        if (pos != s_end) {
            const int sx = IRBPP_PX(start_pt), sy = IRBPP_PY(start_pt);
            const int dx = IRBPP_PX(end_pt) - sx, dy = IRBPP_PY(end_pt) - sy;
            int max_dist = 0;
            while (pos != s_end) {
                const uint8_t pt = pts[pos];
                int dist = (IRBPP_PY(pt) - sy) * dx - (IRBPP_PX(pt) - sx) * dy;
                dist = dist < 0 ? -dist : dist;
                if (dist > max_dist) { max_dist = dist; split = pos; }
                if (++pos >= count) pos = 0;
            }
            le = max_dist * max_dist <= dx * dx + dy * dy;
        } else {
            le = true;
        }
        if (le) {
            dst[new_count++] = start_pt;
        } else {
            if (top + 2 > cap_stk) return false;
            stk[top++] = (uint32_t)split | ((uint32_t)s_end << 16);
            stk[top++] = (uint32_t)s_start | ((uint32_t)split << 16);
        }
    }
    // 4. clean-up of [almost] collinear points, in place as OpenCV does
    {
        const int cnt = new_count;
        pos = cnt - 1;
        start_pt = dst[pos];
        if (++pos >= cnt) pos = 0;
        int wpos = pos;
        uint8_t pt = dst[pos];
        if (++pos >= cnt) pos = 0;
        for (int i = 0; i < cnt && new_count > 2; ++i) {
            const uint8_t end_pt = dst[pos];
            if (++pos >= cnt) pos = 0;
            const int dx = IRBPP_PX(end_pt) - IRBPP_PX(start_pt), dy = IRBPP_PY(end_pt) - IRBPP_PY(start_pt);
            const int ux = IRBPP_PX(pt) - IRBPP_PX(start_pt), uy = IRBPP_PY(pt) - IRBPP_PY(start_pt);
            int dist = ux * dy - uy * dx;
            dist = dist < 0 ? -dist : dist;
            const int inner = ux * (IRBPP_PX(end_pt) - IRBPP_PX(pt)) + uy * (IRBPP_PY(end_pt) - IRBPP_PY(pt));
            if (2 * dist * dist <= dx * dx + dy * dy && dx != 0 && dy != 0 && inner >= 0) {
                --new_count;
                dst[wpos] = start_pt = end_pt;
                if (++wpos >= cnt) wpos = 0;
                pt = dst[pos];
                if (++pos >= cnt) pos = 0;
                ++i;
                continue;
            }
            dst[wpos] = start_pt = pt;
            if (++wpos >= cnt) wpos = 0;
            pt = end_pt;
        }
    }
    // find_convex_vetex
    const int m = new_count;
    if (m <= 3) {
        for (int i = 0; i < m; ++i) atomicOr(&vrows[IRBPP_PY(dst[i])], 1u << IRBPP_PX(dst[i]));
    } else {
        uint8_t a = dst[m - 1], b = dst[0];
        for (int i = 0; i < m; ++i) {
            const uint8_t c = dst[i == m - 1 ? 0 : i + 1];
            const int cross = (IRBPP_PX(b) - IRBPP_PX(a)) * (IRBPP_PY(c) - IRBPP_PY(a)) -
                              (IRBPP_PY(b) - IRBPP_PY(a)) * (IRBPP_PX(c) - IRBPP_PX(a));
            if (cross < 0) atomicOr(&vrows[IRBPP_PY(b)], 1u << IRBPP_PX(b));
            a = b;
            b = c;
        }
    }
    return true;
}

// Whole level image: raster scan (cvFindNextContour) + per-outer-border approximation.
// img[y] bit x = foreground.  Returns 0 ok, 1 capacity overflow (caller retries with a bigger
// slot), 2 iteration guard.
__device__ inline int level_image_vertices(const uint32_t* img, const SlotMem& m, uint32_t* vrows) {
    for (int y = 0; y < 16; ++y) m.lab[y] = 0u;
    for (int y = 0; y < 16; ++y) {
        const uint32_t nz = img[y] & 0xFFFFu;
        if (!nz) continue;
        int cur = 0;
        while (cur < 16) {
            const uint32_t l = m.lab[y];
            const uint32_t vis = l & 0xFFFFu, neg = l >> 16;
            const uint32_t outer = nz & ~vis & ~(nz << 1);                    // prev == 0 && p == 1
            const uint32_t hole = ~nz & (nz << 1) & ~(neg << 1) & 0xFFFFu;    // p == 0 && prev >= 1
            const uint32_t cand = (outer | hole) & ~((1u << cur) - 1u);
            if (!cand) break;
            const int x = __ffs((int)cand) - 1;
            const bool is_hole = ((hole >> x) & 1u) != 0;
            const int ox = is_hole ? x - 1 : x;
            const int n = trace_border(img, m.lab, ox, y, is_hole, !is_hole, m.pts, m.cap);
            if (n < 0) return 2;
            if (!is_hole) {
                if (n > m.cap) return 1;
                if (!approx_and_convex(m.pts, n, m.dst, m.stk, m.cap_stk, vrows)) return 1;
            }
            cur = x + 1;
        }
    }
    return 0;
}

}  // namespace irbpp
