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
// The label image keeps 2 bits per pixel: 0 background, 1 untouched foreground, 2 visited,
// 3 visited + "right bound" (OpenCV's nbd|0x80).  Only those classes steer the scan
// (contours.cpp, cvFindNextContour), so the per-contour label values are not needed.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace irbpp {

struct SlotMem {
    uint32_t* lab;      // [16] 2-bit label rows
    uint8_t*  pts;      // [cap] contour points, x | y<<4
    uint8_t*  dst;      // [cap] approximated polygon
    uint32_t* stk;      // [cap_stk] Douglas-Peucker slices, start | end<<16
    int cap;            // point capacity
    int cap_stk;
};

// direction codes 0..7 = E,NE,N,NW,W,SW,S,SE with y down (OpenCV icvCodeDeltas)
__device__ __forceinline__ int dir_dx(int s) { return (int)((0x901Au >> (2 * s)) & 3u) - 1; }
__device__ __forceinline__ int dir_dy(int s) { return (int)((0xA901u >> (2 * s)) & 3u) - 1; }

__device__ __forceinline__ int lab_get(const uint32_t* lab, int x, int y) {
    return ((unsigned)x < 16u && (unsigned)y < 16u) ? (int)((lab[y] >> (2 * x)) & 3u) : 0;
}
__device__ __forceinline__ void lab_set(uint32_t* lab, int x, int y, uint32_t v) {
    lab[y] = (lab[y] & ~(3u << (2 * x))) | (v << (2 * x));
}

// spread the low 16 bits of v to the even bit positions (pixel x -> bit 2x)
__device__ __forceinline__ uint32_t spread16(uint32_t v) {
    v &= 0xFFFFu;
    v = (v | (v << 8)) & 0x00FF00FFu;
    v = (v | (v << 4)) & 0x0F0F0F0Fu;
    v = (v | (v << 2)) & 0x33333333u;
    v = (v | (v << 1)) & 0x55555555u;
    return v;
}

// icvFetchContourEx with CHAIN_APPROX_SIMPLE.  Returns the number of points produced
// (stored only while they fit in cap), or -1 if the iteration guard tripped.
__device__ inline int trace_border(uint32_t* lab, int x0, int y0, bool is_hole, bool store,
                                   uint8_t* pts, int cap) {
    int s_end = is_hole ? 0 : 4;
    int s = s_end;
    int x1, y1;
    do {
        s = (s - 1) & 7;
        x1 = x0 + dir_dx(s);
        y1 = y0 + dir_dy(s);
    } while (lab_get(lab, x1, y1) == 0 && s != s_end);
    if (s == s_end) {                       // isolated pixel
        lab_set(lab, x0, y0, 3u);
        if (store && cap > 0) pts[0] = (uint8_t)(x0 | (y0 << 4));
        return 1;
    }
    int x3 = x0, y3 = y0, x4 = x0, y4 = y0;
    int prev_s = s ^ 4;
    int px = x0, py = y0;
    int n = 0;
    for (int guard = 0; guard < 4096; ++guard) {
        s_end = s;
        while (s < 15) {
            ++s;
            x4 = x3 + dir_dx(s & 7);
            y4 = y3 + dir_dy(s & 7);
            if (lab_get(lab, x4, y4) != 0) break;
        }
        s &= 7;
        if ((unsigned)(s - 1) < (unsigned)s_end) lab_set(lab, x3, y3, 3u);
        else if (lab_get(lab, x3, y3) == 1) lab_set(lab, x3, y3, 2u);
        if (s != prev_s) {
            if (store && n < cap) pts[n] = (uint8_t)(px | (py << 4));
            ++n;
        }
        prev_s = s;
        px += dir_dx(s);
        py += dir_dy(s);
        if (x4 == x0 && y4 == y0 && x3 == x1 && y3 == y1) return n;
        x3 = x4;
        y3 = y4;
        s = (s + 4) & 7;
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
        pos = (pos + right_start) % count;
        start_pt = pts[pos];
        if (++pos >= count) pos = 0;
        for (int j = 1; j < count; ++j) {
            const uint8_t pt = pts[pos];
            if (++pos >= count) pos = 0;
            const int dx = IRBPP_PX(pt) - IRBPP_PX(start_pt), dy = IRBPP_PY(pt) - IRBPP_PY(start_pt);
            const int dist = dx * dx + dy * dy;
            if (dist > max_dist) { max_dist = dist; right_start = j; }
        }
        le_eps = max_dist <= 1;
    }
    if (!le_eps) {
        const int s0 = pos % count;
        const int far = (right_start + s0) % count;
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
            const int dx = IRBPP_PX(end_pt) - IRBPP_PX(start_pt), dy = IRBPP_PY(end_pt) - IRBPP_PY(start_pt);
            int max_dist = 0;
            while (pos != s_end) {
                const uint8_t pt = pts[pos];
                if (++pos >= count) pos = 0;
                int dist = (IRBPP_PY(pt) - IRBPP_PY(start_pt)) * dx - (IRBPP_PX(pt) - IRBPP_PX(start_pt)) * dy;
                dist = dist < 0 ? -dist : dist;
                if (dist > max_dist) { max_dist = dist; split = (pos + count - 1) % count; }
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
    for (int i = 0; i < m; ++i) {
        const uint8_t b = dst[i];
        bool keep = true;
        if (m > 3) {
            const uint8_t a = dst[i == 0 ? m - 1 : i - 1];
            const uint8_t c = dst[i == m - 1 ? 0 : i + 1];
            const int cross = (IRBPP_PX(b) - IRBPP_PX(a)) * (IRBPP_PY(c) - IRBPP_PY(a)) -
                              (IRBPP_PY(b) - IRBPP_PY(a)) * (IRBPP_PX(c) - IRBPP_PX(a));
            keep = cross < 0;
        }
        if (keep) atomicOr(&vrows[IRBPP_PY(b)], 1u << IRBPP_PX(b));
    }
    return true;
}

// Whole level image: raster scan (cvFindNextContour) + per-outer-border approximation.
// img_rows[y] bit x = foreground.  Returns 0 ok, 1 capacity overflow (caller retries with a
// bigger slot), 2 iteration guard.
__device__ inline int level_image_vertices(const uint32_t* img_rows, const SlotMem& m, uint32_t* vrows) {
    for (int y = 0; y < 16; ++y) m.lab[y] = spread16(img_rows[y]);
    for (int y = 0; y < 16; ++y) {
        int cur = 0;
        while (cur < 16) {
            const uint32_t w = m.lab[y];
            const uint32_t lo = w & 0x55555555u, hi = (w >> 1) & 0x55555555u;
            const uint32_t nz = lo | hi;
            const uint32_t untouched = lo & ~hi;                 // value 1
            const uint32_t pos_lab = (hi & ~lo) | untouched;     // value >= 1 and not right-bound
            const uint32_t outer = untouched & ~(nz << 2);       // prev == 0 && p == 1
            const uint32_t hole = (~nz & 0x55555555u) & (pos_lab << 2);   // p == 0 && prev >= 1
            uint32_t cand = (outer | hole) & ~((1u << (2 * cur)) - 1u);
            if (!cand) break;
            const int x = (__ffs((int)cand) - 1) >> 1;
            const bool is_hole = ((hole >> (2 * x)) & 1u) != 0;
            const int ox = is_hole ? x - 1 : x;
            const int n = trace_border(m.lab, ox, y, is_hole, !is_hole, m.pts, m.cap);
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
