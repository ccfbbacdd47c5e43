// contours_device.h -- candidate vertices of the outer borders of a binary level image.
//
// What the reference does per height level in convexHulls (cvTools.py:83-96):
//   cv2.findContours(RETR_TREE, CHAIN_APPROX_SIMPLE)  -> Suzuki-Abe border following
//   find_out_contour (cvTools.py:7-38)                -> keep outer borders, drop hole borders
//   cv2.approxPolyDP(c, 1, True)                      -> Douglas-Peucker, eps = 1
//   find_convex_vetex (cvTools.py:40-59)              -> vertices with cross(B-A, C-A) < 0
// and np.unique (cvTools.py:101) turns the result into a set, kept here as a 16x16 bit grid.
//
// Decomposition used on the GPU (results identical to the sequential algorithm):
//   * Suzuki-Abe starts exactly one outer border per 8-connected foreground component, at the
//     component's first pixel in raster order (its west neighbour is background and it is still
//     unlabelled when the scan reaches it; every other such pixel has already been labelled by
//     the outer or a hole border).  Hole borders are discarded by find_out_contour and border
//     following itself only tests "non-zero", so neither hole tracing nor label state is
//     needed.  The start pixels are found without any flood fill: a pixel whose W, NW, N and NE
//     neighbours are background is a *candidate* (every component's raster-first pixel is one);
//     the border traced from a candidate is kept iff no pixel on it has a smaller raster index
//     than the candidate -- true for exactly the raster-first pixel of a component, false for
//     other convex corners of the same outer border and for candidates that sit on a hole
//     border (a hole's border always has pixels in rows above it).
//   * one lane then owns one outer border: trace_border() (icvFetchContourEx, SIMPLE
//     approximation) followed by approx_and_convex() (approxPolyDP_ + the convexity test).  All
//     lanes of a wave run the same code, so divergence is limited to loop trip counts.
// The image is 16 rows of 16 bits (bit x of row y = pixel (x,y)).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace irbpp {

struct SlotMem {
    uint8_t*  pts;      // [cap] contour points, x | y<<4
    uint8_t*  dst;      // [cap] approximated polygon
    uint32_t* stk;      // [cap_stk] Douglas-Peucker slices, start | end<<16
    int cap;            // point capacity
    int cap_stk;
};

// direction codes 0..7 = E,NE,N,NW,W,SW,S,SE with y down (OpenCV icvCodeDeltas)
__device__ __forceinline__ int dir_dx(int s) { return (int)((0x901Au >> (2 * s)) & 3u) - 1; }
__device__ __forceinline__ int dir_dy(int s) { return (int)((0xA901u >> (2 * s)) & 3u) - 1; }

// 8-bit neighbour mask from the three rows around a pixel: bit s set iff the neighbour in
// direction s is foreground
__device__ __forceinline__ uint32_t nb_mask(uint32_t a, uint32_t b, uint32_t c, int x) {
    const uint32_t ta = ((a << 1) >> x) & 7u;      // bit0 = x-1, bit1 = x, bit2 = x+1
    const uint32_t tb = ((b << 1) >> x) & 7u;
    const uint32_t tc = ((c << 1) >> x) & 7u;
    return (tb >> 2) | ((ta >> 2) << 1) | (((ta >> 1) & 1u) << 2) | ((ta & 1u) << 3) |
           ((tb & 1u) << 4) | ((tc & 1u) << 5) | (((tc >> 1) & 1u) << 6) | ((tc >> 2) << 7);
}

// Candidate start pixels of row `row` given the row above it (`up`, 0 for the first row).
__device__ __forceinline__ uint32_t start_candidates(uint32_t row, uint32_t up) {
    return row & ~(row << 1) & ~up & ~(up << 1) & ~(up >> 1) & 0xFFFFu;
}

// Straight runs.  After a step in an axis direction d the walk keeps going straight exactly
// while the three neighbours probed before d (directions d+5, d+6, d+7) are background and the
// neighbour in direction d is foreground; no point is emitted inside a run (CHAIN_APPROX_SIMPLE
// keeps direction changes only).  For a horizontal run those are bits of the row below/above,
// for a vertical run bits of the column left/right (imgT holds the transposed image), so the
// run length is one bit scan.  `line` = the row/column being walked, `side` = the row/column
// whose three bits must be clear, `p` = current position along the line.
__device__ __forceinline__ int run_forward(uint32_t line, uint32_t side, int p) {      // towards higher bits
    const uint32_t clear = ~(side | (side << 1) | (side >> 1));
    const uint32_t m = (clear & (line >> 1) & 0xFFFFu) >> p;
    return __ffs((int)~m) - 1;                                                          // consecutive ones from bit p
}
__device__ __forceinline__ int run_backward(uint32_t line, uint32_t side, int p) {     // towards lower bits
    const uint32_t clear = ~(side | (side << 1) | (side >> 1));
    const uint32_t m = clear & (line << 1) & 0xFFFFu;
    const uint32_t z = ~m & ((2u << p) - 1u);                                           // zero bits at or below p
    return z ? p - (31 - __clz((int)z)) : p + 1;
}

// icvFetchContourEx with CHAIN_APPROX_SIMPLE for the OUTER border starting at (x0,y0).  The
// 3-row window around the current pixel stays in registers; axis-aligned runs are jumped in one
// go.  img = 16 row words, imgT = 16 column words (bit y of word x).  Returns the number of
// points produced (stored only while they fit in cap); 0 if the walk met a pixel that precedes
// (x0,y0) in raster order, i.e. (x0,y0) is not the first pixel of its component and the border
// belongs to another start (or is a hole border); -1 if the iteration guard tripped.
__device__ inline int trace_border(const uint16_t* img, const uint16_t* imgT, int x0, int y0, uint8_t* pts, int cap) {
    uint32_t ra = y0 > 0 ? img[y0 - 1] : 0u, rb = img[y0], rc = y0 < 15 ? img[y0 + 1] : 0u;
    uint32_t nb = nb_mask(ra, rb, rc, x0);
    // clockwise search 3,2,1,0,7,6,5 (s_end = 4: the west pixel is background) for the first neighbour
    const uint32_t rot = ((nb << 4) | (nb >> 4)) & 0xFFu;         // direction 3 -> bit 7
    if (rot == 0u) {                                              // isolated pixel
        if (cap > 0) pts[0] = (uint8_t)(x0 | (y0 << 4));
        return 1;
    }
    const int s = (3 - (7 - (31 - __clz((int)rot)))) & 7;
    int x3 = x0, y3 = y0;
    const int x1 = x0 + dir_dx(s), y1 = y0 + dir_dy(s);
    int prev_s = s ^ 4;
    int cur_s = s;
    int n = 0;
    for (int guard = 0; guard < 4096; ++guard) {
        // counter-clockwise search cur_s+1, cur_s+2, ... for the next border pixel
        const int k2 = (cur_s + 1) & 7;
        const uint32_t r2 = ((nb >> k2) | (nb << (8 - k2))) & 0xFFu;   // direction k2 -> bit 0
        const int s2 = (k2 + __ffs((int)r2) - 1) & 7;
        int x4 = x3 + dir_dx(s2), y4 = y3 + dir_dy(s2);
        if (s2 != prev_s) {                                       // CHAIN_APPROX_SIMPLE
            if (n < cap) pts[n] = (uint8_t)(x3 | (y3 << 4));
            ++n;
        }
        prev_s = s2;
        if (x4 == x0 && y4 == y0 && x3 == x1 && y3 == y1) return n;
        // now standing on (x4,y4), arrived by s2: jump to the end of an axis-aligned run.
        // Branch-free on purpose: the lanes of a wave walk different borders in lockstep.
        {
            const bool horiz = (s2 & 3) == 0;                     // E or W: walk a row, else a column
            const bool fwd = s2 == 0 || s2 == 6;                  // E or S: towards higher bits
            const uint16_t* base = horiz ? img : imgT;
            const int li = horiz ? y4 : x4;
            const int si = (horiz == fwd) ? li + 1 : li - 1;      // E: row below, W: row above, S: column left, N: column right
            int p = horiz ? x4 : y4;
            const uint32_t line = base[li];
            const uint32_t side = base[si & 15];
            const uint32_t side_ok = (unsigned)si < 16u ? side : 0u;
            const int lf = run_forward(line, side_ok, p), lb = run_backward(line, side_ok, p);
            const int L = (s2 & 1) ? 0 : (fwd ? lf : -lb);
            p += L;
            x4 = horiz ? p : x4;
            y4 = horiz ? y4 : p;
        }
        // a run that ends on the start pixel, moving opposite to the first step, closes the border
        if (x4 == x0 && y4 == y0 && s2 == (s ^ 4)) return n;
        // run interiors lie between their end points in raster order, so testing end points suffices
        if (y4 * 16 + x4 < y0 * 16 + x0) return 0;
        const uint32_t a = img[(y4 - 1) & 15], b = img[y4], c = img[(y4 + 1) & 15];
        ra = y4 > 0 ? a : 0u;
        rb = b;
        rc = y4 < 15 ? c : 0u;
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
    // 3. Douglas-Peucker as ONE flat loop (pop a slice | visit one point) so that lanes working
    //    on different contours stay converged
    {
        bool in_slice = false;
        int s_start = 0, s_end = 0, sx = 0, sy = 0, dx = 0, dy = 0, max_dist = 0, split = 0;
        for (int guard = 0; guard < 65536; ++guard) {
            if (!in_slice) {
                if (top == 0) break;
                const uint32_t sl = stk[--top];
                s_start = (int)(sl & 0xFFFFu);
                s_end = (int)(sl >> 16);
                pos = s_start;
                start_pt = pts[pos];
                if (++pos >= count) pos = 0;
                if (pos == s_end) {                  // no interior point: accept
                    dst[new_count++] = start_pt;
                    continue;
                }
                const uint8_t end_pt = pts[s_end];
                sx = IRBPP_PX(start_pt);
                sy = IRBPP_PY(start_pt);
                dx = IRBPP_PX(end_pt) - sx;
                dy = IRBPP_PY(end_pt) - sy;
                max_dist = 0;
                in_slice = true;
            }
            const uint8_t pt = pts[pos];
            int dist = (IRBPP_PY(pt) - sy) * dx - (IRBPP_PX(pt) - sx) * dy;
            dist = dist < 0 ? -dist : dist;
            if (dist > max_dist) { max_dist = dist; split = pos; }
            if (++pos >= count) pos = 0;
            if (pos == s_end) {
                in_slice = false;
                if (max_dist * max_dist <= dx * dx + dy * dy) {
                    dst[new_count++] = start_pt;
                } else {
                    if (top + 2 > cap_stk) return false;
                    stk[top++] = (uint32_t)split | ((uint32_t)s_end << 16);
                    stk[top++] = (uint32_t)s_start | ((uint32_t)split << 16);
                }
            }
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

// One outer border, serially: trace, approximate, mark convex vertices.  Returns 0 ok,
// 1 capacity overflow (caller retries with a bigger slot), 2 iteration guard.
__device__ inline int contour_vertices(const uint16_t* img, const uint16_t* imgT, int x0, int y0, const SlotMem& m,
                                       uint32_t* vrows) {
    const int n = trace_border(img, imgT, x0, y0, m.pts, m.cap);
    if (n < 0) return 2;
    if (n == 0) return 0;
    if (n > m.cap) return 1;
    return approx_and_convex(m.pts, n, m.dst, m.stk, m.cap_stk, vrows) ? 0 : 1;
}

// ---------------------------------------------------------------------------------------
// Wave-cooperative approxPolyDP + convexity test for ONE (long) border: all 64 lanes of the
// calling wave take part; the point distances of a pass / slice are computed one point per lane
// and the arg-max ("first strict maximum in traversal order", as the sequential loops find it)
// comes from a wave reduction.  Same results as approx_and_convex().
// ---------------------------------------------------------------------------------------
// wave64 max-reduction with DPP (no LDS crossbar): quad swaps, row mirrors, then row broadcasts;
// the total lands in lane 63 and is read back as a wave-uniform scalar.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint32_t dpp_u32(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, CTRL, ROW_MASK, 0xF, false);
}
__device__ __forceinline__ uint32_t umax(uint32_t a, uint32_t b) { return a > b ? a : b; }
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) {
    v = umax(v, dpp_u32<0xB1, 0xF>(v));      // quad_perm [1,0,3,2]
    v = umax(v, dpp_u32<0x4E, 0xF>(v));      // quad_perm [2,3,0,1]
    v = umax(v, dpp_u32<0x141, 0xF>(v));     // row_half_mirror
    v = umax(v, dpp_u32<0x140, 0xF>(v));     // row_mirror: every lane now holds its row's max
    v = umax(v, dpp_u32<0x142, 0xA>(v));     // row_bcast15 into rows 1 and 3
    v = umax(v, dpp_u32<0x143, 0xC>(v));     // row_bcast31 into rows 2 and 3
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}

__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
// v_writelane equivalent: entry `idx` (wave-uniform) of the lane-indexed register `reg` := val
__device__ __forceinline__ int put_lane(int reg, int idx, int val) { return (int)(threadIdx.x & 63) == idx ? val : reg; }

// Points, the slice stack and the output polygon live in lane-indexed registers (entry i in
// lane i), read and written with v_readlane / v_writelane at wave-uniform indices; only the
// per-lane distance reads go to LDS.  Handles borders of up to 64 points.
__device__ inline bool approx_and_convex_wave(const uint8_t* pts, int count, uint32_t* vrows) {
    const int lane = threadIdx.x & 63;
    count = uni(count);
    const int pv = lane < count ? (int)pts[lane] : 0;            // pv: point j in lane j
    int sv = 0;                                                  // slice stack, entry i in lane i
    int dv = 0;                                                  // output polygon, vertex i in lane i
    int new_count = 0, top = 0;
    // 1. three farthest-point hops; key = dist<<12 | (4095 - j): max dist, then smallest j
    int pos = 0, right_start = 0;
    bool le_eps = false;
    int start_pt = 0;
    // Lane j holds point j, so a pass / slice needs no memory access: each lane works out its own
    // position t along the traversal (t = (j - start) mod count) and whether that lies inside.
    const int px = IRBPP_PX(pv), py = IRBPP_PY(pv);
    for (int it = 0; it < 3; ++it) {
        pos += right_start;
        if (pos >= count) pos -= count;
        start_pt = __builtin_amdgcn_readlane(pv, pos);
        const int sx = IRBPP_PX(start_pt), sy = IRBPP_PY(start_pt);
        int j = lane - pos;
        if (j < 0) j += count;
        uint32_t best = 0u;
        if (lane < count && j >= 1) {
            const int dx = px - sx, dy = py - sy;
            best = ((uint32_t)(dx * dx + dy * dy) << 12) | (uint32_t)(4095 - j);
        }
        best = wave_max_u32(best);
        const int max_dist = (int)(best >> 12);
        if (max_dist > 0) right_start = 4095 - (int)(best & 4095u);
        le_eps = max_dist <= 1;
    }
    if (!le_eps) {
        const int s0 = pos;
        int far = right_start + s0;
        if (far >= count) far -= count;
        sv = put_lane(sv, 0, far | (s0 << 16));     // right slice
        sv = put_lane(sv, 1, s0 | (far << 16));     // slice, processed first
        top = 2;
    } else {
        dv = put_lane(dv, 0, start_pt);
        new_count = 1;
    }
    // 3. Douglas-Peucker: one slice per iteration, its interior points spread over the lanes
    while (top > 0) {
        --top;
        const int sl = __builtin_amdgcn_readlane(sv, top);
        const int s_start = sl & 0xFFFF, s_end = (int)((unsigned)sl >> 16);
        start_pt = __builtin_amdgcn_readlane(pv, s_start);
        int len = s_end - s_start;                  // points from start to end along the closed curve
        if (len <= 0) len += count;
        bool le = true;
        int split = 0;
        if (len > 1) {
            const int end_pt = __builtin_amdgcn_readlane(pv, s_end);
            const int sx = IRBPP_PX(start_pt), sy = IRBPP_PY(start_pt);
            const int dx = IRBPP_PX(end_pt) - sx, dy = IRBPP_PY(end_pt) - sy;
            int t = lane - s_start;
            if (t < 0) t += count;
            uint32_t best = 0u;
            if (lane < count && t >= 1 && t < len) {
                int dist = (py - sy) * dx - (px - sx) * dy;
                dist = dist < 0 ? -dist : dist;
                best = ((uint32_t)dist << 12) | (uint32_t)(4095 - t);
            }
            best = wave_max_u32(best);
            const int max_dist = (int)(best >> 12);
            if (max_dist > 0) {
                split = s_start + 4095 - (int)(best & 4095u);
                if (split >= count) split -= count;
            }
            le = max_dist * max_dist <= dx * dx + dy * dy;
        }
        if (le) {
            dv = put_lane(dv, new_count, start_pt);
            ++new_count;
        } else {
            if (top + 2 > 64) return false;
            sv = put_lane(sv, top, split | (s_end << 16));
            sv = put_lane(sv, top + 1, s_start | (split << 16));
            top += 2;
        }
    }
    // 4. clean-up (inherently sequential, wave-uniform): in place on the register polygon
    {
        const int cnt = new_count;
        int p2 = cnt - 1;
        start_pt = __builtin_amdgcn_readlane(dv, p2);
        if (++p2 >= cnt) p2 = 0;
        int wpos = p2;
        int pt = __builtin_amdgcn_readlane(dv, p2);
        if (++p2 >= cnt) p2 = 0;
        for (int i = 0; i < cnt && new_count > 2; ++i) {
            const int end_pt = __builtin_amdgcn_readlane(dv, p2);
            if (++p2 >= cnt) p2 = 0;
            const int dx = IRBPP_PX(end_pt) - IRBPP_PX(start_pt), dy = IRBPP_PY(end_pt) - IRBPP_PY(start_pt);
            const int ux = IRBPP_PX(pt) - IRBPP_PX(start_pt), uy = IRBPP_PY(pt) - IRBPP_PY(start_pt);
            int dist = ux * dy - uy * dx;
            dist = dist < 0 ? -dist : dist;
            const int inner = ux * (IRBPP_PX(end_pt) - IRBPP_PX(pt)) + uy * (IRBPP_PY(end_pt) - IRBPP_PY(pt));
            if (2 * dist * dist <= dx * dx + dy * dy && dx != 0 && dy != 0 && inner >= 0) {
                --new_count;
                start_pt = end_pt;
                dv = put_lane(dv, wpos, end_pt);
                if (++wpos >= cnt) wpos = 0;
                pt = __builtin_amdgcn_readlane(dv, p2);
                if (++p2 >= cnt) p2 = 0;
                ++i;
                continue;
            }
            start_pt = pt;
            dv = put_lane(dv, wpos, pt);
            if (++wpos >= cnt) wpos = 0;
            pt = end_pt;
        }
    }
    // find_convex_vetex, one vertex per lane (neighbours through the cross-lane network)
    const int m = new_count;
    {
        const int ia = lane == 0 ? m - 1 : lane - 1, ic = lane == m - 1 ? 0 : lane + 1;
        const int a = __shfl(dv, ia < 0 ? 0 : ia), c = __shfl(dv, ic > 63 ? 63 : ic), b = dv;
        if (lane < m) {
            bool keep = true;
            if (m > 3)
                keep = (IRBPP_PX(b) - IRBPP_PX(a)) * (IRBPP_PY(c) - IRBPP_PY(a)) -
                       (IRBPP_PY(b) - IRBPP_PY(a)) * (IRBPP_PX(c) - IRBPP_PX(a)) < 0;
            if (keep) atomicOr(&vrows[IRBPP_PY(b)], 1u << IRBPP_PX(b));
        }
    }
    return true;
}

}  // namespace irbpp
