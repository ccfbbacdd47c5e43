// Host harness (test infrastructure): compiles the lane-serial device routines of
// irbpp_amd/csrc/contours_device.h -- candidate starts, border tracing, Douglas-Peucker + convexity --
// with g++ so that the CPU test-suite can run the very code the GPU executes against the oracle on
// thousands of images.  The wave-cooperative variant needs real lanes and stays a GPU test.
#include <stdint.h>
#include <string.h>

// the device header includes <hip/hip_runtime.h>: tests/host/stub/ holds an empty one, the few names used follow
#define __device__
#define __forceinline__ inline
static inline int __ffs(int v) { return __builtin_ffs(v); }
static inline int __clz(int v) { return v ? __builtin_clz((unsigned)v) : 32; }
static inline unsigned __brev(unsigned v) { unsigned r = 0; for (int i = 0; i < 32; ++i) r |= ((v >> i) & 1u) << (31 - i); return r; }
static inline unsigned atomicOr(uint32_t* p, uint32_t v) { const unsigned o = *p; *p |= v; return o; }
// cross-lane builtins appear only in the cooperative routine, which is compiled but never called here
struct { unsigned x; } threadIdx = {0};
#define __builtin_amdgcn_readlane(v, l) (v)
#define __builtin_amdgcn_readfirstlane(v) (v)
#define __builtin_amdgcn_update_dpp(old, v, ctrl, rm, bm, bc) (v)
#define __shfl(v, l) (v)
#define IRBPP_WAVE_SYNC()
static inline unsigned long long __ballot(bool p) { return p ? 1ull : 0ull; }
static inline unsigned atomicMax(uint32_t* p, uint32_t v) { const unsigned o = *p; if (v > o) *p = v; return o; }
static inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
static inline int __ffsll(long long v) { return __builtin_ffsll(v); }
static inline int __clzll(long long v) { return v ? __builtin_clzll((unsigned long long)v) : 64; }

static long g_trace_iters = 0;
#define IRBPP_TRACE_ITER() (++g_trace_iters)
#include "../../irbpp_amd/csrc/contours_device.h"

using namespace irbpp;

static void transpose(const uint16_t* rows, uint16_t* cols) {
    for (int x = 0; x < 16; ++x) {
        uint16_t c = 0;
        for (int y = 0; y < 16; ++y) c |= (uint16_t)(((rows[y] >> x) & 1u) << y);
        cols[x] = c;
    }
}

extern "C" {

// rows[16]: bit x of word y = pixel (x, y).  out[16] = candidate start bits per row.
void host_start_candidates(const uint16_t* rows, uint32_t* out) {
    for (int y = 0; y < 16; ++y) out[y] = start_candidates(rows[y], y ? rows[y - 1] : 0u);
}

// rows[16] -> cols[16] (bit y of word x = pixel (x, y)) with the register transpose of the trace kernel
void host_transpose16(const uint16_t* rows, uint16_t* cols) {
    uint32_t r[16];
    for (int y = 0; y < 16; ++y) r[y] = rows[y];
    transpose16(r);
    for (int x = 0; x < 16; ++x) cols[x] = (uint16_t)r[x];
}

// tooling: iterations of the border walk since the last call
long host_trace_iters(void) { const long v = g_trace_iters; g_trace_iters = 0; return v; }

// Returns trace_border's value; pts receives min(n, cap) points as x | y<<4.
int host_trace_border(const uint16_t* rows, int x0, int y0, uint8_t* pts, int cap) {
    uint16_t cols[16];
    transpose(rows, cols);
    return trace_border(rows, cols, x0, y0, pts, cap);
}

// trace_border_fast (the trace kernel's walk) on the frames of the image: same contract as host_trace_border
int host_trace_border_fast(const uint16_t* rows, int x0, int y0, uint8_t* pts, int cap) {
    uint32_t r[16], c[16], fr[FRAME_WORDS];
    for (int y = 0; y < 16; ++y) r[y] = c[y] = rows[y];
    transpose16(c);
    frames_store(fr, r, c);
    return trace_border_fast(fr, x0, y0, pts, cap);
}

// the same with the slot split the way the trace kernel splits it: the first cap_lds points in `pts`, the following ones
// (up to spill_cap) through the walk's spill pointer; pts receives them joined
int host_trace_border_fast_spill(const uint16_t* rows, int x0, int y0, uint8_t* pts, int cap_lds, int spill_cap) {
    uint32_t r[16], c[16], fr[FRAME_WORDS];
    for (int y = 0; y < 16; ++y) r[y] = c[y] = rows[y];
    transpose16(c);
    frames_store(fr, r, c);
    static uint8_t head[4096], tail[4096];
    memset(head, 0xEE, sizeof head);
    memset(tail, 0xEE, sizeof tail);
    const int n = trace_border_fast(fr, x0, y0, head, cap_lds, true, tail, spill_cap);
    for (int i = 0; i < n && i < cap_lds + spill_cap; ++i) pts[i] = i < cap_lds ? head[i] : tail[i - cap_lds];
    for (int i = cap_lds + 1; i < 4096; ++i) if (head[i] != 0xEE) return -100;        // wrote behind the slot's dump byte
    for (int i = spill_cap; i < 4096; ++i) if (tail[i] != 0xEE) return -101;           // wrote behind its spill bytes
    return n;
}

// the walk in its start / iteration form (the trace kernel's lane-refill build): same contracts as the two above
int host_trace_border_walk(const uint16_t* rows, int x0, int y0, uint8_t* pts, int cap) {
    uint32_t r[16], c[16], fr[FRAME_WORDS];
    for (int y = 0; y < 16; ++y) r[y] = c[y] = rows[y];
    transpose16(c);
    frames_store(fr, r, c);
    return trace_border_walk(fr, x0, y0, pts, cap);
}
int host_trace_border_walk_spill(const uint16_t* rows, int x0, int y0, uint8_t* pts, int cap_lds, int spill_cap) {
    uint32_t r[16], c[16], fr[FRAME_WORDS];
    for (int y = 0; y < 16; ++y) r[y] = c[y] = rows[y];
    transpose16(c);
    frames_store(fr, r, c);
    static uint8_t head[4096], tail[4096];
    memset(head, 0xEE, sizeof head);
    memset(tail, 0xEE, sizeof tail);
    const int n = trace_border_walk(fr, x0, y0, head, cap_lds, tail, spill_cap);
    for (int i = 0; i < n && i < cap_lds + spill_cap; ++i) pts[i] = i < cap_lds ? head[i] : tail[i - cap_lds];
    for (int i = cap_lds + 1; i < 4096; ++i) if (head[i] != 0xEE) return -100;
    for (int i = spill_cap; i < 4096; ++i) if (tail[i] != 0xEE) return -101;
    return n;
}

// WIDE action grids (up to 32 x 32): candidate starts of every row, and the convex vertices of every component's outer border
// through trace_border_wide + approx_and_convex_t<uint16_t, 5> -- what irbpp_kernels.hip's wide_observe does per level image.
// rows[H]: bit x of word y = pixel (x, y); out_cand[H]: candidate start bits; vrows[H]: vertex bits.  Returns the number of borders
// (candidates that turned out first pixels), negative on a guard / stack failure.
int host_wide_image_vertices(const uint32_t* rows, int W, int H, int cap, int cap_stk, uint32_t* out_cand, uint32_t* vrows, int* longest) {
    static uint16_t pts[4096], dst[4096];
    static uint32_t stk[4096];
    const uint32_t wmask = W >= 32 ? 0xFFFFFFFFu : ((1u << W) - 1u);
    int borders = 0;
    *longest = 0;
    for (int y = 0; y < H; ++y) { vrows[y] = 0; out_cand[y] = start_candidates_wide(rows[y], y ? rows[y - 1] : 0u, wmask); }
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            if (!((out_cand[y] >> x) & 1u)) continue;
            const int n = trace_border_wide<5>(rows, W, H, x, y, pts, cap);
            if (n < 0) return -1;
            if (n == 0) continue;
            if (n > cap) return -2;
            if (n > *longest) *longest = n;
            if (!approx_and_convex_t<uint16_t, 5>(pts, n, dst, stk, cap_stk, vrows)) return -3;
            ++borders;
        }
    return borders;
}

// the wide walk with run jumps against the plain wide walk: every candidate start of the image, same return value and points.
// Returns the number of starts compared, negative at the first difference.
int host_wide_runs_equal_plain(const uint32_t* rows, int W, int H) {
    static uint16_t pa[4096], pb[4096];
    uint32_t cols[32];
    for (int x = 0; x < 32; ++x) { cols[x] = 0; for (int y = 0; y < H; ++y) cols[x] |= ((rows[y] >> x) & 1u) << y; }
    const uint32_t wmask = W >= 32 ? 0xFFFFFFFFu : ((1u << W) - 1u);
    int compared = 0;
    for (int y = 0; y < H; ++y) {
        const uint32_t cand = start_candidates_wide(rows[y], y ? rows[y - 1] : 0u, wmask);
        for (int x = 0; x < W; ++x) {
            if (!((cand >> x) & 1u)) continue;
            const int na = trace_border_wide<5>(rows, W, H, x, y, pa, 4096);
            const int nb = trace_border_wide_runs<5>(rows, cols, W, H, x, y, pb, 4096);
            if (na != nb) return -1 - compared;
            for (int i = 0; i < na; ++i) if (pa[i] != pb[i]) return -100000 - compared;
            ++compared;
        }
    }
    return compared;
}

// approx_and_convex on a point list: vrows[16] gets the vertex bits; returns 1 ok, 0 stack overflow.
int host_approx_and_convex(const uint8_t* pts, int count, int cap_stk, uint32_t* vrows) {
    static uint8_t dst[4096];
    static uint32_t stk[4096];
    memset(vrows, 0, 16 * sizeof(uint32_t));
    return approx_and_convex(pts, count, dst, stk, cap_stk, vrows) ? 1 : 0;
}

// contour_vertices with a slot of the given capacities: 0 ok, 1 overflow, 2 guard.
int host_contour_vertices(const uint16_t* rows, int x0, int y0, int cap, int cap_stk, uint32_t* vrows) {
    static uint8_t pts[4096], dst[4096];
    static uint32_t stk[4096];
    uint16_t cols[16];
    transpose(rows, cols);
    SlotMem m;
    m.pts = pts; m.dst = dst; m.stk = stk; m.cap = cap; m.cap_stk = cap_stk;
    return contour_vertices(rows, cols, x0, y0, m, vrows);
}

// rect_component + rect_vertices for candidate (x0, y0): 0 not an isolated solid rectangle; else w | h << 8, vertex bits in vrows[16]
int host_rect_component(const uint16_t* rows, int x0, int y0, uint32_t* vrows) {
    int w = 0, h = 0;
    memset(vrows, 0, 16 * sizeof(uint32_t));
    if (!rect_component(rows, (uint32_t)rows[y0], x0, y0, w, h)) return 0;
    uint32_t top = 0, bottom = 0;
    rect_vertices(w, h, x0, top, bottom);
    vrows[y0] |= top;
    vrows[y0 + h - 1] |= bottom;
    return w | (h << 8);
}

}  // extern "C"
