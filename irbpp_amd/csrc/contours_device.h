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

#ifndef IRBPP_TRACE_ITER
#define IRBPP_TRACE_ITER()       // host tooling counts the iterations of the border walk here
#endif

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

// Candidate start pixels of row `row` given the row above it (`up`, 0 for the first row): the first pixel of every
// horizontal run that no pixel of the row above touches -- anywhere along the run, diagonals included.  The border that
// findContours reports for a component starts at the component's first pixel in raster order; a run touched from above
// belongs to a component with a pixel in an earlier row, so none of its pixels is one (and the walk from its first pixel,
// were it listed, would meet an earlier pixel and report nothing: listing fewer candidates changes no result).  Looking
// at the whole run instead of the three pixels above its first one drops two thirds of the false starts of lattice data
// (staircases that descend to the left): 33 % -> 13 % of the candidates on BlockOut, 27 % -> 18 % on "general"
// (profiles/r04/LOG.md, session 30) -- fewer lanes and waves for the trace kernel.
__device__ __forceinline__ uint32_t start_candidates(uint32_t row, uint32_t up) {
    const uint32_t upm = up | (up << 1) | (up >> 1);
    const uint32_t first = row & ~(row << 1) & ~upm & 0xFFFFu;       // first pixels of runs, nothing above themselves
    // pixels of the row touched from above, spread towards the first pixel of their run (distances 1, 2, 4, 8)
    uint32_t t = upm & row;
    const uint32_t p2 = row & (row >> 1), p4 = p2 & (p2 >> 2), p8 = p4 & (p4 >> 4);
    t |= (t >> 1) & row;
    t |= (t >> 2) & p2;
    t |= (t >> 4) & p4;
    t |= (t >> 8) & p8;
    return first & ~t;
}

// Transposed copy of a level image, in place on 16 registers: r[y] = row word y (bit x = pixel (x, y)) becomes
// r[x] = column word x (bit y = pixel (x, y)).  Four block-swap stages of the 16 x 16 bit matrix (8, 4, 2, 1).
__device__ __forceinline__ void transpose16(uint32_t (&r)[16]) {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const uint32_t t = ((r[k] >> 8) ^ r[k + 8]) & 0x00FFu;
        r[k + 8] ^= t;
        r[k] ^= t << 8;
    }
#pragma unroll
    for (int g = 0; g < 16; g += 8)
#pragma unroll
        for (int k = g; k < g + 4; ++k) {
            const uint32_t t = ((r[k] >> 4) ^ r[k + 4]) & 0x0F0Fu;
            r[k + 4] ^= t;
            r[k] ^= t << 4;
        }
#pragma unroll
    for (int g = 0; g < 16; g += 4)
#pragma unroll
        for (int k = g; k < g + 2; ++k) {
            const uint32_t t = ((r[k] >> 2) ^ r[k + 2]) & 0x3333u;
            r[k + 2] ^= t;
            r[k] ^= t << 2;
        }
#pragma unroll
    for (int k = 0; k < 16; k += 2) {
        const uint32_t t = ((r[k] >> 1) ^ r[k + 1]) & 0x5555u;
        r[k + 1] ^= t;
        r[k] ^= t << 1;
    }
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

// icvFetchContourEx with CHAIN_APPROX_SIMPLE for the OUTER border starting at (x0,y0), as a resumable
// walk: trace_init() + one trace_step() per direction change / diagonal step, so that a wave can keep all
// its lanes busy by handing a lane the next border as soon as its current one is closed (trace kernel), or
// simply loop to the end (trace_border).  The neighbour mask of the current pixel stays in the state;
// axis-aligned runs are jumped in one go.  img = 16 row words, imgT = 16 column words (bit y of word x).
struct TraceState {
    int x0, y0, x1, y1, s;        // start pixel, its first neighbour, the first direction
    int x3, y3, cur_s, prev_s;    // current pixel, direction back to the previous pixel, last move
    uint32_t nb;                  // 8-bit neighbour mask of the current pixel
    int n;                        // points produced so far (stored only while they fit)
};
constexpr int TRACE_RUNNING = -2;

// Returns 1 for an isolated pixel (its single point stored), else TRACE_RUNNING.
__device__ inline int trace_init(TraceState& t, const uint16_t* img, int x0, int y0, uint8_t* pts, int cap) {
    const uint32_t ra = y0 > 0 ? img[y0 - 1] : 0u, rb = img[y0], rc = y0 < 15 ? img[y0 + 1] : 0u;
    t.nb = nb_mask(ra, rb, rc, x0);
    // clockwise search 3,2,1,0,7,6,5 (s_end = 4: the west pixel is background) for the first neighbour
    const uint32_t rot = ((t.nb << 4) | (t.nb >> 4)) & 0xFFu;     // direction 3 -> bit 7
    // (every field is assigned exactly once, outside any branch: stores under diverging branches end up as one
    // store through a selected address, which keeps the whole state in scratch memory)
    const bool iso = rot == 0u;                                    // isolated pixel
    const int s = iso ? 0 : (3 - (7 - (31 - __clz((int)rot)))) & 7;
    t.x0 = x0;
    t.y0 = y0;
    t.x3 = x0;
    t.y3 = y0;
    t.n = 0;
    t.s = s;
    t.x1 = iso ? 0 : x0 + dir_dx(s);
    t.y1 = iso ? 0 : y0 + dir_dy(s);
    t.prev_s = iso ? 0 : s ^ 4;
    t.cur_s = s;
    if (iso && cap > 0) pts[0] = (uint8_t)(x0 | (y0 << 4));
    return iso ? 1 : TRACE_RUNNING;
}

// The 3 x 3 neighbourhood seen in the TRANSPOSED image (rows = columns of the image) back in image orientation: a
// neighbour at offset (dx, dy) there is the neighbour at (dy, dx) here, i.e. direction s becomes (6 - s) & 7
// (E <-> S, NE <-> SW, N <-> W; NW and SE stay): bits 0..6 reversed, bit 7 kept.
__device__ __forceinline__ uint32_t nb_untranspose(uint32_t m) { return (__brev(m & 0x7Fu) >> 25) | (m & 0x80u); }

// One step of the walk.  Returns TRACE_RUNNING, or the final number of points: 0 if the walk met a pixel
// that precedes (x0,y0) in raster order, i.e. (x0,y0) is not the first pixel of its component and the
// border belongs to another start (or is a hole border).
// A step costs ONE round of LDS reads: three 16-bit words around the pixel just entered -- rows (y-1, y, y+1) after a
// horizontal or diagonal move, columns (x-1, x, x+1) of the transposed copy after a vertical one.  The same three
// words serve the jump to the end of an axis-aligned run (the line walked and the line beside it) and the
// neighbour mask of the pixel where the jump ends (in the transposed frame for a vertical move, mapped back with
// nb_untranspose).
__device__ inline int trace_step(TraceState& t, const uint16_t* img, const uint16_t* imgT, uint8_t* pts, int cap) {
    IRBPP_TRACE_ITER();
    // counter-clockwise search cur_s+1, cur_s+2, ... for the next border pixel
    const int k2 = (t.cur_s + 1) & 7;
    const uint32_t r2 = ((t.nb >> k2) | (t.nb << (8 - k2))) & 0xFFu;   // direction k2 -> bit 0
    const int s2 = (k2 + __ffs((int)r2) - 1) & 7;
    int x4 = t.x3 + dir_dx(s2), y4 = t.y3 + dir_dy(s2);
    if (s2 != t.prev_s) {                                         // CHAIN_APPROX_SIMPLE
        if (t.n < cap) pts[t.n] = (uint8_t)(t.x3 | (t.y3 << 4));
        ++t.n;
    }
    t.prev_s = s2;
    if (x4 == t.x0 && y4 == t.y0 && t.x3 == t.x1 && t.y3 == t.y1) return t.n;
    // now standing on (x4,y4), arrived by s2.  Branch-free on purpose: the lanes of a wave walk different borders in
    // lockstep.
    const bool vert = (s2 & 3) == 2;                              // N or S: work in the transposed image
    const bool axis = (s2 & 1) == 0;                              // E, N, W, S: a straight run may follow
    const bool fwd = s2 == 0 || s2 == 6;                          // E or S: towards higher bits
    const uint16_t* base = vert ? imgT : img;
    const int li = vert ? x4 : y4;                                // the line walked (row, or column of the image)
    int p = vert ? y4 : x4;                                       // position along it
    const uint32_t wm = base[li];
    const uint32_t wa_raw = base[(li - 1) & 15], wb_raw = base[(li + 1) & 15];
    const uint32_t wa = li > 0 ? wa_raw : 0u, wb = li < 15 ? wb_raw : 0u;
    {
        // straight run: E looks at the row below, W at the row above, S at the column to the left, N at the column to
        // the right (the three neighbours probed before the walking direction)
        const uint32_t side = (vert != fwd) ? wb : wa;
        const int lf = run_forward(wm, side, p), lb = run_backward(wm, side, p);
        p += axis ? (fwd ? lf : -lb) : 0;
    }
    x4 = vert ? x4 : p;
    y4 = vert ? p : y4;
    // a run that ends on the start pixel, moving opposite to the first step, closes the border
    if (x4 == t.x0 && y4 == t.y0 && s2 == (t.s ^ 4)) return t.n;
    // run interiors lie between their end points in raster order, so testing end points suffices
    if (y4 * 16 + x4 < t.y0 * 16 + t.x0) return 0;
    const uint32_t nb_local = nb_mask(wa, wm, wb, p);
    t.x3 = x4;
    t.y3 = y4;
    t.cur_s = (s2 + 4) & 7;
    t.nb = vert ? nb_untranspose(nb_local) : nb_local;
    return TRACE_RUNNING;
}

// The whole border in one go: number of points, 0 if (x0,y0) does not start an outer border, -1 if the
// iteration guard tripped.
__device__ inline int trace_border(const uint16_t* img, const uint16_t* imgT, int x0, int y0, uint8_t* pts, int cap) {
    TraceState t;
    int r = trace_init(t, img, x0, y0, pts, cap);
    for (int guard = 0; guard < 4096 && r == TRACE_RUNNING; ++guard) r = trace_step(t, img, imgT, pts, cap);
    return r == TRACE_RUNNING ? -1 : r;
}

// ---------------------------------------------------------------------------------------
// The same walk with a short iteration (the trace kernel's form; trace_border above stays as the plain statement
// of it and serves the in-workgroup contour stage of the hull kernel and the sequential redo).  What is different:
//   * a pixel is ONE number, pos = x | y << 4 -- the byte a contour point is stored as -- and a move adds a per-direction
//     delta; all per-direction constants (move delta, frame, run direction, side line) come out of 8-entry byte tables
//     held in two SGPRs each and read with one v_perm_b32 (the direction code carries 0x0c in its upper bytes, which
//     makes v_perm return the table byte zero-extended);
//   * the level image lies in LDS as two FRAMES of 18 dwords: rows (y = -1 .. 16) and columns (x = -1 .. 16), every
//     line shifted left by one bit, a zero line before and behind (35 dwords: the two frames share the zero line between them): the three lines around a pixel are three reads at
//     one address (no edge cases), and bits p-1, p, p+1 of a line are bits p .. p+2 of the stored word (no shifts by -1);
//   * the contour point is stored unconditionally at index min(n, cap) and n advances only when the direction changed
//     (CHAIN_APPROX_SIMPLE): a point that is not one gets overwritten by the next store;
//   * one loop exit: the three ways a step can end the walk are folded into its result by selects, so the lanes of a
//     wave -- which walk different borders in lockstep -- run straight-line code.
// ~85 instructions per step where trace_step compiles to ~160 (disassembly of the trace kernel).
// ---------------------------------------------------------------------------------------
#ifndef IRBPP_COMPACT_FRAMES
#define IRBPP_COMPACT_FRAMES 0      // 1 (A/B, tools/build_variant.sh): lines as 16-bit halfwords, two per dword -- 76 instead of 140 bytes of LDS per lane
#endif
#if IRBPP_COMPACT_FRAMES
// halfword h of the frame: 0 a zero line, 1 + y row y, 17 a zero line, 18 + x column x, 34 / 35 zero: the three lines around row y are
// halfwords y .. y + 2, around column x halfwords 17 + x .. 19 + x; raw 16-bit lines (the walk shifts them left by one bit after the read)
constexpr int FRAME_LINES = 18, FRAME_COLS = 17, FRAME_WORDS = 19;          // 18 dwords of lines + 1: an odd stride between the lanes' frames
#else
constexpr int FRAME_LINES = 18, FRAME_COLS = FRAME_LINES - 1, FRAME_WORDS = 2 * FRAME_LINES - 1;   // (the zero line behind the rows is the one before the columns)
#endif
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ uint32_t byte_table(uint32_t hi, uint32_t lo, uint32_t sel) { return __builtin_amdgcn_perm(hi, lo, sel); }
#else
__device__ inline uint32_t byte_table(uint32_t hi, uint32_t lo, uint32_t sel) {       // v_perm_b32 as far as it is used here (host passes)
    const unsigned long long t = ((unsigned long long)hi << 32) | lo;
    uint32_t out = 0;
    for (int k = 0; k < 4; ++k) {
        const uint32_t c = (sel >> (8 * k)) & 255u;
        out |= (c < 8 ? (uint32_t)((t >> (8 * c)) & 255u) : (c == 12 ? 0u : 255u)) << (8 * k);
    }
    return out;
}
#endif
// frames of one level image from its row words r[y] (bit x = pixel (x, y)) and column words c[x] (bit y)
#if IRBPP_COMPACT_FRAMES
__device__ __forceinline__ void frames_store(uint32_t* fr, const uint32_t (&r)[16], const uint32_t (&c)[16]) {
    fr[0] = r[0] << 16;                                                        // halfwords 0 (zero), 1 (row 0)
#pragma unroll
    for (int k = 1; k < 8; ++k) fr[k] = r[2 * k - 1] | (r[2 * k] << 16);       // halfwords 2k, 2k + 1 = rows 2k - 1, 2k
    fr[8] = r[15];                                                             // halfwords 16 (row 15), 17 (zero)
#pragma unroll
    for (int k = 0; k < 8; ++k) fr[9 + k] = c[2 * k] | (c[2 * k + 1] << 16);   // halfwords 18 + 2k, 19 + 2k = columns 2k, 2k + 1
    fr[17] = 0u;                                                               // halfwords 34, 35
}
// the three lines around line `li` of the frame that starts at halfword `hoff` (0 rows, 17 columns), shifted left by one bit like the
// stored lines of the wide-frame layout: bit q + 1 = pixel q
__device__ __forceinline__ void frame_lines(const uint32_t* fr, uint32_t li, uint32_t hoff, uint32_t& wa, uint32_t& wm, uint32_t& wb) {
    const uint32_t h0 = li + hoff, sh = (h0 & 1u) << 4;
    const uint32_t* p = fr + (h0 >> 1);
    const uint32_t d0 = p[0], d1 = p[1];
#if defined(__HIP_DEVICE_COMPILE__)
    const uint32_t x = __builtin_amdgcn_alignbit(d1, d0, sh);
#else
    const uint32_t x = (uint32_t)(((((unsigned long long)d1) << 32) | d0) >> sh);
#endif
    wa = (x << 1) & 0x1FFFEu;
    wm = (x >> 15) & 0x1FFFEu;
    wb = ((d1 >> sh) << 1) & 0x1FFFEu;
}
__device__ __forceinline__ uint32_t frame_raw_row(const uint32_t* fr, int y) { return (fr[(1 + y) >> 1] >> (((1 + y) & 1) * 16)) & 0xFFFFu; }
__device__ __forceinline__ uint32_t frame_raw_col(const uint32_t* fr, int x) { return (fr[(18 + x) >> 1] >> (((18 + x) & 1) * 16)) & 0xFFFFu; }
constexpr uint32_t FRAME_COL_OFF = 17u;                                        // what the per-direction table adds to the line index for N / S
#else
__device__ __forceinline__ void frames_store(uint32_t* fr, const uint32_t (&r)[16], const uint32_t (&c)[16]) {
    fr[0] = 0u; fr[FRAME_LINES - 1] = 0u; fr[FRAME_COLS] = 0u; fr[FRAME_WORDS - 1] = 0u;
#pragma unroll
    for (int k = 0; k < 16; ++k) { fr[1 + k] = r[k] << 1; fr[FRAME_COLS + 1 + k] = c[k] << 1; }
}
// the three lines around line `li` of the frame at byte offset `boff`
__device__ __forceinline__ void frame_lines(const uint32_t* fr, uint32_t li, uint32_t boff, uint32_t& wa, uint32_t& wm, uint32_t& wb) {
    const uint32_t* ln = (const uint32_t*)((const char*)fr + ((li << 2) + boff));
    wa = ln[0]; wm = ln[1]; wb = ln[2];
}
__device__ __forceinline__ uint32_t frame_raw_row(const uint32_t* fr, int y) { return fr[1 + y] >> 1; }
__device__ __forceinline__ uint32_t frame_raw_col(const uint32_t* fr, int x) { return fr[FRAME_COLS + 1 + x] >> 1; }
constexpr uint32_t FRAME_COL_OFF = 4u * FRAME_COLS;                            // (a byte offset here)
#endif
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ uint32_t bit_field(uint32_t v, uint32_t off, uint32_t width) { return __builtin_amdgcn_ubfe(v, off, width); }   // v_bfe_u32
// bit `bit` of v as a mask, 0 or ~0 (v_bfe_i32; as asm because the compiler otherwise rewrites mask-and-select as and + compare + cndmask)
template <int BIT>
__device__ __forceinline__ uint32_t bit_flag(uint32_t v) { uint32_t m; asm("v_bfe_i32 %0, %1, %2, 1" : "=v"(m) : "v"(v), "n"(BIT)); return m; }
#else
__device__ inline uint32_t bit_field(uint32_t v, uint32_t off, uint32_t width) { return (v >> (off & 31u)) & ((1u << width) - 1u); }
template <int BIT>
__device__ inline uint32_t bit_flag(uint32_t v) { return ((v >> BIT) & 1u) ? ~0u : 0u; }
#endif
__device__ __forceinline__ uint32_t bit_select(uint32_t mask, uint32_t if_set, uint32_t if_clear) { return (if_set & mask) | (if_clear & ~mask); }   // v_bfi_b32
// 8-bit neighbour mask (bit s = neighbour in direction s, in the frame's own orientation) from the three stored lines
// around a pixel at position p of the middle line: bits p .. p+2 of a stored line are pixels p-1, p, p+1
// (the table words are passed in: the walk keeps them in scalar registers of its own for the whole loop -- the compiler,
// left to itself, re-materialises such constants with an s_mov right in front of every use, and a vector instruction behind
// a fresh scalar one costs a lone wave 13 cycles instead of 6, tools/microbench_int.hip)
struct NbTables { uint32_t rev_hi, rev_lo, ew_hi, ew_lo; };
__device__ __forceinline__ NbTables nb_tables() { return NbTables{0x0e060a02u, 0x0c040800u, 0x11011101u, 0x10001000u}; }
__device__ __forceinline__ uint32_t nb_frame(uint32_t a, uint32_t b, uint32_t c, int p, const NbTables& t = nb_tables()) {
    constexpr uint32_t SEL = 0x0c0c0c00u;
    const uint32_t ra = byte_table(t.rev_hi, t.rev_lo, ((a >> p) & 7u) | SEL);             // NE, N, NW (bits 1..3): the line before, reversed
    const uint32_t rb = byte_table(t.ew_hi, t.ew_lo, ((b >> p) & 7u) | SEL);               // E (bit 0) = pixel p+1, W (bit 4) = pixel p-1
    return (bit_field(c, (uint32_t)p, 3) << 5) | ra | rb;                                  // SW, S, SE (bits 5..7): the line after
}
// All lanes of the wave call it together; `active` says whether the lane has a border to follow (an inactive lane
// passes x0 = y0 = 0 and pointers into memory of its own: it computes along, harmlessly).  The loop is uniform -- it
// runs until no lane of the wave is walking any more -- and its body is straight-line code for every lane: a lane that
// has closed its border keeps computing on whatever its registers hold (every address it forms stays inside its own
// frames and its own slot, and n no longer advances), which costs nothing in lockstep and saves the execution-mask
// bookkeeping of a per-lane exit (two dozen scalar instructions per step in the compiler's rendering of it).
// Returns the number of points, 0 if (x0, y0) does not start an outer border, -1 if the iteration guard tripped.
// `spill`: where the points from index `cap` on go (global memory, `spill_cap` bytes of the lane's own; nullptr / 0: nowhere,
// as before) -- the slot in LDS holds the first `cap` points, which is all of nearly every border; the others cost their
// lanes one more store per step, behind a branch the whole wave skips otherwise.
__device__ inline int trace_border_fast(const uint32_t* fr, int x0, int y0, uint8_t* pts, int cap, bool active = true,
                                        uint8_t* spill = nullptr, int spill_cap = 0) {
    // per-direction tables, entry s in byte s: pos delta + 17; flags (1 axis move, 2 vertical frame, 4 towards higher
    // bits, 8 side line = the line after); bit offset of the line index in pos (4 rows / 0 columns); frame offset in bytes
    uint32_t DELTA_LO = 0x00010212u, DELTA_HI = 0x22212010u;                // E +1, NE -15, N -16, NW -17 | W -1, SW +15, S +16, SE +17
    uint32_t FLAG_LO = 0x000b000du, FLAG_HI = 0x00070001u;                  // E 13, N 11 | W 1, S 7
    uint32_t LSH_LO = 0x04000404u, LSH_HI = LSH_LO;                         // N and S index columns (x = pos & 15), all others rows
    uint32_t FOFF_LO = FRAME_COL_OFF << 16, FOFF_HI = FOFF_LO;              // N and S read the column frame (byte / halfword offset)
    NbTables NT = nb_tables();
#if defined(__HIP_DEVICE_COMPILE__)
    // opaque to the compiler from here on: the table words stay in the scalar registers they are in (see nb_frame)
    // (v_perm_b32 takes one scalar operand: the low halves of the two-word tables live in vector registers)
    asm volatile("" : "+v"(DELTA_LO), "+s"(DELTA_HI), "+v"(FLAG_LO), "+s"(FLAG_HI), "+s"(LSH_LO), "+s"(FOFF_LO));
    asm volatile("" : "+s"(NT.rev_hi), "+v"(NT.rev_lo), "+s"(NT.ew_hi), "+v"(NT.ew_lo));
    LSH_HI = LSH_LO;
    FOFF_HI = FOFF_LO;
#endif
    constexpr uint32_t SEL = 0x0c0c0c00u;
    const int pos0 = x0 | (y0 << 4);
    // first neighbour: clockwise search 3, 2, 1, 0, 7, 6, 5 (the west pixel is background)
    uint32_t wa0, wm0, wb0;
    frame_lines(fr, (uint32_t)y0, 0u, wa0, wm0, wb0);
    const uint32_t nb0 = nb_frame(wa0, wm0, wb0, x0);
    const uint32_t rot = ((nb0 << 4) | (nb0 >> 4)) & 0xFFu;                // direction 3 -> bit 7
    const bool isolated = rot == 0u;
    const int s_first = (3 - (7 - (31 - __builtin_clz(rot | (isolated ? 1u : 0u))))) & 7;
    const int pos1 = pos0 + (int)byte_table(DELTA_HI, DELTA_LO, (uint32_t)s_first | SEL) - 17;
    const uint32_t s_close = (uint32_t)(s_first ^ 4) | SEL;                // a run that ends on the start, moving against the first step
    int pos = pos0, n = 0, result = 0;
    uint32_t nb16 = nb0 | (nb0 << 8), k2 = (uint32_t)(s_first + 1), prev = (uint32_t)(s_first ^ 4) | SEL;
    uint32_t run = (active && !isolated) ? 1u : 0u;                        // (an integer, not a flag: no lane mask is carried around the loop)
    int guard = 4096;                                                      // (uniform; counts down: one scalar subtract and branch per step)
    while (__ballot(run != 0u) != 0ull) {                                  // uniform: somebody still walks
        if (run != 0u) {
        IRBPP_TRACE_ITER();
        // counter-clockwise search k2, k2 + 1, ... for the next border pixel (nb != 0: the pixel we came from)
        k2 &= 7u;
        const uint32_t s2 = ((k2 + (uint32_t)__builtin_ctz(nb16 >> k2)) & 7u) | SEL;
        pts[n < cap ? n : cap] = (uint8_t)pos;                            // CHAIN_APPROX_SIMPLE: kept iff the direction changed
        if (n >= cap && n - cap < spill_cap) spill[n - cap] = (uint8_t)pos;
        n += s2 != prev ? 1 : 0;
        prev = s2;
        int pos4 = pos + (int)byte_table(DELTA_HI, DELTA_LO, s2) - 17;
        const bool closed1 = pos4 == pos0 && pos == pos1;
        // standing on pos4, arrived by s2: the three lines around it in the frame of the move
        const uint32_t flags = byte_table(FLAG_HI, FLAG_LO, s2);
        const uint32_t lsh = byte_table(LSH_HI, LSH_LO, s2), psh = lsh ^ 4u;
        const uint32_t li = bit_field((uint32_t)pos4, lsh, 4);
        int p = (int)bit_field((uint32_t)pos4, psh, 4);
        uint32_t wa, wm, wb;
        frame_lines(fr, li, byte_table(FOFF_HI, FOFF_LO, s2), wa, wm, wb);
        {
            // straight run (see run_forward / run_backward): E looks at the row below, W at the row above, S at the
            // column to the left, N at the column to the right.  Stored lines are shifted by one: bit q + 1 = pixel q.
            const uint32_t side = bit_select(bit_flag<3>(flags), wb, wa);
            const uint32_t blocked = side | (side << 1) | (side >> 1);
            const uint32_t lf = (uint32_t)__builtin_ctz(~(((wm >> 1) & ~blocked) >> (p + 1)));
            const uint32_t lb = (uint32_t)__builtin_clz(~(((wm << 1) & ~blocked) << (30 - p)));
            const int dp = (int)(bit_select(bit_flag<2>(flags), lf, 0u - lb) & bit_flag<0>(flags));
            p += dp;
            pos4 += dp * (1 << psh);
        }
        const bool closed2 = pos4 == pos0 && s2 == s_close;
        const uint32_t nbl = nb_frame(wa, wm, wb, p, NT);
        const uint32_t nb = bit_select(bit_flag<1>(flags), nb_untranspose(nbl), nbl);
        const bool ends = closed1 || closed2;
        result = ends ? n : result;
        run = (ends || pos4 < pos0) ? 0u : 1u;                             // run interiors lie between their end points in raster order
        pos = pos4;
        k2 = s2 + 5u;
        nb16 = nb | (nb << 8);
        }
        if (--guard == 0) return -1;                                       // (uniform)
    }
    if (active && isolated && cap > 0) pts[0] = (uint8_t)pos0;
    return !active ? 0 : (isolated ? 1 : result);
}

// ---------------------------------------------------------------------------------------
// The same walk once more, cut into START and ONE ITERATION, for the trace kernel's lane-refill form (irbpp_kernels.hip:
// trace_refill_body): a wave owns a batch of candidate starts and hands a lane the next one as soon as enough lanes have
// closed their borders, instead of walking until the longest of 64 borders is done (a BlockOut lane walks 9 iterations on
// average, the longest of 64 takes 34: tools/trace_length_study.py).  Statement for statement the loop body of
// trace_border_fast; tests/host/ runs both against the oracle on the same images.
// ---------------------------------------------------------------------------------------
struct WalkTables { uint32_t DELTA_LO, DELTA_HI, FLAG_LO, FLAG_HI, LSH_LO, LSH_HI, FOFF_LO, FOFF_HI; NbTables NT; };
struct Walk {
    int pos, n, result, pos0, pos1;            // current pixel, points so far, points of the closed border (0: none), start, its first neighbour
    uint32_t nb16, k2, prev, s_close, run;     // neighbour mask (twice), next search direction, last move, closing move, 1 while walking
};
#define IRBPP_WALK_TABLES(WT)                                                                                               \
    WalkTables WT;                                                                                                          \
    WT.DELTA_LO = 0x00010212u; WT.DELTA_HI = 0x22212010u; WT.FLAG_LO = 0x000b000du; WT.FLAG_HI = 0x00070001u;               \
    WT.LSH_LO = 0x04000404u; WT.FOFF_LO = FRAME_COL_OFF << 16; WT.NT = nb_tables();                            \
    IRBPP_WALK_PIN(WT)                                                                                                      \
    WT.LSH_HI = WT.LSH_LO; WT.FOFF_HI = WT.FOFF_LO;
#if defined(__HIP_DEVICE_COMPILE__)
#define IRBPP_WALK_PIN(WT)                                                                                                  \
    asm volatile("" : "+v"(WT.DELTA_LO), "+s"(WT.DELTA_HI), "+v"(WT.FLAG_LO), "+s"(WT.FLAG_HI), "+s"(WT.LSH_LO), "+s"(WT.FOFF_LO)); \
    asm volatile("" : "+s"(WT.NT.rev_hi), "+v"(WT.NT.rev_lo), "+s"(WT.NT.ew_hi), "+v"(WT.NT.ew_lo));
#else
#define IRBPP_WALK_PIN(WT)
#endif
// the border that starts at (x0, y0) of the image in `fr`: a lane without one passes active = false.  An isolated pixel is its
// own one-point border: stored, result = 1, nothing to walk.
__device__ __forceinline__ void walk_start(Walk& w, const uint32_t* fr, int x0, int y0, uint8_t* pts, int cap, bool active, const WalkTables& T) {
    constexpr uint32_t SEL = 0x0c0c0c00u;
    const int pos0 = x0 | (y0 << 4);
    uint32_t wa0, wm0, wb0;
    frame_lines(fr, (uint32_t)y0, 0u, wa0, wm0, wb0);
    const uint32_t nb0 = nb_frame(wa0, wm0, wb0, x0, T.NT);
    const uint32_t rot = ((nb0 << 4) | (nb0 >> 4)) & 0xFFu;                // direction 3 -> bit 7
    const bool isolated = rot == 0u;
    const int s_first = (3 - (7 - (31 - __builtin_clz(rot | (isolated ? 1u : 0u))))) & 7;
    w.pos0 = pos0;
    w.pos1 = pos0 + (int)byte_table(T.DELTA_HI, T.DELTA_LO, (uint32_t)s_first | SEL) - 17;
    w.s_close = (uint32_t)(s_first ^ 4) | SEL;
    w.pos = pos0;
    w.n = 0;
    w.result = (active && isolated) ? 1 : 0;
    w.nb16 = nb0 | (nb0 << 8);
    w.k2 = (uint32_t)(s_first + 1);
    w.prev = (uint32_t)(s_first ^ 4) | SEL;
    w.run = (active && !isolated) ? 1u : 0u;
    if (active && isolated && cap > 0) pts[0] = (uint8_t)pos0;
}
// one step of a walking lane (call under `if (w.run != 0u)`); when the border closes w.result = its points and w.run = 0; a
// walk that meets a pixel before its start in raster order (not the first pixel of its component) ends with result 0
__device__ __forceinline__ void walk_iter(Walk& w, const uint32_t* fr, uint8_t* pts, int cap, uint8_t* spill, int spill_cap, const WalkTables& T) {
    constexpr uint32_t SEL = 0x0c0c0c00u;
    IRBPP_TRACE_ITER();
    const uint32_t k2 = w.k2 & 7u;
    const uint32_t s2 = ((k2 + (uint32_t)__builtin_ctz(w.nb16 >> k2)) & 7u) | SEL;
    const int n = w.n, pos = w.pos;
    pts[n < cap ? n : cap] = (uint8_t)pos;                                // CHAIN_APPROX_SIMPLE: kept iff the direction changed
    if (n >= cap && n - cap < spill_cap) spill[n - cap] = (uint8_t)pos;
    const int n1 = n + (s2 != w.prev ? 1 : 0);
    int pos4 = pos + (int)byte_table(T.DELTA_HI, T.DELTA_LO, s2) - 17;
    const bool closed1 = pos4 == w.pos0 && pos == w.pos1;
    const uint32_t flags = byte_table(T.FLAG_HI, T.FLAG_LO, s2);
    const uint32_t lsh = byte_table(T.LSH_HI, T.LSH_LO, s2), psh = lsh ^ 4u;
    const uint32_t li = bit_field((uint32_t)pos4, lsh, 4);
    int p = (int)bit_field((uint32_t)pos4, psh, 4);
    uint32_t wa, wm, wb;
    frame_lines(fr, li, byte_table(T.FOFF_HI, T.FOFF_LO, s2), wa, wm, wb);
    {
        const uint32_t side = bit_select(bit_flag<3>(flags), wb, wa);
        const uint32_t blocked = side | (side << 1) | (side >> 1);
        const uint32_t lf = (uint32_t)__builtin_ctz(~(((wm >> 1) & ~blocked) >> (p + 1)));
        const uint32_t lb = (uint32_t)__builtin_clz(~(((wm << 1) & ~blocked) << (30 - p)));
        const int dp = (int)(bit_select(bit_flag<2>(flags), lf, 0u - lb) & bit_flag<0>(flags));
        p += dp;
        pos4 += dp * (1 << psh);
    }
    const bool closed2 = pos4 == w.pos0 && s2 == w.s_close;
    const uint32_t nbl = nb_frame(wa, wm, wb, p, T.NT);
    const uint32_t nb = bit_select(bit_flag<1>(flags), nb_untranspose(nbl), nbl);
    const bool ends = closed1 || closed2;
    w.result = ends ? n1 : w.result;
    w.run = (ends || pos4 < w.pos0) ? 0u : 1u;                            // run interiors lie between their end points in raster order
    w.n = n1;
    w.prev = s2;
    w.pos = pos4;
    w.k2 = s2 + 5u;
    w.nb16 = nb | (nb << 8);
}
// the whole border through walk_start / walk_iter (tests/host/: must equal trace_border_fast)
__device__ inline int trace_border_walk(const uint32_t* fr, int x0, int y0, uint8_t* pts, int cap, uint8_t* spill = nullptr, int spill_cap = 0) {
    IRBPP_WALK_TABLES(T)
    Walk w;
    walk_start(w, fr, x0, y0, pts, cap, true, T);
    for (int guard = 4096; w.run != 0u; )  {
        walk_iter(w, fr, pts, cap, spill, spill_cap, T);
        if (--guard == 0) return -1;
    }
    return w.result;
}

// a contour point is x | y << SHIFT in a PT: a byte with SHIFT = 4 on action grids of up to 16 x 16 cells, 16 bits with
// SHIFT = 5 on the wide grids (up to 32 x 32, round 6)
#define IRBPP_PX(p) ((int)((p) & ((1 << SHIFT) - 1)))
#define IRBPP_PY(p) ((int)((p) >> SHIFT))

// approxPolyDP_<int>(closed, eps=1) (OpenCV approx.cpp) followed by find_convex_vetex.
// pts[0..count) -> vertex bits ORed into vrows[y] (bit x).  Returns false on stack overflow.
template <typename PT, int SHIFT>
__device__ inline bool approx_and_convex_t(const PT* pts, int count, PT* dst, uint32_t* stk,
                                           int cap_stk, uint32_t* vrows) {
    int new_count = 0;
    int top = 0;
    // 1. three farthest-point hops
    int pos = 0, right_start = 0;
    bool le_eps = false;
    PT start_pt = 0;
    for (int it = 0; it < 3; ++it) {
        int max_dist = 0;
        pos += right_start;
        if (pos >= count) pos -= count;
        start_pt = pts[pos];
        if (++pos >= count) pos = 0;
        const int sx = IRBPP_PX(start_pt), sy = IRBPP_PY(start_pt);
        for (int j = 1; j < count; ++j) {
            const PT pt = pts[pos];
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
                const PT end_pt = pts[s_end];
                sx = IRBPP_PX(start_pt);
                sy = IRBPP_PY(start_pt);
                dx = IRBPP_PX(end_pt) - sx;
                dy = IRBPP_PY(end_pt) - sy;
                max_dist = 0;
                in_slice = true;
            }
            const PT pt = pts[pos];
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
        PT pt = dst[pos];
        if (++pos >= cnt) pos = 0;
        for (int i = 0; i < cnt && new_count > 2; ++i) {
            const PT end_pt = dst[pos];
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
        PT a = dst[m - 1], b = dst[0];
        for (int i = 0; i < m; ++i) {
            const PT c = dst[i == m - 1 ? 0 : i + 1];
            const int cross = (IRBPP_PX(b) - IRBPP_PX(a)) * (IRBPP_PY(c) - IRBPP_PY(a)) -
                              (IRBPP_PY(b) - IRBPP_PY(a)) * (IRBPP_PX(c) - IRBPP_PX(a));
            if (cross < 0) atomicOr(&vrows[IRBPP_PY(b)], 1u << IRBPP_PX(b));
            a = b;
            b = c;
        }
    }
    return true;
}

#undef IRBPP_PX
#undef IRBPP_PY
#define IRBPP_PX(p) ((int)((p) & 15))
#define IRBPP_PY(p) ((int)((p) >> 4))
__device__ inline bool approx_and_convex(const uint8_t* pts, int count, uint8_t* dst, uint32_t* stk,
                                         int cap_stk, uint32_t* vrows) {
    return approx_and_convex_t<uint8_t, 4>(pts, count, dst, stk, cap_stk, vrows);
}

// ---------------------------------------------------------------------------------------
// Isolated solid rectangles (round 6).  Of the ~25 borders per bin that lattice data hands to the trace kernel, 38 % belong to a
// component that is nothing but a solid w x h rectangle with an empty ring around it (tools/rect_component_study.py).  Its border
// is its four corners (CHAIN_APPROX_SIMPLE), and what approxPolyDP(eps = 1) + find_convex_vetex leave of them depends on (w, h)
// alone: the first corner only for the two-pixel components (the second point lies within eps), the first and the opposite corner
// when the thin side is two pixels (the other two lie within eps of the diagonal), all four otherwise -- which for a one-pixel
// line are its two ends.  (tests/test_device_contours_on_host.py checks the rule against the routines above and the oracle for
// every size and position.)  So the transition kernel CAN answer such a candidate start itself, as it does isolated pixels
// (IRBPP_TUNE_RECT).  Measured at 8192 BlockOut bins: trace kernel 36.1 -> 31.1 us, polygon 24.0 -> 22.2, transition 63.2 -> 69.6:
// the divergent per-candidate loop costs the transition kernel what the other two save -- opt-in, parity-tested.
// rect_component: is the component that starts at candidate (x0, y) of `row` -- a start_candidates bit: nothing above the run
// touches it -- the solid rectangle [x0, x0 + w) x [y, y + h) with no other pixel next to it?  rows = the image's 16 row words.
template <typename Rows>
__device__ __forceinline__ bool rect_component(const Rows rows, uint32_t row, int x0, int y, int& w, int& h) {
    w = __builtin_ctz(~(row >> x0));                                       // the run that starts at x0
    const uint32_t m = ((1u << w) - 1u) << x0, mb = (m | (m << 1) | (m >> 1)) & 0xFFFFu;
    h = 1;
    uint32_t nx = 0u;
    for (; y + h < 16; ++h) {
        nx = (uint32_t)rows[y + h] & mb;
        if (nx != m) break;
    }
    return y + h == 16 || nx == 0u;                                        // the row below the last one: clear over the run and beside it
}
// the vertex bits of that rectangle: `top` for row y, `bottom` for row y + h - 1 (the same row when h == 1)
__device__ __forceinline__ void rect_vertices(int w, int h, int x0, uint32_t& top, uint32_t& bottom) {
    const uint32_t first = 1u << x0, last = 1u << (x0 + w - 1);
    if (w * h == 2) { top = first; bottom = 0u; }
    else if (w == 2 || h == 2) { top = first; bottom = last; }
    else { top = first | last; bottom = first | last; }
}

// ---------------------------------------------------------------------------------------
// WIDE action grids (17 .. 32 cells a side: resolutionA = 0.01 on the 0.32 m bin, space.py:19-24; round 6): level images of up
// to 32 rows of 32 bits, contour points of 16 bits (x | y << 5).  Plain statements of the same routines -- candidate starts,
// icvFetchContourEx with CHAIN_APPROX_SIMPLE for the OUTER border that starts at a component's first pixel in raster order,
// approxPolyDP + find_convex_vetex (the template above) -- with no run jumps, tables or frames: a capacity path, lane-serial,
// one border per lane (irbpp_kernels.hip: wide_observe).  tests/host/ runs them against the oracle on 32 x 32 images.
// ---------------------------------------------------------------------------------------
// first pixels of runs with nothing above the run (see start_candidates): `row`, `up` = the row and the one above, `wmask` = the
// image's columns
__device__ __forceinline__ uint32_t start_candidates_wide(uint32_t row, uint32_t up, uint32_t wmask) {
    const uint32_t upm = (up | (up << 1) | (up >> 1)) & wmask;
    const uint32_t first = row & ~(row << 1) & ~upm;
    uint32_t t = upm & row;                                             // pixels touched from above, spread to the first pixel of their run
    const uint32_t p2 = row & (row >> 1), p4 = p2 & (p2 >> 2), p8 = p4 & (p4 >> 4), p16 = p8 & (p8 >> 8);
    t |= (t >> 1) & row;
    t |= (t >> 2) & p2;
    t |= (t >> 4) & p4;
    t |= (t >> 8) & p8;
    t |= (t >> 16) & p16;
    return first & ~t;
}
// 8-bit neighbour mask of pixel (x, y): bit s = the neighbour in direction s (0 E, 1 NE, 2 N, 3 NW, 4 W, 5 SW, 6 S, 7 SE; y down)
__device__ __forceinline__ uint32_t nb_mask_wide(const uint32_t* rows, int H, int x, int y) {
    const unsigned long long ra = y > 0 ? rows[y - 1] : 0u, rb = rows[y], rc = y + 1 < H ? rows[y + 1] : 0u;
    const uint32_t ta = (uint32_t)(((ra << 1) >> x) & 7ull), tb = (uint32_t)(((rb << 1) >> x) & 7ull), tc = (uint32_t)(((rc << 1) >> x) & 7ull);
    // bit 0 of a triple = column x - 1, bit 1 = x, bit 2 = x + 1
    return (tb >> 2) | ((ta >> 2) << 1) | (((ta >> 1) & 1u) << 2) | ((ta & 1u) << 3) | ((tb & 1u) << 4) | ((tc & 1u) << 5) |
           (((tc >> 1) & 1u) << 6) | ((tc >> 2) << 7);
}
// Returns the number of points (all counted, the first `cap` stored), 0 if (x0, y0) is not the first pixel of its component,
// -1 if the iteration guard tripped.  (`rows` holds no bit beyond column W - 1 or row H - 1.)
template <int SHIFT>
__device__ inline int trace_border_wide(const uint32_t* rows, int W, int H, int x0, int y0, uint16_t* pts, int cap) {
    uint32_t nb = nb_mask_wide(rows, H, x0, y0);
    // first non-zero neighbour, clockwise from west (background): 3, 2, 1, 0, 7, 6, 5
    const uint32_t rot = ((nb << 4) | (nb >> 4)) & 0xFFu;                  // direction 3 -> bit 7
    if (rot == 0u) {                                                       // isolated pixel
        if (cap > 0) pts[0] = (uint16_t)(x0 | (y0 << SHIFT));
        return 1;
    }
    int s = (3 - (7 - (31 - __builtin_clz(rot)))) & 7;
    const int x1 = x0 + dir_dx(s), y1 = y0 + dir_dy(s);
    int x3 = x0, y3 = y0, prev_s = s ^ 4, n = 0;
    for (int guard = 0; guard < 8192; ++guard) {
        IRBPP_TRACE_ITER();
        // counter-clockwise search s + 1, s + 2, ... for the next border pixel (the one we came from is set: found within eight)
        const int k2 = (s + 1) & 7;
        const uint32_t r2 = ((nb >> k2) | (nb << (8 - k2))) & 0xFFu;       // direction k2 -> bit 0
        s = (k2 + __builtin_ctz(r2)) & 7;
        const int x4 = x3 + dir_dx(s), y4 = y3 + dir_dy(s);
        if (s != prev_s) {                                                 // CHAIN_APPROX_SIMPLE
            if (n < cap) pts[n] = (uint16_t)(x3 | (y3 << SHIFT));
            ++n;
        }
        prev_s = s;
        if (x4 == x0 && y4 == y0 && x3 == x1 && y3 == y1) return n;
        if (y4 * W + x4 < y0 * W + x0) return 0;                           // an earlier pixel of the component: some other start's border
        x3 = x4;
        y3 = y4;
        nb = nb_mask_wide(rows, H, x3, y3);
        s = (s + 4) & 7;
    }
    return -1;
}

// The wide walk with RUN JUMPS (trace_step's logic on 32-bit lines): after a step in an axis direction the walk keeps going straight
// while the three neighbours probed before that direction are background and the next pixel is foreground -- one bit scan on the
// line walked and the line beside it (rows for E / W, the transposed copy `cols` for N / S; bit y of cols[x] = pixel (x, y)).
// Same points as trace_border_wide (tests/host/: both against the oracle, and against each other).
__device__ __forceinline__ uint32_t nb_mask3_wide(uint32_t a, uint32_t b, uint32_t c, int x) {
    const unsigned long long ra = a, rb = b, rc = c;
    const uint32_t ta = (uint32_t)(((ra << 1) >> x) & 7ull), tb = (uint32_t)(((rb << 1) >> x) & 7ull), tc = (uint32_t)(((rc << 1) >> x) & 7ull);
    return (tb >> 2) | ((ta >> 2) << 1) | (((ta >> 1) & 1u) << 2) | ((ta & 1u) << 3) | ((tb & 1u) << 4) | ((tc & 1u) << 5) |
           (((tc >> 1) & 1u) << 6) | ((tc >> 2) << 7);
}
__device__ __forceinline__ int run_forward_wide(uint32_t line, uint32_t side, int p) {      // towards higher bits
    const uint32_t clear = ~(side | (side << 1) | (side >> 1));
    const uint32_t m = (clear & (line >> 1)) >> p;
    return __builtin_ctz(~m);                                                               // consecutive ones from bit p (bit 31 of line >> 1 is clear)
}
__device__ __forceinline__ int run_backward_wide(uint32_t line, uint32_t side, int p) {     // towards lower bits
    const uint32_t clear = ~(side | (side << 1) | (side >> 1));
    const uint32_t m = clear & (line << 1);
    const uint32_t z = ~m & ((2u << p) - 1u);                                               // zero bits at or below p
    return z ? p - (31 - __builtin_clz(z)) : p + 1;
}
template <int SHIFT>
__device__ inline int trace_border_wide_runs(const uint32_t* rows, const uint32_t* cols, int W, int H, int x0, int y0, uint16_t* pts, int cap) {
    uint32_t nb = nb_mask3_wide(y0 > 0 ? rows[y0 - 1] : 0u, rows[y0], y0 + 1 < H ? rows[y0 + 1] : 0u, x0);
    const uint32_t rot = ((nb << 4) | (nb >> 4)) & 0xFFu;                  // clockwise from west: direction 3 -> bit 7
    if (rot == 0u) {                                                       // isolated pixel
        if (cap > 0) pts[0] = (uint16_t)(x0 | (y0 << SHIFT));
        return 1;
    }
    const int s_first = (3 - (7 - (31 - __builtin_clz(rot)))) & 7;
    const int x1 = x0 + dir_dx(s_first), y1 = y0 + dir_dy(s_first);
    int x3 = x0, y3 = y0, cur_s = s_first, prev_s = s_first ^ 4, n = 0;
    for (int guard = 0; guard < 8192; ++guard) {
        IRBPP_TRACE_ITER();
        const int k2 = (cur_s + 1) & 7;
        const uint32_t r2 = ((nb >> k2) | (nb << (8 - k2))) & 0xFFu;       // direction k2 -> bit 0
        const int s2 = (k2 + __builtin_ctz(r2)) & 7;
        int x4 = x3 + dir_dx(s2), y4 = y3 + dir_dy(s2);
        if (s2 != prev_s) {                                                // CHAIN_APPROX_SIMPLE
            if (n < cap) pts[n] = (uint16_t)(x3 | (y3 << SHIFT));
            ++n;
        }
        prev_s = s2;
        if (x4 == x0 && y4 == y0 && x3 == x1 && y3 == y1) return n;
        const bool vert = (s2 & 3) == 2, axis = (s2 & 1) == 0, fwd = s2 == 0 || s2 == 6;
        const uint32_t* base = vert ? cols : rows;
        const int nl = vert ? W : H;                                       // lines of that frame
        const int li = vert ? x4 : y4;
        int p = vert ? y4 : x4;
        const uint32_t wm = base[li], wa = li > 0 ? base[li - 1] : 0u, wb = li + 1 < nl ? base[li + 1] : 0u;
        if (axis) {
            const uint32_t side = (vert != fwd) ? wb : wa;
            p += fwd ? run_forward_wide(wm, side, p) : -run_backward_wide(wm, side, p);
        }
        x4 = vert ? x4 : p;
        y4 = vert ? p : y4;
        if (x4 == x0 && y4 == y0 && s2 == (s_first ^ 4)) return n;         // a run that ends on the start, against the first step
        if (y4 * W + x4 < y0 * W + x0) return 0;                           // an earlier pixel of the component
        const uint32_t nbl = nb_mask3_wide(wa, wm, wb, p);
        x3 = x4;
        y3 = y4;
        cur_s = (s2 + 4) & 7;
        nb = vert ? nb_untranspose(nbl) : nbl;
    }
    return -1;
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
// approxPolyDP(eps = 1, closed) + find_convex_vetex for SEVERAL borders at once, one contour point
// per lane ("lane = point"): a wave holds up to 64 points of borders packed back to back.  Same
// results as approx_and_convex(), but nothing in it is sequential per border:
//   * the three farthest-point hops are three segmented arg-max rounds (segment = border);
//   * Douglas-Peucker runs level by level: every lane knows the slice (start, end) it is an interior
//     point of, all slices of all borders find their farthest point in ONE segmented arg-max round
//     (segment = slice, keyed by the lane of its start point), split or accept, and the lanes
//     update their slice.  The recursion of approx.cpp emits the start points of accepted slices
//     in traversal order, i.e. the polygon is exactly the set of slice boundaries in contour order
//     starting at the hop phase's start point -- no stack, no output list;
//   * "first strict maximum in traversal order" = max of (dist << 8 | 255 - t);
//   * the clean-up pass of approx.cpp only changes the polygon if some triple of consecutive
//     vertices satisfies its removal test; that is checked for all vertices in parallel and a
//     border where it fires (<1 % of them) is reported back for the sequential routine;
//   * convexity is one cross product per vertex with the neighbouring vertices found by bit scans
//     on the ballot of kept points.
// The segmented arg-max is an LDS ds_max_u32 into one word per segment; a wave's LDS operations
// execute in order, so no barrier is needed between the reset, the max and the read-back.
// ---------------------------------------------------------------------------------------
#ifndef IRBPP_WAVE_SYNC
#define IRBPP_WAVE_SYNC() __builtin_amdgcn_wave_barrier()
#endif

// The clean-up pass of approxPolyDP_ (removal of [almost] collinear vertices, in place as approx.cpp does it,
// including its reads of already rewritten entries after the wrap-around) followed by find_convex_vetex,
// for ONE polygon held in a lane-indexed register: vertex i of `cnt` in lane i.  The pass is sequential
// by nature and wave-uniform here (v_readlane at uniform indices); it only runs for the rare border
// whose polygon it would change.
__device__ __forceinline__ int put_lane(int lane, int reg, int idx, int val) { return lane == idx ? val : reg; }
__device__ inline void cleanup_convex_wave(int lane, int dv, int cnt, uint32_t* vrows) {
    int new_count = cnt;
    int p2 = cnt - 1;
    int start_pt = __builtin_amdgcn_readlane(dv, p2);
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
            dv = put_lane(lane, dv, wpos, end_pt);
            if (++wpos >= cnt) wpos = 0;
            pt = __builtin_amdgcn_readlane(dv, p2);
            if (++p2 >= cnt) p2 = 0;
            ++i;
            continue;
        }
        start_pt = pt;
        dv = put_lane(lane, dv, wpos, pt);
        if (++wpos >= cnt) wpos = 0;
        pt = end_pt;
    }
    const int m = new_count;
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

// The same pass without its loop.  What the sequential pass does, vertex by vertex (v[i] = vertex i of cnt, T(i) = the removal
// test on the triple v[i-1], v[i], v[i+1]): it tests v[0], v[1], ... in turn; a vertex that passes T is removed and the vertex
// behind it is then kept WITHOUT being tested; a kept vertex becomes the next triple's first point.  So every tested triple
// consists of original neighbours -- a vertex is only ever tested when the one before it was kept --, and
//     removed(i) = T(i) and not removed(i-1),   removed(-1) = false:
// inside a run of consecutive vertices that pass T, the first, third, fifth ... are removed.  One exception comes from the
// pass working in place: the last vertex's triple ends on entry 0 of the array AS REWRITTEN, which is v[1] if v[0] was removed.
// The pass also stops when two vertices are left; if this routine arrives at fewer than three it returns false and the
// caller runs the sequential one (then, and only then, the stop can have mattered).  Vertex i in lane i, cnt <= 64.
__device__ inline bool cleanup_convex_parallel(int lane, int dv, int cnt, uint32_t* vrows) {
    const int ip = lane == 0 ? cnt - 1 : lane - 1, in = lane >= cnt - 1 ? 0 : lane + 1;
    const int a = __shfl(dv, ip & 63), c = __shfl(dv, in);
    const int v1 = __builtin_amdgcn_readlane(dv, 1);
    auto removable = [](int s, int p, int e) {
        const int dx = IRBPP_PX(e) - IRBPP_PX(s), dy = IRBPP_PY(e) - IRBPP_PY(s);
        const int ux = IRBPP_PX(p) - IRBPP_PX(s), uy = IRBPP_PY(p) - IRBPP_PY(s);
        int dist = ux * dy - uy * dx;
        dist = dist < 0 ? -dist : dist;
        const int inner = ux * (IRBPP_PX(e) - IRBPP_PX(p)) + uy * (IRBPP_PY(e) - IRBPP_PY(p));
        return 2 * dist * dist <= dx * dx + dy * dy && dx != 0 && dy != 0 && inner >= 0;
    };
    bool T = lane < cnt && removable(a, dv, c);
    unsigned long long tm = __ballot(T);
    if (tm & 1ull) {                                     // (uniform) v[0] goes: the last triple ends on v[1]
        if (lane == cnt - 1) T = removable(a, dv, v1);
        tm = __ballot(T);
    }
    const unsigned long long below = lane == 0 ? 0ull : (~0ull >> (64 - lane));
    const unsigned long long z = ~tm & below;                                       // vertices before me that stay for sure
    const int run_start = z != 0ull ? 64 - __clzll((long long)z) : 0;               // first vertex of my run of T's
    const bool removed = T && ((lane - run_start) & 1) == 0;
    const unsigned long long valid = cnt >= 64 ? ~0ull : ((1ull << cnt) - 1ull);
    const unsigned long long kept = valid & ~__ballot(removed);
    const int m = __popcll(kept);
    if (m < 3) return false;
    // find_convex_vetex on the kept vertices: neighbours = the kept vertices next to me, cyclically
    const unsigned long long lo = kept & below, hi = kept & ~below & ~(1ull << lane);
    const int pi = lo != 0ull ? 63 - __clzll((long long)lo) : 63 - __clzll((long long)kept);
    const int ni = hi != 0ull ? __ffsll((long long)hi) - 1 : __ffsll((long long)kept) - 1;
    const int pa = __shfl(dv, pi), pc = __shfl(dv, ni), pb = dv;
    if ((kept >> lane) & 1ull) {
        bool mark = true;
        if (m > 3)
            mark = (IRBPP_PX(pb) - IRBPP_PX(pa)) * (IRBPP_PY(pc) - IRBPP_PY(pa)) -
                   (IRBPP_PY(pb) - IRBPP_PY(pa)) * (IRBPP_PX(pc) - IRBPP_PX(pa)) < 0;
        if (mark) atomicOr(&vrows[IRBPP_PY(pb)], 1u << IRBPP_PX(pb));
    }
    return true;
}

// The same clean-up + convexity for a polygon of more than 64 vertices (a border of more than 64 points whose
// polygon the clean-up changes: rarest of the rare), by one lane on a byte array it may overwrite.
__device__ inline void cleanup_convex_serial(uint8_t* dst, int cnt, uint32_t* vrows) {
    int new_count = cnt;
    int pos = cnt - 1;
    uint8_t start_pt = dst[pos];
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
}

// The arg-max words take one ds_max_u32 per point; all points of a border (hop rounds) or a slice (Douglas-Peucker rounds) hit
// ONE word and are serialised by the LDS, which made 66 % of the polygon kernel's LDS cycles bank conflicts (profiles/r03/final/
// sq2_summary.json).  MEASURED AND NOT TAKEN (profiles/r04/s10, a build that pre-reduced the keys of a segment inside each row
// of 16 lanes on the DPP network and sent one key per run to the word): the conflict share fell to 15 % and the LDS cycles by
// 58 %, and the kernel got SLOWER, 24.1 -> 27.9 us on BlockOut and 26.9 -> 33.3 us on "general": it is bound by its vector
// instructions (4.4 cycles each per SIMD), and the ~21 added per point set and round cost more than the serialised atomics did
// -- the LDS takes those in its stride.  (The hook of that build was removed in round 5 with the straight-line rounds.)

// Set bits of a ballot below this lane (v_mbcnt_lo / v_mbcnt_hi take the mask as a scalar pair: two instructions).
__device__ __forceinline__ int wave_count_below(unsigned long long mask, int lane) {
#if defined(__HIP_DEVICE_COMPILE__)
    (void)lane;
    return (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
#else
    return __popcll(mask & (lane == 0 ? 0ull : (~0ull >> (64 - lane))));
#endif
}

// A wave serves 64 * P contour points per round: position q = u * 64 + lane, u < P, borders packed back to
// back over the positions (a border of up to 64 * P points fits).  Per position:
//   live: a point sits here;  pv: the point (x | y<<4);  j, n: its index in / the size of its border;
//   sb: position of its border's point 0;  pts: its border's point list in LDS;  rot: row block of its
//   border's vertex bits in `vmask`.
// slots: 64 * P words of LDS private to this wave;  scratch: 64 * P bytes of LDS private to this wave.
template <int P>
__device__ inline void approx_convex_segmented(int lane, const bool (&live)[P], const int (&pv)[P], const int (&j)[P],
                                               const int (&n)[P], const int (&sb)[P], const uint8_t* const (&pts)[P],
                                               const int (&rot)[P], uint32_t* slots, uint8_t* scratch, uint32_t* vmask
#ifdef IRBPP_AB_POLY_ACCOUNT
                                               , long long* acct = nullptr
#endif
                                               ) {
#ifdef IRBPP_AB_POLY_ACCOUNT
#define IRBPP_POLY_STAMP(k) if (acct) acct[k] = (long long)clock64()
#define IRBPP_POLY_COUNT(k) if (acct) acct[k] += 1
#else
#define IRBPP_POLY_STAMP(k)
#define IRBPP_POLY_COUNT(k)
#endif
    int px[P], py[P];
#pragma unroll
    for (int u = 0; u < P; ++u) { px[u] = IRBPP_PX(pv[u]); py[u] = IRBPP_PY(pv[u]); }
    // 1. three farthest-point hops.  The arg-max key carries the winner's coordinates in its low byte
    // (dist << 16 | 255 - t << 8 | x | y << 4: t is unique inside a segment, so the byte never decides), which
    // makes the winner's point known to every lane without another read: the next hop starts there.
    int pos[P], right_start[P], sxy[P], fxy[P];
    bool le_eps[P];
#pragma unroll
    for (int u = 0; u < P; ++u) {
        pos[u] = 0; right_start[u] = 0; le_eps[u] = false;
        sxy[u] = live[u] ? (int)pts[u][0] : 0;
        fxy[u] = sxy[u];
    }
    // (straight-line code like the Douglas-Peucker level below: a position without a bid -- no point, or the hop's start point
    // itself -- sends 0, which changes no word; a position without a point has sb = 0, n = 1 and carries values nobody reads)
    for (int it = 0; it < 3; ++it) {
#pragma unroll
        for (int u = 0; u < P; ++u) slots[u * 64 + lane] = 0u;
        IRBPP_WAVE_SYNC();
#pragma unroll
        for (int u = 0; u < P; ++u) {
            int t = j[u] - pos[u];
            t += t < 0 ? n[u] : 0;
            const int dx = px[u] - IRBPP_PX(sxy[u]), dy = py[u] - IRBPP_PY(sxy[u]);
            const uint32_t key = ((uint32_t)(dx * dx + dy * dy) << 16) | ((uint32_t)(255 - t) << 8) | (uint32_t)pv[u];
            const bool bids = live[u] && t >= 1;
            atomicMax(&slots[bids ? sb[u] : u * 64 + lane], bids ? key : 0u);      // (no bid: 0 at the position's OWN word -- the empty
                                                                                  //  positions of a round all sending to word 0 were serialised there)
        }
        IRBPP_WAVE_SYNC();
        uint32_t best[P];
#pragma unroll
        for (int u = 0; u < P; ++u) best[u] = slots[sb[u]];
        IRBPP_WAVE_SYNC();
#pragma unroll
        for (int u = 0; u < P; ++u) {
            const int max_dist = (int)(best[u] >> 16);
            const bool found = max_dist > 0;
            right_start[u] = found ? 255 - (int)((best[u] >> 8) & 255u) : right_start[u];
            fxy[u] = found ? (int)(best[u] & 255u) : fxy[u];
            le_eps[u] = max_dist <= 1;
            if (it < 2) {                                    // (uniform) the next hop starts at the farthest point found
                int p2 = pos[u] + right_start[u];
                p2 -= p2 >= n[u] ? n[u] : 0;
                pos[u] = found ? p2 : pos[u];
                sxy[u] = found ? fxy[u] : sxy[u];
            }
        }
    }
    IRBPP_POLY_STAMP(1);
    // 2. Douglas-Peucker, all slices of one recursion level per round; every lane keeps the end points of its
    // slice in registers, the split point's coordinates arrive with the arg-max.
    // A position's whole state is four words: seg >= 0: it is an interior point of the slice whose arg-max word is slots[seg]
    // (seg = position of the slice's start point), -1: dropped, -2: kept; axy / bxy: the slice's end points; tk: the low half
    // of its arg-max key, (255 - t) << 8 | point with t = its distance from the slice's start along the border.  The level
    // is STRAIGHT-LINE code for every position -- a masked-off vector instruction costs the issue slot it saves nothing of, and
    // the kernel is bound by issue --: a position that is not active bids 0 (which changes no word) at its own word, and its
    // state passes through selects.  (Until round 5 session 33 the level was three `if (active)` regions per position with
    // two flags carried in vector registers: 112 vector instructions per level of two positions, a third of them moves and
    // flag tests.)
    int s0[P], seg[P], axy[P], bxy[P];
    uint32_t tk[P];
    int any_seg = -1;                                               // sign bit clear iff some position is active
#pragma unroll
    for (int u = 0; u < P; ++u) {
        s0[u] = pos[u];
        int far = right_start[u] + s0[u];
        if (far >= n[u]) far -= n[u];
        int t0 = j[u] - s0[u], len_a = far - s0[u];
        if (t0 < 0) t0 += n[u];
        if (len_a < 0) len_a += n[u];
        const bool keep0 = live[u] && (le_eps[u] ? j[u] == s0[u] : (j[u] == s0[u] || j[u] == far));
        const bool act0 = live[u] && !le_eps[u] && !keep0;
        const bool first = t0 < len_a;                              // my slice: (s0, far) or (far, s0)
        axy[u] = first ? sxy[u] : fxy[u];
        bxy[u] = first ? fxy[u] : sxy[u];
        tk[u] = ((uint32_t)(255 - (first ? t0 : t0 - len_a)) << 8) | (uint32_t)pv[u];
        seg[u] = act0 ? sb[u] + (first ? s0[u] : far) : (keep0 ? -2 : -1);
        any_seg &= seg[u];
    }
    while (__ballot(any_seg >= 0) != 0ull) {
        IRBPP_POLY_COUNT(5);
#pragma unroll
        for (int u = 0; u < P; ++u) slots[u * 64 + lane] = 0u;
        IRBPP_WAVE_SYNC();
        int dx[P], dy[P], word[P];
#pragma unroll
        for (int u = 0; u < P; ++u) {
            const int ax = IRBPP_PX(axy[u]), ay = IRBPP_PY(axy[u]);
            dx[u] = IRBPP_PX(bxy[u]) - ax;
            dy[u] = IRBPP_PY(bxy[u]) - ay;
            int dist = (py[u] - ay) * dx[u] + (ax - px[u]) * dy[u];
            dist = dist < 0 ? -dist : dist;
            const bool act = seg[u] >= 0;
            word[u] = act ? seg[u] : u * 64 + lane;
            atomicMax(&slots[word[u]], act ? (((uint32_t)dist << 16) | tk[u]) : 0u);
        }
        IRBPP_WAVE_SYNC();
        uint32_t best[P];
#pragma unroll
        for (int u = 0; u < P; ++u) best[u] = slots[word[u]];
        IRBPP_WAVE_SYNC();
        any_seg = -1;
#pragma unroll
        for (int u = 0; u < P; ++u) {
            const bool act = seg[u] >= 0;
            const int md = (int)(best[u] >> 16);
            const bool accept = md * md <= dx[u] * dx[u] + dy[u] * dy[u];      // slice accepted: its interior points are dropped
            const uint32_t bt = best[u] & 0xFF00u, mt = tk[u] & 0xFF00u;       // 255 - t of the split point / of me
            const bool split = act && !accept;
            const bool right = split && mt < bt;                              // I lie behind the split point: my slice starts there now
            const bool left = split && mt > bt;                               //   before it: my slice ends there
            const int np = (int)(best[u] & 255u);
            const int ts = 255 - (int)(bt >> 8);
            int s2 = seg[u] + ts;
            s2 -= s2 >= sb[u] + n[u] ? n[u] : 0;
            axy[u] = right ? np : axy[u];
            bxy[u] = left ? np : bxy[u];
            tk[u] = right ? tk[u] + ((uint32_t)ts << 8) : tk[u];
            seg[u] = !act ? seg[u] : accept ? -1 : right ? s2 : left ? seg[u] : -2;      // (neither side: I am the split point, kept)
            any_seg &= seg[u];
        }
    }
    bool keep[P];
#pragma unroll
    for (int u = 0; u < P; ++u) keep[u] = seg[u] == -2;
    IRBPP_POLY_STAMP(2);
    // 3. the polygon = kept points in contour order: every kept point files its index at its rank among the
    // kept points of its border (bit counts on the ballots), neighbours are the entries next to it
    unsigned long long kept[P];
#pragma unroll
    for (int u = 0; u < P; ++u) kept[u] = __ballot(keep[u]);
    // kept points at positions below mine = bit counts on the ballots; every position files that count, and a point's rank
    // among the kept points of its border (and the border's polygon size) are differences to the counts filed at the
    // border's first position and behind its last one.  (Until round 5 session 32 every lane counted the bits of its
    // border's range in every ballot word: ~60 vector instructions per position of 64-bit mask arithmetic.)
    int below[P], total = 0;
#pragma unroll
    for (int u = 0; u < P; ++u) {
        below[u] = total + wave_count_below(kept[u], lane);
        total += __popcll(kept[u]);
        slots[u * 64 + lane] = (uint32_t)below[u];
    }
    IRBPP_WAVE_SYNC();
    int m[P], rank[P];
#pragma unroll
    for (int u = 0; u < P; ++u) {
        const int first = (int)slots[sb[u]], e = sb[u] + n[u];
        const int behind = e < 64 * P ? (int)slots[e < 64 * P ? e : 0] : total;
        m[u] = behind - first;
        rank[u] = below[u] - first;
    }
    IRBPP_WAVE_SYNC();
#pragma unroll
    for (int u = 0; u < P; ++u)
        if (keep[u]) slots[sb[u] + rank[u]] = (uint32_t)j[u] | ((uint32_t)pv[u] << 16);   // index and point
    IRBPP_WAVE_SYNC();
    int ea[P], ec[P];
#pragma unroll
    for (int u = 0; u < P; ++u) {
        ea[u] = ec[u] = 0;
        if (keep[u]) {
            ea[u] = (int)slots[sb[u] + (rank[u] == 0 ? m[u] - 1 : rank[u] - 1)];
            ec[u] = (int)slots[sb[u] + (rank[u] == m[u] - 1 ? 0 : rank[u] + 1)];
        }
    }
    bool redo[P], mark[P];
    bool any_redo = false;
#pragma unroll
    for (int u = 0; u < P; ++u) {
        redo[u] = mark[u] = false;
        if (keep[u]) {
            const int pa = ea[u] >> 16, pc = ec[u] >> 16;
            const int ax = IRBPP_PX(pa), ay = IRBPP_PY(pa), cx = IRBPP_PX(pc), cy = IRBPP_PY(pc);
            if (m[u] > 2) {                              // removal test of the clean-up pass (start = A, pt = me, end = C)
                const int ddx = cx - ax, ddy = cy - ay, ux = px[u] - ax, uy = py[u] - ay;
                int dist = ux * ddy - uy * ddx;
                dist = dist < 0 ? -dist : dist;
                const int inner = ux * (cx - px[u]) + uy * (cy - py[u]);
                redo[u] = 2 * dist * dist <= ddx * ddx + ddy * ddy && ddx != 0 && ddy != 0 && inner >= 0;
            }
            mark[u] = m[u] <= 3 || (px[u] - ax) * (cy - ay) - (py[u] - ay) * (cx - ax) < 0;       // find_convex_vetex
        }
        any_redo |= redo[u];
    }
    // a border with a removable vertex tells all its points through its first slot word ... which holds the
    // polygon list; use the scratch bytes instead: one flag byte per border at its first position
    IRBPP_POLY_STAMP(3);
    if (__ballot(any_redo) == 0ull) {                    // the common case: no clean-up anywhere in the wave
#pragma unroll
        for (int u = 0; u < P; ++u)
            if (mark[u]) atomicOr(&vmask[rot[u] * 16 + py[u]], 1u << px[u]);
        return;
    }
#pragma unroll
    for (int u = 0; u < P; ++u) scratch[u * 64 + lane] = 0;
    IRBPP_WAVE_SYNC();
#pragma unroll
    for (int u = 0; u < P; ++u)
        if (redo[u]) scratch[sb[u]] = 1;
    IRBPP_WAVE_SYNC();
    bool flagged[P];
#pragma unroll
    for (int u = 0; u < P; ++u) {
        flagged[u] = live[u] && scratch[sb[u]] != 0;
        if (mark[u] && !flagged[u]) atomicOr(&vmask[rot[u] * 16 + py[u]], 1u << px[u]);
    }
    IRBPP_WAVE_SYNC();
    // borders whose polygon the clean-up pass changes (<1 %): one at a time, the polygon in its own order (it
    // starts at the hop phase's start point) in a lane-indexed register, or serially if it has > 64 vertices
#pragma unroll
    for (int u = 0; u < P; ++u) {
        unsigned long long pending = __ballot(flagged[u] && j[u] == 0);       // first points of flagged borders in word u
        while (pending != 0ull) {
            const int l0 = __ffsll((long long)pending) - 1;
            pending &= pending - 1ull;
            IRBPP_POLY_COUNT(6);
            const int sb0 = __builtin_amdgcn_readlane(sb[u], l0), cnt = __builtin_amdgcn_readlane(m[u], l0);
            const int s00 = __builtin_amdgcn_readlane(s0[u], l0), rot0 = __builtin_amdgcn_readlane(rot[u], l0);
            // rank of the start point s0 in the polygon list: the list is sorted by index
            int r0 = 0;
            if (cnt <= 64) {
                const bool before = lane < cnt && (int)(slots[sb0 + (lane < cnt ? lane : 0)] & 0xFFFFu) < s00;
                r0 = __popcll(__ballot(before));
                int k = lane + r0;
                if (k >= cnt) k -= cnt;
                const int dv = lane < cnt ? (int)(slots[sb0 + k] >> 16) : 0;
                if (!cleanup_convex_parallel(lane, dv, cnt, vmask + rot0 * 16))
                    cleanup_convex_wave(lane, dv, cnt, vmask + rot0 * 16);
            } else {
                for (int i = 0; i < cnt; ++i) r0 += (int)(slots[sb0 + i] & 0xFFFFu) < s00 ? 1 : 0;
                for (int i = lane; i < cnt; i += 64) {
                    int k = i + r0;
                    if (k >= cnt) k -= cnt;
                    scratch[i] = (uint8_t)(slots[sb0 + k] >> 16);
                }
                IRBPP_WAVE_SYNC();
                if (lane == 0) cleanup_convex_serial(scratch, cnt, vmask + rot0 * 16);
                IRBPP_WAVE_SYNC();
            }
        }
    }
}

}  // namespace irbpp
