// irbpp_device.h -- device-visible data layout of the batched packing environment.
//
// HBM layout (N = bins on this device, Hc = Hx*Hy, K = bufferSize, S = selectedAction):
//   hm        f64 [N][Hc]        Space.heightmapC of every bin (space.py:26), contiguous per bin
//   queue     i32 [N][K]         ItemCreator.item_list (IRcreator.py:6-24), always full
//   cand      u32 [N][S]         (rot<<16 | lx<<8 | ly) of the candidate rows handed out last
//   bs        BinState [N]       64-byte line of per-bin scalars (cursor, episode, cur_item, nvalid,
//                                order_action, item_idx, ep_len, ratio_acc, ep_reward); totals f64 [N][4]
//   ShapeRot  104 B per (shape, rot) + compact lists of the masked-in footprint cells, of uniform b x b blocks and
//                                of the generic path's positions (read-only, shared by all bins -> L2 / MALL resident)
//   seq       i32 [n_traj][L]    pre-drawn item ids
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace irbpp {

// Contour stage: every thread of the 256-thread workgroup owns CONTOUR_IPT (image, row) pairs, i.e. a
// batch holds CONTOUR_IPT * 16 level images of 16x16 pixels (row words + column words, 16 bit each).
#ifndef IRBPP_CONTOUR_IPT
#define IRBPP_CONTOUR_IPT 2
#endif
constexpr int CONTOUR_IPT = IRBPP_CONTOUR_IPT;

// Split pipeline: the transition kernel stops after the overlap test and hands the contour work of a bin on through
// global memory -- its level images (at most R * 64) and the candidate start pixels of their outer borders (at most
// R * AC), sized for the worst case.  The trace kernel serves the candidates of all bins as one flat list per XCD,
// which is what keeps its lanes busy: one bin alone has ~20 borders to follow.
constexpr int WMETA = 8;
constexpr int NXCD = 8;                            // accelerator dies of the MI355X: one flat candidate list each
constexpr int XCD_STRIDE = 64;                     // ints between the lists' counters: a 256-byte line each
#ifndef IRBPP_TRACE_P
#define IRBPP_TRACE_P 2                            // contour points per lane of a polygon round (A/B: 4 measured in profiles/r04)
#endif
constexpr int ROUND_POINTS = 64 * IRBPP_TRACE_P, ROUND_BYTES = ROUND_POINTS * 7;   // a round record of the polygon kernel: [points | border length | border
                                                                     // start] per position, then the vertex-row index of the position's border
constexpr int MAX_BINS = 1 << 20;                 // bins per device (irbpp_create refuses more): keeps every per-launch index an int32

struct ShapeRot {
    int32_t fx, fy;        // footprint in heightmap cells: ceil(round(extents,6)/resH)  (space.py:105)
    int32_t ax, ay;        // footprint in action cells:    ceil(round(extents,6)/resA)  (space.py:106)
    int32_t nb, nt;        // number of maskB==1 / maskH==1 cells
    int32_t ob, ot;        // offsets of this (shape, rot) in the bottom / top cell lists
    int32_t has_out;       // 1 iff some maskB==0 cell exists: the window max then includes (H-B)*0 = 0
    int32_t pad;
    int32_t nblk, oblk;    // block list (Params.block_b > 0): uniform-bottom b x b tiles of the footprint
    int32_t bx, by;        // box path (Params.box): maskB is the solid rectangle [0,bx) x [0,by) with one bottom height bc
    double bc;
    double ext_x, ext_y, ext_z;   // raw mesh.extents (prejudge, simulateHeight)
    double ext_z_r;               // round(extents,6)[2] (space.py:104,120)
    double com_x, com_y;          // centre of mass of the column solid between the two tables, in heightmap cells from the
                                  // footprint's corner (stability proxy only)
};

// Footprint tables are stored as compact lists of the masked-in cells only.  A masked-out cell
// contributes (H-B)*0 = +-0 to np.max at space.py:118-119, i.e. the constant 0 (has_out), and
// (T+z)*0 = +-0 to np.maximum at space.py:213, a no-op on a non-negative heightmap.
struct Cell {              // 16 B
    double v;              // heightMapB / heightMapT value
    int32_t ij;            // i | j << 16: the cell's row and column in its [fx][fy] table
    int32_t pad;           // row-major index i*fy+j of the cell in its [fx][fy] table
};

// Generic overlap test: the masked-in bottom cells of a footprint once more, in the order of the bottom list (entry e of
// a (shape, rotation) is cell bcell[ob + e]), as what the inner loop consumes with ONE scalar load: the cell's heightMapB
// and the BYTE offset of heightmap cell (i, j) relative to an action cell's own entry in the LDS tile.
struct alignas(16) GCell { double b; int32_t off; int32_t pad; };

struct Tables {
    const ShapeRot* sr;    // [n_shapes][R]
    const Cell* bcell;     // bottom cells (maskB == 1)
    const Cell* tcell;     // top cells    (maskH == 1)
    const Cell* blkcell;   // bottom tiles (block path): v = the tile's heightMapB, ij = offset into the block-max grid
    const GCell* gcell;    // generic path: [same indexing as bcell] bottom height + LDS byte offset of every masked-in cell
    const double* volume;  // [n_shapes]
    const int32_t* seq;    // [n_traj][seq_len]
    int32_t n_shapes, n_traj, seq_len;
    int32_t stream;        // irbpp_config::item_stream: rows of seq are per-bin item rings instead of trajectories
};

// Per-bin scalars, one 64-byte line per bin: a transition touches exactly one line of it.
struct alignas(64) BinState {
    int32_t cursor;        // next index into the bin's trajectory (LoadItemCreator.item_index)
    int32_t episode;       // episodes started by this bin minus one
    int32_t cur_item;      // item the last location observation was built for (next_item_ID)
    int32_t nvalid;        // np.sum(naiveMask) of that observation (prejudge, binPhy.py:243)
    int32_t order_action;  // buffer slot chosen by get_action_candidates (binPhy.py:168)
    int32_t item_idx;      // items packed in this episode (info['counter'])
    int32_t ep_len;        // steps in this episode (Monitor 'l')
    int32_t traj_row;      // row of the current episode's trajectory in seq: (traj_start + g + e*G) mod n_traj
    double ratio_acc;      // sequential sum of packed volumes (get_ratio, binPhy.py:149-153)
    double ep_reward;      // sequential sum of rewards (Monitor 'r')
    int32_t nrows;         // candidate rows of the last location observation (entries of `cand` that are current)
    int32_t pad0;
    double pad1[1];
};

struct State {
    double* hm;            // [N][Hc]
    int32_t* queue;        // [N][K]
    uint32_t* cand;        // [N][S]
    BinState* bs;          // [N]
    double* totals;        // [N][4]: episodes, sum ratio, sum counter, sum reward
    uint32_t* log_meta;    // optional [N][log_cap]: item | rot<<16 | lx<<20 | ly<<24 of the episode's placements
    double* log_z;         // optional [N][log_cap]: drop height of each placement
    int32_t log_cap;
    int32_t* order;        // [N] launch order of the bins (irbpp_item_order_kernel); identity until it has run
    int32_t* err;          // [1] device error word
    // split pipeline: per-bin hand-over between the transition, trace and emit kernels (L2 / Infinity Cache resident)
    double* w_posz;        // [N][R*AC] posZmap of the observed item, written only where naiveMask is set (w_valid says where)
    uint32_t* w_valid;     // [N][R*16] naiveMask of the observed item as bit rows: bit Y of word r*16 + X
    uint32_t* w_vmask;     // [N][R*16] vertex bits: isolated pixels from the transition kernel, the rest from the trace kernel
    int32_t* w_meta;       // [N][WMETA]: level images handed over, candidates handed over, np.sum(naiveMask), observed item
    uint16_t* w_img;       // [N][wimg][16] level images: 16 row words (bit x of word y = pixel (x, y))
    uint8_t* w_imgrot;     // [N][wimg] rotation of each level image
    uint2* w_cand;         // [NXCD][seg_cap] flat lists of the launch's candidate starts, in arrival order:
                           // (bin, image<<8 | y0<<4 | x0)
    uint8_t* w_big;        // [trace waves][TRACE_WAVE_BYTES] scratch of the sequential redo of a border with more than 128 points, then 64 x 72 spill bytes (points 56.. of the lanes' borders)
    uint8_t* w_round;      // [NXCD][round_cap][ROUND_BYTES] round records, trace kernel -> polygon kernel
    int32_t* w_nround;     // [NXCD * XCD_STRIDE] records in each XCD's list
    int32_t* w_heavy;      // [2][XCD_STRIDE + heavy_cap] bins with many border starts, listed by the transition kernel for the emit kernel
                           // to serve first: counter, then the bins; two lists used in turn (StepIO::heavy_turn)
    int32_t* w_total;      // [NXCD * XCD_STRIDE] candidates in each XCD's list (XCD-local atomicAdd in the transition kernel, zeroed by the emit kernel)
};

struct Params {
    int32_t N, Hx, Hy, Hc, Ax, Ay, AC, step, R, S, K;
    double res_a, res_h, res_z, inv_res_z;
    double txs[32];                // np.round(k * resolutionA, 6) for k = 0..31 (binPhy.py:236)
    double bin_x, bin_y, bin_z, bin_vol;
    double scale_z, ibin_z;        // Interface scale and round(bin_z*scale, 6)
    int32_t traj_start, goff, gbins;
    int32_t obs_len0, obs_len1;
    // dynamic-LDS carve-up (byte offsets, all multiples of 16)
    // heightmap tile in LDS ("phase planes" of period pp = step, see irbpp_kernels.hip): LX x LY = Ax x Ay entries per plane
    int32_t pp, LX, LY, PL, tile_words;
    uint32_t mg_pp, mg_ly;
    int32_t g_ysh;            // generic overlap test: a wave's 64 lanes are (64 >> g_ysh) rows of (1 << g_ysh) >= Ay action cells
    int32_t box, o_m1;        // box path (every footprint a solid box): LDS offset of the row-maxima grid [Hx][Ay]
    int32_t o_vbits;          // naiveMask bit rows [R][16] of the observation being built
    int32_t o_sr;             // the R ShapeRots of the observed item
    int32_t o_hm, o_posz, o_lev, o_present, o_taskidx, o_tasklist, o_img, o_clist, o_vmask, o_scratch, o_red, o_dps;
    int32_t nslot, slot_cap, slot_bytes, scratch_bytes, lds_bytes;   // lds_bytes: transition kernel (no posZValid region)
    int32_t lds_bytes_full;   // + posZValid at o_posz: heuristic kernel
    int32_t e_vmask, e_red, e_hist, e_keys, emit_lds_bytes;   // the emit kernel's own carve-up
    int32_t ew_bytes, e_need; // wave-per-bin emit kernel: bytes of LDS per wave (vertex bits + candidate keys), offset of its flag words
    int32_t big_slot_bytes;   // bytes of the scratch region the serial redo of an oversized border may use
    // Block path of the overlap test: when every footprint of the dataset is a union of b x b tiles
    // (b a multiple of step) that are either masked out or have one bottom height, the per-bin grid
    // mb[pi][pj] = max of the b x b heightmap block at (pi*step, pj*step) replaces the cell list:
    // max over the tile of (H - B) == (max over the tile of H) - B exactly (rounding is monotone).
    int32_t block_b, mb_w, mb_h, o_mb, o_c2;
    // ... decided per ROTATION (round 6): bit r set = rotation r of every item has such footprints and takes the block loop;
    // the other rotations (BlockOut at eight rotations: the 45-degree ones) walk their cell lists like free-form data, in the
    // same kernel (PATH_MIXED).  All R bits set on pure lattice data; 0 when block_b == 0.
    int32_t block_rots;
    // Division by the runtime grid sizes costs ~25 VALU instructions each; n / d == umulhi(n, mg_d) exactly
    // for n, d < 2^16 with mg_d = floor(2^32 / d) + 1 (d >= 2), see fdiv() in irbpp_kernels.hip.
    uint32_t mg_hy, mg_step, mg_ay, mg_ax, mg_ac, mg_mbw;
    // Tooling (IRBPP_DEBUG_REPEAT): bit k set = run phase k twice (idempotent), so that the slow-down of a
    // launch prices that phase at full chip load.  1 trace, 2 Douglas-Peucker, 4 overlap loops, 8 emit,
    // 16 whole contour stage.
    int32_t dbg_repeat;
    int32_t split;         // 1: transition kernel -> trace kernel -> emit kernel; 0: everything in the transition kernel
    int32_t wimg;          // level images a bin can hand over: R * 64
    int32_t seg_cap;       // entries of one XCD's flat candidate list: twice the worst case (R*AC per bin) of its share of the bins
    int32_t round_cap;     // round records of one XCD's list (a full list makes the trace kernel approximate in place)
    int32_t heavy_cap, heavy_thr;   // bins the emit kernel can serve first (0: off); border starts + isolated pixels that make a bin one of them
    int32_t stability;     // 0 off, 1 rate accepted placements, 2 refuse unstable ones (irbpp_config::stability)
    int32_t wide;          // action grid of 17 .. 32 cells a side: the capacity path of irbpp_wide.hip (one kernel per observation)
    int32_t vrow;          // words per rotation of w_valid (naiveMask bit rows): 16, or 32 on a wide grid
    int32_t rect;          // IRBPP_TUNE_RECT: isolated solid rectangles are answered by the transition kernel instead of the trace kernel (A/B)
};

// ---------------------------------------------------------------------------------------------------------------------
// Derived integers of a configuration (dynamic-LDS carve-up, division constants) -- constexpr, so that the host computes
// them for whatever geometry it is given (irbpp_create) and the device code can have them at COMPILE time for the
// geometries of BASELINE.json's configs (spec_params / specialise below).
// ---------------------------------------------------------------------------------------------------------------------
constexpr int32_t align16(int32_t v) { return (v + 15) & ~15; }
constexpr uint32_t div_magic(int32_t d) { return d >= 2 ? (uint32_t)((1ull << 32) / (uint64_t)d + 1ull) : 0u; }

// dynamic-LDS carve-up of the transition kernel (and the division constants of its grid sizes); `pad`: tooling only
constexpr void layout_lds(Params& P, int32_t pad = 0) {
    P.mg_hy = div_magic(P.Hy);
    P.mg_step = div_magic(P.step);
    P.mg_ay = div_magic(P.Ay);
    P.mg_ax = div_magic(P.Ax);
    P.mg_ac = div_magic(P.AC);
    P.mg_mbw = div_magic(P.mb_w);
    // heightmap tile: phase planes of period step, one entry per action cell (= per lane of the generic overlap test)
    P.pp = P.step;
    P.LX = P.Ax;
    P.LY = P.Ay;
    P.PL = P.LX * P.LY;
    P.tile_words = P.pp * P.pp * P.PL;                                 // == Hc: no padding entries
    P.mg_pp = div_magic(P.pp);
    P.mg_ly = div_magic(P.LY);
    P.g_ysh = 0;
    while ((1 << P.g_ysh) < P.Ay) ++P.g_ysh;                           // Ay <= 16: at least four rows of action cells per wave
    P.nslot = 64;                                                     // candidate starts traced per pass (16 per wave)
    P.slot_cap = 64;                                                  // points of a border: one lane each in the segmented Douglas-Peucker
    P.slot_bytes = P.slot_cap + 4;                                    // 68 B = 17 dwords: odd stride, lanes hit distinct LDS banks
    int32_t off = 0;
    P.o_sr = off;        off += align16(P.R * (int32_t)sizeof(ShapeRot));
    P.o_lev = off;       off += align16(P.R * P.AC);
    P.o_present = off;   off += align16(P.R * 8);
    P.o_taskidx = off;   off += align16(P.R * 64 * 2);
    P.o_tasklist = off;  off += align16(P.R * 64 * 2);
    const int32_t img_bytes = align16(2 * CONTOUR_IPT * 16 * 16 * 2);  // level images of a batch: 16-bit row words + column words
    P.o_img = off;       off += img_bytes;
    // block-max grid of the overlap test (0 bytes on the generic path): dead before the contour
    // stage builds its images, so it shares their bytes when it fits
    const int32_t mb_bytes = align16(P.mb_w * P.mb_h * 8);
    if (mb_bytes <= img_bytes) P.o_mb = P.o_img;
    else { P.o_mb = off; off += mb_bytes; }
    P.o_c2 = off;        off += mb_bytes ? align16(P.AC * 8) : 0;     // block path: per-action-cell maxima, the grid's first step
    P.o_vmask = off;     off += align16(P.R * 16 * 4);
    P.o_vbits = P.o_taskidx;                           // naiveMask bit rows: handed over before the task index is built (split_handover)
    P.o_m1 = off;        off += P.box ? align16(P.Hx * P.Ay * 8) : 0; // box path: row maxima of the tile, [Hx][Ay]
    P.o_red = off;       off += 256;                                  // reductions, flags, queue copy
    // one region serves, in turn, the heightmap tile (apply + overlap test), the contour stage (border
    // slots, the arg-max words of the segmented Douglas-Peucker, and at its end the 256 candidate starts of
    // an image batch) and the candidate keys: the tile's float32 copy is written out before the reuse
    const int32_t slots = align16(P.nslot * P.slot_bytes);
    const int32_t dps = 4 * 64 * 4 + 4 * 64;           // arg-max words and scratch bytes of the segmented Douglas-Peucker
    int32_t scratch = slots + dps + 512;
    const int32_t keys = (P.R * P.AC + P.S) * 4 + 64;
    if (scratch < keys) scratch = keys;
    if (scratch < P.tile_words * 8) scratch = P.tile_words * 8;
    if (scratch < 2 * CONTOUR_IPT * 16 * 16 * 2 + 512) scratch = 2 * CONTOUR_IPT * 16 * 16 * 2 + 512;   // hand-over: a batch's row words + candidate words, the list
    P.scratch_bytes = align16(scratch);
    P.o_hm = off;
    P.o_scratch = off;
    P.o_dps = off + slots;
    P.o_clist = off + P.scratch_bytes - 512;
    P.big_slot_bytes = P.scratch_bytes - 512;          // the serial redo of a border may use everything below the candidate list
    off += P.scratch_bytes;
    off += align16(pad);                               // tooling build only (IRBPP_LDS_PAD): caps workgroups per CU
    P.lds_bytes = off;
    P.o_posz = off;                                    // only the heuristic kernel keeps posZValid in LDS
    P.lds_bytes_full = off + align16(P.R * P.AC * 8);
    // emit kernel: vertex bits, reductions, the radix-select counters / sort keys / row values, candidate keys + selected keys
    int32_t e = 0;
    P.e_vmask = e;  e += align16(P.R * 16 * 4);
    P.e_red = e;    e += 256;
    {   // 256 radix counters, later the sort keys of the selected rows: 10 bytes per entry of the next power of two
        int32_t npad = 64;
        while (npad < P.S) npad <<= 1;
        P.e_hist = e;   e += align16(10 * npad > 1024 ? 10 * npad : 1024);       // >= 4 * S bytes for the rows' values too
    }
    P.e_keys = e;   e += align16(keys);
    // wave-per-bin form of the emit kernel (lattice / box data): per wave the bin's vertex bits and up to S candidate keys
    P.ew_bytes = align16((P.R * 16 + P.S + 64) * 4);
    if (e < 4 * P.ew_bytes) e = 4 * P.ew_bytes;
    P.e_need = e;   e += 16;                           // its four "this bin needs the workgroup" words, behind everything else
    P.emit_lds_bytes = e;
}

// ---------------------------------------------------------------------------------------------------------------------
// Specialised builds.  The step's kernels take every size from Params at run time: grid sizes, rotation count, LDS
// offsets, division constants -- a few hundred scalar loads, index multiplications and trip-count computations per wave
// in kernels that are bound by instruction issue.  For the geometries BASELINE.json names (16 x 16 action cells, step 2
// or 4, R = 2 / 4 / 8, S = 500) a second build of each kernel has these integers as compile-time constants: the
// kernel copies its Params and overwrites the pinned fields (specialise<SPEC>), and constant propagation does the rest.
// The host launches a specialised build only if every pinned field of the environment's Params equals the constant the
// build was compiled with (spec_matches) -- same constexpr layout function on both sides, compared field by field -- and
// the run-time build otherwise; results are identical by construction (tests/test_gpu_features.py runs both).
// ---------------------------------------------------------------------------------------------------------------------
struct SpecKey { int32_t Ax, Ay, step, R, S, block_b, box, block_rots; };
constexpr Params spec_params(const SpecKey& k) {
    Params P{};
    P.Ax = k.Ax; P.Ay = k.Ay; P.step = k.step; P.R = k.R; P.S = k.S;
    P.Hx = k.Ax * k.step; P.Hy = k.Ay * k.step; P.Hc = P.Hx * P.Hy; P.AC = k.Ax * k.Ay;
    P.block_b = k.block_b; P.box = k.box; P.block_rots = k.block_rots;
    P.mb_h = k.block_b ? (P.Hx - k.block_b) / k.step + 1 : 0;
    P.mb_w = k.block_b ? (P.Hy - k.block_b) / k.step + 1 : 0;
    P.split = 1;
    P.stability = 0;
    P.wide = 0;
    P.vrow = 16;
    P.dbg_repeat = 0;
    P.wimg = k.R * 64;
    P.heavy_thr = (k.S * 3) / 5;
    P.obs_len1 = 5 * k.S + 9 + P.Hc;
    layout_lds(P);
    return P;
}
// the fields a specialised build has as constants (everything else -- bins, buffer size, trajectories, capacities, every
// float64 -- stays a run-time value)
#define IRBPP_PINNED_FIELDS(X)                                                                                          \
    X(Hx) X(Hy) X(Hc) X(Ax) X(Ay) X(AC) X(step) X(R) X(S) X(pp) X(LX) X(LY) X(PL) X(tile_words) X(mg_pp) X(mg_ly)      \
    X(g_ysh) X(box) X(o_m1) X(o_vbits) X(o_sr) X(o_hm) X(o_posz) X(o_lev) X(o_present) X(o_taskidx) X(o_tasklist)      \
    X(o_img) X(o_clist) X(o_vmask) X(o_scratch) X(o_red) X(o_dps) X(nslot) X(slot_cap) X(slot_bytes) X(scratch_bytes)  \
    X(lds_bytes) X(lds_bytes_full) X(e_vmask) X(e_red) X(e_hist) X(e_keys) X(emit_lds_bytes) X(ew_bytes) X(e_need) X(big_slot_bytes)         \
    X(block_b) X(block_rots) X(mb_w) X(mb_h) X(o_mb) X(o_c2) X(mg_hy) X(mg_step) X(mg_ay) X(mg_ax) X(mg_ac) X(mg_mbw) X(dbg_repeat)  \
    X(split) X(wimg) X(heavy_thr) X(stability) X(obs_len1) X(wide) X(vrow)
constexpr bool spec_matches(const Params& P, const Params& C) {
#define IRBPP_X(f) if (P.f != C.f) return false;
    IRBPP_PINNED_FIELDS(IRBPP_X)
#undef IRBPP_X
    return true;
}
// SPEC 0 is the run-time build
constexpr SpecKey SPEC_KEYS[] = {
    {0, 0, 0, 0, 0, 0, 0, 0},
    {16, 16, 2, 4, 500, 4, 0, 0x0F},   // 1: BlockOut, R = 4 (BASELINE configs 2 and 4): block path, 4 x 4 cells
    {16, 16, 2, 2, 500, 0, 1, 0},      // 2: Cube, R = 2 (config 1): box path
    {16, 16, 2, 8, 500, 0, 0, 0},      // 3: free-form solids, R = 8, 32 x 32 heightmap (config 3)
    {16, 16, 4, 8, 500, 0, 0, 0},      // 4: free-form solids, R = 8, 64 x 64 heightmap (config 5)
    {16, 16, 2, 8, 500, 4, 0, 0x0F},   // 5: BlockOut at the README command's eight rotations (README.md:100): rotations 0 .. 3 block path, 4 .. 7 cell lists
};
constexpr int N_SPECS = (int)(sizeof(SPEC_KEYS) / sizeof(SPEC_KEYS[0]));
template <int SPEC>
__host__ __device__ __forceinline__ Params specialise(const Params& P) {
    if constexpr (SPEC == 0) {
        return P;
    } else {
        constexpr Params C = spec_params(SPEC_KEYS[SPEC]);
        Params Q = P;
#define IRBPP_X(f) Q.f = C.f;
        IRBPP_PINNED_FIELDS(IRBPP_X)
#undef IRBPP_X
        return Q;
    }
}

enum Mode : int32_t {
    MODE_RESET = 0,       // reset(): new episodes everywhere, first observation
    MODE_STEP = 1,        // step(): apply action, auto-reset, next observation
    MODE_CANDS = 2,       // get_action_candidates(): location observation of the chosen slot
    MODE_POSSIBLE = 3,    // get_possible_position() only, results to global memory
    MODE_OBSERVE = 4      // second half of a split step(): irbpp_apply_kernel has applied the actions; observe the queue's first item (K == 1)
};

struct StepIO {
    const int32_t* actions;     // MODE_STEP: candidate index; MODE_CANDS: buffer slot; MODE_POSSIBLE: item id
    float* obs;                 // observation rows
    int32_t obs_stride;
    double* reward;
    uint8_t* done;
    int32_t* counter;
    double* ratio;
    double* ep_reward;
    int32_t* ep_len;
    uint8_t* stable;            // stability proxy verdict of this step's placement
    double* posz_out;           // MODE_POSSIBLE
    uint8_t* mask_out;
    long long* phase_cycles;    // optional [N][8] shader-clock stamps per phase (tooling)
    int32_t* heur_out;          // MODE_HEURISTIC: [N][3] = rot, lx, ly
    int32_t heur_method;        // 1 MINZ, 2 DBLF, 3 FIRSTFIT, 4 HM (space.py:168-218)
    int32_t heur_dir;           // dirIdx 0..3: (Xflip, Yflip) (space.py:163-166)
    const int32_t* bin_list;    // MODE_RESET on a subset (reset_specific): workgroup i resets bin bin_list[i]
    int32_t reset_next;         // MODE_RESET of all bins: 0 = episode 0 (first reset), 1 = every bin moves on to its next episode
    int32_t* obs_rows;          // optional [N]: this obs buffer is registered (irbpp_register_obs_buffer): candidate rows it holds per bin
                                // from the last call that wrote it (-1 unknown); rows beyond are known to be zero
    int32_t* auto_action;       // optional [N]: the scripted MINZ policy's choice for the observation just emitted (row with the
                                // lowest H among V == 1, first on ties; 0 if none) -- irbpp_set_auto_policy
    int32_t* err_out;           // optional [1]: copy of the device error word, written by the emit kernel (split pipeline)
    int32_t use_order;          // 1: launch slot -> bin through State::order (bins grouped by observed item per die); 0: identity
    int32_t block_off;          // grouped stepping: this launch covers launch slots block_off .. block_off + gridDim.x - 1
    int32_t heavy_turn;         // which of State::w_heavy's two lists this launch fills and serves (-1: none: listed resets, block / box data)
    int32_t n_slots;            // launch slots of this launch (irbpp_apply_kernel: one wave per slot, four per workgroup)
    int32_t fixed_slot;         // MODE_CANDS: 0 = the buffer slot of bin b is actions[b]; j + 1 = slot j for every bin, and the bins' chosen slot
                                // (BinState::order_action, which the next step pops) stays as it is: irbpp_get_all_possible_observation
};

}  // namespace irbpp
