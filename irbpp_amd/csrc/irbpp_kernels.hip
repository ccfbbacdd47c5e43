// irbpp_kernels.hip -- CDNA4 (gfx950) kernels of the batched packing environment.
//
// A transition of all bins is four kernels on the caller's stream:
//   irbpp_env_kernel      one 256-thread workgroup (4 wave64) per bin: apply the action -> (rot,lx,ly) -> prejudge ->
//                         drop height -> height check -> heightmap update / episode end + auto-reset (binPhy.py:248-337);
//                         overlap test of the next item over (rot,X,Y) (space.py:98-129); height levels -> per-level
//                         binary images -> candidate start pixels of their outer borders (cvTools.py:61-84)
//   irbpp_trace_kernel    border following over the candidates of ALL bins, one per lane (cv2.findContours)
//   irbpp_polygon_kernel  approxPolyDP + convex vertices of 128 contour points per wave (cvTools.py:40-59,91-96)
//   irbpp_emit_kernel     one workgroup per bin: candidate set -> select/pad S rows -> float32 observation (binPhy.py:183-232)
// The float64 heightmap tile of a bin lives in LDS for the whole transition kernel (8 KiB at 32x32, 32 KiB at 64x64);
// footprint tables are wave-uniform reads served from L2; levels, level images and vertex bit grids stay in LDS;
// posZValid, the images, the candidate lists and the polygon rounds are handed from kernel to kernel through
// global memory (L2 / Infinity Cache).  Arithmetic is float64 in the reference's operation order (compile with
// -ffp-contract=off), so results are bit-identical to the numpy code.  No MFMA: this is subtract/max/compare and
// integer border following, not a contraction.
//
// LDS heightmap tile layout ("phase planes" of period pp = step): heightmap cell (row, col) with
// row = X*step + u, col = Y*step + v (u, v < step) lives at plane (u*step + v), entry X*Ay + Y; a plane has one entry per
// action cell.  In the generic overlap test a lane owns ONE action cell (X, Y): for a fixed footprint cell all lanes of
// a wave read ONE plane at their own entries plus a wave-uniform offset -- the lanes of a half-wave are two (or more)
// whole rows of the action grid, i.e. 32 distinct bank pairs: the per-lane ds_read_b64 is bank-conflict-free and the
// footprint cell's tile offset is a scalar.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "contours_device.h"
#include "irbpp_device.h"
#include "../../include/irbpp.h"

// This file is compiled in TWO PASSES (it includes itself at its end): pass 1 is everything, for 256-thread workgroups; pass 2
// compiles the transition kernel's code once more inside namespace irbpp::wg512 with BLOCK = 512 -- eight waves share one
// bin's heightmap tile -- for the data whose tile is so large (64 x 64 float64 = 32 KB) that only four 256-thread workgroups
// fit a CU's LDS, i.e. four waves per SIMD.  Everything that does not depend on BLOCK is compiled in pass 1 only and found by
// pass 2 in the enclosing namespace.
#ifndef IRBPP_PASS
#define IRBPP_PASS 1
#endif

// (a call from twice-compiled code to a twice-compiled function names its namespace: argument-dependent lookup would
// otherwise find the pass-1 function beside the pass-2 one)
// (round 6: a THIRD pass, namespace irbpp::wg128, BLOCK = 128 -- two waves per bin for the block path of lattice data, whose
// per-wave prologue, scalar bookkeeping and hand-over control are then executed twice per bin instead of four times)
#undef IRBPP_HERE
#undef IRBPP_PASS_NS
#if IRBPP_PASS == 1
#define IRBPP_HERE ::irbpp::
#elif IRBPP_PASS == 2
#define IRBPP_HERE ::irbpp::wg512::
#define IRBPP_PASS_NS wg512
#else
#define IRBPP_HERE ::irbpp::wg128::
#define IRBPP_PASS_NS wg128
#endif

namespace irbpp {
#if IRBPP_PASS != 1
namespace IRBPP_PASS_NS {
#endif

#if IRBPP_PASS == 1
constexpr int BLOCK = 256;
#elif IRBPP_PASS == 2
constexpr int BLOCK = 512;                       // second pass: the transition kernel's code again for 512-thread workgroups (see the end of the file)
#else
constexpr int BLOCK = 128;                       // third pass: two waves per bin
#endif
constexpr int WAVES = BLOCK / 64;

// Tooling build only (tools/build_variant.sh NAME -DIRBPP_ABLATE): phase `bit` runs twice when Params.dbg_repeat has the
// bit set.  Every repeated phase is idempotent, so results do not change and the slow-down of a launch
// prices the phase at full chip load (and the instruction counters of a PMC pass its instruction count).  In the product
// build the trip count is the constant 1.  Bits: 0 in-workgroup trace, 1 Douglas-Peucker (hull kernel), 2 overlap loops,
// 3 emit stores, 5 tile staging, 6 block-max grid, 7 level codes + masks, 8 float32 heightmap copy, 9 level images +
// candidate bits, 10 candidate list, 11 drop height of the placement, 12 heightmap update, 13 overlap-test set-up,
// 14 hand-over stores.
#ifndef IRBPP_REPS
#ifdef IRBPP_ABLATE
#define IRBPP_REPS(bit) (1 + ((P.dbg_repeat >> (bit)) & 1))
#else
#define IRBPP_REPS(bit) 1
#endif
#endif

#if IRBPP_PASS == 1
// np.round(x, 6): multiply, round-half-even, true-divide (numpy around for decimals > 0)
__device__ __forceinline__ double round6(double x) { return rint(x * 1e6) / 1e6; }

// Sign of np.round(x, 6) without the divide: rint(x*1e6)/1e6 has the sign (and zero-ness) of
// rint(x*1e6), and float64 division is a quarter-rate multi-instruction sequence on CDNA.
__device__ __forceinline__ double round6_scaled(double x) { return rint(x * 1e6); }

// (int) np.floor_divide(a, b) for b > 0 (numpy npy_divmod on float64; decides which cells share a
// height level, cvTools.py:78 -- e.g. 0.06 // 0.01 == 5) WITHOUT any division.  numpy computes
// mod = fmod(a,b), div = (a-mod)/b [-1 if mod has the other sign], and rounds div to the nearest
// integer: that integer is the exact quotient floor(a/b).  The exact quotient comes from a
// reciprocal-multiply guess (off by at most one) corrected with the exact FMA remainder
// (fmod's result is always representable, so fma(-q, b, |a|) is exact for the right q).
__device__ __forceinline__ int np_floor_divide_int(double a, double b, double inv_b) {
    const double x = fabs(a);
    const double q0 = trunc(x * inv_b);
    const double r0 = fma(-q0, b, x);
    // the guess is off by at most one either way: a remainder below 0 -> one less, at or above b -> one more
    int q = (int)q0 + (r0 >= b ? 1 : 0) - (r0 < 0.0 ? 1 : 0);
#if defined(__HIP_DEVICE_COMPILE__)
    if (__ballot(a < 0.0) == 0ull) return q;                 // (wave-uniform: placement heights are practically never negative,
#endif                                                       //  and the sign's share -- a second exact remainder -- was two thirds of this routine)
    if (a < 0.0) {
        const double r = fma(-(double)q, b, x);              // the exact remainder of the corrected quotient
        q = r == 0.0 ? -q : -q - 1;
    }
    return q;
}

// wave64 inclusive scans on the DPP network (no LDS round trips): prefix inside each row of 16 lanes by four
// row shifts, then the row totals are carried over with the two row broadcasts.  Operands are >= 0, so the
// 0 that a shift brings in from outside the row is the identity of both sum and max.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp_shift(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, ROW_MASK, 0xF, false); }
__device__ __forceinline__ int wave_inclusive_sum(int v) {
    v += dpp_shift<0x111, 0xF>(v);           // row_shr:1
    v += dpp_shift<0x112, 0xF>(v);           // row_shr:2
    v += dpp_shift<0x114, 0xF>(v);           // row_shr:4
    v += dpp_shift<0x118, 0xF>(v);           // row_shr:8
    v += dpp_shift<0x142, 0xA>(v);           // row_bcast15 into rows 1 and 3
    v += dpp_shift<0x143, 0xC>(v);           // row_bcast31 into rows 2 and 3
    return v;
}
__device__ __forceinline__ int imax(int a, int b) { return a > b ? a : b; }
__device__ __forceinline__ int wave_inclusive_max(int v) {
    v = imax(v, dpp_shift<0x111, 0xF>(v));
    v = imax(v, dpp_shift<0x112, 0xF>(v));
    v = imax(v, dpp_shift<0x114, 0xF>(v));
    v = imax(v, dpp_shift<0x118, 0xF>(v));
    v = imax(v, dpp_shift<0x142, 0xA>(v));
    v = imax(v, dpp_shift<0x143, 0xC>(v));
    return v;
}

// wave64 maximum of a float64 on the DPP network (lane 63 ends up with it; every lane gets it back by v_readlane): a
// butterfly of __shfl_xor is six ds_bpermute_b32 per word, 24 LDS cycles each and an LDS round trip of latency per step
// (tools/microbench_int.hip)
__device__ __forceinline__ double wave_max_f64(double v) {
#define IRBPP_MAX_STEP(CTRL, ROWS)                                                                                   \
    {                                                                                                                \
        const int lo = __builtin_amdgcn_update_dpp(__double2loint(v), __double2loint(v), CTRL, ROWS, 0xF, false);    \
        const int hi = __builtin_amdgcn_update_dpp(__double2hiint(v), __double2hiint(v), CTRL, ROWS, 0xF, false);    \
        v = fmax(v, __hiloint2double(hi, lo));                                                                       \
    }
    IRBPP_MAX_STEP(0x111, 0xF) IRBPP_MAX_STEP(0x112, 0xF) IRBPP_MAX_STEP(0x114, 0xF) IRBPP_MAX_STEP(0x118, 0xF)
    IRBPP_MAX_STEP(0x142, 0xA) IRBPP_MAX_STEP(0x143, 0xC)
#undef IRBPP_MAX_STEP
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 63), __builtin_amdgcn_readlane(__double2loint(v), 63));
}

#endif  // IRBPP_PASS == 1
__device__ inline double block_max_f64(double v, double* red) {
    v = wave_max_f64(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    double r = red[0];
    for (int w = 1; w < WAVES; ++w) r = fmax(r, red[w]);
    return r;
}

__device__ inline int block_sum_int(int v, int* red) {
    v = __builtin_amdgcn_readlane(wave_inclusive_sum(v), 63);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    int r = 0;
    for (int w = 0; w < WAVES; ++w) r += red[w];
    return r;
}

// exclusive prefix count of `flag` over the block in thread order; total in `total`
__device__ inline int block_scan_flag(bool flag, int* red, int& total) {
    const unsigned long long bal = __ballot(flag);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int pre = __popcll(bal & ((1ull << lane) - 1ull));
    __syncthreads();
    if (lane == 0) red[w] = __popcll(bal);
    __syncthreads();
    int base = 0;
    total = 0;
    for (int i = 0; i < WAVES; ++i) {
        const int c = red[i];
        if (i < w) base += c;
        total += c;
    }
    return base + pre;
}

#if IRBPP_PASS == 1
// LoadItemCreator.generate_item (IRcreator.py:97-103) on the pre-drawn trajectories; the
// trajectory of global bin g in its e-th episode is (traj_start + g + e*global_bins) % n_traj.
// The 64-bit modulo runs once per episode (trajectory_row, kept in BinState::traj_row), not per item.
// Stream mode (irbpp_config::item_stream, the Random*Creator classes of IRcreator.py:26-72): bin b of this device owns
// row b % n_traj for good, the row is a ring the host keeps ahead of the bin, and `cursor` counts the items the bin has
// drawn since the row was loaded -- a new episode goes on where the last one stopped (ItemCreator.reset only clears
// the queue).
__device__ inline int trajectory_row(const Params& P, const Tables& T, int b, int episode) {
    long long row = T.stream ? (long long)b : (long long)P.traj_start + P.goff + b + (long long)episode * P.gbins;
    row %= T.n_traj;
    if (row < 0) row += T.n_traj;
    return (int)row;
}
// Device error bits are sticky in S.err; a step that has an error word in its outputs (StepIO::err_out) gets every bit
// there too, at the moment it is raised -- the buffered step (K > 1) ends with the transition kernel, so nobody copies
// the word afterwards (the online step's emit kernel stores S.err into it once more at the end).  Cold path: the
// pointer is read from the kernarg segment where it is needed.
struct KernArgs;
__device__ __forceinline__ void raise_error(const State& S, int bit);
constexpr int STREAM_CONSUMED = -3;     // stream mode: a ring slot the bin has read and the host has not rewritten yet (irbpp_stream_write
                                        // turns any id below -1 it is handed into -1, so the host cannot write this value)
// Stream mode: the id that was read from ring slot (row, cursor) is consumed -- the slot is marked, in the item table
// itself (Tables::seq is the bins' own rings then: every slot has one reader, the bin that owns the row, and is read once
// per lap) -- and validated.  A slot that still carries the mark means the ring ran dry: caught at the fetch, not a refill
// later.  One routine for the fetch and for the ids a step requested ahead of time.
__device__ inline int consume_item(const Tables& T, const State& S, int row, int cursor, int id) {
    if (T.stream) {
        if (id == STREAM_CONSUMED) { raise_error(S, IRBPP_DEVERR_STREAM_DRY); return -1; }
        const_cast<int32_t*>(T.seq)[(long long)row * T.seq_len + (int)((uint32_t)cursor % (uint32_t)T.seq_len)] = STREAM_CONSUMED;
    }
    if (id >= T.n_shapes) { raise_error(S, IRBPP_DEVERR_BAD_ITEM); return -1; }
    return id < 0 ? -1 : id;
}
__device__ inline int fetch_item(const Tables& T, const State& S, int row, int cursor) {
    if (!T.stream && cursor >= T.seq_len) return -1;
    const int at = T.stream ? (int)((uint32_t)cursor % (uint32_t)T.seq_len) : cursor;
    return consume_item(T, S, row, cursor, T.seq[(long long)row * T.seq_len + at]);
}

// n / d for 0 <= n < 2^16 and the runtime grid sizes 1 <= d < 2^16, without the ~25-instruction integer
// division sequence: with m = floor(2^32 / d) + 1, m*d = 2^32 + e, 0 < e <= d, so
// floor(n*m / 2^32) = floor(n/d + n*e / (d * 2^32)) = floor(n/d) because n*e < 2^32.  (m would overflow for d = 1.)
__device__ __forceinline__ int fdiv(int n, int d, uint32_t m) {
    return d == 1 ? n : (int)__umulhi((uint32_t)n, m);
}

// Kernel arguments that are used once, by one thread, at the end of a phase (step outputs, episode totals,
// placement log, scheduling hints).  Read normally they are loaded at kernel entry together with
// everything else -- as 8- and 16-dword tuples that do not fit the SGPR file and get spilled to VGPR
// lanes and restored, tuple-wise, eleven times along the way (176 v_readlane for the six output
// pointers alone).  Read through this laundered pointer to the kernarg segment they are scalar loads
// right where they are needed.  Both transition kernels take (Params, Tables, State, StepIO, int).
struct KernArgs { Params P; Tables T; State S; StepIO io; int mode; };
typedef const __attribute__((address_space(4))) KernArgs* KernArgsPtr;
__device__ __forceinline__ KernArgsPtr cold_args() {
    KernArgsPtr p = (KernArgsPtr)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(p));
    return p;
}

__device__ __forceinline__ void raise_error(const State& S, int bit) {
    atomicOr(S.err, bit);
    int32_t* mirror = cold_args()->io.err_out;
    if (mirror != nullptr) atomicOr(mirror, bit);
}

__device__ __forceinline__ uint32_t div_magic_dev(int d) { return d >= 2 ? (uint32_t)(0x100000000ull / (uint64_t)d + 1ull) : 0u; }

// heightmap cell (row, col) / linear heightmap index (row*Hy + col) -> LDS tile index in the phase-plane layout.
// The index is a sum of a row part and a column part, so loops over rows x columns compute each part once.
__device__ __forceinline__ int tile_row_part(const Params& P, int row) {
    const int q = fdiv(row, P.pp, P.mg_pp);
    return (row - q * P.pp) * P.pp * P.PL + q * P.LY;
}
__device__ __forceinline__ int tile_col_part(const Params& P, int col) {
    const int q = fdiv(col, P.pp, P.mg_pp);
    return (col - q * P.pp) * P.PL + q;
}
__device__ __forceinline__ int tile_rc(const Params& P, int row, int col) { return tile_row_part(P, row) + tile_col_part(P, col); }
__device__ __forceinline__ int tile_of_linear(const Params& P, int g) {
    const int row = fdiv(g, P.Hy, P.mg_hy);
    return tile_rc(P, row, g - row * P.Hy);
}
#endif  // IRBPP_PASS == 1
// Walk of the whole heightmap by the workgroup, element i = tid + k*BLOCK of the row-major map: when BLOCK is a
// multiple of Hy the column of a thread's elements never changes and its row advances by BLOCK/Hy per trip.
struct TileWalk {
    int col_part, row, row_step, lin;
    bool regular;
};
__device__ __forceinline__ TileWalk tile_walk_begin(const Params& P, int tid) {
    TileWalk w;
    const int row = fdiv(tid, P.Hy, P.mg_hy), col = tid - row * P.Hy;
    w.regular = BLOCK % P.Hy == 0;
    w.row = row;
    w.row_step = BLOCK / P.Hy;
    w.col_part = tile_col_part(P, col);
    w.lin = tid;
    return w;
}
__device__ __forceinline__ int tile_walk_index(const Params& P, const TileWalk& w) {      // tile index of element w.lin
    return w.regular ? tile_row_part(P, w.row) + w.col_part : tile_of_linear(P, w.lin);
}
__device__ __forceinline__ void tile_walk_next(TileWalk& w) { w.lin += BLOCK; w.row += w.row_step; }
#if IRBPP_PASS == 1
// footprint cell `ij` (i | j << 16) of an item whose corner sits on action cell (lx, ly)
__device__ __forceinline__ int tile_of_cell(const Params& P, int lx, int ly, int ij) {
    return tile_rc(P, lx * P.step + (ij & 0xFFFF), ly * P.step + (ij >> 16));
}

// Read-only tables are addressed through the constant address space so that wave-uniform
// reads become scalar loads (s_load_dwordx4 per Cell) instead of per-lane vector loads.
typedef const __attribute__((address_space(4))) Cell* ConstCellPtr;
__device__ __forceinline__ ConstCellPtr as_const(const Cell* p) { return (ConstCellPtr)(unsigned long long)p; }

// optional per-phase shader-clock stamps (s_memtime), one row of PHASE_ROW per bin:
// [0..4] phase boundaries, [5..7] contour-stage detail, [8]/[9] 100 MHz wall clock at entry/exit,
// [10] HW_ID | XCC_ID << 32 (which CU the bin ran on)
constexpr int PHASE_ROW = 16;
__device__ __forceinline__ void stamp(const StepIO& io, int b, int k, bool leader) {
    if (io.phase_cycles && leader) {
        long long* row = io.phase_cycles + (size_t)b * PHASE_ROW;
        row[k] = (long long)clock64();
        if (k == 0) {
            row[8] = (long long)wall_clock64();
            const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);      // HW_REG_HW_ID
            const unsigned xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);    // HW_REG_XCC_ID
            row[10] = (long long)hw | ((long long)xcc << 32);
        }
        if (k == 4) row[9] = (long long)wall_clock64();
    }
}

__device__ __forceinline__ void stamp(const StepIO& io, int b, int k) { stamp(io, b, k, threadIdx.x == 0); }

__device__ inline SlotMem carve_slot(unsigned char* base, int cap, int cap_stk) {
    SlotMem m;
    m.pts = base;
    m.dst = m.pts + cap;
    m.stk = (uint32_t*)(m.dst + cap);
    m.cap = cap;
    m.cap_stk = cap_stk;
    return m;
}

struct Lds {
    int* sr;                // the R ShapeRots of the observed item, as dwords
    double* hm;
    double* mb;             // block-max grid of the tile (block path of the overlap test)
    double* c2;             // maxima of the action cells' own step x step heightmap cells, [Ax][Ay] (block path)
    double* m1;             // row maxima of the tile, [Hx][Ay] (box path of the overlap test)
    double* posz;
    uint8_t* lev;
    unsigned long long* present;
    uint16_t* taskidx;      // [R][64] level code -> task index
    uint16_t* tasklist;     // [ntasks] rot<<8 | level code
    uint16_t* img;          // [2][IMGS][16] level images of the current batch: 16-bit row words, then column words
    uint16_t* clist;        // [256] candidate starts of the image (sub-)batch: image | x0<<6 | y0<<10
    uint32_t* dps;          // [WAVES][64] arg-max words of the segmented Douglas-Peucker, one set per wave
    uint32_t* vmask;
    uint32_t* vbits;        // [R][16] naiveMask as bit rows (bit Y of word r*16 + X)
    unsigned char* scratch;
    double* redd;
    int* redi;
};

__device__ inline Lds carve_lds(unsigned char* smem, const Params& P) {
    Lds L;
    L.sr = (int*)(smem + P.o_sr);
    L.hm = (double*)(smem + P.o_hm);
    L.mb = (double*)(smem + P.o_mb);
    L.c2 = (double*)(smem + P.o_c2);
    L.m1 = (double*)(smem + P.o_m1);
    L.posz = (double*)(smem + P.o_posz);
    L.lev = smem + P.o_lev;
    L.present = (unsigned long long*)(smem + P.o_present);
    L.taskidx = (uint16_t*)(smem + P.o_taskidx);
    L.tasklist = (uint16_t*)(smem + P.o_tasklist);
    L.img = (uint16_t*)(smem + P.o_img);
    L.clist = (uint16_t*)(smem + P.o_clist);
    L.dps = (uint32_t*)(smem + P.o_dps);
    L.vmask = (uint32_t*)(smem + P.o_vmask);
    L.vbits = (uint32_t*)(smem + P.o_vbits);
    L.scratch = smem + P.o_scratch;
    L.redd = (double*)(smem + P.o_red);
    L.redi = (int*)(L.redd + 8);
    return L;
}

#endif  // IRBPP_PASS == 1
// ---------------------------------------------------------------------------------------
// cvTools.getConvexHullActions on the grids in LDS: L.posz = posZValid [R][AC] (1e3 where
// invalid), `valid` given per thread/rotation through L.lev (255 = masked).  On return
// L.vmask[r*16 + row] has bit col set for every candidate (row, col) of rotation r.
// ---------------------------------------------------------------------------------------
constexpr int CONTOUR_IMGS = CONTOUR_IPT * (BLOCK / 16);         // level images per batch (contour stage of the hull kernel)
constexpr int CONTOUR_CLIST = 256;                               // candidate starts listed at a time
// (A hand-over batch of 128 images and 1024 listed candidates for the generic path, R = 8 with ~100 speckled level
// images per bin -- one batch instead of four -- was measured and lost: the wider per-thread loops cost the kernel its
// register allocation, general 12.8 -> 11.1 M steps/s.)

// task list: one task (level image) per (rotation, present level), in (rotation, level) order
__device__ inline int contour_tasks(const Params& P, const Lds& L) {
    const int tid = threadIdx.x, R = P.R;
    int ntasks = 0;
    for (int r = 0; r < R; ++r) ntasks += __popcll(L.present[r]);
    // (`#pragma unroll 1` on the workgroup-strided loops that run once or twice: left to itself the compiler unrolls them
    // sixteen-fold with a remainder loop, and a wave that makes one trip still walks ~25 vector and ~20 scalar instructions of
    // trip-count arithmetic and skipped bodies per loop -- a dozen such loops in a kernel bound by instruction issue)
#pragma unroll 1
    for (int t = tid; t < R * 64; t += BLOCK) {
        const int r = t >> 6, l = t & 63;
        const unsigned long long m = L.present[r];
        if ((m >> l) & 1ull) {
            int base = 0;
            for (int q = 0; q < r; ++q) base += __popcll(L.present[q]);
            const int idx = base + __popcll(m & ((1ull << l) - 1ull));
            L.tasklist[idx] = (uint16_t)((r << 8) | l);
            L.taskidx[t] = (uint16_t)idx;
        }
    }
    return ntasks;
}

// level images of the batch starting at task `base`: 16-bit row words and, for the in-kernel trace of the hull
// kernel only, the transposed copy (the trace kernel of the split pipeline transposes the images it follows itself:
// 86 k traced images per step instead of 32 slots of every bin)
template <int IPT>
__device__ inline void contour_images(const Params& P, const Lds& L, uint16_t* const rows, int base, bool with_cols) {
    const int tid = threadIdx.x, R = P.R, AC = P.AC;
    constexpr int IMGS = IPT * (BLOCK / 16);
    const int g = tid >> 4, y = tid & 15;            // this thread holds row y of the batch's images g, g+16, ...
    uint16_t* const cols = rows + IMGS * 16;         // [IMGS][16] column words (bit y of word x), after the row words
#pragma unroll 1
    for (int i = tid; i < IMGS * 16; i += BLOCK) rows[i] = 0;             // the column copy is derived below
    __syncthreads();
    // Image rows: thread tid holds action cell (X, Y), bit Y of row word X of the image its level belongs to.  One
    // LDS atomic OR per rotation (16-bit half of a dword).  The lanes of a row hit one word and are serialised inside
    // the LDS unit, which has the time; building the words from wave ballots instead (one loop trip per distinct
    // image of the wave) cost ~30 VALU instructions per rotation and wave on a kernel that is VALU-bound.
    // (two-wave build: two cells per thread)
    constexpr int CELLS = BLOCK >= 256 ? 1 : 256 / BLOCK;
#pragma unroll 1
    for (int ci = 0; ci < CELLS; ++ci) {
    const int cell = CELLS == 1 ? tid : tid + ci * BLOCK;
    const int X = fdiv(cell, P.Ay, P.mg_ay), Y = cell - X * P.Ay;
    for (int r = 0; r < R; ++r) {
        const int code = cell < AC ? (int)L.lev[r * AC + cell] : 255;
        int ti = code != 255 ? (int)L.taskidx[r * 64 + code] - base : -1;
        if (ti >= 0 && ti < IMGS) {
            const int w = ti * 16 + X;
            atomicOr((uint32_t*)rows + (w >> 1), (1u << Y) << ((w & 1) * 16));
        }
    }
    }
    __syncthreads();
    if (!with_cols) return;
    // transposed copy (column words) for the vertical run jumps: thread (g, y) gathers column y
    for (int h = 0; h < IPT; ++h) {
        const int gg = g + h * (BLOCK / 16);
        uint32_t col = 0u;
        for (int k = 0; k < 16; ++k) col |= (((uint32_t)rows[gg * 16 + k] >> y) & 1u) << k;
        cols[gg * 16 + y] = (uint16_t)col;
    }
    __syncthreads();
}

// candidate starts of row y of batch image gg (pure bit operations on three row words) and, among them, the isolated
// pixels: no foreground neighbour at all, i.e. a one-point border -- approxPolyDP returns the point and
// find_convex_vetex keeps every vertex of a polygon with <= 3 of them (cvTools.py:42-43), so it is marked as a vertex
// directly and never needs a trace lane
__device__ __forceinline__ uint32_t row_candidates(const uint16_t* rows, int gg, int y, bool live, uint32_t& iso) {
    const uint32_t row_bits = live ? (uint32_t)rows[gg * 16 + y] : 0u;
    const uint32_t up_bits = (live && y > 0) ? (uint32_t)rows[gg * 16 + y - 1] : 0u;
    const uint32_t down_bits = (live && y < 15) ? (uint32_t)rows[gg * 16 + y + 1] : 0u;
    const uint32_t cand = start_candidates(row_bits, up_bits);
    iso = cand & ~(row_bits >> 1) & ~down_bits & ~(down_bits << 1) & ~(down_bits >> 1);
    return cand & ~iso;
}

// first pass over the batch, one thread per (image, row): isolated pixels are marked as vertices on the spot; returns
// the number of candidate starts of the whole batch (low 16 bits) and of its isolated pixels (high 16 bits).  The candidate
// words go to `keep` ([images of the batch][16], next to the row words in LDS) for contour_list; without one (the hull
// kernel's contour stage) contour_list computes them again.
template <int IPT>
__device__ inline int contour_candidates(const Params& P, const Lds& L, const uint16_t* const rows, int base, int ntasks,
                                         uint16_t* const keep = nullptr) {
    const int tid = threadIdx.x;
    const int g = tid >> 4, y = tid & 15;
    int my_count = 0;
#pragma unroll 1
    for (int h = 0; h < IPT; ++h) {
        const int gg = g + h * (BLOCK / 16);
        uint32_t iso;
        uint32_t cand = row_candidates(rows, gg, y, base + gg < ntasks, iso);
        if (iso) atomicOr(&L.vmask[(L.tasklist[base + gg] >> 8) * 16 + y], iso);
        int nrect = 0;
        if (keep && cand && P.rect) {
            // isolated solid rectangles (contours_device.h: rect_component): their vertices follow from (w, h) -- marked here like
            // the isolated pixels, and the start leaves the list.  (Only where the candidate words are kept: the split pipeline; opt-in,
            // IRBPP_TUNE_RECT: the loop costs the transition kernel what it saves the trace and polygon kernels.)
            const uint16_t* const im = rows + gg * 16;
            const uint32_t row_bits = (uint32_t)im[y];
            uint32_t* const vm = &L.vmask[(L.tasklist[base + gg] >> 8) * 16];
            uint32_t todo = cand;
            while (todo) {
                const int x0 = __ffs((int)todo) - 1;
                todo &= todo - 1u;
                int rw, rh;
                if (rect_component(im, row_bits, x0, y, rw, rh)) {
                    uint32_t top, bottom;
                    rect_vertices(rw, rh, x0, top, bottom);
                    atomicOr(vm + y, top);
                    if (bottom) atomicOr(vm + y + rh - 1, bottom);
                    cand &= ~(1u << x0);
                    ++nrect;
                }
            }
        }
        if (keep) keep[gg * 16 + y] = (uint16_t)cand;
        my_count += __popc(cand) + ((__popc(iso) + nrect) << 16);
    }
    return block_sum_int(my_count, L.redi);
}

// the candidates of the batch (nsub == 1) or of its image `sub` into `clist` (image in batch | x0 << 7 | y0 << 11);
// returns their number
template <int IPT>
__device__ inline int contour_list(const Lds& L, const uint16_t* const rows, uint16_t* const clist, int base, int ntasks,
                                   int nsub, int sub, const uint16_t* const kept = nullptr) {
    const int tid = threadIdx.x;
    const int g = tid >> 4, y = tid & 15;
    __syncthreads();
    if (tid == 0) L.redi[10] = 0;
    __syncthreads();
#pragma unroll 1
    for (int h = 0; h < IPT; ++h) {
        const int gg = g + h * (BLOCK / 16);
        if (nsub == 1 || gg == sub) {
            uint32_t iso;
            uint32_t cand = kept ? (uint32_t)kept[gg * 16 + y] : row_candidates(rows, gg, y, base + gg < ntasks, iso);
            while (cand) {
                const int x = __ffs((int)cand) - 1;
                cand &= cand - 1u;
                clist[atomicAdd(&L.redi[10], 1)] = (uint16_t)(gg | (x << 7) | (y << 11));
            }
        }
    }
    __syncthreads();
    return L.redi[10];
}

#if IRBPP_PASS == 1
__device__ inline void contour_stage(const Params& P, const State& S, const Lds& L, long long* prof) {
    const int tid = threadIdx.x;
    constexpr int IMGS = CONTOUR_IMGS, CLIST = CONTOUR_CLIST;
    const int ntasks = IRBPP_HERE contour_tasks(P, L);
    uint16_t* const rows = L.img;                    // [IMGS][16] row words (bit x of word y)
    uint16_t* const cols = L.img + IMGS * 16;        // [IMGS][16] column words (bit y of word x)
    for (int base = 0; base < ntasks; base += IMGS) {
        const long long t_img = prof ? (long long)clock64() : 0;
        contour_images<CONTOUR_IPT>(P, L, L.img, base, true);
        // (a) candidate starts.  The list holds CLIST entries; a batch with more candidates (pathological
        // speckle) is walked one image at a time (an image has at most 64: every other pixel of every other row).
        const int batch_total = contour_candidates<CONTOUR_IPT>(P, L, L.img, base, ntasks) & 0xFFFF;
        const int nsub = batch_total <= CLIST ? 1 : IMGS;
        for (int sub = 0; sub < nsub; ++sub) {
        const int total = contour_list<CONTOUR_IPT>(L, L.img, L.clist, base, ntasks, nsub, sub);
        if (prof && tid == 0) prof[5] += (long long)clock64() - t_img;       // images, transposes, candidate list
        // (b) 64 candidates per pass, spread over the four waves: candidate c is traced by lane c / 4 of wave
        // c % 4, and each wave then runs approxPolyDP + convexity on the borders it traced itself, one
        // contour POINT per lane (approx_convex_segmented), so a pass needs no block barrier between the two
        constexpr int NPASS = 64, PER_WAVE = NPASS / WAVES;
        const int wave = tid >> 6, lane = tid & 63;
        uint32_t* const dps = L.dps + wave * 64;                     // this wave's arg-max words
        for (int c0 = 0; c0 < total; c0 += NPASS) {
            const int count = total - c0 < NPASS ? total - c0 : NPASS;
            const long long t_b = prof ? (long long)clock64() : 0;
            if (tid == 0) { L.redi[8] = 0; L.redi[9] = 0; }
            __syncthreads();
            // (b1) trace; a candidate that is not the first pixel of its component returns 0 points
            const int c = lane * WAVES + wave;
            int my_n = 0, my_r = 0;
            if (lane < PER_WAVE && c < count) {
                const uint32_t e = L.clist[c0 + c];
                const int gi = e & 127u;
                my_r = L.tasklist[base + gi] >> 8;
                int n = 0;
                for (int rep = 0; rep < IRBPP_REPS(0); ++rep)
                    n = trace_border(rows + gi * 16, cols + gi * 16, (e >> 7) & 15u, (e >> 11) & 15u,
                                     L.scratch + c * P.slot_bytes, P.slot_cap);
                if (n < 0) atomicOr(S.err, IRBPP_DEVERR_TRACE_GUARD);
                else if (n > P.slot_cap) atomicOr(&L.redi[8 + (c >> 5)], 1 << (c & 31));
                else my_n = n;
            }
            const long long t_tr = prof ? (long long)clock64() : 0;
            // (b2) this wave's borders, packed back to back over the lanes: as many whole borders per
            // round as fit into 64 points
            for (int rep = 0; rep < IRBPP_REPS(1); ++rep) {
            uint32_t todo = (uint32_t)__ballot(my_n > 0);
            while (todo) {
                uint32_t sel = 0u;
                int sum = 0, mine = -1, sb = 0, nn = 1;
                for (uint32_t bb = todo; bb; bb &= bb - 1u) {
                    const int i = __ffs((int)bb) - 1;
                    const int ni = __builtin_amdgcn_readlane(my_n, i);
                    if (sum + ni > 64) break;
                    if (lane >= sum && lane < sum + ni) { mine = i; sb = sum; nn = ni; }
                    sel |= 1u << i;
                    sum += ni;
                }
                todo &= ~sel;
                const bool live = mine >= 0;
                const int mc = (live ? mine : 0) * WAVES + wave;          // my border's slot
                const uint8_t* pts = L.scratch + mc * P.slot_bytes;
                const int jj = lane - sb;
                const int pv = live ? (int)pts[jj] : 0;
                const int rr = __shfl(my_r, live ? mine : 0);
                {
                    const bool live1[1] = {live};
                    const int pv1[1] = {pv}, j1[1] = {jj}, n1[1] = {nn}, sb1[1] = {sb}, r1[1] = {rr};
                    const uint8_t* const p1[1] = {pts};
                    approx_convex_segmented<1>(lane, live1, pv1, j1, n1, sb1, p1, r1, dps, (uint8_t*)(L.dps + WAVES * 64) + wave * 64, L.vmask);
                }
            }
            }
            const long long t_dp = prof ? (long long)clock64() : 0;
            __syncthreads();
            const long long t_bar = prof ? (long long)clock64() : 0;
            // (b3) a border that outgrew its slot (more than 64 points: pathological speckle): lane 0 redoes it
            // with the sequential routine in one big slot
            {
                const unsigned long long redo = ((unsigned long long)(unsigned)L.redi[9] << 32) | (unsigned)L.redi[8];
                if (redo != 0ull) {
                    if (tid == 0) {
                        const int cap = (P.big_slot_bytes / 6) & ~3;
                        const SlotMem m = carve_slot(L.scratch, cap, cap);
                        for (int cc = 0; cc < count; ++cc) {
                            if (!((redo >> cc) & 1ull)) continue;
                            const uint32_t e = L.clist[c0 + cc];
                            const int gi = e & 127u;
                            const int r = L.tasklist[base + gi] >> 8;
                            if (contour_vertices(rows + gi * 16, cols + gi * 16, (e >> 7) & 15u, (e >> 11) & 15u, m,
                                                 L.vmask + r * 16) != 0)
                                atomicOr(S.err, IRBPP_DEVERR_TRACE_GUARD);          // never silently drop a border
                        }
                    }
                    __syncthreads();
                }
            }
            if (prof && tid == 0) {                 // wave 0's view of the pass
                const long long t_e = (long long)clock64();
                prof[6] += t_e - t_b; prof[7] += count;
                prof[11] += t_tr - t_b; prof[12] += t_dp - t_tr; prof[13] += t_bar - t_dp; prof[14] += t_e - t_bar;
                prof[15] += __popcll(((unsigned long long)(unsigned)L.redi[9] << 32) | (unsigned)L.redi[8]);
            }
        }
        }
    }
}

// np.sum over the window of np.max(((heightMapT + posZ) * maskH, heightmapC[window]), axis=0)
// (space.py:213-214) in numpy's own summation order: float64 pairwise sum with eight
// accumulators per block of <= 128 elements, halves split at multiples of 8 (numpy
// loops_utils.h.src, @TYPE@_pairwise_sum).  The recursion visits the elements strictly in order, each once, so
// `next()` is a generator (the caller's cursor over the window) and the recursion an explicit post-order walk
// of the split tree with its few frames in registers (selected by unrolled compares: a frame is touched once per
// block of 128 elements).  Until round 5 session 37 this was a recursive device function -- 784 bytes of stack per
// lane -- over an `at(i)` that divided i by the window's width and searched the masked-in list by bisection for every
// element.
template <typename G>
__device__ __forceinline__ double np_sum_block(G& next, int n) {               // n <= 128
    if (n < 8) {
        double res = 0.0;
        for (int i = 0; i < n; ++i) res += next();
        return res;
    }
    double r[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) r[k] = next();
    int i = 8;
    for (; i < n - (n % 8); i += 8)
#pragma unroll
        for (int k = 0; k < 8; ++k) r[k] += next();
    double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; ++i) res += next();
    return res;
}
template <typename G>
__device__ __forceinline__ double np_pairwise_sum(G& next, int n) {
    constexpr int DEPTH = 10;                          // 128 << 9 elements: far beyond any window (64 x 64 cells: depth 6)
    int fn[DEPTH], phase[DEPTH];                       // frame d: elements of its subtree; 0 = entered, 1 = left half done, 2 = both
    double left[DEPTH];
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) { fn[d] = 0; phase[d] = 0; left[d] = 0.0; }
    fn[0] = n;
    int sp = 0;
    double ret = 0.0;
    while (sp >= 0) {
        int cn = 0, cp = 0;
        double cl = 0.0;
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) if (d == sp) { cn = fn[d]; cp = phase[d]; cl = left[d]; }
        if (cp == 0 && (cn <= 128 || sp == DEPTH - 1)) {                     // a leaf (the depth bound is never reached: see DEPTH)
            ret = np_sum_block(next, cn);
            --sp;
            continue;
        }
        int n2 = cn / 2;
        n2 -= n2 % 8;
        if (cp == 2) {                                                       // S(left) + S(right)
            ret = cl + ret;
            --sp;
            continue;
        }
        const int child = cp == 0 ? n2 : cn - n2;
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            if (d == sp) { phase[d] = cp + 1; if (cp == 1) left[d] = ret; }
            if (d == sp + 1) { fn[d] = child; phase[d] = 0; }
        }
        ++sp;
    }
    return ret;
}

// Space.get_heuristic_action (space.py:162-218) for one action cell; invalid cells score 1e6.
__device__ inline double heuristic_score(const Params& P, const Tables& T, const StepIO& io, const Lds& L,
                                         const ShapeRot& sr, int r, int X, int Y) {
    const double z = L.posz[r * P.AC + X * P.Ay + Y];
    if (!(z < 1e3)) return 1e6;
    const bool xf = (io.heur_dir & 2) != 0, yf = (io.heur_dir & 1) != 0;
    const double cx = xf ? (double)(P.Ax - X) : (double)X, cy = yf ? (double)(P.Ay - Y) : (double)Y;
    double score;
    switch (io.heur_method) {
        case 1: score = z; break;                                            // MINZ
        case 2: score = (cx + cy) * P.res_a + 100.0 * z; break;              // DBLF
        case 3: score = cx + cy; break;                                      // FIRSTFIT
        default: {                                                           // HM
            score = (cx + cy) * P.res_a;
            const Cell* top = T.tcell + sr.ot;
            // dense row-major walk over the fx x fy window; the compact top list (masked-in cells, `pad` = row-major index
            // in the window) is in the same order: one cursor over it
            const int fy = sr.fy, nt = sr.nt;
            int e = 0, i = 0, j = 0, cur = 0;
            int next_in = nt > 0 ? top[0].pad : -1;                          // window index of the next masked-in cell
            auto next = [&]() -> double {
                const double h = L.hm[tile_rc(P, X * P.step + i, Y * P.step + j)];
                double v = 0.0;                                             // (T + z) * 0 for masked-out cells
                if (e == next_in) {
                    v = top[cur].v + z;
                    ++cur;
                    next_in = cur < nt ? top[cur].pad : -1;
                }
                ++e;
                if (++j == fy) { j = 0; ++i; }
                return fmax(v, h);
            };
            score += np_pairwise_sum(next, sr.fx * sr.fy) * 100.0;
        }
    }
    return round6(score);
}

// wave64 OR-reduction on the DPP network; lane 63 ends up with the OR of all lanes (0 shifted in = identity)
__device__ __forceinline__ uint32_t wave_or_to_lane63(uint32_t x) {
    int v = (int)x;
    v |= __builtin_amdgcn_update_dpp(0, v, 0x111, 0xF, 0xF, false);     // row_shr:1
    v |= __builtin_amdgcn_update_dpp(0, v, 0x112, 0xF, 0xF, false);     // row_shr:2
    v |= __builtin_amdgcn_update_dpp(0, v, 0x114, 0xF, 0xF, false);     // row_shr:4
    v |= __builtin_amdgcn_update_dpp(0, v, 0x118, 0xF, 0xF, false);     // row_shr:8
    v |= __builtin_amdgcn_update_dpp(0, v, 0x142, 0xA, 0xF, false);     // row_bcast15 into rows 1 and 3
    v |= __builtin_amdgcn_update_dpp(0, v, 0x143, 0xC, 0xF, false);     // row_bcast31 into rows 2 and 3
    return (uint32_t)v;
}

// a GCell as ONE 16-byte scalar load (the struct's fields would be fetched one s_load_dword(x2) each)
typedef int32_t gcell_words __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(4))) gcell_words* ConstGCellPtr;
__device__ __forceinline__ double gcell_b(const gcell_words c) { return __hiloint2double(c.y, c.x); }

// Space.get_possible_position (space.py:98-129) for `item` on the tile in LDS: posZmap where naiveMask is set goes to
// `zdst` ([R][AC]; global memory in the transition kernel, where the emit kernel reads the rows it needs, LDS in the
// heuristic kernel, which asks for `dense`: 1e3 everywhere else), naiveMask itself to L.vbits (bit rows), height-level
// codes to L.lev, the levels present to L.present; returns np.sum(naiveMask).
// PATH: the overlap path compiled in -- one of the three, so that a transition kernel carries (and allocates registers
// for) only the path its data set takes, or PATH_ANY: decided at run time from Params (heuristic kernel, fallback build).
enum OverlapPath : int { PATH_ANY = 0, PATH_BLOCK = 1, PATH_BOX = 2, PATH_GENERIC = 3,
                         PATH_MIXED = 4 };   // block loop for the rotations of Params::block_rots, cell lists for the others
constexpr int GENERIC_TICKET = 40;               // word of Lds::redi that deals the generic path's tasks
// One footprint cell list walked for G row groups at once (overlap_test's generic path): per cell one 16-byte scalar
// load (bottom height, byte offset in the tile), per row group one LDS read at lane base + offset, one subtract and
// one max; four cells (one 64-byte scalar load) per trip, two max chains per row group.
// The scalar load of the NEXT four cells is requested before the LDS reads of the current four and awaited after their
// max chains, so that its latency (about 300 cycles from L2, tools/microbench_issue.hip) runs beside the trip's own
// LDS and VALU work instead of in front of it.  The compiler cannot be talked into this order (it moves the request
// back to the head of the trip: scalar loads return out of order with LDS reads on one counter), hence the two
// asm statements: the request is invisible to the compiler's wait-count bookkeeping, which stays correct -- one more
// operation in flight than it knows of makes each of its LDS waits stricter, never looser -- and the explicit wait for
// everything precedes the first use of the requested cells.  The list storage is padded for the reads past the end.
typedef int32_t gcell_quad __attribute__((ext_vector_type(16)));
__device__ __forceinline__ gcell_quad gcell_request(ConstGCellPtr p) {
    gcell_quad q;
    asm volatile("s_load_dwordx16 %0, %1, 0x0" : "=&s"(q) : "s"(p));
    return q;
}
template <int G>
__device__ __forceinline__ void gcell_await(gcell_quad& q, double (&a0)[G], double (&a1)[G]) {
    // (the accumulators ride along so that the trip's arithmetic stays in front of the wait)
    if (G == 1) asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(q), "+v"(a0[0]), "+v"(a1[0]));
    else if (G == 2) asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(q), "+v"(a0[0]), "+v"(a1[0]), "+v"(a0[G > 1 ? 1 : 0]), "+v"(a1[G > 1 ? 1 : 0]));
    else asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(q), "+v"(a0[0]), "+v"(a1[0]), "+v"(a0[G > 1 ? 1 : 0]), "+v"(a1[G > 1 ? 1 : 0]),
                      "+v"(a0[G > 2 ? 2 : 0]), "+v"(a1[G > 2 ? 2 : 0]));
}
// One heightmap read of row group g at byte offset `off`.  STRIDE > 0: the row groups' lane bases are STRIDE bytes apart
// (an action grid whose rows are a power of two wide: a row group is 64 consecutive doubles of a plane), so a trip adds
// each cell's offset to ONE base and the groups differ by the instruction's immediate offset: 4 + 24 VALU instructions
// per trip of four cells and three groups instead of 12 + 24.  (volatile: left alone, the compiler pairs two groups'
// reads into ds_read2st64_b64, which takes 8 LDS cycles where two ds_read_b64 take 4 -- MI355X_MICROARCH.md, LDS table.
// The low word of a generic pointer into the LDS is its LDS byte address.)
template <int G, int STRIDE>
__device__ __forceinline__ double gcell_height(const char* const (&hb)[G], int g, int off) {
#if defined(__HIP_DEVICE_COMPILE__)
    if (STRIDE > 0) {
        typedef const volatile __attribute__((address_space(3))) double* LdsF64;
        return *(LdsF64)((uint32_t)(uintptr_t)hb[0] + off + g * STRIDE);
    }
#endif
    return *(const double*)(hb[STRIDE > 0 ? 0 : g] + off + (STRIDE > 0 ? g * STRIDE : 0));
}
template <int G, int STRIDE>
__device__ __forceinline__ void gcell_quad_apply(const gcell_quad& q, const char* const (&hb)[G], double (&a0)[G], double (&a1)[G]) {
    double h0[G], h1[G], h2[G], h3[G];
#pragma unroll
    for (int g = 0; g < G; ++g) {
        h0[g] = gcell_height<G, STRIDE>(hb, g, q[2]); h1[g] = gcell_height<G, STRIDE>(hb, g, q[6]);
        h2[g] = gcell_height<G, STRIDE>(hb, g, q[10]); h3[g] = gcell_height<G, STRIDE>(hb, g, q[14]);
    }
#pragma unroll
    for (int g = 0; g < G; ++g) {
        a0[g] = fmax(a0[g], h0[g] - __hiloint2double(q[1], q[0]));
        a1[g] = fmax(a1[g], h1[g] - __hiloint2double(q[5], q[4]));
    }
#pragma unroll
    for (int g = 0; g < G; ++g) {
        a0[g] = fmax(a0[g], h2[g] - __hiloint2double(q[9], q[8]));
        a1[g] = fmax(a1[g], h3[g] - __hiloint2double(q[13], q[12]));
    }
}
template <int G, int STRIDE>
__device__ inline void gcell_walk(ConstGCellPtr gc, int nb, const char* const (&hb)[G], double init, double (&z)[G]) {
    double a0[G], a1[G];
#pragma unroll
    for (int g = 0; g < G; ++g) a0[g] = a1[g] = init;
    gcell_quad A = gcell_request(gc), B;
    gcell_await<G>(A, a0, a1);
    int e = 0;                                               // A holds cells e .. e + 3
    bool in_a = true;
    while (e + 4 <= nb) {
        B = gcell_request(gc + e + 4);
        gcell_quad_apply<G, STRIDE>(A, hb, a0, a1);
        gcell_await<G>(B, a0, a1);
        e += 4;
        in_a = false;
        if (e + 4 > nb) break;
        A = gcell_request(gc + e + 4);
        gcell_quad_apply<G, STRIDE>(B, hb, a0, a1);
        gcell_await<G>(A, a0, a1);
        e += 4;
        in_a = true;
    }
    if (e < nb) {                                            // the last one to three cells are in the quad already here
        const gcell_quad q = in_a ? A : B;
        const int left = nb - e;
#pragma unroll
        for (int g = 0; g < G; ++g) {
            a0[g] = fmax(a0[g], gcell_height<G, STRIDE>(hb, g, q[2]) - __hiloint2double(q[1], q[0]));
            if (left > 1) a1[g] = fmax(a1[g], gcell_height<G, STRIDE>(hb, g, q[6]) - __hiloint2double(q[5], q[4]));
            if (left > 2) a0[g] = fmax(a0[g], gcell_height<G, STRIDE>(hb, g, q[10]) - __hiloint2double(q[9], q[8]));
        }
    }
#pragma unroll
    for (int g = 0; g < G; ++g) z[g] = fmax(a0[g], a1[g]);
}

#endif  // IRBPP_PASS == 1
template <int PATH>
__device__ inline int overlap_test(const Params& P, const Tables& T, const State& S, const StepIO& io,
                                   const Lds& L, int b, int item, bool debug_out, double* zdst, bool sr_staged, bool dense) {
    const bool use_block = PATH == PATH_BLOCK || PATH == PATH_MIXED || (PATH == PATH_ANY && P.block_b > 0);
    const bool use_box = PATH == PATH_BOX || (PATH == PATH_ANY && P.box != 0);
    // rotations that take the block loop (all of them on pure lattice data); the others walk their cell lists below
    const uint32_t rots_all = (1u << P.R) - 1u;
    const uint32_t brots = PATH == PATH_BLOCK ? rots_all : ((PATH == PATH_MIXED || PATH == PATH_ANY) ? (uint32_t)P.block_rots : 0u);
    const bool use_lists = PATH == PATH_GENERIC || PATH == PATH_MIXED || (PATH == PATH_ANY && !use_box && brots != rots_all);
    const int tid = threadIdx.x;
    const int R = P.R, AC = P.AC, Ax = P.Ax, Ay = P.Ay;
    constexpr int SRW = sizeof(ShapeRot) / 4;                // ShapeRot as dwords
    int* srw = L.sr;                                         // all R ShapeRots of the item in ONE coalesced load
    if (item >= 0 && !sr_staged)                             // (the transition kernel may have them in place already)
#pragma unroll 1
        for (int t = tid; t < R * SRW; t += BLOCK) srw[t] = ((const int*)(T.sr + (size_t)item * R))[t];
    for (int rep = 0; rep < IRBPP_REPS(13); ++rep) {
    if (tid < R) L.present[tid] = 0ull;
    if (tid == 0) L.redi[GENERIC_TICKET] = WAVES;            // generic path: the first WAVES tasks are taken without a ticket
#pragma unroll 1
    for (int i = tid; i < R * 16; i += BLOCK) { L.vmask[i] = 0u; L.vbits[i] = 0u; }
#pragma unroll 1
    for (int i = tid; i < (R * AC + 3) / 4; i += BLOCK) ((uint32_t*)L.lev)[i] = 0xFFFFFFFFu;     // 255: no level
    }
    if (dense) for (int i = tid; i < R * AC; i += BLOCK) zdst[i] = 1e3;
    int bl_first = 0, bl_total = 0;
    Cell bl_entry = {};
    if (use_block)
    for (int rep = 0; rep < IRBPP_REPS(6); ++rep) {
        if (rep) __syncthreads();
        // Block-max grid of the current tile, in two steps.  (1) the maximum of every action cell's own step x step cells:
        // in the phase-plane layout those are entry X*Ay + Y of every plane -- step^2 conflict-free reads at one index.
        // (2) a b x b block is (b / step)^2 neighbouring action cells.  (Reading the b^2 cells of every block straight from
        // the planes took 16 scattered reads with their index arithmetic per block at b = 4, step = 2.)
        const int planes = P.pp * P.pp, mq = P.block_b / P.step;
#pragma unroll 1
        for (int t = tid; t < AC; t += BLOCK) {
            double m = L.hm[t];
            for (int pl = 1; pl < planes; ++pl) m = fmax(m, L.hm[pl * P.PL + t]);
            L.c2[t] = m;
        }
        __syncthreads();
        // The item's block lists -- all rotations, contiguous in the table, a few dozen 16-byte entries -- are requested now, a
        // thread per entry: they arrive during the grid's second step and go to LDS behind it (the bytes of c2, dead by then),
        // where the rotation loops read them with ONE broadcast ds_read_b128 per entry instead of three v_readlane.
        if (rep == 0 && item >= 0) {
            const ShapeRot* sa = (const ShapeRot*)srw;
            bl_first = __builtin_amdgcn_readfirstlane(sa->oblk);
            bl_total = __builtin_amdgcn_readfirstlane(sa[R - 1].oblk) + __builtin_amdgcn_readfirstlane(sa[R - 1].nblk) - bl_first;
            if (bl_total <= AC / 2 && tid < bl_total) bl_entry = T.blkcell[bl_first + tid];
        }
#pragma unroll 1
        for (int t = tid; t < P.mb_h * P.mb_w; t += BLOCK) {
            const int pi = fdiv(t, P.mb_w, P.mg_mbw), pj = t - pi * P.mb_w;
            const double* c = L.c2 + pi * Ay + pj;
            double m = -1e300;
            for (int i = 0; i < mq; ++i)
                for (int j = 0; j < mq; ++j) m = fmax(m, c[i * Ay + j]);
            L.mb[t] = m;
        }
    }
    __syncthreads();
    const bool bl_lds = use_block && item >= 0 && bl_total <= AC / 2;        // (block-uniform)
    typedef int32_t bl_words __attribute__((ext_vector_type(4)));
    bl_words* const bl = (bl_words*)L.c2;                                    // [bl_total] (bottom height, byte offset in the block-max grid)
    if (bl_lds) {
        if (tid < bl_total) {
            bl_words w;
            w.x = __double2loint(bl_entry.v); w.y = __double2hiint(bl_entry.v); w.z = bl_entry.ij * 8; w.w = 0;
            bl[tid] = w;
        }
        __syncthreads();
    }

    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int my_valid = 0;
    if (use_block || use_box) {
    // ---- one action cell per thread, all rotations: block path (footprint = list of uniform b x b blocks over the
    // block-max grid) or box path (footprint = one solid box: separable rectangle maximum) ---------------------------
    // (the two-wave build, BLOCK = 128: two cells per thread, one after the other; block path only)
    constexpr int CELLS = BLOCK >= 256 ? 1 : 256 / BLOCK;
    if (CELLS > 1 && use_box) __builtin_trap();
#pragma unroll 1
    for (int ci = 0; ci < CELLS; ++ci) {
    const int cell = CELLS == 1 ? tid : tid + ci * BLOCK;
    const int X = fdiv(cell, Ay, P.mg_ay), Y = cell - X * Ay;
    double zs[8];
    bool vs[8];
    if (use_box) {
        // max over the bx x by window of the heightmap, rows first: m1[i][Y] = max_j H[i][Y*step + j], j < by, then
        // z[X][Y] = max_i m1[X*step + i][Y] - bc, i < bx: (bx + by) reads per cell where the pair loop has bx * by
        double* const m1 = L.m1;
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            zs[r] = 1e3;
            vs[r] = false;
            if (r >= R || item < 0) continue;
            const ShapeRot* sp = (const ShapeRot*)srw + r;
            const int s_ax = __builtin_amdgcn_readfirstlane(sp->ax), s_ay = __builtin_amdgcn_readfirstlane(sp->ay);
            const int bx = __builtin_amdgcn_readfirstlane(sp->bx), by = __builtin_amdgcn_readfirstlane(sp->by);
            const int has_out = __builtin_amdgcn_readfirstlane(sp->has_out);
            const double ext_z_r = sp->ext_z_r, bc = sp->bc;
            for (int idx = tid; idx < P.Hx * Ay; idx += BLOCK) {
                const int i = fdiv(idx, Ay, P.mg_ay), yy = idx - i * Ay;
                if (yy <= Ay - s_ay) {
                    const double* rowp = L.hm + tile_row_part(P, i);
                    double m = -1e300;
                    for (int j = 0; j < by; ++j) m = fmax(m, rowp[tile_col_part(P, yy * P.step + j)]);
                    m1[idx] = m;
                }
            }
            __syncthreads();
            const bool in_range = cell < AC && X <= Ax - s_ax && Y <= Ay - s_ay;
            if (in_range) {
                const double* colp = m1 + X * P.step * Ay + Y;
                double m = -1e300;
                for (int i = 0; i < bx; ++i) m = fmax(m, colp[i * Ay]);
                m = m - bc;                                      // max(H - c) == max(H) - c: rounding is monotone
                if (has_out) m = fmax(m, 0.0);                   // masked-out cells of the table: (H - B) * 0
                zs[r] = m;
                vs[r] = round6_scaled(m + ext_z_r - P.bin_z) <= 0.0;     // np.round(.,6) <= 0 (space.py:120)
            }
            __syncthreads();                                     // the next rotation rewrites m1
        }
    } else {
    // (One coalesced load of all R rotations' block lists -- they are contiguous and total a few dozen entries -- ahead of
    // the block-max grid, instead of this chain of prefetches, was measured: 28.25 vs 28.42 M steps/s, not taken.)
    int ncell_next = 0, off_next = 0;
    Cell pre = {};                                           // first cell chunk of the next rotation, in flight
    if (bl_lds) {
    for (int rep = 0; rep < IRBPP_REPS(2); ++rep) {
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        zs[r] = 1e3;
        vs[r] = false;
        if (r >= R || !((brots >> r) & 1u)) continue;
        const ShapeRot* sp = (const ShapeRot*)srw + r;
        const int s_ax = __builtin_amdgcn_readfirstlane(sp->ax), s_ay = __builtin_amdgcn_readfirstlane(sp->ay);
        const int has_out = __builtin_amdgcn_readfirstlane(sp->has_out);
        const int ne = __builtin_amdgcn_readfirstlane(sp->nblk), e0 = __builtin_amdgcn_readfirstlane(sp->oblk) - bl_first;
        const double ext_z_r = sp->ext_z_r;
        const bool in_range = cell < AC && X <= Ax - s_ax && Y <= Ay - s_ay;
        if (in_range) {
            const char* const hb = (const char*)(L.mb + X * P.mb_w + Y);
            const bl_words* const be = bl + e0;                       // (one address for the whole wave: a broadcast read)
            double m = has_out ? 0.0 : -1e300;
            int u = 0;
            for (; u + 4 <= ne; u += 4) {
                bl_words q[4];
                double hv[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) q[k] = be[u + k];
#pragma unroll
                for (int k = 0; k < 4; ++k) hv[k] = *(const double*)(hb + q[k].z);
#pragma unroll
                for (int k = 0; k < 4; ++k) m = fmax(m, hv[k] - __hiloint2double(q[k].y, q[k].x));
            }
            for (; u < ne; ++u) {
                const bl_words q = be[u];
                m = fmax(m, *(const double*)(hb + q.z) - __hiloint2double(q.y, q.x));
            }
            zs[r] = m;
            vs[r] = round6_scaled(m + ext_z_r - P.bin_z) <= 0.0;     // np.round(.,6) <= 0 (space.py:120)
        }
    }
    }
    } else
    for (int rep = 0; rep < IRBPP_REPS(2); ++rep) {
    if (item >= 0) {
        const ShapeRot* s0 = (const ShapeRot*)srw;
        ncell_next = __builtin_amdgcn_readfirstlane(s0->nblk);
        off_next = __builtin_amdgcn_readfirstlane(s0->oblk);
        const Cell* c0 = T.blkcell + off_next;
        if (ncell_next > 0) pre = c0[lane < ncell_next ? lane : ncell_next - 1];
    }
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        zs[r] = 1e3;
        vs[r] = false;
        if (r >= R || item < 0) continue;
        const ShapeRot* sp = (const ShapeRot*)srw + r;
        const int s_ax = __builtin_amdgcn_readfirstlane(sp->ax), s_ay = __builtin_amdgcn_readfirstlane(sp->ay);
        const int has_out = __builtin_amdgcn_readfirstlane(sp->has_out);
        const double ext_z_r = sp->ext_z_r;
        const int ncell = ((brots >> r) & 1u) ? ncell_next : 0, off0 = off_next;     // (a list rotation has no blocks: nblk = 0 anyway)
        Cell c = pre;
        if (r + 1 < R) {                                     // issue the next rotation's first chunk now
            const ShapeRot* sn = sp + 1;
            ncell_next = __builtin_amdgcn_readfirstlane(sn->nblk);
            off_next = __builtin_amdgcn_readfirstlane(sn->oblk);
            const Cell* cn = T.blkcell + off_next;
            if (ncell_next > 0) pre = cn[lane < ncell_next ? lane : ncell_next - 1];
        }
        const bool in_range = cell < AC && X <= Ax - s_ax && Y <= Ay - s_ay;
        const double* h0 = L.mb + X * P.mb_w + Y;
        double m = has_out ? 0.0 : -1e300;
        // The block list is fetched 64 entries at a time, one 16-byte entry per lane (a coalesced
        // vector load), and broadcast entry by entry with v_readlane: the loop then has only LDS reads
        // in flight, which the hardware returns in order and the compiler can pipeline.
        const Cell* cells = T.blkcell + off0;
        for (int base = 0; base < ncell; base += 64) {
            if (base > 0) {
                const int idx = base + lane < ncell ? base + lane : ncell - 1;
                c = cells[idx];
            }
            const int cnt = ncell - base < 64 ? ncell - base : 64;
            int c_off = c.ij;
            int c_lo = (int)__double2loint(c.v), c_hi = (int)__double2hiint(c.v);
            // every lane must really hold its entry (v_readlane reads lanes that are masked off
            // below): keep the compiler from sinking the load into the in_range branch
            asm volatile("" : "+v"(c_off), "+v"(c_lo), "+v"(c_hi));
            if (in_range) {
                int u = 0;
                for (; u + 8 <= cnt; u += 8) {                  // 8 LDS reads in flight per trip
                    double hv[8], vv[8];
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        const int off = __builtin_amdgcn_readlane(c_off, u + k);
                        vv[k] = __hiloint2double(__builtin_amdgcn_readlane(c_hi, u + k),
                                                 __builtin_amdgcn_readlane(c_lo, u + k));
                        hv[k] = h0[off];
                    }
#pragma unroll
                    for (int k = 0; k < 8; ++k) m = fmax(m, hv[k] - vv[k]);
                }
                for (; u < cnt; ++u) {
                    const int off = __builtin_amdgcn_readlane(c_off, u);
                    const double v = __hiloint2double(__builtin_amdgcn_readlane(c_hi, u),
                                                      __builtin_amdgcn_readlane(c_lo, u));
                    m = fmax(m, h0[off] - v);
                }
            }
        }
        if (in_range && ((brots >> r) & 1u)) {
            zs[r] = m;
            vs[r] = round6_scaled(m + ext_z_r - P.bin_z) <= 0.0;     // np.round(.,6) <= 0 (space.py:120)
        }
    }
    }
    }
    int level_code[8];
    for (int rep = 0; rep < IRBPP_REPS(7); ++rep) {
    if (rep) my_valid = 0;
    const bool rows_align = (1 << P.g_ysh) == Ay;            // a wave's 64 consecutive cells are whole rows of the action grid
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        level_code[r] = 255;
        if (r >= R || (!use_box && !((brots >> r) & 1u))) continue;      // (a list rotation: below)
        const double z = zs[r];
        const bool valid = vs[r];
        if (cell < AC) {
            if (debug_out) {
                const KernArgsPtr ka = cold_args();
                ka->io.posz_out[((size_t)b * R + r) * AC + cell] = z;
                ka->io.mask_out[((size_t)b * R + r) * AC + cell] = valid ? 1 : 0;
            }
            int code = 255;
            if (valid) {
                if (!debug_out) zdst[r * AC + cell] = z;                // (irbpp_possible_position leaves the last observation's hand-over alone: the next step reads its drop height there)
                const int li = np_floor_divide_int(z, P.res_z, P.inv_res_z);   // cvTools.py:78
                if (li != -1) {                                        // level -1 is skipped (cvTools.py:84)
                    const int idx = li + 32;
                    if (idx < 0 || idx > 63) raise_error(S, IRBPP_DEVERR_LEVEL_RANGE);
                    else code = idx;
                }
                ++my_valid;
                L.lev[r * AC + cell] = (uint8_t)code;
                if (!rows_align) atomicOr(&L.vbits[r * 16 + X], 1u << Y);
            }
            level_code[r] = code;
        }
        if (rows_align) {                                    // naiveMask bit rows straight from the ballot
            const unsigned long long bal = __ballot(valid && cell < AC);
            if (Y == 0 && cell < AC) L.vbits[r * 16 + X] = (uint32_t)(bal >> (lane & ~(Ay - 1))) & ((1u << Ay) - 1u);
        }
    }
    // presence masks: one LDS atomic per distinct level per wave (64 lanes ORing into one word would
    // be serialised by the LDS lane by lane)
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        if (r >= R || (!use_box && !((brots >> r) & 1u))) continue;
        const int code = level_code[r];
        unsigned long long todo = __ballot(code != 255), bits = 0ull;
        while (todo) {
            const int c = __builtin_amdgcn_readlane(code, __ffsll((long long)todo) - 1);
            bits |= 1ull << c;
            todo &= ~__ballot(code == c);
        }
        if ((tid & 63) == 0 && bits) atomicOr(&L.present[r], bits);
    }
    }
    }       // cells of this thread
    }
    if (use_lists) {
    // ---- generic path: ONE action cell per lane, and only cells that can be in range.  A wave task is (rotation r,
    // row group q): the wave's lanes are 64 >> ysh consecutive rows X of the action grid times 1 << ysh >= Ay
    // columns Y, and only the rows X <= Ax - ax_r get a task at all.  The lane walks the rotation's masked-in bottom
    // cells: per cell one scalar load (bottom height + byte offset in the tile, wave-uniform), one LDS read of the
    // heightmap at lane base + offset, one subtract, one max -- the pairs (action cell, footprint cell) that
    // np.max((H - B) * mask) ranges over, nothing else.  In the tile (phase planes of period step) lane (X, Y) sits
    // at entry X*Ay + Y of every plane, so the 32 lanes of a half-wave read 32 consecutive-or-disjoint bank pairs:
    // conflict-free ds_read_b64.  A lane outside the grid or outside the rotation's range reads some float64 of the
    // workgroup's LDS (or nothing: beyond the allocation reads return 0) and is discarded.
    const int ysh = P.g_ysh, rpw = 64 >> ysh;
    const int rs = lane >> ysh, Y = lane & ((1 << ysh) - 1);
    const bool rows_whole = (1 << ysh) == Ay;                // wave-uniform
    // Register blocking over row groups: a task is up to three consecutive row groups of one rotation, walked
    // together, so that one scalar load of a footprint cell (16 bytes, the wave-uniform operand) serves up to 192
    // action cells instead of 64.  The scalar side was the loop's bound: a CU gets one 64-byte scalar load per 14-18
    // cycles (tools/microbench_issue.hip: 3.5-4.6 bytes per cycle per CU, cache hit or L2), i.e. 4 cycles per cell
    // and wave, against 2.2 for the LDS read and 3.3 for the three VALU instructions.  Blocked three deep the VALU
    // work is the largest term.  (Walking two INDEPENDENT lists in one loop instead -- same scalar traffic, twice the
    // work per wait -- changed nothing: abc_fine 5.4 -> 5.1 M.)  On grids whose rows are a power of two wide the row
    // groups share one address add per cell (gcell_height): 28 instead of 36 VALU instructions per trip.  That form
    // LOST 10 % while the loop still waited for its scalar loads one trip at a time (abc_fine 6.67 -> 6.0 M, cause not
    // found: neither LDS cycles nor conflicts differ) and wins 2-3 % now that the loads are pipelined (7.12 -> 7.34 M,
    // general 14.6 -> 14.9 M; SQ_INSTS_VALU of the kernel -14 %).
    // A rotation's groups are split evenly over ceil(groups / gmax) tasks; gmax drops when that would leave waves
    // without a task.
    // The task table lives in the lanes of every wave, one rotation per lane (lanes 0 .. 7): row groups in range, tasks, tasks
    // before the rotation.  (As three unrolled 8-element arrays in scalar registers with their select chains it cost the
    // capped build most of its 137 SGPR spills.)
    int my_grp = 0;
    if (lane < R && item >= 0 && !((brots >> lane) & 1u)) {            // (PATH_MIXED: the block loop above served the rotations of brots)
        const ShapeRot* sp = (const ShapeRot*)srw + lane;
        const int wx = Ax - sp->ax + 1, wy = Ay - sp->ay + 1;
        if (wx > 0 && wy > 0) my_grp = (wx + rpw - 1) >> (6 - ysh);
    }
    // (my_grp <= 4 -- an action grid is at most 256 cells, four waves' worth -- so the ceilings are spelled out)
    const int t3 = my_grp > 3 ? 2 : (my_grp > 0 ? 1 : 0), t2 = (my_grp + 1) >> 1;
    const int n3 = __builtin_amdgcn_readlane(wave_inclusive_sum(t3), 63), n2 = __builtin_amdgcn_readlane(wave_inclusive_sum(t2), 63);
    const int gmax = n3 >= WAVES ? 3 : (n2 >= WAVES ? 2 : 1);
    const int my_tasks = gmax == 3 ? t3 : (gmax == 2 ? t2 : my_grp);
    const int my_incl = wave_inclusive_sum(my_tasks), my_first = my_incl - my_tasks;          // tasks before this lane's rotation
    const int ntask = __builtin_amdgcn_readlane(my_incl, 63);
    auto task_rotation = [&](int t) {                        // the last rotation whose tasks start at or before t
        return __popcll(__ballot(lane >= 1 && lane < 8 && t >= my_first));
    };
    int pref = 0;
    uint32_t task_lo = 0u, task_hi = 0u;                     // level bits of the lane's cells in the current task
    auto task_finish = [&](int r, int X, int s_ax, int s_ay, double ext_z_r, double z) {
        const bool in_range = X <= Ax - s_ax && Y <= Ay - s_ay;
        const bool valid = in_range && round6_scaled(z + ext_z_r - P.bin_z) <= 0.0;     // np.round(.,6) <= 0 (space.py:120)
        const int cell = X * Ay + Y;
        if (debug_out && in_range) {
            const KernArgsPtr ka = cold_args();
            ka->io.posz_out[((size_t)b * R + r) * AC + cell] = z;
            ka->io.mask_out[((size_t)b * R + r) * AC + cell] = valid ? 1 : 0;
        }
        uint32_t bits_lo = 0u, bits_hi = 0u;                 // my level code as a bit
        if (valid) {
            if (!debug_out) zdst[r * AC + cell] = z;
            const int li = np_floor_divide_int(z, P.res_z, P.inv_res_z);   // cvTools.py:78
            if (li != -1) {                                        // level -1 is skipped (cvTools.py:84)
                const int idx = li + 32;
                if (idx < 0 || idx > 63) raise_error(S, IRBPP_DEVERR_LEVEL_RANGE);
                else {
                    L.lev[r * AC + cell] = (uint8_t)idx;
                    if (idx < 32) bits_lo = 1u << idx; else bits_hi = 1u << (idx - 32);
                }
            }
            ++my_valid;
        }
        // naiveMask bit rows from the ballot: lane (row slot rs, Y == 0) stores its row's word
        const unsigned long long bal = __ballot(valid);
        if (Y == 0 && X < Ax) L.vbits[r * 16 + X] = (uint32_t)(bal >> (rs << ysh)) & ((1u << Ay) - 1u);
        // presence mask of the rotation: the lane's bits join those of the task's other row groups (task_presence below)
        task_lo |= bits_lo;
        task_hi |= bits_hi;
    };
    // ... OR over the wave on the DPP network once per TASK (up to three row groups), LDS atomics by the wave's last lane (a
    // rotation can have several tasks)
    auto task_presence = [&](int r) {
        const uint32_t lo = wave_or_to_lane63(task_lo), hi = wave_or_to_lane63(task_hi);
        if (lane == 63) {
            uint32_t* pw = (uint32_t*)&L.present[r];
            if (lo) atomicOr(pw, lo);
            if (hi) atomicOr(pw + 1, hi);
        }
        task_lo = task_hi = 0u;
    };
    auto prefetch_list = [&](int ob, int nb) {
        const char* lv = (const char*)(T.gcell + ob);
        for (int o = lane * 128; o < nb * 16; o += 64 * 128) pref |= *(const int*)(lv + o);
    };
    for (int t = wave, mine = 0; ; ++mine) {
        // Tasks are dealt by a ticket in LDS: a wave that drew a short list (or few row groups) comes back for the next task
        // while the others are still walking (the static deal t = wave, wave + 4, ... was 8 % off an even split).  The first
        // task of a wave is its own number: no round trip before the first walk.
        if (mine > 0) {
            int tk = 0;
            if (lane == 0) tk = atomicAdd(&L.redi[GENERIC_TICKET], 1);
            t = __builtin_amdgcn_readfirstlane(tk);
        }
        if (t >= ntask) break;
        const int r = task_rotation(t);
        const int tq = t - __builtin_amdgcn_readlane(my_first, r), ng = __builtin_amdgcn_readlane(my_grp, r);
        const int nt = __builtin_amdgcn_readlane(my_tasks, r);
        // even split of ng groups over nt tasks: the first ng % nt tasks take one more
        const int base = nt == 1 ? ng : (nt == ng ? 1 : ng >> 1), rem = ng - base * nt;   // (nt is 1, 2 or ng)
        const int G = base + (tq < rem ? 1 : 0), g0 = tq * base + (tq < rem ? tq : rem);
        const ShapeRot* sp = (const ShapeRot*)srw + r;
        const int s_ax = __builtin_amdgcn_readfirstlane(sp->ax), s_ay = __builtin_amdgcn_readfirstlane(sp->ay);
        const int has_out = __builtin_amdgcn_readfirstlane(sp->has_out);
        const int nb = __builtin_amdgcn_readfirstlane(sp->nb), ob = __builtin_amdgcn_readfirstlane(sp->ob);
        const double ext_z_r = sp->ext_z_r;
        const ConstGCellPtr gc = (ConstGCellPtr)(unsigned long long)(T.gcell + ob);
        // The lists (tens of MB per dataset: beyond L2, and what is in L2 is flushed by every step's observation
        // stores) are consumed by scalar loads, of which a wave has only one chunk in flight: one vector load per
        // 128-byte line, issued a task ahead (the wave's first task: up front), brings a list into the XCD's L2 so that
        // the scalar loads wait for L2 instead of HBM.  (Requesting all of the next item's lists earlier still -- beside
        // its ShapeRots, before the placement is applied -- was measured and is worse: abc_fine 7.1 -> 6.9 M, general
        // 14.6 -> 14.1 M; the lines do not survive in L2 until they are needed, and the kept-alive register costs the
        // kernel its seventh wave per SIMD.)
        if (mine == 0) prefetch_list(ob, nb);
        if (t + WAVES < ntask) {                             // (whoever draws it: the list is pulled into this die's L2)
            const int rn = task_rotation(t + WAVES);
            if (rn != r) {
                const ShapeRot* sn = (const ShapeRot*)srw + rn;
                prefetch_list(__builtin_amdgcn_readfirstlane(sn->ob), __builtin_amdgcn_readfirstlane(sn->nb));
            }
        }
        const double init = has_out ? 0.0 : -1e300;
        const int yc = Y < Ay ? Y : Ay - 1;
        const int X0 = g0 * rpw + rs;
        auto lane_base = [&](int X) { return (const char*)(L.hm + (X < Ax ? X : Ax - 1) * Ay + yc); };
        // whole rows per wave (Ay a power of two): the row groups' bases are 512 bytes apart and need no clamp -- a lane
        // outside the grid reads some float64 of the workgroup's LDS, or 0 beyond its allocation, and is discarded
        double z[3];
        if (rows_whole) {
            const char* h1[1] = {(const char*)(L.hm + X0 * Ay + Y)};
            const char* h2[2] = {h1[0], nullptr};
            const char* h3[3] = {h1[0], nullptr, nullptr};
            if (G == 1) gcell_walk<1, 512>(gc, nb, h1, init, (double(&)[1])z[0]);
            else if (G == 2) gcell_walk<2, 512>(gc, nb, h2, init, (double(&)[2])z[0]);
            else gcell_walk<3, 512>(gc, nb, h3, init, z);
        } else {
            const char* h1[1] = {lane_base(X0)};
            const char* h2[2] = {h1[0], lane_base(X0 + rpw)};
            const char* h3[3] = {h1[0], h2[1], lane_base(X0 + 2 * rpw)};
            if (G == 1) gcell_walk<1, 0>(gc, nb, h1, init, (double(&)[1])z[0]);
            else if (G == 2) gcell_walk<2, 0>(gc, nb, h2, init, (double(&)[2])z[0]);
            else gcell_walk<3, 0>(gc, nb, h3, init, z);
        }
        task_finish(r, X0, s_ax, s_ay, ext_z_r, z[0]);
        if (G > 1) task_finish(r, X0 + rpw, s_ax, s_ay, ext_z_r, z[1]);
        if (G > 2) task_finish(r, X0 + 2 * rpw, s_ax, s_ay, ext_z_r, z[2]);
        task_presence(r);
    }
    asm volatile("" :: "v"(pref));                           // the prefetch loads are complete by now; nothing uses their data
    if (debug_out)                                           // posZmap / naiveMask outside every rotation's range
        for (int i = tid; i < R * AC; i += BLOCK) {
            const int r = fdiv(i, AC, P.mg_ac), cell = i - r * AC, X = fdiv(cell, Ay, P.mg_ay), Y = cell - X * Ay;
            const ShapeRot* sp = (const ShapeRot*)srw + r;
            if (((brots >> r) & 1u) == 0u && (item < 0 || X > Ax - sp->ax || Y > Ay - sp->ay)) {
                const KernArgsPtr ka = cold_args();
                ka->io.posz_out[((size_t)b * R + r) * AC + cell] = 1e3;
                ka->io.mask_out[((size_t)b * R + r) * AC + cell] = 0;
            }
        }
    }
    return block_sum_int(my_valid, L.redi);                  // np.sum(naiveMask) for prejudge
}

__device__ inline void emit_observation(const Params& P, const State& S, const StepIO& io, const Lds& L, int b, int item,
                                        int nvalid, float* obs, const double* zsrc, const uint32_t* gvalid);
template <int IPT>
__device__ inline void split_handover(const Params& P, const State& S, const Lds& L, int b, int slot, int item, int nvalid);

// ---------------------------------------------------------------------------------------
// Location observation for `item` on the heightmap tile in LDS (binPhy.py:188-227).
// ---------------------------------------------------------------------------------------
template <int PATH, bool CHAIN = false>
__device__ inline void observe_location(const Params& P, const Tables& T, const State& S, const StepIO& io,
                                        const Lds& L, int b, int item, float* obs, bool debug_out, bool sr_staged,
                                        unsigned char* chain_lds = nullptr) {
    item = __builtin_amdgcn_readfirstlane(item);     // block-uniform: footprint reads become scalar loads
    const int tid = threadIdx.x;
    const int R = P.R, AC = P.AC, Ax = P.Ax, Ay = P.Ay;
    const int X = fdiv(tid, Ay, P.mg_ay), Y = tid - X * Ay;
    const int nvalid = IRBPP_HERE overlap_test<PATH>(P, T, S, io, L, b, item, debug_out, S.w_posz + (size_t)b * R * AC, sr_staged, false);
    if (debug_out) return;
    // the tile is done with: write its float32 copy and the item vector now, because the
    // contour scratch and the candidate keys reuse the tile's LDS
    if (tid < 9) obs[5 * P.S + tid] = tid == 0 ? (float)item : 0.0f;
    for (int rep = 0; rep < IRBPP_REPS(8); ++rep)
    for (TileWalk w = IRBPP_HERE tile_walk_begin(P, tid); w.lin < P.Hc; tile_walk_next(w)) obs[5 * P.S + 9 + w.lin] = (float)L.hm[tile_walk_index(P, w)];
    __syncthreads();
    stamp(io, b, 2);

    if (io.phase_cycles && tid == 0)
        for (int k = 5; k < PHASE_ROW; ++k)
            if (k < 8 || k > 10) io.phase_cycles[(size_t)b * PHASE_ROW + k] = 0;
    if constexpr (CHAIN) {
        // Small launches (irbpp_capi.hip: chain_launch): the bin's own workgroup finishes the observation -- level images, border
        // following, approxPolyDP + convexity (contour_stage: the hull kernel's in-workgroup form of the same routines), then the
        // candidate rows (emit_observation) -- with LDS hand-over: ONE launch per observation instead of four, where the chip has
        // idle CUs anyway and the step is a chain of launch latencies.  Same results: same routines on the same level codes.
        {   // naiveMask's bit rows: what the next apply looks its drop height up with, and the no-candidate fallback below
            // (they share their LDS bytes with the task index the contour stage builds next)
            uint32_t* gb = S.w_valid + (size_t)b * P.R * 16;
#pragma unroll 1
            for (int i = tid; i < P.R * 16; i += BLOCK) gb[i] = L.vbits[i];
        }
        __syncthreads();
        ::irbpp::contour_stage(P, S, L, nullptr);
        // posZmap (w_posz) and the bit rows were stored by this workgroup's threads and are read back by others below: workgroup
        // scope (one CU, one L1: the release is a wait for the stores)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __syncthreads();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        Lds Le = L;
        Le.img = (uint16_t*)chain_lds;               // radix counters / sort keys of a > S selection: behind the transition kernel's own carve-up
        ::irbpp::emit_observation(P, S, io, Le, b, item, nvalid, obs, S.w_posz + (size_t)b * R * AC, S.w_valid + (size_t)b * R * 16);
        return;
    }
    // the trace and emit kernels take it from here
    IRBPP_HERE split_handover<(CONTOUR_IPT * 256) / BLOCK>(P, S, L, b, (int)blockIdx.x + io.block_off, item, nvalid);    // (32 level images per batch)
    if (io.phase_cycles && tid == 0)             // tooling: cycles of the hand-over (images, candidates, stores)
        io.phase_cycles[(size_t)b * PHASE_ROW + 5] = (long long)clock64() - io.phase_cycles[(size_t)b * PHASE_ROW + 2];
}

#if IRBPP_PASS == 1
// ---------------------------------------------------------------------------------------
// The `want` smallest of n <= R*AC values in (value, position) order -- np.argsort(...)[:S] with ties by
// ascending position (binPhy.py:209-212, 217-225) -- without the n^2 ranking of everything against
// everything: (1) radix select on the order-preserving 64-bit image of the float64 values finds the
// want-th smallest value T in eight histogram rounds, (2) the elements below T plus the first few equal to
// T are compacted in position order, (3) only those `want` elements are sorted.
// Element e is the candidate key(e) = rot<<16 | lx<<8 | ly with value zsrc[rot, lx, ly] (posZValid, read where it
// lies in global memory: this path is rare); out[rank] = its key.
// `sel` ([n] words, may be the array key() reads) and `hist` ([256] words) are LDS.  All threads call it.
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned long long sortable_f64(double v) {
    const unsigned long long b = (unsigned long long)__double_as_longlong(v);
    return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
constexpr int SEL_PER_THREAD = (8 * 256 + BLOCK - 1) / BLOCK;           // R*AC <= 2048 elements

template <typename KEY>
__device__ inline void select_smallest(const Params& P, const Lds& L, const double* zsrc, const uint32_t* vbits, int n, int want,
                                       const KEY& key, uint32_t* out, uint32_t* sel, uint32_t* hist) {
    const int tid = threadIdx.x;
    // posZValid of element k: posZmap where naiveMask is set (only there has the transition kernel written it), 1e3
    // elsewhere (space.py:123-125).  vbits == nullptr: every element is a candidate, i.e. a valid cell.
    auto value = [&](uint32_t k) {
        const uint32_t r = k >> 16, x = (k >> 8) & 255u, y = k & 255u;
        if (vbits != nullptr && !((vbits[r * 16 + x] >> y) & 1u)) return 1e3;
        return zsrc[r * P.AC + x * P.Ay + y];
    };
    unsigned long long sk[SEL_PER_THREAD];
    uint32_t ky[SEL_PER_THREAD];
#pragma unroll
    for (int k = 0; k < SEL_PER_THREAD; ++k) {
        const int e = tid + k * BLOCK;
        ky[k] = e < n ? key(e) : 0u;
        sk[k] = e < n ? sortable_f64(value(ky[k])) : ~0ull;
    }
    // (1) the want-th smallest value, one byte per round from the top
    unsigned long long prefix = 0ull;
    int remaining = want;
    for (int d = 7; d >= 0; --d) {
        hist[tid] = 0u;                                                  // BLOCK == 256 bins
        __syncthreads();
        // posZ values of one bin share their high bytes, so in most rounds every lane of a wave counts into the SAME bin:
        // plain LDS atomics would be serialised lane by lane (2048 of them on one address, ~8 k cycles per round,
        // which made this path set the emit kernel's duration).  A wave whose lanes agree on the digit adds its count
        // once; a few distinct digits per wave are conflicts the LDS takes in its stride.
#pragma unroll
        for (int k = 0; k < SEL_PER_THREAD; ++k) {
            if (k * BLOCK >= n) continue;                                    // block-uniform
            const int e = tid + k * BLOCK;
            const bool in = e < n && (d == 7 || (sk[k] >> (8 * (d + 1))) == prefix);
            const int digit = (int)((uint32_t)(sk[k] >> (8 * d)) & 255u);
            const unsigned long long act = __ballot(in);
            if (act == 0ull) continue;
            const int c0 = __builtin_amdgcn_readlane(digit, __ffsll((long long)act) - 1);
            if (__ballot(in && digit != c0) == 0ull) {
                if ((tid & 63) == __ffsll((long long)act) - 1) atomicAdd(&hist[c0], (uint32_t)__popcll(act));
            } else if (in) {
                atomicAdd(&hist[digit], 1u);
            }
        }
        __syncthreads();
        if (tid < 64) {                                                   // one wave scans the 256 counts
            int c[4], sum = 0;
#pragma unroll
            for (int i = 0; i < 4; ++i) { c[i] = (int)hist[tid * 4 + i]; sum += c[i]; }
            const int incl = wave_inclusive_sum(sum);                    // (DPP network: no LDS round trips)
            int run = incl - sum;                                        // elements in the bins below this lane's four
            if (run < remaining && remaining <= incl) {                  // the want-th element falls into one of my bins
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    if (run < remaining && remaining <= run + c[i]) { L.redi[32] = tid * 4 + i; L.redi[33] = remaining - run; }
                    run += c[i];
                }
            }
        }
        __syncthreads();
        prefix = (prefix << 8) | (unsigned long long)(uint32_t)L.redi[32];
        remaining = L.redi[33];
    }
    // (2) compaction in position order: everything below T, and the first `remaining` elements equal to T.
    // `sel` may be the array key() reads: chunk k only writes below the positions it has read.
    int nsel = 0, neq = 0;
#pragma unroll
    for (int k = 0; k < SEL_PER_THREAD; ++k) {
        if (k * BLOCK < n) {
            const int e = tid + k * BLOCK;
            const bool eq = e < n && sk[k] == prefix;
            int teq;
            const int eqb = block_scan_flag(eq, L.redi, teq);
            const bool take = e < n && (sk[k] < prefix || (eq && neq + eqb < remaining));
            int tsel;
            const int pos = block_scan_flag(take, L.redi, tsel);
            if (take) sel[nsel + pos] = ky[k];
            nsel += tsel;
            neq += teq;
        }
    }
    __syncthreads();
    // (3) sort the selected elements by (value, position in the list): bitonic network over the next power of two
    // (entries beyond `want` are +infinity), keys as three LDS arrays (high word, low word, list position) in the
    // region the radix counters used.  log2(n)(log2(n)+1)/2 rounds of one compare-exchange per thread.
    int npad = 64;
    while (npad < want) npad <<= 1;
    uint32_t* const khi = hist;                                      // [npad]
    uint32_t* const klo = hist + npad;                               // [npad]
    uint16_t* const kps = (uint16_t*)(hist + 2 * npad);              // [npad]
    for (int i = tid; i < npad; i += BLOCK) {
        unsigned long long v = ~0ull;
        if (i < want) v = sortable_f64(value(sel[i]));
        khi[i] = (uint32_t)(v >> 32);
        klo[i] = (uint32_t)v;
        kps[i] = (uint16_t)(i < want ? i : 0xFFFF);
    }
    __syncthreads();
    for (int k = 2; k <= npad; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            const int sh = __ffs(j) - 1;
            for (int t = tid; t < (npad >> 1); t += BLOCK) {
                const int i = ((t >> sh) << (sh + 1)) | (t & (j - 1)), l = i | j;
                const uint32_t ah = khi[i], al = klo[i], bh = khi[l], bl = klo[l];
                const uint32_t ap = kps[i], bp = kps[l];
                const bool a_gt_b = ah > bh || (ah == bh && (al > bl || (al == bl && ap > bp)));
                if (a_gt_b == ((i & k) == 0)) {
                    khi[i] = bh; klo[i] = bl; kps[i] = (uint16_t)bp;
                    khi[l] = ah; klo[l] = al; kps[l] = (uint16_t)ap;
                }
            }
            // A stride below 128 pairs elements of one 128-element block, and block t >> 6 belongs to the wave of thread t
            // (t = tid + m * BLOCK: blocks w, w + 4, ...): such a round reads and writes only what this wave wrote, and a
            // wave's LDS operations execute in order.  The workgroup meets only around the rounds that cross blocks: 3 of
            // the 45 rounds of a 512-element sort.
            const int j_next = j > 1 ? (j >> 1) : k;                 // (the next merge starts at stride 2k / 2)
            if (j >= 128 || j_next >= 128 || (j == 1 && k == npad)) __syncthreads();
            else IRBPP_WAVE_SYNC();
        }
    for (int r = tid; r < want; r += BLOCK) out[r] = sel[kps[r]];
    __syncthreads();
}

// ---------------------------------------------------------------------------------------
// Candidate rows, selection / padding and the float32 observation (binPhy.py:204-227) from L.vmask, the posZmap values
// in `zsrc` (global memory, written where naiveMask is set) and naiveMask's bit rows `gvalid`; also records the
// candidate keys for the next step's action_to_position.
// ---------------------------------------------------------------------------------------
#ifdef IRBPP_AB_EMIT_ACCOUNT
#define IRBPP_EMIT_STAMP(k) do { if (io.phase_cycles && threadIdx.x == 0) io.phase_cycles[(size_t)b * PHASE_ROW + (k)] = (long long)clock64(); } while (0)
#else
#define IRBPP_EMIT_STAMP(k) do {} while (0)
#endif
__device__ inline void emit_observation(const Params& P, const State& S, const StepIO& io, const Lds& L, int b, int item,
                                        int nvalid, float* obs, const double* zsrc, const uint32_t* gvalid) {
    const int tid = threadIdx.x;
    const int R = P.R, AC = P.AC, Ax = P.Ax, Ay = P.Ay;
    const int X = fdiv(tid, Ay, P.mg_ay), Y = tid - X * Ay;
    // ---- candidate rows: per rotation, vertices ordered by (col, row) (np.unique, cvTools.py:101)
    // the candidate rows the buffer holds from last time (registered buffers): asked for first -- nothing depends on it until
    // the rows are stored, and as a load in front of them it was one more memory round trip on a path made of those
    const int prev_rows = io.obs_rows != nullptr ? io.obs_rows[b] : -1;
    if (tid == 0) ((unsigned long long*)L.redd)[7] = ~0ull;         // the MINZ bids' word (see the end; several barriers lie between)
    uint32_t* keys = (uint32_t*)L.scratch;          // [R*AC]
    uint32_t* okey = keys + R * AC;                 // [S]
    uint32_t* hist = (uint32_t*)L.img;              // [256] counters of the radix select: the level images are done with
    // One wave lists them, a lane per (rotation, column) -- four rotations at a time --: the lane gathers its column of the
    // rotation's vertex bits (16 LDS reads of row words, bit cx of each), the columns' counts are summed over the lanes on
    // the DPP network, and every lane writes the keys of its column, rows ascending, behind those of the columns before
    // it.  (Until session 35: a thread per action cell and one workgroup-wide scan of the flags per rotation -- two
    // barriers and a round of LDS each, 3.9 k of the workgroup's 11.5 k cycles, profiles/r04/s34.)
    int n = 0;
    if (tid < 64) {
        const int lane = tid, cx = lane & 15;
        for (int r0 = 0; r0 < R; r0 += 4) {
            const int r = r0 + (lane >> 4);
            uint32_t col = 0u;
            if (r < R && cx < Ay) {
#pragma unroll
                for (int cy = 0; cy < 16; ++cy) col |= ((L.vmask[r * 16 + cy] >> cx) & 1u) << cy;
                col &= (1u << Ax) - 1u;
            }
            const int cnt = __popc(col), incl = wave_inclusive_sum(cnt);
            int at = n + incl - cnt;
            while (col != 0u) {
                const int cy = __ffs((int)col) - 1;
                col &= col - 1u;
                keys[at++] = ((uint32_t)r << 16) | ((uint32_t)cy << 8) | (uint32_t)cx;
            }
            n += __builtin_amdgcn_readlane(incl, 63);
        }
        if (tid == 0) L.redi[34] = n;
    }
    __syncthreads();
    n = L.redi[34];
    IRBPP_EMIT_STAMP(11);
    int nrows;
    bool fallback = false;
    const uint32_t* rows;
    if (n > 0 && n <= P.S) {
        nrows = n;
        rows = keys;
    } else if (n > P.S) {
        // np.argsort(candidates[:,3])[:S] (binPhy.py:209-212), ties by ascending index
        select_smallest(P, L, zsrc, nullptr, n, P.S, [&](int e) { return keys[e]; }, okey, keys, hist);
        nrows = P.S;
        rows = okey;
        __syncthreads();
    } else {
        // no candidate at all: the S smallest of posZValid.reshape(-1) (binPhy.py:217-225)
        fallback = true;
        const int total_cells = R * AC;
        const int want = total_cells < P.S ? total_cells : P.S;
        auto cell_key = [&](int e) {
            const int r = fdiv(e, AC, P.mg_ac), c = e - r * AC, x = fdiv(c, Ay, P.mg_ay);
            return ((uint32_t)r << 16) | ((uint32_t)x << 8) | (uint32_t)(c - x * Ay);
        };
        if (nvalid == 0) {                   // nothing fits (the usual end of an episode): every value is 1e3, position order
            for (int e = tid; e < want; e += BLOCK) okey[e] = cell_key(e);
            __syncthreads();
        } else {
            select_smallest(P, L, zsrc, gvalid, total_cells, want, cell_key, okey, keys, hist);
        }
        nrows = total_cells < P.S ? total_cells : P.S;
        rows = okey;
        __syncthreads();
    }

    IRBPP_EMIT_STAMP(12);
    // ---- emit: candidate block [S][5] (item vector and heightmap were written by the transition kernel); float32 cast last.
    // A registered observation buffer (irbpp_register_obs_buffer) is only ever written by this library, which
    // remembers per bin how many rows it wrote last time: the rows beyond are still zero and are not written again
    // (typically 100 of 500 rows exist).
    int write_rows = P.S;
    if (io.obs_rows != nullptr) {
        if (prev_rows >= 0) write_rows = prev_rows > nrows ? prev_rows : nrows;
        if (tid == 0) io.obs_rows[b] = nrows;        // (everyone read the old count at the top of the function, barriers ago)
    }
    // A thread per ROW, once: the row's one data-dependent float -- H = (float) posZValid of its cell, read where it lies
    // (global memory, L2 / Infinity Cache: only the ~100 rows of the bin, not the whole [R][AC] grid), or in the fallback the
    // cell's validity flag --, its five floats (five stores, 20 bytes apart between lanes), its key for the next apply and
    // its bid for the MINZ policy.  The hundred rows of an ordinary bin are the work of two waves.  (As separate loops -- row
    // values into LDS, a barrier, a thread per FLOAT with a division by five and a five-way select, keys, policy -- this was
    // half of the workgroup's cycles in a kernel bound by instruction issue, profiles/r04/s34 - s38.)
    float best = INFINITY;                            // MINZ: the lowest float32 H among V == 1, first on ties (ascending rows per thread)
    int bi = 0x7fffffff;
    for (int rep = 0; rep < IRBPP_REPS(3); ++rep)
    for (int row = tid; row < write_rows; row += BLOCK) {
        float v0 = 0.0f, v1 = 0.0f, v2 = 0.0f, v3 = 0.0f, v4 = 0.0f;
        if (row < nrows) {
            const uint32_t k = rows[row];
            float rv;
            if (fallback) rv = (float)((gvalid[(k >> 16) * 16 + ((k >> 8) & 255u)] >> (k & 255u)) & 1u);
            else rv = (float)zsrc[(k >> 16) * AC + ((k >> 8) & 255u) * Ay + (k & 255u)];       // a candidate is a valid cell
            v0 = (float)(k >> 16);
            v1 = (float)((k >> 8) & 255u);
            v2 = (float)(k & 255u);
            v3 = fallback ? (float)P.bin_z : rv;
            v4 = fallback ? rv : 1.0f;
            S.cand[(size_t)b * P.S + row] = k;       // candidate keys for the next apply: only the rows that exist (BinState::nrows tells apply where they end)
            if (!fallback && rv < best) { best = rv; bi = row; }
        }
        float* const o = obs + 5 * row;
        o[0] = v0; o[1] = v1; o[2] = v2; o[3] = v3; o[4] = v4;
    }
    IRBPP_EMIT_STAMP(13);
    if (tid == 0) {
        S.bs[b].cur_item = item;
        S.bs[b].nvalid = nvalid;
        S.bs[b].nrows = nrows;
    }
    // The scripted MINZ policy on the rows just written (irbpp_policy_minz on this observation gives the same).  Fallback
    // rows all carry H = bin_z and are sorted valid-first, so the answer there is row 0 (or "none" = 0).
    if (io.auto_action != nullptr) {
        {   // the wave's (lowest H, lowest row) on the DPP network: lane 63 ends up with it (see wave_max_f64)
#define IRBPP_ARGMIN_STEP(CTRL, ROWS)                                                                                          \
            {                                                                                                                  \
                const float ob = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(best), __float_as_int(best), CTRL, ROWS, 0xF, false)); \
                const int oi = __builtin_amdgcn_update_dpp(bi, bi, CTRL, ROWS, 0xF, false);                                     \
                if (ob < best || (ob == best && oi < bi)) { best = ob; bi = oi; }                                               \
            }
            IRBPP_ARGMIN_STEP(0x111, 0xF) IRBPP_ARGMIN_STEP(0x112, 0xF) IRBPP_ARGMIN_STEP(0x114, 0xF) IRBPP_ARGMIN_STEP(0x118, 0xF)
            IRBPP_ARGMIN_STEP(0x142, 0xA) IRBPP_ARGMIN_STEP(0x143, 0xC)
#undef IRBPP_ARGMIN_STEP
        }
        // the waves' bids meet in one 64-bit LDS word: H as an unsigned number that orders like the float (-0 made +0 first:
        // the two compare equal) above the row; a wave without a bid offers +inf | 0x7fffffff, the word starts above that
        if ((tid & 63) == 63) {
            const uint32_t fb = (uint32_t)__float_as_int(best + 0.0f);
            const uint32_t su = fb ^ ((fb >> 31) != 0u ? 0xFFFFFFFFu : 0x80000000u);
            atomicMin((unsigned long long*)L.redd + 7, ((unsigned long long)su << 32) | (uint32_t)bi);
        }
        __syncthreads();
        if (tid == 0) {
            const int won = (int)(uint32_t)(((const unsigned long long*)L.redd)[7] & 0xFFFFFFFFull);
            io.auto_action[b] = won == 0x7fffffff ? 0 : won;
        }
    }
    stamp(io, b, 4);
}

#endif  // IRBPP_PASS == 1
// ---------------------------------------------------------------------------------------
// Split pipeline, hand-over of one bin from the transition kernel: level images, candidate starts, the
// vertex bits of isolated pixels and the scalars of the observation go to global memory for the trace and
// emit kernels (posZValid is there already).  The hand-over holds whatever a bin can produce: a rotation has at
// most 64 levels, and every candidate start is a pixel of exactly one level image of its rotation, so a bin has at
// most R*64 images and R*AC candidates; an XCD's candidate list holds twice the worst case of its share of the bins
// (IRBPP_DEVERR_CAPACITY if the dispatcher ever gave one die more than twice its share of such bins).
// ---------------------------------------------------------------------------------------
template <int IPT>
__device__ inline void split_handover(const Params& P, const State& S, const Lds& L, int b, int slot, int item, int nvalid) {
    const int tid = threadIdx.x;
    const KernArgsPtr ka = cold_args();
    constexpr int IMGS = IPT * (BLOCK / 16);
    // the batch's row words and the candidate list live in the bytes of the heightmap tile (its float32 copy is out)
    constexpr int CLIST = CONTOUR_CLIST;
    uint16_t* const rows = (uint16_t*)L.scratch;
    uint16_t* const cwords = rows + IMGS * 16;               // the batch's candidate words, behind its row words
    uint16_t* const clist = L.clist;
    {   // naiveMask's bit rows first: they share their LDS bytes with the task index built next
        uint32_t* gb = ka->S.w_valid + (size_t)b * P.R * 16;
        for (int rep = 0; rep < IRBPP_REPS(14); ++rep)
#pragma unroll 1
        for (int i = tid; i < P.R * 16; i += BLOCK) gb[i] = L.vbits[i];
        __syncthreads();
    }
    int ntasks = 0;
    for (int rep = 0; rep < IRBPP_REPS(9); ++rep) { if (rep) __syncthreads(); ntasks = IRBPP_HERE contour_tasks(P, L); }
    int ncand = 0, niso = 0;
    uint32_t* gi = (uint32_t*)(ka->S.w_img + (size_t)b * P.wimg * 16);
    uint8_t* gr = ka->S.w_imgrot + (size_t)b * P.wimg;
    const int xcd = (int)(__builtin_amdgcn_s_getreg((31 << 11) | 20) & (NXCD - 1));     // HW_REG_XCC_ID: the die this workgroup runs on
    for (int base = 0; base < ntasks; base += IMGS) {                        // one batch of level images at a time
        int batch_total = 0;
        for (int rep = 0; rep < IRBPP_REPS(9); ++rep) {
            IRBPP_HERE contour_images<IPT>(P, L, rows, base, false);
            batch_total = IRBPP_HERE contour_candidates<IPT>(P, L, rows, base, ntasks, cwords);
        }
        niso += batch_total >> 16;
        batch_total &= 0xFFFF;
        const int nb = ntasks - base < IMGS ? ntasks - base : IMGS;
        // rows [IMGS][16] in LDS -> [image][16 row words] in global, as dwords
        const uint32_t* lr = (const uint32_t*)rows;
        for (int rep = 0; rep < IRBPP_REPS(14); ++rep) {
#pragma unroll 1
        for (int i = tid; i < nb * 8; i += BLOCK) gi[(size_t)base * 8 + i] = lr[i];
#pragma unroll 1
        for (int i = tid; i < nb; i += BLOCK) gr[base + i] = (uint8_t)(L.tasklist[base + i] >> 8);
        }
        // The candidates join a flat list of (bin, image<<8 | y0<<4 | x0) pairs, in whatever order the bins arrive -- the
        // trace kernel's results do not depend on it.  One list per XCD: the line of a counter that only the
        // workgroups of one XCD touch stays in that XCD's L2, whereas the line of one device-wide counter travels
        // between the eight L2s with every allocation (measured: +22 us per launch).  The LDS list holds CLIST
        // entries; a batch with more candidates (speckle) goes image by image (an image has at most 64).
        const int nsub = batch_total <= CLIST ? 1 : IMGS;
        for (int sub = 0; sub < nsub; ++sub) {
            int total = 0;
            for (int rep = 0; rep < IRBPP_REPS(10); ++rep) total = IRBPP_HERE contour_list<IPT>(L, rows, clist, base, ntasks, nsub, sub, cwords);
            if (tid == 0) {
                // This die's list; should it be full (the dispatcher gave this die far more than its share of speckled
                // bins, or the device runs in a partition mode where XCC_ID does not spread the workgroups over eight
                // values) the batch goes to the next list that has room.  A failed reservation is taken back, so the
                // counters end up exact; the lists together hold twice the worst case of all bins.
                int at = -1, seg = xcd;
                if (total > 0) {
                    for (int k = 0; k < NXCD && at < 0; ++k) {
                        seg = (xcd + k) & (NXCD - 1);
                        const int got = atomicAdd(ka->S.w_total + seg * XCD_STRIDE, total);
                        if (got + total <= P.seg_cap) at = got;
                        else atomicSub(ka->S.w_total + seg * XCD_STRIDE, total);
                    }
                    if (at < 0) raise_error(S, IRBPP_DEVERR_CAPACITY);
                }
                L.redi[11] = at;
                L.redi[12] = seg;
            }
            __syncthreads();
            if (L.redi[11] >= 0) {
                uint2* flat = ka->S.w_cand + (size_t)L.redi[12] * P.seg_cap + L.redi[11];
#pragma unroll 1
                for (int i = tid; i < total; i += BLOCK) {
                    const uint32_t e = clist[i];
                    flat[i] = make_uint2((uint32_t)b, ((uint32_t)(base + (e & 127u)) << 8) | (((e >> 11) & 15u) << 4) | ((e >> 7) & 15u));
                }
            }
            ncand += total;
        }
        __syncthreads();                             // the next batch rebuilds the images and the list
    }
    uint32_t* gv = ka->S.w_vmask + (size_t)b * P.R * 16;
    for (int rep = 0; rep < IRBPP_REPS(14); ++rep)
#pragma unroll 1
    for (int i = tid; i < P.R * 16; i += BLOCK) gv[i] = L.vmask[i];
    if (tid == 0) {
        int32_t* m = ka->S.w_meta + (size_t)b * WMETA;
        m[0] = ntasks;
        m[1] = ncand;
        m[2] = nvalid;
        m[3] = item;
        // A bin with many border starts (speckled level images) is a bin whose observation is expensive: more than S
        // candidates mean a radix select and a sort, five times the time of an ordinary bin, and the emit kernel lasts as
        // long as the last of them.  Such bins enter a list that the emit kernel serves FIRST (its leading workgroups); the
        // count of starts + isolated pixels tracks the number of candidates closely (r = 0.997 on the "general" data set, every
        // bin with more than S candidates has >= 350 of them against a median of 84; listed from 0.6 S on).
        int heavy = 0;
        const int turn = ka->io.heavy_turn;
        if (turn >= 0 && ncand + niso >= P.heavy_thr) {
            int32_t* hv = ka->S.w_heavy + (size_t)turn * (XCD_STRIDE + P.heavy_cap);
            const int pos = atomicAdd(hv, 1);
            if (pos < P.heavy_cap) { hv[XCD_STRIDE + pos] = b; heavy = 1; }
        }
        m[4] = heavy;
    }
}

#if IRBPP_PASS == 1
// ---------------------------------------------------------------------------------------
// Split pipeline, last kernel: the observation of one bin from what the other two left in global memory.
// ---------------------------------------------------------------------------------------
// HEAVY_FIRST: the grid may lead with the workgroups of the speckled bins (StepIO::heavy_turn >= 0: free-form level images).
// One build serves all data: with this code in, the register allocator happens to fit the kernel into its 64 VGPRs; the
// build without it spilled four of them and cost the BlockOut step 1.1 us (19.1 vs 18.0 us, profiles/r04/s8) --
// tests/test_kernel_asm.py watches the scratch sizes of the step's kernels.
template <bool HEAVY_FIRST, int SPEC>
__device__ __forceinline__ void emit_body(const Params& P_run, const Tables& T, const State& S, const StepIO& io, const int mode, unsigned char* smem) {
    Params P_spec;
    const Params& P = SPEC == 0 ? P_run : (P_spec = specialise<SPEC>(P_run), P_spec);
    Lds L = {};                                      // the emit kernel's own, small carve-up (Params.e_*)
    L.vmask = (uint32_t*)(smem + P.e_vmask);
    L.redd = (double*)(smem + P.e_red);
    L.redi = (int*)(L.redd + 8);
    L.img = (uint16_t*)(smem + P.e_hist);            // the 256 counters of the radix select, the sort keys, the rows' values
    L.scratch = smem + P.e_keys;
    const bool some = mode == MODE_RESET && io.bin_list != nullptr;
    const int tid = threadIdx.x;
    // Workgroup 0 also retires the launch's flat candidate list (the trace kernel is done with it) and hands the
    // device error word to the step outputs: every bit of this step was raised by the transition or the trace
    // kernel, which have completed (the emit kernel raises none).
    if (blockIdx.x == 0 && tid < NXCD) { S.w_total[tid * XCD_STRIDE] = 0; S.w_nround[tid * XCD_STRIDE] = 0; }
    if (blockIdx.x == 0 && tid == 0 && io.err_out != nullptr) *io.err_out = *S.err;
    // The grid's first heavy_cap workgroups serve the bins the transition kernel listed as expensive (see split_handover),
    // the rest the bins in launch order minus those: the expensive ones start at once instead of wherever their index
    // puts them.  The list of the next launch (the other one of two, used in turn) is cleared here.
#ifdef IRBPP_AB_EMIT_ACCOUNT
    const long long t_entry = (long long)clock64();
#endif
    int slot = (int)blockIdx.x + io.block_off, b = 0;
    bool mapped = false;
    if constexpr (HEAVY_FIRST) {
        const int hg = io.heavy_turn >= 0 ? P.heavy_cap : 0;
        if ((int)blockIdx.x < hg) {
            const int32_t* hv = S.w_heavy + (size_t)io.heavy_turn * (XCD_STRIDE + P.heavy_cap);
            if (blockIdx.x == 0 && tid == 0) S.w_heavy[(size_t)(io.heavy_turn ^ 1) * (XCD_STRIDE + P.heavy_cap)] = 0;
            int cnt = hv[0];
            cnt = cnt < P.heavy_cap ? cnt : P.heavy_cap;
            if ((int)blockIdx.x >= cnt) return;
            b = hv[XCD_STRIDE + blockIdx.x];
            slot = b;
            mapped = true;
        } else {
            slot -= hg;
        }
    }
    if (!mapped) {
        b = ((mode == MODE_STEP || mode == MODE_CANDS) && io.use_order) ? S.order[slot] : some ? io.bin_list[slot] : slot;
        if constexpr (HEAVY_FIRST)
            if (io.heavy_turn >= 0 && b >= 0 && b < P.N && S.w_meta[(size_t)b * WMETA + 4] != 0) return;       // served by a leading workgroup
    }
    if (b < 0 || b >= P.N) return;                   // the transition kernel has flagged it already
#ifdef IRBPP_AB_EMIT_ACCOUNT
    if (io.phase_cycles && tid == 0) { io.phase_cycles[(size_t)b * PHASE_ROW + 6] = t_entry; io.phase_cycles[(size_t)b * PHASE_ROW + 7] = (long long)clock64(); }
#endif
    float* obs = io.obs + (size_t)(some ? slot : b) * io.obs_stride;
    const uint32_t* gv = S.w_vmask + (size_t)b * P.R * 16;
#pragma unroll 1
    for (int i = tid; i < P.R * 16; i += BLOCK) L.vmask[i] = gv[i];
    const int nvalid = S.w_meta[(size_t)b * WMETA + 2], item = S.w_meta[(size_t)b * WMETA + 3];
    __syncthreads();
    stamp(io, b, 3);
    emit_observation(P, S, io, L, b, item, nvalid, obs, S.w_posz + (size_t)b * P.R * P.AC, S.w_valid + (size_t)b * P.R * 16);
    IRBPP_EMIT_STAMP(14);
}
#define IRBPP_EMIT_KERNEL(NAME, HF, SPEC)                                                                                \
    extern "C" __global__ void __launch_bounds__(BLOCK) __attribute__((amdgpu_waves_per_eu(8, 8)))                      \
    NAME(const Params P, const Tables T, const State S, const StepIO io, const int mode) {                              \
        extern __shared__ __attribute__((aligned(16))) unsigned char smem[];                                            \
        emit_body<HF, SPEC>(P, T, S, io, mode, smem);                                                                   \
    }
IRBPP_EMIT_KERNEL(irbpp_emit_kernel, true, 0)
#ifndef IRBPP_NO_SPEC
IRBPP_EMIT_KERNEL(irbpp_emit_kernel_s1, true, 1)
IRBPP_EMIT_KERNEL(irbpp_emit_kernel_s2, true, 2)
IRBPP_EMIT_KERNEL(irbpp_emit_kernel_s3, true, 3)
IRBPP_EMIT_KERNEL(irbpp_emit_kernel_s4, true, 4)
IRBPP_EMIT_KERNEL(irbpp_emit_kernel_s5, true, 5)
#endif

// ---------------------------------------------------------------------------------------
// The emit kernel with ONE WAVE per bin (four bins per workgroup), for data whose bins practically never have more than S
// candidates (lattice and box data).  An ordinary bin has ~100 candidate rows: as a 256-thread workgroup two waves did the
// rows, two idled through the barriers, and 8192 bins were four rounds of workgroups, each a chain of dependent reads (vertex
// bits -> keys -> posZ of the rows -> stores).  With a wave per bin there is no barrier on the ordinary path, 32 bins share
// a CU and 8192 bins are ONE round.  A bin that needs the workgroup -- more than S candidates, or no candidate but valid
// cells (both: radix select + sort over LDS) -- is flagged and served by all four waves with emit_observation behind one
// barrier, after the waves' own bins are out.  Same rows, same order, same stores as emit_observation.
// ---------------------------------------------------------------------------------------
template <int SPEC>
__device__ __forceinline__ void emit_wave_body(const Params& P_run, const Tables& T, const State& S, const StepIO& io, const int mode,
                                               unsigned char* smem) {
    Params P_spec;
    const Params& P = SPEC == 0 ? P_run : (P_spec = specialise<SPEC>(P_run), P_spec);
    int* const need = (int*)(smem + P.e_need);       // (dynamic LDS only: the kernel's limit is raised to the CU's whole 160 KB)
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int R = P.R, AC = P.AC, Ax = P.Ax, Ay = P.Ay;
    const bool some = mode == MODE_RESET && io.bin_list != nullptr;
    // workgroup 0 retires the launch's flat lists and hands on the error word (see emit_body)
    if (blockIdx.x == 0 && tid < NXCD) { S.w_total[tid * XCD_STRIDE] = 0; S.w_nround[tid * XCD_STRIDE] = 0; }
    if (blockIdx.x == 0 && tid == 0 && io.err_out != nullptr) *io.err_out = *S.err;
    const int local = (int)blockIdx.x * WAVES + wave;
    const int slot = local + io.block_off;
    int b = -1;
    if (local < io.n_slots) b = ((mode == MODE_STEP || mode == MODE_CANDS) && io.use_order) ? S.order[slot] : some ? io.bin_list[slot] : slot;
    if (b >= P.N) b = -1;                            // (the transition kernel has flagged it already)
    b = __builtin_amdgcn_readfirstlane(b);
    if (lane == 0) need[wave] = -1;
    if (b >= 0) {
        uint32_t* const vm = (uint32_t*)(smem + wave * P.ew_bytes);    // [R * 16] vertex bits
        uint32_t* const keys = vm + R * 16;                            // [S] candidate keys
        const uint32_t* gv = S.w_vmask + (size_t)b * R * 16;
        for (int i = lane; i < R * 16; i += 64) vm[i] = gv[i];
        const int nvalid = S.w_meta[(size_t)b * WMETA + 2], item = S.w_meta[(size_t)b * WMETA + 3];
        const int prev_rows = io.obs_rows != nullptr ? io.obs_rows[b] : -1;
        float* const obs = io.obs + (size_t)(some ? slot : b) * io.obs_stride;
        const double* const zsrc = S.w_posz + (size_t)b * R * AC;
        const uint32_t* const gvalid = S.w_valid + (size_t)b * R * 16;
        stamp(io, b, 3, lane == 0);
        IRBPP_WAVE_SYNC();
        // candidate rows per rotation, ordered by (col, row) (np.unique, cvTools.py:101): a lane per (rotation, column), four
        // rotations at a time; counted first -- only S keys fit, and more than S is the workgroup's business
        const int cx = lane & 15;
        auto column = [&](int r) {                                     // my column of rotation r: bit cy = vertex at (row cy, col cx)
            uint32_t col = 0u;
            if (r < R && cx < Ay) {
#pragma unroll
                for (int cy = 0; cy < 16; ++cy) col |= ((vm[r * 16 + cy] >> cx) & 1u) << cy;
                col &= (1u << Ax) - 1u;
            }
            return col;
        };
        int n = 0;
#pragma unroll 1
        for (int r0 = 0; r0 < R; r0 += 4) n += __builtin_amdgcn_readlane(wave_inclusive_sum(__popc(column(r0 + (lane >> 4)))), 63);
        if (n > P.S || (n == 0 && nvalid > 0)) {
            if (lane == 0) need[wave] = b;
        } else {
            int at0 = 0;
#pragma unroll 1
            for (int r0 = 0; r0 < R; r0 += 4) {
                const int r = r0 + (lane >> 4);
                uint32_t col = column(r);
                const int cnt = __popc(col), incl = wave_inclusive_sum(cnt);
                int at = at0 + incl - cnt;
                while (col != 0u) {
                    const int cy = __ffs((int)col) - 1;
                    col &= col - 1u;
                    keys[at++] = ((uint32_t)r << 16) | ((uint32_t)cy << 8) | (uint32_t)cx;
                }
                at0 += __builtin_amdgcn_readlane(incl, 63);
            }
            IRBPP_WAVE_SYNC();
            const bool fallback = n == 0;            // nothing fits (nvalid == 0): the first S cells in position order, H = bin height, V = 0
            const int total_cells = R * AC;
            const int nrows = fallback ? (total_cells < P.S ? total_cells : P.S) : n;
            int write_rows = P.S;
            if (io.obs_rows != nullptr) {
                if (prev_rows >= 0) write_rows = prev_rows > nrows ? prev_rows : nrows;
                if (lane == 0) io.obs_rows[b] = nrows;
            }
            float best = INFINITY;                   // MINZ: the lowest float32 H among V == 1, first on ties
            int bi = 0x7fffffff;
            for (int row = lane; row < write_rows; row += 64) {
                float v0 = 0.0f, v1 = 0.0f, v2 = 0.0f, v3 = 0.0f, v4 = 0.0f;
                if (row < nrows) {
                    uint32_t k;
                    if (fallback) {
                        const int r = fdiv(row, AC, P.mg_ac), c = row - r * AC, x = fdiv(c, Ay, P.mg_ay);
                        k = ((uint32_t)r << 16) | ((uint32_t)x << 8) | (uint32_t)(c - x * Ay);
                    } else {
                        k = keys[row];
                    }
                    float rv;
                    if (fallback) rv = (float)((gvalid[(k >> 16) * 16 + ((k >> 8) & 255u)] >> (k & 255u)) & 1u);
                    else rv = (float)zsrc[(k >> 16) * AC + ((k >> 8) & 255u) * Ay + (k & 255u)];       // a candidate is a valid cell
                    v0 = (float)(k >> 16);
                    v1 = (float)((k >> 8) & 255u);
                    v2 = (float)(k & 255u);
                    v3 = fallback ? (float)P.bin_z : rv;
                    v4 = fallback ? rv : 1.0f;
                    S.cand[(size_t)b * P.S + row] = k;
                    if (!fallback && rv < best) { best = rv; bi = row; }
                }
                float* const o = obs + 5 * row;
                o[0] = v0; o[1] = v1; o[2] = v2; o[3] = v3; o[4] = v4;
            }
            if (lane == 0) {
                S.bs[b].cur_item = item;
                S.bs[b].nvalid = nvalid;
                S.bs[b].nrows = nrows;
            }
            if (io.auto_action != nullptr) {
#define IRBPP_ARGMIN_STEP(CTRL, ROWS)                                                                                          \
                {                                                                                                              \
                    const float ob = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(best), __float_as_int(best), CTRL, ROWS, 0xF, false)); \
                    const int oi = __builtin_amdgcn_update_dpp(bi, bi, CTRL, ROWS, 0xF, false);                                 \
                    if (ob < best || (ob == best && oi < bi)) { best = ob; bi = oi; }                                           \
                }
                IRBPP_ARGMIN_STEP(0x111, 0xF) IRBPP_ARGMIN_STEP(0x112, 0xF) IRBPP_ARGMIN_STEP(0x114, 0xF) IRBPP_ARGMIN_STEP(0x118, 0xF)
                IRBPP_ARGMIN_STEP(0x142, 0xA) IRBPP_ARGMIN_STEP(0x143, 0xC)
#undef IRBPP_ARGMIN_STEP
                if (lane == 63) io.auto_action[b] = bi == 0x7fffffff ? 0 : bi;
            }
            stamp(io, b, 4, lane == 0);
        }
    }
    __syncthreads();
    // the bins that need the workgroup (rare): one after the other, all four waves, the 256-thread routine
    if ((need[0] & need[1] & need[2] & need[3]) == -1) return;          // (-1 = all bits set: nobody)
    Lds L = {};
    L.vmask = (uint32_t*)(smem + P.e_vmask);
    L.redd = (double*)(smem + P.e_red);
    L.redi = (int*)(L.redd + 8);
    L.img = (uint16_t*)(smem + P.e_hist);
    L.scratch = smem + P.e_keys;
#pragma unroll 1
    for (int w = 0; w < WAVES; ++w) {
        const int bw = __builtin_amdgcn_readfirstlane(need[w]);       // (uniform: addresses stay scalar)
        if (bw < 0) continue;
        const int slot_w = (int)blockIdx.x * WAVES + w + io.block_off;
        float* const obs = io.obs + (size_t)(some ? slot_w : bw) * io.obs_stride;
        const uint32_t* gv = S.w_vmask + (size_t)bw * R * 16;
        __syncthreads();
#pragma unroll 1
        for (int i = tid; i < R * 16; i += BLOCK) L.vmask[i] = gv[i];
        const int nvalid = S.w_meta[(size_t)bw * WMETA + 2], item = S.w_meta[(size_t)bw * WMETA + 3];
        __syncthreads();
        stamp(io, bw, 3);
        emit_observation(P, S, io, L, bw, item, nvalid, obs, S.w_posz + (size_t)bw * R * AC, S.w_valid + (size_t)bw * R * 16);
    }
}
#define IRBPP_EMIT_WAVE_KERNEL(NAME, SPEC)                                                                                  \
    extern "C" __global__ void __launch_bounds__(BLOCK) __attribute__((amdgpu_waves_per_eu(8, 8)))                      \
    NAME(const Params P, const Tables T, const State S, const StepIO io, const int mode) {                              \
        extern __shared__ __attribute__((aligned(16))) unsigned char smem[];                                            \
        emit_wave_body<SPEC>(P, T, S, io, mode, smem);                                                                  \
    }
IRBPP_EMIT_WAVE_KERNEL(irbpp_emit_wave_kernel, 0)
#ifndef IRBPP_NO_SPEC
IRBPP_EMIT_WAVE_KERNEL(irbpp_emit_wave_kernel_s1, 1)
IRBPP_EMIT_WAVE_KERNEL(irbpp_emit_wave_kernel_s2, 2)
IRBPP_EMIT_WAVE_KERNEL(irbpp_emit_wave_kernel_s5, 5)
#endif

// ---------------------------------------------------------------------------------------
// Split pipeline, middle kernels: border following + approxPolyDP + convexity over the candidate starts of ALL
// bins of the launch as one flat list.  One bin has ~25 borders to follow, a handful of them long: traced inside
// the bin's own workgroup, most lanes idle, and the bins with many or long borders set the duration of the
// launch.  Here the candidates of all bins form one flat list (every bin appends its batch with one atomicAdd on
// the list's counter while it hands over) that is cut into chunks of 64: one wave per chunk, one candidate per lane, whichever bins they
// come from -- every wave has the same amount of work.  A lane copies its level image (64 bytes) into LDS,
// follows its border (trace_border), and the wave then runs approx_convex_segmented on all the closed borders,
// 128 contour points per round; vertex bits go to the bins' rows in global memory with one atomic OR each.
// ---------------------------------------------------------------------------------------
#ifndef IRBPP_TRACE_SHORT
#define IRBPP_TRACE_SHORT 0
#endif
constexpr int TRACE_P = IRBPP_TRACE_P;                                // contour points per lane and polygon round
constexpr int TRACE_CAP = 128;                                        // points of a border the wave-parallel path takes
constexpr int TRACE_LDS_CAP = 56, TRACE_SLOT = TRACE_LDS_CAP + 4;     // of which in the lane's slot in LDS (15 dwords: odd stride); the rest
constexpr int TRACE_SPILL = TRACE_CAP - TRACE_LDS_CAP;                //   in global scratch (a border of more than 56 points: one in ~10^3)
constexpr int TRACE_SHORT = IRBPP_TRACE_SHORT;                        // borders of up to this many points get rounds of their own, ahead of the
                                                                      // long ones (0 = one class: measured 27.46 vs 27.25 M steps/s for 8)
static_assert(TRACE_CAP <= 64 * TRACE_P, "a border must fit one polygon round");
static_assert(ROUND_POINTS == 64 * TRACE_P, "a round record holds one polygon round");
constexpr int TRACE_BIG = 768;                                        // point capacity of the sequential redo (global scratch)
constexpr int TRACE_FSTRIDE = FRAME_WORDS;                            // dwords per staged image: the two frames (35: odd, lanes on distinct banks)
static_assert(TRACE_FSTRIDE % 2 == 1 && (TRACE_SLOT / 4) % 2 == 1 && TRACE_SLOT % 4 == 0, "odd dword strides in LDS");
constexpr int TRACE_BIG_BYTES = 6 * TRACE_BIG + 64;                   // scratch of the sequential redo: points, polygon, stack, 16-bit image
constexpr int TRACE_WAVE_BYTES = TRACE_BIG_BYTES + 64 * TRACE_SPILL;  // per wave of the grid: the redo's scratch, then the lanes' spill bytes
// Candidates per wave (chunk) of the trace kernel: 64 at full width; 32 or 16 when the launch has too few candidates to
// give every SIMD a wave of 64 (a wave lasts as long as its longest border: with fewer borders per wave the mean wave is
// shorter and the idle SIMDs take the extra waves -- launch_group in irbpp_capi.hip picks by the number of bins).
template <int TRACE_CPW>
__device__ __forceinline__ void trace_body(const Params& P, const State& S, long long* prof) {
    constexpr int SLOT = TRACE_SLOT, CAP = TRACE_CAP, LCAP = TRACE_LDS_CAP, PP = TRACE_P;
    __shared__ __attribute__((aligned(16))) uint8_t slots[TRACE_CPW * SLOT];            // one border per tracing lane
    __shared__ __attribute__((aligned(16))) uint32_t sfr[TRACE_CPW * TRACE_FSTRIDE];    // one level image per tracing lane: row + column frames
    __shared__ uint32_t dps[64 * PP];
    __shared__ uint8_t dpscratch[64 * PP];
    const int lane = threadIdx.x;
    // the eight lists, one after the other, cut into chunks of TRACE_CPW candidates (a list's last chunk may be short)
    const int seg_cap = P.seg_cap;
    int seg_n[NXCD], seg_first[NXCD + 1];                                    // candidates of list s, its first chunk
    seg_first[0] = 0;
#pragma unroll
    for (int s = 0; s < NXCD; ++s) {
        const int n = S.w_total[s * XCD_STRIDE];
        seg_n[s] = n < seg_cap ? n : seg_cap;
        seg_first[s + 1] = seg_first[s] + (seg_n[s] + TRACE_CPW - 1) / TRACE_CPW;
    }
    for (int chunk = blockIdx.x; chunk < seg_first[NXCD]; chunk += gridDim.x) {
        const long long t_start = prof ? (long long)clock64() : 0;
        int seg = 0;
#pragma unroll
        for (int s = 1; s < NXCD; ++s) seg += chunk >= seg_first[s] ? 1 : 0;
        int first_chunk = 0, count = 0;
#pragma unroll
        for (int s = 0; s < NXCD; ++s) if (s == seg) { first_chunk = seg_first[s]; count = seg_n[s]; }
        const int gi = (chunk - first_chunk) * TRACE_CPW + lane;              // my candidate in list `seg`
        const bool have = lane < TRACE_CPW && gi < count;
        const size_t g = (size_t)seg * seg_cap + gi;
        // ---- my candidate, its level image into LDS
        int my_n = 0, rk = 0, x0 = 0, y0 = 0;
        uint32_t* const fr = sfr + (lane < TRACE_CPW ? lane : 0) * TRACE_FSTRIDE;
        uint8_t* const my_slot = slots + (lane < TRACE_CPW ? lane : 0) * SLOT;
        uint8_t* const wave_spill = S.w_big + (size_t)blockIdx.x * TRACE_WAVE_BYTES + TRACE_BIG_BYTES;    // [64][TRACE_SPILL]
        if (have) {
            const uint2 ce = S.w_cand[g];
            const uint32_t e = ce.y;
            const int b = (int)ce.x, img = (int)((e >> 8) & 511u);
            x0 = e & 15u;
            y0 = (e >> 4) & 15u;
            rk = b * P.R + (int)S.w_imgrot[(size_t)b * P.wimg + img];
            const uint4* gi = (const uint4*)(S.w_img + ((size_t)b * P.wimg + img) * 16);
            const uint4 v0 = gi[0], v1 = gi[1];
            const uint32_t w[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
            uint32_t r[16], c[16];
#pragma unroll
            for (int q = 0; q < 8; ++q) {                             // row words 2q, 2q + 1
                r[2 * q] = c[2 * q] = w[q] & 0xFFFFu;
                r[2 * q + 1] = c[2 * q + 1] = w[q] >> 16;
            }
            transpose16(c);                                           // column words for the vertical moves
            frames_store(fr, r, c);
        }
        const long long t_staged = prof ? (long long)clock64() : 0;
        // ---- follow the borders, all lanes in lockstep
        {
            const int n = trace_border_fast(fr, x0, y0, my_slot, LCAP, have, wave_spill + lane * TRACE_SPILL, TRACE_SPILL);
            if (n < 0) { if (lane == 0) atomicOr(S.err, IRBPP_DEVERR_TRACE_GUARD); }
            else my_n = n;                                            // 0: not the first pixel of its component
        }
        // ---- a border of more than 128 points (not seen in any workload): sequential, in global scratch
        {
            unsigned long long big = __ballot(my_n > CAP);
            while (big != 0ull) {
                const int l0 = __ffsll((long long)big) - 1;
                big &= big - 1ull;
                if (lane == l0) {
                    uint8_t* gsc = S.w_big + (size_t)blockIdx.x * TRACE_WAVE_BYTES;       // one scratch per wave of the grid
                    SlotMem m;
                    m.pts = gsc; m.dst = gsc + TRACE_BIG; m.stk = (uint32_t*)(gsc + 2 * TRACE_BIG); m.cap = TRACE_BIG; m.cap_stk = TRACE_BIG;
                    uint16_t* im = (uint16_t*)(gsc + 6 * TRACE_BIG);                      // the plain walk reads 16-bit row and column words
                    for (int q = 0; q < 16; ++q) { im[q] = (uint16_t)frame_raw_row(fr, q); im[16 + q] = (uint16_t)frame_raw_col(fr, q); }
                    if (contour_vertices(im, im + 16, x0, y0, m, S.w_vmask + (size_t)rk * 16) != 0)
                        atomicOr(S.err, IRBPP_DEVERR_TRACE_GUARD);
                    my_n = 0;
                }
            }
        }
        const bool spilled = __ballot(my_n > LCAP) != 0ull;           // (uniform) somebody's points lie partly in global scratch:
        if (spilled) {                                                //   written by one lane, read by others below
            // (workgroup scope = this wave: writer and readers share the CU's L1, the release is a wait for the stores.  Until
            // session 26 this was an AGENT-scope pair, whose release writes back the L2: paid by every wave with a border of
            // more than 56 points, one wave in sixteen)
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        }
        const long long t_traced = prof ? (long long)clock64() : 0;
        // ---- the closed borders go to the polygon kernel in rounds of 64 * PP contour points, borders packed back
        // to back (optionally in two classes, TRACE_SHORT).  A wave traces for as long as its longest border takes and
        // has 2 to 4 rounds' worth of points: approximating them here would stretch the slowest waves, which set the
        // kernel's duration; as records in global memory every round is one work item of irbpp_polygon_kernel.
        auto next_round = [&](int cls, int left, int& wn, int& excl, int& base) -> unsigned long long {
            wn = (cls == 0 ? left <= TRACE_SHORT : true) ? left : 0;
            const int incl = wave_inclusive_sum(wn);
            excl = incl - wn;
            if (__builtin_amdgcn_readlane(incl, 63) == 0) return 0ull;
            const unsigned long long todo = __ballot(wn > 0);
            base = __builtin_amdgcn_readlane(excl, __ffsll((long long)todo) - 1);
            return __ballot(wn > 0 && incl - base <= 64 * PP);
        };
        int n_rounds = 0;                                             // first pass: how many rounds
        {
            int left = my_n;
            for (int cls = 0; cls < 2; ++cls)
                for (;;) {
                    int wn, excl, base = 0;
                    const unsigned long long sel = next_round(cls, left, wn, excl, base);
                    if (sel == 0ull) break;
                    if ((sel >> lane) & 1ull) left = 0;
                    ++n_rounds;
                }
        }
        // their records: one allocation in this XCD's list (L2-local atomic, like the candidate lists)
        const int round_cap = P.round_cap;
        const int xcd = (int)(__builtin_amdgcn_s_getreg((31 << 11) | 20) & (NXCD - 1));
        int at = 0;
        if (n_rounds > 0) {
            if (lane == 0) at = atomicAdd(S.w_nround + xcd * XCD_STRIDE, n_rounds);
            at = __builtin_amdgcn_readfirstlane(at);
        }
        const bool inline_dp = at + n_rounds > round_cap;             // list full: approximate here (never changes results)
        uint8_t* const rec0 = S.w_round + ((size_t)xcd * round_cap + at) * ROUND_BYTES;
        int left = my_n, n_dp = 0;
        for (int cls = 0; cls < 2; ++cls)
        for (;;) {
            int wn, excl, base = 0;
            const unsigned long long sel = next_round(cls, left, wn, excl, base);
            if (sel == 0ull) break;
            // which border does the point at position q = u * 64 + lane belong to: border lanes drop their id at
            // the position of their first point, a running maximum over the positions spreads it
#pragma unroll
            for (int u = 0; u < PP; ++u) dps[u * 64 + lane] = 0u;
            IRBPP_WAVE_SYNC();
            if ((sel >> lane) & 1ull) dps[excl - base] = (uint32_t)lane + 1u;
            IRBPP_WAVE_SYNC();
            int mark[PP];
#pragma unroll
            for (int u = 0; u < PP; ++u) mark[u] = (int)dps[u * 64 + lane];
            IRBPP_WAVE_SYNC();
            bool live[PP];
            int pv[PP], jj[PP], nn[PP], sbq[PP], prk[PP];
            const uint8_t* pts[PP];
            int carry = 0;
#pragma unroll
            for (int u = 0; u < PP; ++u) {
                const int run = imax(wave_inclusive_max(mark[u]), carry);
                carry = __builtin_amdgcn_readlane(run, 63);
                const int owner = run - 1;                                    // lane that traced this position's border
                const int on = owner >= 0 ? owner : 0;
                const int ns = __shfl(wn | ((excl - base) << 8), on);           // (one gather for both: wn <= 128, start < 128)
                nn[u] = ns & 255;
                sbq[u] = ns >> 8;
                prk[u] = __shfl(rk, on);
                live[u] = owner >= 0 && u * 64 + lane < sbq[u] + nn[u];
                pts[u] = slots + on * SLOT;
                jj[u] = u * 64 + lane - sbq[u];
                if (!live[u]) { nn[u] = 1; sbq[u] = 0; jj[u] = 0; }
                pv[u] = live[u] ? (int)pts[u][jj[u] < LCAP ? jj[u] : LCAP] : 0;
                if (spilled && live[u] && jj[u] >= LCAP) pv[u] = (int)wave_spill[on * TRACE_SPILL + jj[u] - LCAP];
            }
            if (inline_dp) {
                approx_convex_segmented<PP>(lane, live, pv, jj, nn, sbq, pts, prk, dps, dpscratch, S.w_vmask);
            } else {
                uint8_t* const rec = rec0 + (size_t)n_dp * ROUND_BYTES;       // [pts | n | sb][64 * PP] bytes, then rk words
#pragma unroll
                for (int u = 0; u < PP; ++u) {
                    const int q = u * 64 + lane;
                    rec[q] = (uint8_t)pv[u];
                    rec[64 * PP + q] = (uint8_t)(live[u] ? nn[u] : 0);         // 0: no point at this position
                    rec[2 * 64 * PP + q] = (uint8_t)sbq[u];
                    ((uint32_t*)(rec + 3 * 64 * PP))[q] = (uint32_t)prk[u];
                }
            }
            if ((sel >> lane) & 1ull) left = 0;
            ++n_dp;
        }
        if (inline_dp && n_rounds > 0 && at < round_cap) {            // reserved but unused slots of a full list: no points
            const int lim = at + n_rounds < round_cap ? n_rounds : round_cap - at;
            for (int i = 0; i < lim; ++i)
#pragma unroll
                for (int u = 0; u < PP; ++u) rec0[(size_t)i * ROUND_BYTES + 64 * PP + u * 64 + lane] = 0;
        }
#if defined(IRBPP_AB_POLY_ACCOUNT) || defined(IRBPP_AB_EMIT_ACCOUNT)
        if (false) {
#else
        if (prof && lane == 0) {                 // tooling: this wave's account of its first chunk, in the row of that chunk's first bin
#endif
            const int b0 = (int)S.w_cand[(size_t)seg * seg_cap + (size_t)(chunk - first_chunk) * TRACE_CPW].x;
            long long* row = prof + (size_t)b0 * PHASE_ROW;
            const long long t_end = (long long)clock64();
            row[11] = t_end - t_start;
            row[12] = t_staged - t_start;
            row[13] = t_traced - t_staged;
            row[14] = t_end - t_traced;
            row[15] = 1 | ((long long)n_dp << 20) | ((long long)(count - (chunk - first_chunk) * TRACE_CPW < TRACE_CPW ? count - (chunk - first_chunk) * TRACE_CPW : TRACE_CPW) << 40);
        }
        IRBPP_WAVE_SYNC();
    }
}

// ---------------------------------------------------------------------------------------
// Border following with LANE REFILL (round 6; opt-in: IRBPP_TUNE_TRACE_REFILL -- measured SLOWER, see pick_trace_cpw in irbpp_capi.hip).  A wave of trace_body walks
// until the LONGEST of its 64 borders is closed: on BlockOut a lane walks 9.4 iterations on average and the longest of 64
// takes 34 (tools/trace_length_study.py; profiles/r06), i.e. lanes idle for three quarters of the loop, and no cheap
// property of a level image predicts which border is the long one (correlations <= 0.4).  Here a wave owns a BATCH of
// candidate starts: it starts 64 walks and, whenever THR lanes have closed their borders (or nobody walks any more), (1)
// hands the finished borders on -- the same round records, packed by the same code -- and (2) gives every idle lane the
// next candidate of the batch: the candidate's image words were requested a refill earlier by the lane of the same RANK
// among the idle ones and change lanes with a handful of ds_bpermute, the 16 x 16 transpose and the frame stores run once per
// refill for all takers together.  The loop executes sum / 64 iterations (+ the refills) instead of max per 64.
// Results are the old kernel's, record for record up to their order in the lists (which no consumer looks at).
// ---------------------------------------------------------------------------------------
constexpr int TRACE_REFILL_BATCH = 128, TRACE_REFILL_THR = 32;
template <int BATCH, int THR>
__device__ __forceinline__ void trace_refill_body(const Params& P, const State& S, long long* prof) {
    constexpr int SLOT = TRACE_SLOT, CAP = TRACE_CAP, LCAP = TRACE_LDS_CAP, PP = TRACE_P, NCE = BATCH / 64;
    static_assert(BATCH % 64 == 0 && NCE >= 1 && NCE <= 4, "a batch is a whole number of wave-wide candidate loads");
    __shared__ __attribute__((aligned(16))) uint8_t slots[64 * SLOT];
    __shared__ __attribute__((aligned(16))) uint32_t sfr[64 * TRACE_FSTRIDE];
    __shared__ uint32_t dps[64 * PP];
    __shared__ uint8_t dpscratch[64 * PP];
    const int lane = threadIdx.x;
    const int seg_cap = P.seg_cap;
    int seg_n[NXCD], seg_first[NXCD + 1];
    seg_first[0] = 0;
#pragma unroll
    for (int s = 0; s < NXCD; ++s) {
        const int n = S.w_total[s * XCD_STRIDE];
        seg_n[s] = n < seg_cap ? n : seg_cap;
        seg_first[s + 1] = seg_first[s] + (seg_n[s] + BATCH - 1) / BATCH;
    }
    uint32_t* const fr = sfr + lane * TRACE_FSTRIDE;
    uint8_t* const my_slot = slots + lane * SLOT;
    uint8_t* const wave_spill = S.w_big + (size_t)blockIdx.x * TRACE_WAVE_BYTES + TRACE_BIG_BYTES;    // [64][TRACE_SPILL]
    uint8_t* const my_spill = wave_spill + lane * TRACE_SPILL;
    IRBPP_WALK_TABLES(WT)
    const int round_cap = P.round_cap;
    const int xcd = (int)(__builtin_amdgcn_s_getreg((31 << 11) | 20) & (NXCD - 1));
    for (int chunk = blockIdx.x; chunk < seg_first[NXCD]; chunk += gridDim.x) {
        const long long t_start = prof ? (long long)clock64() : 0;
        int seg = 0;
#pragma unroll
        for (int s = 1; s < NXCD; ++s) seg += chunk >= seg_first[s] ? 1 : 0;
        int first_chunk = 0, count = 0;
#pragma unroll
        for (int s = 0; s < NXCD; ++s) if (s == seg) { first_chunk = seg_first[s]; count = seg_n[s]; }
        const int gi0 = (chunk - first_chunk) * BATCH;
        const int nb = count - gi0 < BATCH ? count - gi0 : BATCH;        // candidates of this batch (uniform, >= 1)
        const uint2* const cl = S.w_cand + (size_t)seg * seg_cap + gi0;
        uint2 ce[NCE];                                                    // candidate u * 64 + lane of the batch: (bin, image << 8 | y0 << 4 | x0)
#pragma unroll
        for (int u = 0; u < NCE; ++u) ce[u] = u * 64 + lane < nb ? cl[u * 64 + lane] : make_uint2(0u, 0u);
        Walk w;
        w.pos = w.n = w.result = w.pos0 = w.pos1 = 0;
        w.nb16 = w.k2 = w.prev = w.s_close = w.run = 0u;
        int rk = 0, x0 = 0, y0 = 0, nxt = 0, n_dp_total = 0, refills = 0;
        // the image of candidate nxt + lane, requested a refill ahead of its use
        uint2 pf_e = make_uint2(0u, 0u);
        uint4 pf0 = make_uint4(0u, 0u, 0u, 0u), pf1 = pf0;
        uint32_t pf_rot = 0u;
        auto prefetch = [&]() {
            const int c = nxt + lane;
            uint2 e = make_uint2(0u, 0u);
#pragma unroll
            for (int u = 0; u < NCE; ++u) {
                const uint32_t ex = (uint32_t)__shfl((int)ce[u].x, c & 63), ey = (uint32_t)__shfl((int)ce[u].y, c & 63);
                if ((c >> 6) == u) e = make_uint2(ex, ey);
            }
            pf_e = e;
            if (c < nb) {
                const size_t at = (size_t)e.x * P.wimg + ((e.y >> 8) & 511u);
                pf_rot = S.w_imgrot[at];
                const uint4* gi = (const uint4*)(S.w_img + at * 16);
                pf0 = gi[0];
                pf1 = gi[1];
            }
        };
        // the closed borders of the lanes that do not walk go to the polygon kernel (trace_body's code, on w.result)
        auto flush = [&]() {
            int my_n = w.run == 0u ? w.result : 0;
            w.result = w.run == 0u ? 0 : w.result;
            if (__ballot(my_n != 0) == 0ull) return;
            {   // a border of more than 128 points (not seen in any workload): sequential, in global scratch
                unsigned long long big = __ballot(my_n > CAP);
                while (big != 0ull) {
                    const int l0 = __ffsll((long long)big) - 1;
                    big &= big - 1ull;
                    if (lane == l0) {
                        uint8_t* gsc = S.w_big + (size_t)blockIdx.x * TRACE_WAVE_BYTES;
                        SlotMem m;
                        m.pts = gsc; m.dst = gsc + TRACE_BIG; m.stk = (uint32_t*)(gsc + 2 * TRACE_BIG); m.cap = TRACE_BIG; m.cap_stk = TRACE_BIG;
                        uint16_t* im = (uint16_t*)(gsc + 6 * TRACE_BIG);
                        for (int q = 0; q < 16; ++q) { im[q] = (uint16_t)frame_raw_row(fr, q); im[16 + q] = (uint16_t)frame_raw_col(fr, q); }
                        if (contour_vertices(im, im + 16, x0, y0, m, S.w_vmask + (size_t)rk * 16) != 0)
                            atomicOr(S.err, IRBPP_DEVERR_TRACE_GUARD);
                        my_n = 0;
                    }
                }
            }
            const bool spilled = __ballot(my_n > LCAP) != 0ull;
            if (spilled) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            }
            auto next_round = [&](int left, int& wn, int& excl, int& base) -> unsigned long long {
                wn = left;
                const int incl = wave_inclusive_sum(wn);
                excl = incl - wn;
                if (__builtin_amdgcn_readlane(incl, 63) == 0) return 0ull;
                const unsigned long long todo = __ballot(wn > 0);
                base = __builtin_amdgcn_readlane(excl, __ffsll((long long)todo) - 1);
                return __ballot(wn > 0 && incl - base <= 64 * PP);
            };
            int n_rounds = 0;
            {
                int left = my_n;
                for (;;) {
                    int wn, excl, base = 0;
                    const unsigned long long sel = next_round(left, wn, excl, base);
                    if (sel == 0ull) break;
                    if ((sel >> lane) & 1ull) left = 0;
                    ++n_rounds;
                }
            }
            int at = 0;
            if (n_rounds > 0) {
                if (lane == 0) at = atomicAdd(S.w_nround + xcd * XCD_STRIDE, n_rounds);
                at = __builtin_amdgcn_readfirstlane(at);
            }
            const bool inline_dp = at + n_rounds > round_cap;
            uint8_t* const rec0 = S.w_round + ((size_t)xcd * round_cap + at) * ROUND_BYTES;
            int left = my_n, n_dp = 0;
            for (;;) {
                int wn, excl, base = 0;
                const unsigned long long sel = next_round(left, wn, excl, base);
                if (sel == 0ull) break;
#pragma unroll
                for (int u = 0; u < PP; ++u) dps[u * 64 + lane] = 0u;
                IRBPP_WAVE_SYNC();
                if ((sel >> lane) & 1ull) dps[excl - base] = (uint32_t)lane + 1u;
                IRBPP_WAVE_SYNC();
                int mark[PP];
#pragma unroll
                for (int u = 0; u < PP; ++u) mark[u] = (int)dps[u * 64 + lane];
                IRBPP_WAVE_SYNC();
                bool live[PP];
                int pv[PP], jj[PP], nn[PP], sbq[PP], prk[PP];
                const uint8_t* pts[PP];
                int carry = 0;
#pragma unroll
                for (int u = 0; u < PP; ++u) {
                    const int run = imax(wave_inclusive_max(mark[u]), carry);
                    carry = __builtin_amdgcn_readlane(run, 63);
                    const int owner = run - 1;
                    const int on = owner >= 0 ? owner : 0;
                    const int ns = __shfl(wn | ((excl - base) << 8), on);
                    nn[u] = ns & 255;
                    sbq[u] = ns >> 8;
                    prk[u] = __shfl(rk, on);
                    live[u] = owner >= 0 && u * 64 + lane < sbq[u] + nn[u];
                    pts[u] = slots + on * SLOT;
                    jj[u] = u * 64 + lane - sbq[u];
                    if (!live[u]) { nn[u] = 1; sbq[u] = 0; jj[u] = 0; }
                    pv[u] = live[u] ? (int)pts[u][jj[u] < LCAP ? jj[u] : LCAP] : 0;
                    if (spilled && live[u] && jj[u] >= LCAP) pv[u] = (int)wave_spill[on * TRACE_SPILL + jj[u] - LCAP];
                }
                if (inline_dp) {
                    approx_convex_segmented<PP>(lane, live, pv, jj, nn, sbq, pts, prk, dps, dpscratch, S.w_vmask);
                } else {
                    uint8_t* const rec = rec0 + (size_t)n_dp * ROUND_BYTES;
#pragma unroll
                    for (int u = 0; u < PP; ++u) {
                        const int q = u * 64 + lane;
                        rec[q] = (uint8_t)pv[u];
                        rec[64 * PP + q] = (uint8_t)(live[u] ? nn[u] : 0);
                        rec[2 * 64 * PP + q] = (uint8_t)sbq[u];
                        ((uint32_t*)(rec + 3 * 64 * PP))[q] = (uint32_t)prk[u];
                    }
                }
                if ((sel >> lane) & 1ull) left = 0;
                ++n_dp;
            }
            if (inline_dp && n_rounds > 0 && at < round_cap) {
                const int lim = at + n_rounds < round_cap ? n_rounds : round_cap - at;
                for (int i = 0; i < lim; ++i)
#pragma unroll
                    for (int u = 0; u < PP; ++u) rec0[(size_t)i * ROUND_BYTES + 64 * PP + u * 64 + lane] = 0;
            }
            n_dp_total += n_dp;
            IRBPP_WAVE_SYNC();
        };
        prefetch();
        int guard = 16384;                                                // (uniform: iterations of this batch's loop)
        for (;;) {
            const unsigned long long walking = __ballot(w.run != 0u);
            const int idle = 64 - __popcll(walking);
            if ((idle >= THR && nxt < nb) || walking == 0ull) {
                flush();
                if (nxt >= nb) break;                                     // (nobody walks and the batch is handed out)
                // every idle lane takes the next candidate: rank r among the idle lanes gets candidate nxt + r, whose image
                // lane r requested at the last refill
                const int rank = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(~walking >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)~walking, 0u));
                const bool take = w.run == 0u && nxt + rank < nb;
                const uint32_t ex = (uint32_t)__shfl((int)pf_e.x, rank), ey = (uint32_t)__shfl((int)pf_e.y, rank);
                const uint32_t rot = (uint32_t)__shfl((int)pf_rot, rank);
                uint32_t wd[8];
                wd[0] = (uint32_t)__shfl((int)pf0.x, rank); wd[1] = (uint32_t)__shfl((int)pf0.y, rank);
                wd[2] = (uint32_t)__shfl((int)pf0.z, rank); wd[3] = (uint32_t)__shfl((int)pf0.w, rank);
                wd[4] = (uint32_t)__shfl((int)pf1.x, rank); wd[5] = (uint32_t)__shfl((int)pf1.y, rank);
                wd[6] = (uint32_t)__shfl((int)pf1.z, rank); wd[7] = (uint32_t)__shfl((int)pf1.w, rank);
                if (take) {
                    x0 = (int)(ey & 15u);
                    y0 = (int)((ey >> 4) & 15u);
                    rk = (int)ex * P.R + (int)rot;
                    uint32_t r[16], c[16];
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        r[2 * q] = c[2 * q] = wd[q] & 0xFFFFu;
                        r[2 * q + 1] = c[2 * q + 1] = wd[q] >> 16;
                    }
                    transpose16(c);
                    frames_store(fr, r, c);
                    walk_start(w, fr, x0, y0, my_slot, LCAP, true, WT);
                }
                nxt += idle < nb - nxt ? idle : nb - nxt;
                ++refills;
                if (nxt < nb) prefetch();
                continue;
            }
            if (w.run != 0u) walk_iter(w, fr, my_slot, LCAP, my_spill, TRACE_SPILL, WT);
            if (--guard == 0) { if (lane == 0) atomicOr(S.err, IRBPP_DEVERR_TRACE_GUARD); break; }
        }
        if (prof && lane == 0) {                 // tooling: this wave's account of its first batch, in the row of that batch's first bin
            const int b0 = (int)cl[0].x;
            long long* row = prof + (size_t)b0 * PHASE_ROW;
            const long long t_end = (long long)clock64();
            row[11] = t_end - t_start;
            row[12] = 0;
            row[13] = t_end - t_start;
            row[14] = (long long)refills | ((long long)(16384 - guard) << 16);     // refills, loop iterations
            row[15] = 1 | ((long long)n_dp_total << 20) | ((long long)nb << 40);
        }
        IRBPP_WAVE_SYNC();
    }
}
extern "C" __global__ void __launch_bounds__(64) irbpp_trace_kernel_refill(const Params P, const State S, long long* prof) {
    trace_refill_body<TRACE_REFILL_BATCH, TRACE_REFILL_THR>(P, S, prof);
}

extern "C" __global__ void __launch_bounds__(64) irbpp_trace_kernel(const Params P, const State S, long long* prof) { trace_body<64>(P, S, prof); }
extern "C" __global__ void __launch_bounds__(64) irbpp_trace_kernel_c32(const Params P, const State S, long long* prof) { trace_body<32>(P, S, prof); }
extern "C" __global__ void __launch_bounds__(64) irbpp_trace_kernel_c16(const Params P, const State S, long long* prof) { trace_body<16>(P, S, prof); }

// Split pipeline, after the trace kernel: approxPolyDP + find_convex_vetex of one round of borders per wave
// (approx_convex_segmented on the 64 * TRACE_P contour points of a record); vertex bits go to the bins' rows in
// global memory with one atomic OR each.  The rounds of the eight lists are numbered through like the chunks of
// the trace kernel; every round is the same amount of work, so the waves finish together.
extern "C" __global__ void __launch_bounds__(64)
irbpp_polygon_kernel(const Params P, const State S
#ifdef IRBPP_AB_POLY_ACCOUNT
                     , long long* prof
#endif
                     ) {
    constexpr int PP = TRACE_P;
#ifdef IRBPP_AB_POLY_ACCOUNT
    long long acct[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    long long* const ac = prof ? acct : nullptr;
    const long long t_begin = (long long)clock64(), w_begin = (long long)wall_clock64();
    long long t_first_loaded = 0, cyc_hops = 0, cyc_dp = 0, cyc_rank = 0, cyc_tail = 0, rounds_done = 0;
#endif
    __shared__ uint32_t dps[64 * PP];
    __shared__ uint8_t dpscratch[64 * PP];
    __shared__ __attribute__((aligned(16))) uint8_t lpts[64 * PP];
    const int lane = threadIdx.x;
    const int round_cap = P.round_cap;
    int seg_n[NXCD], seg_first[NXCD + 1];
    seg_first[0] = 0;
#pragma unroll
    for (int s = 0; s < NXCD; ++s) {
        const int n = S.w_nround[s * XCD_STRIDE];
        seg_n[s] = n < round_cap ? n : round_cap;
        seg_first[s + 1] = seg_first[s] + seg_n[s];
    }
    for (int rd = blockIdx.x; rd < seg_first[NXCD]; rd += gridDim.x) {
        int seg = 0;
#pragma unroll
        for (int s = 1; s < NXCD; ++s) seg += rd >= seg_first[s] ? 1 : 0;
        int first = 0;
#pragma unroll
        for (int s = 0; s < NXCD; ++s) if (s == seg) first = seg_first[s];
        const uint8_t* const rec = S.w_round + ((size_t)seg * round_cap + (rd - first)) * ROUND_BYTES;
        bool live[PP];
        int pv[PP], jj[PP], nn[PP], sbq[PP], prk[PP];
        const uint8_t* pts[PP];
#pragma unroll
        for (int u = 0; u < PP; ++u) {
            const int q = u * 64 + lane;
            pv[u] = rec[q];
            nn[u] = rec[64 * PP + q];
            sbq[u] = rec[2 * 64 * PP + q];
            prk[u] = (int)((const uint32_t*)(rec + 3 * 64 * PP))[q];
            lpts[q] = (uint8_t)pv[u];
        }
        IRBPP_WAVE_SYNC();
#pragma unroll
        for (int u = 0; u < PP; ++u) {
            live[u] = nn[u] > 0;
            jj[u] = u * 64 + lane - sbq[u];
            if (!live[u]) { nn[u] = 1; sbq[u] = 0; jj[u] = 0; pv[u] = 0; }
            pts[u] = lpts + sbq[u];
        }
#ifdef IRBPP_AB_POLY_ACCOUNT
        const long long t_in = (long long)clock64();
        if (rounds_done == 0) t_first_loaded = t_in - t_begin;
        acct[1] = acct[2] = acct[3] = 0;
        approx_convex_segmented<PP>(lane, live, pv, jj, nn, sbq, pts, prk, dps, dpscratch, S.w_vmask, ac);
        const long long t_out = (long long)clock64();
        cyc_hops += acct[1] - t_in; cyc_dp += acct[2] - acct[1]; cyc_rank += acct[3] - acct[2]; cyc_tail += t_out - acct[3];
        ++rounds_done;
#else
        approx_convex_segmented<PP>(lane, live, pv, jj, nn, sbq, pts, prk, dps, dpscratch, S.w_vmask);
#endif
        IRBPP_WAVE_SYNC();
    }
#ifdef IRBPP_AB_POLY_ACCOUNT
    if (prof && lane == 0 && (int)blockIdx.x < P.N) {
        long long* row = prof + (size_t)blockIdx.x * PHASE_ROW;
        row[11] = (long long)clock64() - t_begin;
        row[12] = t_first_loaded | (((long long)(__builtin_amdgcn_s_getreg((31 << 11) | 20) & (NXCD - 1))) << 32);
        row[10] = w_begin;                       // 100 MHz wall clock when this wave started / ended (one clock for all XCDs)
        row[9] = (long long)wall_clock64();
        row[13] = cyc_hops | (cyc_dp << 32);
        row[14] = cyc_rank | (cyc_tail << 32);
        row[15] = rounds_done | (acct[5] << 8) | (acct[6] << 24);
    }
#endif
}

#endif  // IRBPP_PASS == 1
// ---------------------------------------------------------------------------------------
// The environment transition kernel: one workgroup per bin.
// ---------------------------------------------------------------------------------------
// One body, one build per overlap path.  irbpp_env_kernel (block path of lattice data, whose per-bin work is short and
// latency-bound) and the *8 builds are held to 64 VGPRs: eight waves per SIMD, eight workgroups per CU, i.e. 4096 bins
// are exactly two rounds of the chip (measured on the block path: 24.5 / 25.1 / 25.6 M steps/s at 6 / 7 / 8 workgroups
// per CU).  irbpp_env_kernel_wide decides the path at run time and lets the register allocator have what it wants: the
// fallback that irbpp_config::tuning can force for A/B measurements (see pick_env_kernel in irbpp_capi.hip).
template <int PATH, int SPEC, bool FUSED, bool CHAIN = false>
__device__ __forceinline__ void env_transition(const Params& P_run, const Tables& T, const State& S, const StepIO& io,
                                               const int mode, unsigned char* smem);

#if IRBPP_PASS == 1
#ifndef IRBPP_ENV_WAVES
#define IRBPP_ENV_WAVES 8
#endif
// FUSED: the build can apply a step's actions itself (MODE_STEP) as well as observe behind irbpp_apply_kernel (MODE_OBSERVE):
// which of the two a step takes is decided per launch (irbpp_capi.hip: split_apply).  (An observe-only build of the block
// path allocates five SGPR spills instead of sixteen and is not measurably faster.)
#define IRBPP_ENV_KERNEL(NAME, PATH, SPEC, ATTR)                                                                        \
    extern "C" __global__ void __launch_bounds__(BLOCK) ATTR                                                           \
    NAME(const Params P, const Tables T, const State S, const StepIO io, const int mode) {                             \
        extern __shared__ __attribute__((aligned(16))) unsigned char smem[];                                           \
        env_transition<PATH, SPEC, true>(P, T, S, io, mode, smem);                                                     \
    }
// (eight waves per SIMD need <= 64 VGPRs AND <= 96 SGPRs of the SIMD's 800: the cap is on both)
#define IRBPP_CAPPED __attribute__((amdgpu_waves_per_eu(IRBPP_ENV_WAVES, IRBPP_ENV_WAVES)))
IRBPP_ENV_KERNEL(irbpp_env_kernel, PATH_BLOCK, 0, IRBPP_CAPPED)
IRBPP_ENV_KERNEL(irbpp_env_kernel_box8, PATH_BOX, 0, IRBPP_CAPPED)
IRBPP_ENV_KERNEL(irbpp_env_kernel_box, PATH_BOX, 0, )
IRBPP_ENV_KERNEL(irbpp_env_kernel_generic8, PATH_GENERIC, 0, IRBPP_CAPPED)
// (pinning the generic build to seven waves per SIMD with amdgpu_waves_per_eu(7, 7) -- it asks for 71 VGPRs of its own
// accord -- changes the compiler's scheduling for the worse: general 12.8 -> 11.7, abc_fine 5.2 -> 3.8 M steps/s)
IRBPP_ENV_KERNEL(irbpp_env_kernel_generic, PATH_GENERIC, 0, )
IRBPP_ENV_KERNEL(irbpp_env_kernel_wide, PATH_ANY, 0, )
IRBPP_ENV_KERNEL(irbpp_env_kernel_mixed8, PATH_MIXED, 0, IRBPP_CAPPED)
// CHAIN builds (launches of up to CHAIN_BINS bins): the whole observation in the bin's workgroup, one launch (observe_location)
#define IRBPP_ENV_KERNEL_CHAIN(NAME, PATH, SPEC)                                                                        \
    extern "C" __global__ void __launch_bounds__(BLOCK)                                                                \
    NAME(const Params P, const Tables T, const State S, const StepIO io, const int mode) {                             \
        extern __shared__ __attribute__((aligned(16))) unsigned char smem[];                                           \
        env_transition<PATH, SPEC, true, true>(P, T, S, io, mode, smem);                                               \
    }
IRBPP_ENV_KERNEL_CHAIN(irbpp_env_kernel_chain, PATH_ANY, 0)
// specialised builds (irbpp_device.h: SPEC_KEYS): the geometries of BASELINE.json's configs as compile-time constants
#ifndef IRBPP_NO_SPEC
IRBPP_ENV_KERNEL(irbpp_env_kernel_s1, PATH_BLOCK, 1, IRBPP_CAPPED)
IRBPP_ENV_KERNEL(irbpp_env_kernel_s2, PATH_BOX, 2, IRBPP_CAPPED)
IRBPP_ENV_KERNEL(irbpp_env_kernel_s3, PATH_GENERIC, 3, IRBPP_CAPPED)
IRBPP_ENV_KERNEL(irbpp_env_kernel_s4, PATH_GENERIC, 4, )
IRBPP_ENV_KERNEL(irbpp_env_kernel_s5, PATH_MIXED, 5, IRBPP_CAPPED)
IRBPP_ENV_KERNEL_CHAIN(irbpp_env_kernel_chain_s1, PATH_BLOCK, 1)
#endif

#elif IRBPP_PASS == 3
// two waves per bin: the block path of BlockOut at R = 4 (SPEC 1), uncapped (LDS admits five waves per SIMD at most)
extern "C" __global__ void __launch_bounds__(128)
irbpp_env_kernel_s1_w128(const Params P, const Tables T, const State S, const StepIO io, const int mode) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    IRBPP_HERE env_transition<PATH_BLOCK, 1, true>(P, T, S, io, mode, smem);
}
#else
// 512-thread builds of the generic path: the 64 x 64 heightmap's geometry as constants (SPEC 4) and the run-time build
#define IRBPP_ENV_KERNEL_512(NAME, PATH, SPEC, ATTR)                                                                    \
    extern "C" __global__ void __launch_bounds__(512) ATTR                                                             \
    NAME(const Params P, const Tables T, const State S, const StepIO io, const int mode) {                             \
        extern __shared__ __attribute__((aligned(16))) unsigned char smem[];                                           \
        IRBPP_HERE env_transition<PATH, SPEC, true>(P, T, S, io, mode, smem);                                          \
    }
IRBPP_ENV_KERNEL_512(irbpp_env_kernel_generic_w512, PATH_GENERIC, 0, )
#ifndef IRBPP_NO_SPEC
IRBPP_ENV_KERNEL_512(irbpp_env_kernel_s4_w512, PATH_GENERIC, 4, )
IRBPP_ENV_KERNEL_512(irbpp_env_kernel_s4_w512c, PATH_GENERIC, 4, __attribute__((amdgpu_waves_per_eu(8, 8))))
#endif
#endif  // IRBPP_PASS

template <int PATH, int SPEC, bool FUSED, bool CHAIN>
__device__ __forceinline__ void env_transition(const Params& P_run, const Tables& T, const State& S, const StepIO& io,
                                               const int mode, unsigned char* smem) {
    if constexpr (!FUSED) { if (mode == MODE_STEP) return; }         // (never launched that way: irbpp_capi.hip)
    // SPEC > 0: sizes, LDS offsets and division constants become literals (the run-time build keeps reading the kernarg
    // segment where it needs a field: a copy costs it its register allocation)
    Params P_spec;
    const Params& P = SPEC == 0 ? P_run : (P_spec = specialise<SPEC>(P_run), P_spec);
    const Lds L = carve_lds(smem, P);
    // Launch slot -> bin: identity, the caller's list (reset_specific), or -- online steps on large generic data sets --
    // S.order, which irbpp_item_order_kernel groups by observed item per die.
    const bool some = mode == MODE_RESET && io.bin_list != nullptr;          // reset_specific
    const int slot = (int)blockIdx.x + io.block_off;
    const int b = ((mode == MODE_STEP || mode == MODE_CANDS || mode == MODE_OBSERVE) && io.use_order) ? S.order[slot] : some ? io.bin_list[slot] : slot;
    const int tid = threadIdx.x;
    if (b < 0 || b >= P.N) {                                                 // whole workgroup leaves
        if (tid == 0) raise_error(S, IRBPP_DEVERR_BAD_BIN);
        return;
    }
    const long long t_begin = (long long)clock64();
    double* ghm = S.hm + (size_t)b * P.Hc;
    int32_t* q = S.queue + (size_t)b * P.K;
    float* obs = io.obs ? io.obs + (size_t)(some ? slot : b) * io.obs_stride : nullptr;

    stamp(io, b, 0);
    // MODE_STEP: a bin's transition is a chain of dependent global reads (action -> candidate key -> ShapeRot ->
    // footprint cells; bin state -> next item id -> its ShapeRots -> its lists), each a round trip of 1-2 k cycles
    // under load -- more than the arithmetic of the whole apply phase.  The reads are therefore issued as early as
    // their addresses are known, in four rounds, and carried in registers to where they are used.
    int st_a = 0, st_item0 = -1, st_oa = 0, st_nvalid = 0, st_nrows = 0, st_cursor = 0, st_trow = 0;
    int ob_item = -1;                                 // MODE_OBSERVE: the item irbpp_apply_kernel left at the head of the queue
    if (mode == MODE_OBSERVE) ob_item = q[0];
    if (FUSED && mode == MODE_STEP) {                 // round 1: needs only b; in flight together with the tile
        st_a = io.actions[b];
        const BinState* ps0 = S.bs + b;
        st_item0 = ps0->cur_item;
        st_oa = ps0->order_action;
        st_nvalid = ps0->nvalid;
        st_nrows = ps0->nrows;
        st_cursor = ps0->cursor;
        st_trow = ps0->traj_row;
    }
    // stage the heightmap tile
    if (mode == MODE_RESET) {
        for (int i = tid; i < P.tile_words; i += BLOCK) L.hm[i] = 0.0;
        for (int i = tid; i < P.Hc; i += BLOCK) ghm[i] = 0.0;
    } else {
        if (P.tile_words > P.Hc) {                       // odd action grid: the planes have padding entries
            for (int i = tid; i < P.tile_words; i += BLOCK) L.hm[i] = 0.0;
            __syncthreads();
        }
        for (int rep = 0; rep < IRBPP_REPS(5); ++rep)
        for (TileWalk w = IRBPP_HERE tile_walk_begin(P, tid); w.lin < P.Hc; tile_walk_next(w)) L.hm[tile_walk_index(P, w)] = ghm[w.lin];
    }
    uint32_t st_key = 0u;
    int st_next = -2;                                 // -2: not prefetched
    constexpr int SRW_ = sizeof(ShapeRot) / 4;
    int* const sr0 = (int*)(smem + P.o_img);          // the R ShapeRots of the item being placed: the level images' bytes, idle until
    bool ob_staged = false;
    if (mode == MODE_OBSERVE) {                       // round 2 of an observation: the item's ShapeRots, in place before the barrier
        ob_staged = ob_item >= 0 && ob_item < T.n_shapes;
        if (ob_staged && tid < P.R * SRW_) L.sr[tid] = ((const int*)(T.sr + (size_t)ob_item * P.R))[tid];
    }
    if (FUSED && mode == MODE_STEP) {                 //   the overlap test.  round 2: candidate key, the next item of the trajectory,
        // candidates[action] (binPhy.py:235): a negative index counts from the end, anything else outside [0, S) is the
        // reference's IndexError -- here IRBPP_DEVERR_BAD_ACTION (the step then uses a clamped index: results are void)
        st_a += st_a < 0 ? P.S : 0;                                        //   and those ShapeRots (one dword per thread, coalesced)
        if (st_a < 0 || st_a >= P.S) { if (tid == 0) raise_error(S, IRBPP_DEVERR_BAD_ACTION); st_a = st_a < 0 ? 0 : P.S - 1; }
        st_key = st_a < st_nrows ? S.cand[(size_t)b * P.S + st_a] : 0u;     // rows beyond the last are zeros
        if (P.K == 1) {
            const int at = T.stream ? (int)((uint32_t)st_cursor % (uint32_t)T.seq_len) : st_cursor;
            st_next = at < T.seq_len ? T.seq[(long long)st_trow * T.seq_len + at] : -1;
        }
        if (tid < P.R * SRW_) sr0[tid] = st_item0 >= 0 ? ((const int*)(T.sr + (size_t)st_item0 * P.R))[tid] : 0;
    }
    __syncthreads();

    // every mode ends in at most one call of observe_location (single call site: small code)
    int obs_item = -1;
    bool do_observe = true, debug_out = false, sr_staged = false;

    if (mode == MODE_POSSIBLE) {
        obs_item = io.actions[b];
        if (obs_item >= T.n_shapes) obs_item = -1;
        debug_out = true;
    } else if (mode == MODE_CANDS) {     // PackingGame.get_action_candidates (binPhy.py:161-169)
        int oa = io.fixed_slot ? io.fixed_slot - 1 : io.actions[b];
        oa += oa < 0 ? P.K : 0;                  // next_k_item_ID[orderAction] (binPhy.py:163): a list index
        if (oa < 0 || oa >= P.K) { if (tid == 0) raise_error(S, IRBPP_DEVERR_BAD_ACTION); oa = oa < 0 ? 0 : P.K - 1; }
        obs_item = q[oa];
        if (tid == 0 && !io.fixed_slot) S.bs[b].order_action = oa;     // (get_all_possible_observation, binPhy.py:171-180, leaves self.orderAction alone)
    } else if (mode == MODE_OBSERVE) {   // cur_observation of the item the split step left at the head of the queue
        obs_item = ob_item;
        sr_staged = ob_staged;
    } else
    if (mode == MODE_RESET) {            // PackingGame.reset (binPhy.py:128-147)
        if (tid == 0) {
            BinState* ps = S.bs + b;
            // The first reset() starts episode 0.  Every later reset() and reset_specific() is the env's own
            // reset: the item creator moves on to its next trajectory (IRcreator.py:86-92) and the running
            // episode is dropped without statistics (monitor.py reset)
            const int ep = (some || io.reset_next) ? ps->episode + 1 : 0;
            const int trow = trajectory_row(P, T, b, ep);
            const int c0 = T.stream ? ps->cursor : 0;                // a stream goes on, a trajectory starts at its first item
            for (int i = 0; i < P.K; ++i) q[i] = fetch_item(T, S, trow, c0 + i);
            ps->episode = ep;
            ps->traj_row = trow;
            ps->cursor = c0 + P.K;
            ps->cur_item = -1;
            ps->nvalid = 0;
            ps->order_action = 0;
            ps->item_idx = 0;
            ps->ep_len = 0;
            ps->ratio_acc = 0.0;
            ps->ep_reward = 0.0;
            if (!some) for (int i = 0; i < 4; ++i) cold_args()->S.totals[(size_t)b * 4 + i] = 0.0;
            for (int i = 0; i < P.K; ++i) L.redi[16 + i] = q[i];
        }
        __syncthreads();
    } else if constexpr (FUSED) {        // PackingGame.step (binPhy.py:248-337)
        const uint32_t key = st_key;                                 // action_to_position (:234-236)
        const int rot = key >> 16, lx = (key >> 8) & 255, ly = key & 255;
        const int item0 = st_item0;
        const int oa = st_oa;
        bool ok = item0 >= 0 && st_nvalid > 0 && rot < P.R;          // prejudge (:238-245)
        // the placed item's ShapeRot: staged in LDS with round 2 (until round 5: a 112-byte copy per thread from global memory,
        // one more dependent round trip behind the candidate key)
        const bool have_sr = item0 >= 0 && rot < P.R;
        const ShapeRot& sr = ((const ShapeRot*)sr0)[have_sr ? rot : 0];
        // round 3: the volume, and -- speculatively, for the observation that follows a successful placement -- the R
        // ShapeRots of the next item (one dword per thread, stored to LDS further down)
        double vol0 = 0.0;
        if (tid == 0 && item0 >= 0) vol0 = T.volume[item0];
        int sr_pref = 0;
        const bool pref_ok = st_next >= 0 && st_next < T.n_shapes;
        if (pref_ok && tid < P.R * SRW_) sr_pref = ((const int*)(T.sr + (size_t)st_next * P.R))[tid];
        if (ok) {
            const double tx = P_run.txs[lx & 31], ty = P_run.txs[ly & 31];   // np.round(lx*resA, 6), precomputed (indexed in the kernarg segment: the local copy must stay in registers)
            if (round6_scaled(tx + sr.ext_x - P.bin_x) > 0.0 || round6_scaled(ty + sr.ext_y - P.bin_y) > 0.0) ok = false;
        }
        double z = 1e3;                                              // posZmap[rot, lx, ly] (:266)
        // The reference reads the drop height before it looks at `success`, and appends the placement to
        // self.packed either way (binPhy.py:266,296): with a placement log attached, a refused placement needs
        // its height too.
        const bool in_grid = have_sr && lx <= P.Ax - sr.ax && ly <= P.Ay - sr.ay;
        const bool want_z = in_grid && (ok || cold_args()->S.log_meta != nullptr);
        // ... which is an entry of the posZmap the last observation of this bin computed (self.space.posZmap, binPhy.py:266):
        // where naiveMask was set -- every candidate row with V = 1 -- the overlap test left it in w_posz, and the bit row
        // of w_valid says so.  Also round 3, together with the thread's first top cell (heightmap update).
        uint32_t vword = 0u;
        double zc = 1e3;
        if (want_z) {
            vword = S.w_valid[((size_t)b * P.R + rot) * 16 + lx];
            zc = S.w_posz[((size_t)b * P.R + rot) * P.AC + lx * P.Ay + ly];
        }
        Cell tc0 = {};
        const int s_nt = ok ? sr.nt : 0, s_ot = sr.ot;
        if (tid < s_nt) tc0 = T.tcell[s_ot + tid];
        if (want_z) {
            if ((vword >> ly) & 1u) {
                z = zc;
            } else {
                // A cell the observation did not list as valid (a zero-padded row, an action beyond the rows -- nothing a policy
                // that honours the mask column ever picks): its posZmap entry is recomputed from the footprint's bottom cells
                // (space.py:118-119), by ONE thread -- the workgroup-parallel form of this loop cost the kernel registers on a
                // path that practically never runs.
                if (tid == 0) {
                    const Cell* cells = T.bcell + sr.ob;
                    const int s_nb = sr.nb;
                    double m = sr.has_out ? 0.0 : -1e300;
#pragma unroll 1
                    for (int e = 0; e < s_nb; ++e) m = fmax(m, L.hm[tile_of_cell(P, lx, ly, cells[e].ij)] - cells[e].v);
                    L.redd[0] = m;
                }
                __syncthreads();
                z = L.redd[0];
                __syncthreads();
            }
        }
        if (ok) {
            // Interface.simulateHeight (Interface.py:365-369) on the kinematic AABB, x scale
            const double top = z * P.scale_z + sr.ext_z * P.scale_z;
            if (round6_scaled(top - P.ibin_z) > 0.0) ok = false;
        }
        // Stability proxy (irbpp_config::stability; not part of the reference's no-physics path): the item rests on
        // the bottom cells whose gap to the heightmap is within half a height level of the drop height; it is rated
        // stable iff its centre of mass lies inside the octagonal hull (8 support directions) of those cells.
        bool stable = false;
        if (P.stability != 0 && ok) {
            int* sup = L.redi + 20;                                   // support function of the contact cells, 8 directions
            if (tid < 8) sup[tid] = -0x40000000;
            __syncthreads();
            const Cell* cells = T.bcell + sr.ob;
            const double tol = 0.5 * P.res_z;
            int m8[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) m8[k] = -0x40000000;
            for (int e = tid; e < sr.nb; e += BLOCK) {
                if (L.hm[tile_of_cell(P, lx, ly, cells[e].ij)] - cells[e].v >= z - tol) {
                    const int ci = fdiv(cells[e].pad, sr.fy, div_magic_dev(sr.fy)), cj = cells[e].pad - ci * sr.fy;
                    m8[0] = max(m8[0], ci + 1); m8[1] = max(m8[1], -ci); m8[2] = max(m8[2], cj + 1); m8[3] = max(m8[3], -cj);
                    m8[4] = max(m8[4], ci + cj + 2); m8[5] = max(m8[5], -(ci + cj));
                    m8[6] = max(m8[6], ci + 1 - cj); m8[7] = max(m8[7], cj + 1 - ci);
                }
            }
#pragma unroll
            for (int k = 0; k < 8; ++k)
                if (m8[k] > -0x40000000) atomicMax(&sup[k], m8[k]);
            __syncthreads();
            const double cx = sr.com_x, cy = sr.com_y;
            stable = cx <= sup[0] && -cx <= sup[1] && cy <= sup[2] && -cy <= sup[3] && cx + cy <= sup[4] &&
                     -(cx + cy) <= sup[5] && cx - cy <= sup[6] && cy - cx <= sup[7];
            __syncthreads();
            if (P.stability == 2 && !stable) ok = false;
        }
        if (ok) {
            // heightmap update, closed form of space.py:213 (np.maximum with (T + z) * maskH)
            const Cell* cells = T.tcell + s_ot;
            for (int rep = 0; rep < IRBPP_REPS(12); ++rep)
#pragma unroll 1
            for (int e = tid; e < s_nt; e += BLOCK) {
                const Cell tc = e == tid ? tc0 : cells[e];
                const int ij = tc.ij;
                const int row = lx * P.step + (ij & 0xFFFF), col = ly * P.step + (ij >> 16);
                const int c = tile_rc(P, row, col);
                const double h = fmax(L.hm[c], tc.v + z);
                L.hm[c] = h;
                ghm[row * P.Hy + col] = h;
            }
        } else {
            for (int i = tid; i < P.tile_words; i += BLOCK) L.hm[i] = 0.0;                 // Space.reset (space.py:49-52)
            for (int i = tid; i < P.Hc; i += BLOCK) ghm[i] = 0.0;
        }
        if (tid == 0) {
            BinState* ps = S.bs + b;                                 // field-wise: no struct copy (keeps scratch at 0)
            const KernArgsPtr ka = cold_args();                      // outputs, totals, log: loaded here, not at entry
            if (ok) {
                const double vol = vol0;
                const double reward = (vol / P.bin_vol) * 10.0;      // binPhy.py:321-322
                const double epr = ps->ep_reward + reward;
                const int epl = ps->ep_len + 1;
                const int cursor = st_cursor;
                const int slot = ps->item_idx;                       // self.packed.append(...) (binPhy.py:296)
                if (ka->S.log_meta && slot < ka->S.log_cap) {
                    ka->S.log_meta[(size_t)b * ka->S.log_cap + slot] = (uint32_t)item0 | ((uint32_t)rot << 16) |
                                                                ((uint32_t)lx << 20) | ((uint32_t)ly << 24);
                    ka->S.log_z[(size_t)b * ka->S.log_cap + slot] = z;
                }
                ps->ep_reward = epr;
                ps->ep_len = epl;
                ps->item_idx += 1;
                ps->ratio_acc += vol;
                for (int i = oa; i < P.K - 1; ++i) q[i] = q[i + 1];  // update_item_queue (IRcreator.py:22-24)
                int nxt = st_next;                                    // generate_item (:325): prefetched for K == 1
                if (nxt == -2) nxt = fetch_item(T, S, st_trow, cursor);
                else nxt = consume_item(T, S, st_trow, cursor, nxt);   // requested with round 2, consumed here
                q[P.K - 1] = nxt;
                ps->cursor = cursor + 1;
                if (ka->io.reward) ka->io.reward[b] = reward;
                if (ka->io.done) ka->io.done[b] = 0;
                if (ka->io.counter) ka->io.counter[b] = -1;
                if (ka->io.ratio) ka->io.ratio[b] = -1.0;
                if (ka->io.ep_reward) ka->io.ep_reward[b] = epr;
                if (ka->io.ep_len) ka->io.ep_len[b] = epl;
                if (ka->io.stable) ka->io.stable[b] = stable ? 1 : 0;
            } else {
                const int counter = ps->item_idx;                    // info (binPhy.py:306-309)
                const double ratio = ps->ratio_acc / P.bin_vol;      // get_ratio (:149-153)
                const double epr = ps->ep_reward + 0.0;
                const int epl = ps->ep_len + 1;
                if (ka->io.stable) ka->io.stable[b] = 0;
                if (ka->io.reward) ka->io.reward[b] = 0.0;
                if (ka->io.done) ka->io.done[b] = 1;
                if (ka->io.counter) ka->io.counter[b] = counter;
                if (ka->io.ratio) ka->io.ratio[b] = ratio;
                if (ka->io.ep_reward) ka->io.ep_reward[b] = epr;
                if (ka->io.ep_len) ka->io.ep_len[b] = epl;
                if (ka->S.log_meta && counter < ka->S.log_cap) {       // the refused placement is in self.packed too (binPhy.py:296)
                    ka->S.log_meta[(size_t)b * ka->S.log_cap + counter] = (uint32_t)(item0 & 0xFFFF) | ((uint32_t)(rot & 15) << 16) |
                                                                   ((uint32_t)(lx & 15) << 20) | ((uint32_t)(ly & 15) << 24);
                    ka->S.log_z[(size_t)b * ka->S.log_cap + counter] = z;
                }
                double* tot = ka->S.totals + (size_t)b * 4;
                tot[0] += 1.0; tot[1] += ratio; tot[2] += (double)counter; tot[3] += epr;
                // auto-reset (shmem_vec_env.py:142-144) -> PackingGame.reset
                const int ep = ps->episode + 1;
                const int trow = trajectory_row(P, T, b, ep);
                ps->episode = ep;
                ps->traj_row = trow;
                const int c0 = T.stream ? st_cursor : 0;
                for (int i = 0; i < P.K; ++i) q[i] = fetch_item(T, S, trow, c0 + i);
                ps->cursor = c0 + P.K;
                ps->item_idx = 0;
                ps->ratio_acc = 0.0;
                ps->ep_reward = 0.0;
                ps->ep_len = 0;
            }
            for (int i = 0; i < P.K; ++i) L.redi[16 + i] = q[i];
        }
        // the placement stood, so the item observed next is the one whose ShapeRots came with round 3
        if (ok && pref_ok) {
            if (tid < P.R * SRW_) L.sr[tid] = sr_pref;
            sr_staged = true;
        }
        __syncthreads();
    }

    if (mode == MODE_OBSERVE) stamp(io, b, 1);
    if (mode == MODE_RESET || mode == MODE_STEP) {
        stamp(io, b, 1);
        if (P.K == 1) {                  // online: cur_observation with a fresh item (binPhy.py:188-227)
            obs_item = L.redi[16];
            __syncthreads();
        } else {                         // buffer branch (binPhy.py:228-230): [k ids | heightmap]
            do_observe = false;
            for (int i = tid; i < P.K; i += BLOCK) obs[i] = (float)L.redi[16 + i];
            for (int i = tid; i < P.Hc; i += BLOCK) obs[P.K + i] = (float)L.hm[tile_of_linear(P, i)];
        }
    }
    if (do_observe) IRBPP_HERE observe_location<PATH, CHAIN>(P, T, S, io, L, b, obs_item, obs, debug_out, sr_staged, smem + P.lds_bytes);
}

#if IRBPP_PASS == 1
// ---------------------------------------------------------------------------------------
// PackingGame.step without its observation (binPhy.py:248-337 up to the cur_observation call): ONE WAVE per bin.
// Applying an action is a chain of dependent global reads -- action and bin state; candidate key, the placed item's
// ShapeRots; drop height and top cells; the heightmap cells under them -- with a few dozen instructions of arithmetic in
// between: latency, no work.  Inside the transition kernel that chain held a 256-thread workgroup with its 15 KB of LDS
// for 21 k of its 47 k cycles, eight workgroups per CU at a time (profiles/r05/s3, s4).  As a kernel of its own with a wave
// per bin every bin of the launch is resident at once (32 waves per CU: 8192 bins fill the chip exactly), the chain is
// paid once per launch instead of once per round of workgroups, and the heightmap is updated where it lives: the
// footprint's cells of the float64 map in HBM (a few hundred bytes), not a tile.  The observation of the next item is the
// transition kernel's MODE_OBSERVE, launched behind this one.  A buffered environment's step ends here: the order
// observation [k ids | heightmap] is written by the bin's wave.
// Same arithmetic, same order of operations, same outputs as the fused apply phase of env_transition (which the
// stability proxy and IRBPP_TUNE_FUSED_APPLY still run: tests/test_gpu_features.py plays both side by side).
// ---------------------------------------------------------------------------------------
// PER_WORKGROUP (buffered environments): a WORKGROUP per bin -- wave 0 applies the action exactly as below, then all four waves
// write the order observation (the tile's 1024 ... 4096 cells as float32: one wave alone took 16 ... 64 dependent round trips
// for it, which is what made the wave-per-bin form lose for K > 1).
template <bool PER_WORKGROUP>
__device__ __forceinline__ void apply_body(const Params& P, const Tables& T, const State& S, const StepIO& io) {
    const int lane = threadIdx.x & 63;
    const int slot = PER_WORKGROUP ? (int)blockIdx.x : (int)blockIdx.x * WAVES + (int)(threadIdx.x >> 6);
    if (slot >= io.n_slots) return;                                  // (wave-uniform; workgroup-uniform when PER_WORKGROUP)
    const int b = __builtin_amdgcn_readfirstlane(slot + io.block_off);
    double* const ghm = S.hm + (size_t)b * P.Hc;
    int32_t* const q = S.queue + (size_t)b * P.K;
    if (!PER_WORKGROUP || threadIdx.x < 64) {
    // round 1: the action and the bin's scalars
    int a = io.actions[b];
    const BinState* ps0 = S.bs + b;
    const int item0 = ps0->cur_item, oa = ps0->order_action, nvalid0 = ps0->nvalid, nrows = ps0->nrows;
    const int cursor = ps0->cursor, trow = ps0->traj_row;
    // round 2: candidate key; ALL rotations' ShapeRots of the placed item, sixteen bytes per lane (R x 112 bytes are
    // contiguous), so that the one the key names needs no round trip of its own; its volume; the next item of the trajectory
    a += a < 0 ? P.S : 0;                                            // candidates[action] (binPhy.py:235): a negative index counts from the end
    if (a < 0 || a >= P.S) { if (lane == 0) raise_error(S, IRBPP_DEVERR_BAD_ACTION); a = a < 0 ? 0 : P.S - 1; }   // the reference's IndexError
    const uint32_t key = (uint32_t)__builtin_amdgcn_readfirstlane(a < nrows ? (int)S.cand[(size_t)b * P.S + a] : 0);   // action_to_position (:234-236)
    int nxt = -2;
    if (P.K == 1) {
        const int at = T.stream ? (int)((uint32_t)cursor % (uint32_t)T.seq_len) : cursor;
        nxt = at < T.seq_len ? T.seq[(long long)trow * T.seq_len + at] : -1;
    }
    static_assert(sizeof(ShapeRot) == 112, "seven 16-byte chunks per ShapeRot");
    int4 srq = make_int4(0, 0, 0, 0);
    if (item0 >= 0 && lane < P.R * 7) srq = ((const int4*)(T.sr + (size_t)item0 * P.R))[lane];
    const double vol0 = item0 >= 0 ? T.volume[item0] : 0.0;
    const int rot = key >> 16, lx = (key >> 8) & 255, ly = key & 255;
    const bool have_sr = item0 >= 0 && rot < P.R;
    const int rot7 = have_sr ? rot * 7 : 0;
    // dword W (a compile-time offset) of the chosen rotation's ShapeRot: component W % 4 of lane rot * 7 + W / 4
#define IRBPP_SRW(W) __builtin_amdgcn_readlane(((W) & 3) == 0 ? srq.x : ((W) & 3) == 1 ? srq.y : ((W) & 3) == 2 ? srq.z : srq.w, rot7 + ((W) >> 2))
#define IRBPP_SRI(FIELD) IRBPP_SRW(offsetof(ShapeRot, FIELD) / 4)
#define IRBPP_SRD(FIELD) __hiloint2double(IRBPP_SRW(offsetof(ShapeRot, FIELD) / 4 + 1), IRBPP_SRW(offsetof(ShapeRot, FIELD) / 4))
    const int s_ax = IRBPP_SRI(ax), s_ay = IRBPP_SRI(ay), s_nb = IRBPP_SRI(nb), s_ob = IRBPP_SRI(ob), s_has_out = IRBPP_SRI(has_out);
    const int s_ot = IRBPP_SRI(ot);
    const double ext_x = IRBPP_SRD(ext_x), ext_y = IRBPP_SRD(ext_y), ext_z = IRBPP_SRD(ext_z);
    bool ok = item0 >= 0 && nvalid0 > 0 && rot < P.R;                // prejudge (:238-245)
    if (ok) {
        const double tx = P.txs[lx & 31], ty = P.txs[ly & 31];       // np.round(lx*resA, 6), precomputed
        if (round6_scaled(tx + ext_x - P.bin_x) > 0.0 || round6_scaled(ty + ext_y - P.bin_y) > 0.0) ok = false;
    }
    const int s_nt = ok ? IRBPP_SRI(nt) : 0;
#undef IRBPP_SRD
#undef IRBPP_SRI
#undef IRBPP_SRW
    // round 3: the drop height -- posZmap[rot, lx, ly] of the last observation (binPhy.py:266), which the overlap test left in
    // w_posz wherever naiveMask was set (w_valid's bit row says so) -- and the lane's first top cell
    const KernArgsPtr ka = cold_args();
    const bool in_grid = have_sr && lx <= P.Ax - s_ax && ly <= P.Ay - s_ay;
    const bool want_z = in_grid && (ok || ka->S.log_meta != nullptr);
    uint32_t vword = 0u;
    double zc = 1e3;
    if (want_z) {
        vword = S.w_valid[((size_t)b * P.R + rot) * P.vrow + lx];
        zc = S.w_posz[((size_t)b * P.R + rot) * P.AC + lx * P.Ay + ly];
    }
    const Cell* const tcells = T.tcell + s_ot;
    Cell tc0 = {};
    if (lane < s_nt) tc0 = tcells[lane];
    const int row0 = lx * P.step, col0 = ly * P.step;
    // round 4: the heightmap cell under it
    double h0 = 0.0;
    const int i0 = (row0 + (tc0.ij & 0xFFFF)) * P.Hy + col0 + (tc0.ij >> 16);
    if (lane < s_nt) h0 = ghm[i0];
    double z = 1e3;
    if (want_z) {
        if ((vword >> ly) & 1u) {
            z = zc;
        } else {
            // a cell the observation did not list as valid (a zero-padded row, an action beyond the rows): its posZmap entry
            // from the footprint's bottom cells (space.py:118-119)
            const Cell* cells = T.bcell + s_ob;
            double m = s_has_out ? 0.0 : -1e300;
            for (int e = lane; e < s_nb; e += 64) {
                const Cell c = cells[e];
                m = fmax(m, ghm[(row0 + (c.ij & 0xFFFF)) * P.Hy + col0 + (c.ij >> 16)] - c.v);
            }
            z = wave_max_f64(m);
        }
    }
    if (ok) {
        // Interface.simulateHeight (Interface.py:365-369) on the kinematic AABB, x scale
        const double top = z * P.scale_z + ext_z * P.scale_z;
        if (round6_scaled(top - P.ibin_z) > 0.0) ok = false;
    }
    if (ok) {
        // heightmap update, closed form of space.py:213 (np.maximum with (T + z) * maskH)
        if (lane < s_nt) ghm[i0] = fmax(h0, tc0.v + z);
        for (int e0 = 64; e0 < s_nt; e0 += 4 * 64) {                 // (free-form footprints: four cells per lane in flight)
            Cell tc[4];
            int ix[4];
            double hv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int e = e0 + u * 64 + lane;
                tc[u] = tcells[e < s_nt ? e : s_nt - 1];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                ix[u] = (row0 + (tc[u].ij & 0xFFFF)) * P.Hy + col0 + (tc[u].ij >> 16);
                hv[u] = ghm[ix[u]];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (e0 + u * 64 + lane < s_nt) ghm[ix[u]] = fmax(hv[u], tc[u].v + z);
        }
    } else {
        for (int i = lane; i < P.Hc; i += 64) ghm[i] = 0.0;            // Space.reset (space.py:49-52)
    }
    if (lane == 0) {
        BinState* ps = S.bs + b;
        if (ok) {
            const double vol = vol0;
            const double reward = (vol / P.bin_vol) * 10.0;          // binPhy.py:321-322
            const double epr = ps->ep_reward + reward;
            const int epl = ps->ep_len + 1;
            const int slot_i = ps->item_idx;                         // self.packed.append(...) (binPhy.py:296)
            if (ka->S.log_meta && slot_i < ka->S.log_cap) {
                ka->S.log_meta[(size_t)b * ka->S.log_cap + slot_i] = (uint32_t)item0 | ((uint32_t)rot << 16) |
                                                              ((uint32_t)lx << 20) | ((uint32_t)ly << 24);
                ka->S.log_z[(size_t)b * ka->S.log_cap + slot_i] = z;
            }
            ps->ep_reward = epr;
            ps->ep_len = epl;
            ps->item_idx += 1;
            ps->ratio_acc += vol;
            for (int i = oa; i < P.K - 1; ++i) q[i] = q[i + 1];      // update_item_queue (IRcreator.py:22-24)
            if (nxt == -2) nxt = fetch_item(T, S, trow, cursor);     // generate_item (:325): requested with round 2 for K == 1
            else nxt = consume_item(T, S, trow, cursor, nxt);
            q[P.K - 1] = nxt;
            ps->cursor = cursor + 1;
            if (ka->io.reward) ka->io.reward[b] = reward;
            if (ka->io.done) ka->io.done[b] = 0;
            if (ka->io.counter) ka->io.counter[b] = -1;
            if (ka->io.ratio) ka->io.ratio[b] = -1.0;
            if (ka->io.ep_reward) ka->io.ep_reward[b] = epr;
            if (ka->io.ep_len) ka->io.ep_len[b] = epl;
            if (ka->io.stable) ka->io.stable[b] = 0;
        } else {
            const int counter = ps->item_idx;                        // info (binPhy.py:306-309)
            const double ratio = ps->ratio_acc / P.bin_vol;          // get_ratio (:149-153)
            const double epr = ps->ep_reward + 0.0;
            const int epl = ps->ep_len + 1;
            if (ka->io.stable) ka->io.stable[b] = 0;
            if (ka->io.reward) ka->io.reward[b] = 0.0;
            if (ka->io.done) ka->io.done[b] = 1;
            if (ka->io.counter) ka->io.counter[b] = counter;
            if (ka->io.ratio) ka->io.ratio[b] = ratio;
            if (ka->io.ep_reward) ka->io.ep_reward[b] = epr;
            if (ka->io.ep_len) ka->io.ep_len[b] = epl;
            if (ka->S.log_meta && counter < ka->S.log_cap) {           // the refused placement is in self.packed too (binPhy.py:296)
                ka->S.log_meta[(size_t)b * ka->S.log_cap + counter] = (uint32_t)(item0 & 0xFFFF) | ((uint32_t)(rot & 15) << 16) |
                                                               ((uint32_t)(lx & 15) << 20) | ((uint32_t)(ly & 15) << 24);
                ka->S.log_z[(size_t)b * ka->S.log_cap + counter] = z;
            }
            double* tot = ka->S.totals + (size_t)b * 4;
            tot[0] += 1.0; tot[1] += ratio; tot[2] += (double)counter; tot[3] += epr;
            // auto-reset (shmem_vec_env.py:142-144) -> PackingGame.reset
            const int ep = ps->episode + 1;
            const int trow2 = trajectory_row(P, T, b, ep);
            ps->episode = ep;
            ps->traj_row = trow2;
            const int c0 = T.stream ? cursor : 0;
            for (int i = 0; i < P.K; ++i) q[i] = fetch_item(T, S, trow2, c0 + i);
            ps->cursor = c0 + P.K;
            ps->item_idx = 0;
            ps->ratio_acc = 0.0;
            ps->ep_reward = 0.0;
            ps->ep_len = 0;
        }
    }
    }
    if (P.K > 1 && io.obs != nullptr) {  // buffer branch of cur_observation (binPhy.py:228-230): [k ids | heightmap]
        // (the queue was written by lane 0, heightmap cells by whichever lane held their top cell: made visible to the
        // whole wave / workgroup first.  WORKGROUP scope: the waves of a workgroup share their CU's L1, so the release is a
        // wait for the stores and the acquire nothing -- an AGENT-scope release writes back the whole L2 on this chip,
        // which made this step take a millisecond whatever the number of bins, profiles/r05/s25)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        if (PER_WORKGROUP) __syncthreads();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        float* const obs = io.obs + (size_t)b * io.obs_stride;
        const int t0 = PER_WORKGROUP ? (int)threadIdx.x : lane, stride = PER_WORKGROUP ? BLOCK : 64;
        for (int i = t0; i < P.K; i += stride) obs[i] = (float)q[i];
        for (int i0 = 0; i0 < P.Hc; i0 += 8 * stride) {              // eight loads in flight per thread
            double v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) { const int i = i0 + u * stride + t0; v[u] = ghm[i < P.Hc ? i : P.Hc - 1]; }
#pragma unroll
            for (int u = 0; u < 8; ++u) { const int i = i0 + u * stride + t0; if (i < P.Hc) obs[P.K + i] = (float)v[u]; }
        }
    }
}
extern "C" __global__ void __launch_bounds__(BLOCK)
irbpp_apply_kernel(const Params P, const Tables T, const State S, const StepIO io, const int mode) { apply_body<false>(P, T, S, io); }
extern "C" __global__ void __launch_bounds__(BLOCK)
irbpp_apply_wg_kernel(const Params P, const Tables T, const State S, const StepIO io, const int mode) { apply_body<true>(P, T, S, io); }

// Launch order of an online step on the generic path: bins that are about to observe the SAME item run on the same
// die, one after the other.  The item a bin observes next is known before the step (the next entry of its trajectory;
// wrong only where the step ends the episode, which costs nothing but the hint), the footprint lists of an item are
// what the overlap test reads with scalar loads, and a data set's lists (tens of MB at fine resolutions) do not fit a
// die's 4 MB L2: grouped like this a list is fetched from memory once per die and step instead of once per bin.
// Counting sort by item (1024 buckets, monotone in the id), then sorted position s goes to launch slot
// 8 * (s mod N/8) + s div (N/8): workgroup p runs on die p mod 8, so die x gets the contiguous sorted range x.
// One workgroup; a few microseconds.
extern "C" __global__ void __launch_bounds__(1024)
irbpp_item_order_kernel(const Tables T, const State S, int N) {
    __shared__ int hist[1024];
    __shared__ int start[1024];
    const int tid = threadIdx.x;
    hist[tid] = 0;
    __syncthreads();
    // A thread's bins are b = tid + k * 1024.  Each needs two DEPENDENT global reads (cursor and trajectory row, then the item):
    // taken one bin at a time that was 16 + 8 round trips per thread at 8192 bins (40 us, 5 % of a fine-heightmap step); taken
    // EIGHT bins at a time the loads of a chunk are in flight together: two round trips per chunk.
    constexpr int PER = 8;
    auto buckets_of = [&](int k0, int (&out)[PER]) {
        int cur[PER], row[PER], item[PER];
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const int bb = tid + (k0 + u) * 1024;
            const BinState* ps = S.bs + (bb < N ? bb : 0);
            cur[u] = ps->cursor;
            row[u] = ps->traj_row;
        }
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            int c = cur[u];
            if (T.stream) c = (int)((uint32_t)c % (uint32_t)T.seq_len);
            item[u] = c < T.seq_len ? T.seq[(long long)row[u] * T.seq_len + c] : 0;
        }
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const int it = item[u] < 0 ? 0 : (item[u] >= T.n_shapes ? T.n_shapes - 1 : item[u]);
            out[u] = (int)(((long long)it * 1024) / T.n_shapes);
        }
    };
    int mine[PER];                                   // the first chunk's buckets stay in registers (N <= 8192: all of them)
    for (int k0 = 0; k0 * 1024 < N; k0 += PER) {
        int bk[PER];
        buckets_of(k0, bk);
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            if (k0 == 0) mine[u] = bk[u];
            if (tid + (k0 + u) * 1024 < N) atomicAdd(&hist[bk[u]], 1);
        }
    }
    __syncthreads();
    if (tid < 64) {                                  // exclusive scan of the 1024 counts by one wave
        int v[16], sum = 0;
        for (int i = 0; i < 16; ++i) { v[i] = hist[tid * 16 + i]; sum += v[i]; }
        const int inc = wave_inclusive_sum(sum);
        int acc = inc - sum;
        for (int i = 0; i < 16; ++i) { start[tid * 16 + i] = acc; acc += v[i]; }
    }
    __syncthreads();
    const int chunk = N / NXCD;                      // N is a multiple of 8 (irbpp_capi.hip)
    for (int k0 = 0; k0 * 1024 < N; k0 += PER) {
        int bk[PER];
        if (k0 == 0) {
#pragma unroll
            for (int u = 0; u < PER; ++u) bk[u] = mine[u];
        } else {
            buckets_of(k0, bk);
        }
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const int bb = tid + (k0 + u) * 1024;
            if (bb < N) {
                const int s_pos = atomicAdd(&start[bk[u]], 1);
                S.order[NXCD * (s_pos % chunk) + s_pos / chunk] = bb;
            }
        }
    }
}

// Space.get_heuristic_action (space.py:162-218) for the item of the last observation: its own
// kernel, so that the recursion of numpy's pairwise sum (HM) costs the transition kernel nothing.
extern "C" __global__ void __launch_bounds__(BLOCK)
irbpp_heuristic_kernel(const Params P, const Tables T, const State S, const StepIO io) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const Lds L = carve_lds(smem, P);
    const int b = blockIdx.x, tid = threadIdx.x;
    const int R = P.R, AC = P.AC, Ay = P.Ay;
    const int X = fdiv(tid, Ay, P.mg_ay), Y = tid - X * Ay;
    const double* ghm = S.hm + (size_t)b * P.Hc;
    for (int i = tid; i < P.Hc; i += BLOCK) L.hm[tile_of_linear(P, i)] = ghm[i];
    __syncthreads();
    const int item = __builtin_amdgcn_readfirstlane(S.bs[b].cur_item);
    overlap_test<PATH_ANY>(P, T, S, io, L, b, item, false, L.posz, false, true);
    __syncthreads();
    double best = 1e300;
    int best_i = 0x7fffffff;
    for (int r = 0; r < R; ++r) {
        if (tid < AC) {
            double sc = 1e6;
            if (item >= 0) sc = heuristic_score(P, T, io, L, T.sr[item * R + r], r, X, Y);
            const int idx = r * AC + tid;
            if (sc < best || (sc == best && idx < best_i)) { best = sc; best_i = idx; }
        }
    }
    for (int o = 32; o > 0; o >>= 1) {
        const double ob = __shfl_xor(best, o);
        const int oi = __shfl_xor(best_i, o);
        if (ob < best || (ob == best && oi < best_i)) { best = ob; best_i = oi; }
    }
    __syncthreads();
    if ((tid & 63) == 0) { L.redd[tid >> 6] = best; L.redi[4 + (tid >> 6)] = best_i; }
    __syncthreads();
    if (tid == 0) {                      // np.argmin: the first minimum in C order (rot, X, Y)
        for (int w = 1; w < WAVES; ++w)
            if (L.redd[w] < best || (L.redd[w] == best && L.redi[4 + w] < best_i)) { best = L.redd[w]; best_i = L.redi[4 + w]; }
        const int hr = fdiv(best_i, AC, P.mg_ac), hrem = best_i - hr * AC;
        const int hx = fdiv(hrem, Ay, P.mg_ay);
        io.heur_out[b * 3 + 0] = hr;
        io.heur_out[b * 3 + 1] = hx;
        io.heur_out[b * 3 + 2] = hrem - hx * Ay;
    }
}

// shot_item (tools.py:98-135): footprint tables of one mesh by vertical ray casting.  The mesh is
// already rotated and translated so that its bounding-box minimum sits at the origin; ray (i, j)
// passes through (i*res + shift, j*res + shift).  One thread per ray; the triangles go through LDS in chunks of
// SHOT_CHUNK (vertices gathered once per workgroup instead of once per ray, edge-on triangles dropped while
// staging): the bottom table is the lowest intersection, the top table the highest, masks flag a hit.
constexpr int SHOT_CHUNK = 128;
extern "C" __global__ void __launch_bounds__(BLOCK)
irbpp_shot_item_kernel(const double* verts, const int32_t* faces, int n_faces, int fx, int fy, double res, double shift,
                       double* top, double* bottom, double* mtop, double* mbot, int32_t* any_hit) {
    __shared__ double tri[SHOT_CHUNK][10];           // a, b, d (x, y, z each) and the signed double area
    __shared__ int n_staged;
    const int c = blockIdx.x * BLOCK + threadIdx.x;
    const bool mine = c < fx * fy;
    const double px = (double)(c / fy) * res + shift, py = (double)(c % fy) * res + shift;
    double zmin = 1e300, zmax = -1e300;
    bool hit = false;
    for (int f0 = 0; f0 < n_faces; f0 += SHOT_CHUNK) {
        __syncthreads();
        if (threadIdx.x == 0) n_staged = 0;
        __syncthreads();
        if (threadIdx.x < SHOT_CHUNK && f0 + threadIdx.x < n_faces) {
            const int f = f0 + threadIdx.x;
            const double* a = verts + 3 * faces[3 * f + 0];
            const double* b = verts + 3 * faces[3 * f + 1];
            const double* d = verts + 3 * faces[3 * f + 2];
            const double area = (b[0] - a[0]) * (d[1] - a[1]) - (b[1] - a[1]) * (d[0] - a[0]);
            if (area != 0.0) {                                  // edge-on triangle: a vertical ray cannot cross it
                double* t = tri[atomicAdd(&n_staged, 1)];       // (order within a chunk is irrelevant: min / max)
                t[0] = a[0]; t[1] = a[1]; t[2] = a[2]; t[3] = b[0]; t[4] = b[1]; t[5] = b[2];
                t[6] = d[0]; t[7] = d[1]; t[8] = d[2]; t[9] = area;
            }
        }
        __syncthreads();
        if (mine)
            for (int k = 0; k < n_staged; ++k) {
                const double* a = tri[k];
                const double* b = a + 3;
                const double* d = a + 6;
                const double area = a[9];
                const double w0 = (b[0] - px) * (d[1] - py) - (b[1] - py) * (d[0] - px);
                const double w1 = (d[0] - px) * (a[1] - py) - (d[1] - py) * (a[0] - px);
                const double w2 = (a[0] - px) * (b[1] - py) - (a[1] - py) * (b[0] - px);
                const bool inside = area > 0.0 ? (w0 >= 0.0 && w1 >= 0.0 && w2 >= 0.0) : (w0 <= 0.0 && w1 <= 0.0 && w2 <= 0.0);
                if (!inside) continue;
                const double z = (a[2] == b[2] && b[2] == d[2]) ? a[2] : (w0 * a[2] + w1 * b[2] + w2 * d[2]) / area;
                zmin = fmin(zmin, z);
                zmax = fmax(zmax, z);
                hit = true;
            }
    }
    if (!mine) return;
    top[c] = hit ? zmax : 0.0;
    bottom[c] = hit ? zmin : 0.0;
    mtop[c] = hit ? 1.0 : 0.0;
    mbot[c] = hit ? 1.0 : 0.0;
    if (hit) atomicOr(any_hit, 1);
}

// the no-hit-at-all fallback of shot_item (tools.py:112-117,126-131)
extern "C" __global__ void __launch_bounds__(BLOCK)
irbpp_shot_item_fallback_kernel(int n, double extent_z, double* top, double* bottom, double* mtop, double* mbot,
                                const int32_t* any_hit) {
    const int c = blockIdx.x * BLOCK + threadIdx.x;
    if (c >= n || *any_hit) return;
    top[c] = extent_z;
    bottom[c] = 0.0;
    mtop[c] = 1.0;
    mbot[c] = 1.0;
}

// getConvexHullActions on caller-supplied grids (parity tests of the contour stage).
extern "C" __global__ void __launch_bounds__(BLOCK)
irbpp_hull_kernel(const Params P, const State S, const double* posz_valid, const uint8_t* mask,
                  uint32_t* vertex_rows) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const Lds L = carve_lds(smem, P);
    const int g = blockIdx.x, tid = threadIdx.x;
    const int R = P.R, AC = P.AC;
    if (tid < R) L.present[tid] = 0ull;
    for (int i = tid; i < R * 16; i += BLOCK) L.vmask[i] = 0u;
    __syncthreads();
    if (tid < AC) {
        for (int r = 0; r < R; ++r) {
            const size_t gi = ((size_t)g * R + r) * AC + tid;
            const double z = posz_valid[gi];
            const bool valid = mask[gi] != 0;
            int code = 255;
            if (valid) {
                const int li = np_floor_divide_int(z, P.res_z, P.inv_res_z);
                if (li != -1) {
                    const int idx = li + 32;
                    if (idx < 0 || idx > 63) atomicOr(S.err, IRBPP_DEVERR_LEVEL_RANGE);
                    else code = idx;
                }
            }
            L.lev[r * AC + tid] = (uint8_t)code;
            if (code != 255) atomicOr(&L.present[r], 1ull << code);
        }
    }
    __syncthreads();
    contour_stage(P, S, L, nullptr);
    for (int i = tid; i < R * 16; i += BLOCK) vertex_rows[(size_t)g * R * 16 + i] = L.vmask[i];
}

// Scripted policy: lowest-H row with V == 1, first on ties; 0 if none.  One wave per bin.
extern "C" __global__ void __launch_bounds__(BLOCK)
irbpp_policy_minz_kernel(const float* obs, int obs_stride, int S_rows, int N, int32_t* actions) {
    const int wave = (blockIdx.x * BLOCK + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    if (wave >= N) return;
    const float* c = obs + (size_t)wave * obs_stride;
    float best = INFINITY;
    int bi = 0x7fffffff;
    for (int i = lane; i < S_rows; i += 64) {
        const float h = c[i * 5 + 3], v = c[i * 5 + 4];
        if (v == 1.0f && (h < best || (h == best && i < bi))) { best = h; bi = i; }
    }
    for (int o = 32; o > 0; o >>= 1) {
        const float ob = __shfl_xor(best, o);
        const int oi = __shfl_xor(bi, o);
        if (ob < best || (ob == best && oi < bi)) { best = ob; bi = oi; }
    }
    if (lane == 0) actions[wave] = bi == 0x7fffffff ? 0 : bi;
}

// irbpp_stream_cursors / irbpp_stream_write: the host side of the item ring of stream mode
extern "C" __global__ void __launch_bounds__(BLOCK)
irbpp_stream_cursor_kernel(BinState* bs, int32_t* cursors, int N, int set) {
    const int b = blockIdx.x * BLOCK + threadIdx.x;
    if (b >= N) return;
    if (set) bs[b].cursor = cursors[b];
    else cursors[b] = bs[b].cursor;
}
extern "C" __global__ void __launch_bounds__(BLOCK)
irbpp_stream_write_kernel(int32_t* seq, int n_traj, int seq_len, const int32_t* ids, const int32_t* first, const int32_t* count,
                          int width) {
    const long long t = (long long)blockIdx.x * BLOCK + threadIdx.x;
    if (t >= (long long)n_traj * width) return;
    const int row = (int)(t / width), c = (int)(t - (long long)row * width);
    if (c < count[row]) seq[(long long)row * seq_len + (int)((uint32_t)(first[row] + c) % (uint32_t)seq_len)] = ids[t] < -1 ? -1 : ids[t];   // (-3 is the bins' own mark)
}

extern "C" __global__ void __launch_bounds__(BLOCK)
irbpp_totals_kernel(const double* totals, int N, double* out) {
    __shared__ double red[4][WAVES];
    double acc[4] = {0, 0, 0, 0};
    for (int b = threadIdx.x; b < N; b += BLOCK)
        for (int k = 0; k < 4; ++k) acc[k] += totals[(size_t)b * 4 + k];
    for (int k = 0; k < 4; ++k) {
        double v = acc[k];
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
        if ((threadIdx.x & 63) == 0) red[k][threadIdx.x >> 6] = v;
    }
    __syncthreads();
    if (threadIdx.x < 4) {
        double v = 0;
        for (int w = 0; w < WAVES; ++w) v += red[threadIdx.x][w];
        out[threadIdx.x] = v;
    }
}

#endif  // IRBPP_PASS == 1

#if IRBPP_PASS != 1
}  // namespace wg512 / wg128
#endif
}  // namespace irbpp

#if IRBPP_PASS == 1
#undef IRBPP_PASS
#define IRBPP_PASS 2
#include "irbpp_kernels.hip"
#undef IRBPP_PASS
#define IRBPP_PASS 3
#include "irbpp_kernels.hip"
#undef IRBPP_PASS
#define IRBPP_PASS 1
#endif
