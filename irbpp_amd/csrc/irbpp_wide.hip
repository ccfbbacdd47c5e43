// irbpp_wide.hip -- WIDE action grids: 17 .. 32 cells a side (resolutionA = 0.01 on the 0.32 m bin: space.py:19-24 takes any
// resolutionA with an integral stepSize; every README command of the reference uses 0.02 = 16 x 16).
//
// The step's tuned kernels hold a contour point in ONE byte (x | y << 4), a level image in sixteen 16-bit row words and an
// action cell per thread: all of that is the 16 x 16 grid's.  This file is the CAPACITY path for larger grids -- correct first,
// simple on purpose: ONE kernel per observation, one 256-thread workgroup per bin, everything in LDS:
//
//   [irbpp_apply_kernel in front for a step: it is geometry-free]
//   irbpp_wide_kernel: reset / get_action_candidates / observe bookkeeping (env_transition's), overlap test over the footprint's
//   cell lists (space.py:98-129), level codes (cvTools.py:78-79), one level image after the other as 32 rows of 32 bits,
//   candidate starts, the plain border walk and approxPolyDP + convexity lane-serially (contours_device.h:
//   trace_border_wide, approx_and_convex_t<uint16_t, 5>: host-tested against the oracle), candidate rows in np.unique order,
//   the > S selection / the no-candidate fallback by a radix select over keys in LDS, rows, candidate keys, fused MINZ policy.
//
// Same arithmetic, same order of operations as the 16 x 16 pipeline; parity against both oracles in tests/test_gpu_wide.py.
#pragma once

namespace irbpp {

constexpr int WIDE_VROW = 32;                     // words per rotation of w_valid / vertex bits on a wide grid
// (sized so that TWO workgroups share a CU's LDS at four rotations -- two waves per SIMD instead of one: the whole kernel is a chain
// of dependent latencies -- profiles/r06/LOG.md s16)
constexpr int WIDE_TL = 32;                       // lanes of a workgroup that follow borders (eight per wave)
constexpr int WIDE_LCAP = 96;                     // contour points a tracing lane holds in LDS; longer borders: thread 0, global scratch
constexpr int WIDE_IB = 32;                       // level images built and followed at a time
constexpr int WIDE_BIG = 4096;                    // ... of this many points
constexpr int WIDE_BIG_BYTES = WIDE_BIG * (2 + 2 + 4);
// global scratch of a bin: the redo's points / polygon / stack, later the sortable images of the > S selection's values (8 bytes per cell)
__host__ __device__ inline size_t wide_scratch_bytes(const Params& P) {
    const size_t sel = (size_t)P.R * P.AC * 8;
    return sel > (size_t)WIDE_BIG_BYTES ? sel : (size_t)WIDE_BIG_BYTES;
}
constexpr int WIDE_CLIST = 1024;                  // candidate starts of a batch of images: image in batch << 10 | y0 << 5 | x0

struct WideLayout {
    int o_sr, o_present, o_vmask, o_vbits, o_red, o_keys, o_hist, o_rows, o_cnt, o_clist, o_redo, o_lut, o_imglist, o_lev, o_over, o_hm, o_trace, o_skl, bytes;
};
__host__ __device__ inline WideLayout wide_layout(const Params& P) {
    // Three phases share most of the bytes (a workgroup of four rotations needs 49 KB: three per CU):
    //   region K: phase 1 the footprint's staged cells, phase 2 the batch of level images + their transposed copies, the candidate
    //             list and the image tables, phase 3 the candidate keys [R*AC] and the selected keys [S]
    //   region T: phase 1 the heightmap tile, phase 2 the tracing lanes' slots, phase 3 the radix counters / sort keys
    WideLayout w{};
    int off = 0;
    w.o_sr = off;       off += align16(P.R * (int)sizeof(ShapeRot));
    w.o_present = off;  off += align16(P.R * 8);
    w.o_vmask = off;    off += align16(P.R * WIDE_VROW * 4);
    w.o_vbits = off;    off += align16(P.R * WIDE_VROW * 4);
    w.o_red = off;      off += 512;
    w.o_cnt = off;      off += 16;
    w.o_redo = off;     off += 64 * 2;
    w.o_lev = off;      off += align16(P.R * P.AC);
    // region K
    const int k0 = off;
    w.o_keys = k0;
    int k2 = k0;
    w.o_rows = k2;      k2 += 2 * WIDE_IB * 32 * 4;
    w.o_clist = k2;     k2 += WIDE_CLIST * 2;
    w.o_lut = k2;       k2 += align16(P.R * 64 * 2);
    w.o_imglist = k2;   k2 += align16(P.R * 64 * 2);
    const int k3 = k0 + align16((P.R * P.AC + P.S) * 4 + 64);
    off = k2 > k3 ? k2 : k3;
    // region T
    w.o_over = off;
    w.o_hm = w.o_trace = w.o_hist = off;
    w.o_skl = 0;                                                               // (global scratch)
    int npad = 64;
    while (npad < P.S) npad <<= 1;
    const int t1 = align16(P.Hc * 8), t2 = WIDE_TL * (WIDE_LCAP * 2 * 2 + WIDE_LCAP * 4), t3 = align16(10 * npad > 1024 ? 10 * npad : 1024);
    off += t1 > t2 ? (t1 > t3 ? t1 : t3) : (t2 > t3 ? t2 : t3);
    w.bytes = off;
    return w;
}

// The `want` smallest of n keys by (value, position) -- np.argsort(...)[:want] with ties by ascending position (binPhy.py:209-212,
// 217-225) -- keys, their sortable values and the selection all in LDS (select_smallest keeps them in registers: 2048 at most).
template <typename KEY>
__device__ inline void wide_select(const Params& P, const double* zsrc, const uint32_t* vbits, int n, int want, const KEY& key,
                                   uint32_t* out, uint32_t* sel, uint32_t* hist, unsigned long long* skl, int* redi) {
    const int tid = threadIdx.x;
    auto value = [&](uint32_t k) {
        const uint32_t r = k >> 16, x = (k >> 8) & 255u, y = k & 255u;
        if (vbits != nullptr && !((vbits[r * WIDE_VROW + x] >> y) & 1u)) return 1e3;
        return zsrc[r * P.AC + x * P.Ay + y];
    };
    for (int e = tid; e < n; e += BLOCK) skl[e] = sortable_f64(value(key(e)));
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");                // (skl lies in global scratch: stored by one thread, read by others)
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    unsigned long long prefix = 0ull;
    int remaining = want;
    for (int d = 7; d >= 0; --d) {                                       // the want-th smallest value, a byte per round from the top
        hist[tid] = 0u;
        __syncthreads();
        // (the heights of one bin share their high bytes: a wave whose lanes agree on the digit adds its count once instead of 64
        // serialised LDS atomics on one word -- select_smallest's trick)
        for (int e0 = 0; e0 < n; e0 += BLOCK) {
            const int e = e0 + tid;
            const unsigned long long v = e < n ? skl[e] : 0ull;
            const bool in = e < n && (d == 7 || (v >> (8 * (d + 1))) == prefix);
            const int digit = (int)((uint32_t)(v >> (8 * d)) & 255u);
            const unsigned long long act = __ballot(in);
            if (act == 0ull) continue;
            const int first = __ffsll((long long)act) - 1;
            const int c0 = __builtin_amdgcn_readlane(digit, first);
            if (__ballot(in && digit != c0) == 0ull) {
                if ((tid & 63) == first) atomicAdd(&hist[c0], (uint32_t)__popcll(act));
            } else if (in) {
                atomicAdd(&hist[digit], 1u);
            }
        }
        __syncthreads();
        if (tid < 64) {                                                   // one wave scans the 256 counts
            int c[4], sum = 0;
#pragma unroll
            for (int i = 0; i < 4; ++i) { c[i] = (int)hist[tid * 4 + i]; sum += c[i]; }
            const int incl = wave_inclusive_sum(sum);
            int run = incl - sum;
            if (run < remaining && remaining <= incl) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    if (run < remaining && remaining <= run + c[i]) { redi[32] = tid * 4 + i; redi[33] = remaining - run; }
                    run += c[i];
                }
            }
        }
        __syncthreads();
        prefix = (prefix << 8) | (unsigned long long)(uint32_t)redi[32];
        remaining = redi[33];
        __syncthreads();
    }
    // compaction in position order: everything below T, and the first `remaining` elements equal to T (sel may be the array
    // key() reads: a chunk only writes below the positions it has read)
    int nsel = 0, neq = 0;
    for (int e0 = 0; e0 < n; e0 += BLOCK) {
        const int e = e0 + tid;
        const unsigned long long v = e < n ? skl[e] : ~0ull;
        const uint32_t k = e < n ? key(e) : 0u;
        const bool eq = e < n && v == prefix;
        int teq;
        const int eqb = block_scan_flag(eq, redi, teq);
        const bool take = e < n && (v < prefix || (eq && neq + eqb < remaining));
        int tsel;
        const int pos = block_scan_flag(take, redi, tsel);
        __syncthreads();
        if (take) sel[nsel + pos] = k;
        nsel += tsel;
        neq += teq;
    }
    __syncthreads();
    int npad = 64;
    while (npad < want) npad <<= 1;
    uint32_t* const khi = hist;
    uint32_t* const klo = hist + npad;
    uint16_t* const kps = (uint16_t*)(hist + 2 * npad);
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
            __syncthreads();
        }
    for (int r = tid; r < want; r += BLOCK) out[r] = sel[kps[r]];
    __syncthreads();
}

extern "C" __global__ void __launch_bounds__(BLOCK)
irbpp_wide_kernel(const Params P, const Tables T, const State S, const StepIO io, const int mode) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const WideLayout W = wide_layout(P);
    int* const srw = (int*)(smem + W.o_sr);
    uint32_t* const present = (uint32_t*)(smem + W.o_present);        // [R][2]: bit `code` of rotation r
    uint32_t* const vmask = (uint32_t*)(smem + W.o_vmask);            // [R][32] vertex bits: word row, bit col
    uint32_t* const vbits = (uint32_t*)(smem + W.o_vbits);            // [R][32] naiveMask bit rows
    double* const redd = (double*)(smem + W.o_red);
    int* const redi = (int*)(redd + 8);
    uint32_t* const keys = (uint32_t*)(smem + W.o_keys);
    uint32_t* const hist = (uint32_t*)(smem + W.o_hist);
    uint32_t* const rows = (uint32_t*)(smem + W.o_rows);
    uint32_t* const cols = rows + WIDE_IB * 32;                       // [image][x]: bit y = pixel (x, y) (the vertical run jumps)
    int* const cnt = (int*)(smem + W.o_cnt);
    uint16_t* const clist = (uint16_t*)(smem + W.o_clist);
    uint16_t* const redo_list = (uint16_t*)(smem + W.o_redo);       // borders that outgrew a tracing lane's slot
    uint16_t* const lut = (uint16_t*)(smem + W.o_lut);
    uint16_t* const imglist = (uint16_t*)(smem + W.o_imglist);
    double* const hm = (double*)(smem + W.o_hm);
    uint8_t* const lev = smem + W.o_lev;
    const int tid = threadIdx.x, lane = tid & 63;
    const int R = P.R, AC = P.AC, Ax = P.Ax, Ay = P.Ay;
    constexpr int VR = WIDE_VROW;

    const bool some = mode == MODE_RESET && io.bin_list != nullptr;
    const int slot = (int)blockIdx.x + io.block_off;
    const int b = some ? io.bin_list[slot] : slot;
    if (b < 0 || b >= P.N) {
        if (tid == 0) raise_error(S, IRBPP_DEVERR_BAD_BIN);
        return;
    }
    double* const ghm = S.hm + (size_t)b * P.Hc;
    int32_t* const q = S.queue + (size_t)b * P.K;
    float* const obs = io.obs ? io.obs + (size_t)(some ? slot : b) * io.obs_stride : nullptr;
    stamp(io, b, 0);                                 // tooling (irbpp_debug_phase_cycles): 0 start, 1 tile staged, 2 overlap test done,
                                                     // 3 contours done, 4 observation written; 5 / 6: cycles of image building / border following
    long long t_img = 0, t_trace = 0;

    // ---- bookkeeping of the transition (env_transition's, binPhy.py:128-147, 161-169)
    if (mode == MODE_RESET) {
        for (int i = tid; i < P.Hc; i += BLOCK) { hm[i] = 0.0; ghm[i] = 0.0; }
        if (tid == 0) {
            BinState* ps = S.bs + b;
            const int ep = (some || io.reset_next) ? ps->episode + 1 : 0;
            const int trow = trajectory_row(P, T, b, ep);
            const int c0 = T.stream ? ps->cursor : 0;
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
            if (!some) for (int i = 0; i < 4; ++i) S.totals[(size_t)b * 4 + i] = 0.0;
        }
    } else {
        for (int i = tid; i < P.Hc; i += BLOCK) hm[i] = ghm[i];
    }
    __syncthreads();
    int obs_item = -1;
    if (mode == MODE_CANDS) {
        int oa = io.fixed_slot ? io.fixed_slot - 1 : io.actions[b];
        oa += oa < 0 ? P.K : 0;
        if (oa < 0 || oa >= P.K) { if (tid == 0) raise_error(S, IRBPP_DEVERR_BAD_ACTION); oa = oa < 0 ? 0 : P.K - 1; }
        obs_item = q[oa];
        if (tid == 0 && !io.fixed_slot) S.bs[b].order_action = oa;
    } else if (P.K == 1) {                       // MODE_RESET / MODE_OBSERVE of an online environment: the queue's first item
        obs_item = q[0];
    } else {                                     // buffer branch of cur_observation (binPhy.py:228-230): [k ids | heightmap]
        for (int i = tid; i < P.K; i += BLOCK) obs[i] = (float)q[i];
        for (int i = tid; i < P.Hc; i += BLOCK) obs[P.K + i] = (float)hm[i];
        return;
    }
    stamp(io, b, 1);
    int item = __builtin_amdgcn_readfirstlane(obs_item);
    if (item >= T.n_shapes) { if (tid == 0) raise_error(S, IRBPP_DEVERR_BAD_ITEM); item = -1; }

    // ---- Space.get_possible_position (space.py:98-129) over the footprint's masked-in bottom cells
    constexpr int SRW = sizeof(ShapeRot) / 4;
    if (item >= 0) for (int t = tid; t < R * SRW; t += BLOCK) srw[t] = ((const int*)(T.sr + (size_t)item * R))[t];
    for (int i = tid; i < R * 2; i += BLOCK) present[i] = 0u;
    for (int i = tid; i < R * VR; i += BLOCK) { vmask[i] = 0u; vbits[i] = 0u; }
    for (int i = tid; i < R * AC; i += BLOCK) lev[i] = 255;
    __syncthreads();
    double* const zdst = S.w_posz + (size_t)b * R * AC;
    int my_valid = 0;
    for (int r = 0; r < R && item >= 0; ++r) {
        const ShapeRot* sp = (const ShapeRot*)srw + r;
        const int s_ax = sp->ax, s_ay = sp->ay, nb = sp->nb, has_out = sp->has_out;
        const double ext_z_r = sp->ext_z_r;
        // the rotation's masked-in bottom cells, staged in LDS (the candidate keys' bytes, idle until the observation is emitted)
        // as (bottom height, offset of the cell in the row-major tile): every thread walks the same list
        const Cell* cells = T.bcell + sp->ob;
        typedef double cell_pair __attribute__((ext_vector_type(2)));
        cell_pair* const lc = (cell_pair*)keys;
        const int lcap = ((R * AC + P.S) * 4) / 16;
        __syncthreads();
        for (int e = tid; e < nb && e < lcap; e += BLOCK) {
            const Cell ce = cells[e];
            cell_pair v;
            v.x = ce.v;
            v.y = __hiloint2double(0, (ce.ij & 0xFFFF) * P.Hy + (ce.ij >> 16));
            lc[e] = v;
        }
        __syncthreads();
        const int nl = nb < lcap ? nb : lcap;
        for (int c0 = 0; c0 < AC; c0 += BLOCK) {                         // (uniform trip count: np_floor_divide_int votes)
            const int c = c0 + tid;
            const int X = fdiv(c, Ay, P.mg_ay), Y = c - X * Ay;
            const bool in_range = c < AC && X <= Ax - s_ax && Y <= Ay - s_ay;
            double m = has_out ? 0.0 : -1e300;
            if (in_range) {
                const double* h0 = hm + (X * P.step) * P.Hy + Y * P.step;
                int e = 0;
                for (; e + 4 <= nl; e += 4) {
                    cell_pair q[4];
                    double hv[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) q[k] = lc[e + k];
#pragma unroll
                    for (int k = 0; k < 4; ++k) hv[k] = h0[__double2loint(q[k].y)];
#pragma unroll
                    for (int k = 0; k < 4; ++k) m = fmax(m, hv[k] - q[k].x);
                }
                for (; e < nl; ++e) {
                    const cell_pair q = lc[e];
                    m = fmax(m, h0[__double2loint(q.y)] - q.x);
                }
                for (e = nl; e < nb; ++e) {                                // (a list beyond the staging bytes: from global memory)
                    const Cell ce = cells[e];
                    m = fmax(m, h0[(ce.ij & 0xFFFF) * P.Hy + (ce.ij >> 16)] - ce.v);
                }
            }
            const bool valid = in_range && round6_scaled(m + ext_z_r - P.bin_z) <= 0.0;      // np.round(.,6) <= 0 (space.py:120)
            const int li = np_floor_divide_int(valid ? m : 0.0, P.res_z, P.inv_res_z);       // cvTools.py:78
            if (valid) {
                zdst[r * AC + c] = m;
                if (li != -1) {                                            // level -1 is skipped (cvTools.py:84)
                    const int idx = li + 32;
                    if (idx < 0 || idx > 63) raise_error(S, IRBPP_DEVERR_LEVEL_RANGE);
                    else {
                        lev[r * AC + c] = (uint8_t)idx;
                        atomicOr(&present[r * 2 + (idx >> 5)], 1u << (idx & 31));
                    }
                }
                atomicOr(&vbits[r * VR + X], 1u << Y);
                ++my_valid;
            }
        }
    }
    const int nvalid = block_sum_int(my_valid, redi);
    stamp(io, b, 2);
    // the tile is done with after its float32 copy: item vector and heightmap of the observation (binPhy.py:196-203)
    if (tid < 9) obs[5 * P.S + tid] = tid == 0 ? (float)item : 0.0f;
    for (int i = tid; i < P.Hc; i += BLOCK) obs[5 * P.S + 9 + i] = (float)hm[i];
    {   // naiveMask's bit rows: what the next apply looks its drop height up with
        uint32_t* gb = S.w_valid + (size_t)b * R * VR;
        for (int i = tid; i < R * VR; i += BLOCK) gb[i] = vbits[i];
    }
    __syncthreads();

    // ---- getConvexHullActions (cvTools.py:61-102): the level images of all rotations, WIDE_IB at a time -- built as 32 rows of 32
    // bits, their candidate starts listed together, one candidate per tracing lane (sixteen lanes per wave)
    const uint32_t wmask = Ay >= 32 ? 0xFFFFFFFFu : ((1u << Ay) - 1u);
    const int tl = (lane < WIDE_TL / WAVES) ? (tid >> 6) * (WIDE_TL / WAVES) + lane : -1;
    uint8_t* const tbase = smem + W.o_trace + (tl >= 0 ? tl : 0) * (WIDE_LCAP * 2 * 2 + WIDE_LCAP * 4);   // (the tile is done with)
    uint16_t* const tpts = (uint16_t*)tbase;
    uint16_t* const tdst = tpts + WIDE_LCAP;
    uint32_t* const tstk = (uint32_t*)(tdst + WIDE_LCAP);
    uint8_t* const big = S.w_big + (size_t)b * wide_scratch_bytes(P);
    unsigned long long* const skl = (unsigned long long*)big;       // (the > S selection's sortable values: the same global scratch, later)
    for (int i = tid; i < R * 64; i += BLOCK) lut[i] = 0xFFFF;
    __syncthreads();
    if (tid == 0) {                                  // images in (rotation, level) order
        int ni = 0;
        for (int r = 0; r < R; ++r)
            for (int half = 0; half < 2; ++half)
                for (uint32_t pm = present[r * 2 + half]; pm != 0u; pm &= pm - 1u) {
                    const int code = half * 32 + __ffs((int)pm) - 1;
                    imglist[ni] = (uint16_t)((r << 8) | code);
                    lut[r * 64 + code] = (uint16_t)ni;
                    ++ni;
                }
        cnt[2] = ni;
    }
    __syncthreads();
    const int nimg = cnt[2];
    for (int base = 0; base < nimg; base += WIDE_IB) {
        const int nbi = nimg - base < WIDE_IB ? nimg - base : WIDE_IB;
        const long long t0 = io.phase_cycles ? (long long)clock64() : 0;
        for (int i = tid; i < nbi * 32; i += BLOCK) rows[i] = 0u;
        if (tid == 0) { cnt[0] = 0; cnt[1] = 0; }
        __syncthreads();
        for (int r = 0; r < R; ++r)
            for (int c = tid; c < AC; c += BLOCK) {
                const int code = lev[r * AC + c];
                if (code != 255) {
                    const int slot = (int)lut[r * 64 + code] - base;
                    if (slot >= 0 && slot < nbi) {
                        const int X = fdiv(c, Ay, P.mg_ay), Y = c - X * Ay;
                        atomicOr(&rows[slot * 32 + X], 1u << Y);
                    }
                }
            }
        __syncthreads();
        for (int i = tid; i < nbi * 32; i += BLOCK) {                      // column word x of image `slot`
            const uint32_t* im = rows + (i & ~31);
            const int x = i & 31;
            uint32_t cw = 0u;
            for (int y = 0; y < Ax; ++y) cw |= ((im[y] >> x) & 1u) << y;
            cols[i] = cw;
        }
        // candidate starts, one list for the whole batch -- or, should a batch of speckle have more starts than the list holds, one
        // image at a time (an image has at most 16 x 32 of them: every other pixel of a row)
        long long t1 = 0;
        int nsub = 1;
        for (int sub = 0; sub < nsub; ++sub) {
            if (sub > 0 || nsub > 1) { __syncthreads(); if (tid == 0) { cnt[0] = 0; cnt[1] = 0; } __syncthreads(); }
            for (int i = tid; i < nbi * 32; i += BLOCK) {                  // candidate starts of row y of image `slot`
                const int slot = i >> 5, y = i & 31;
                if (y >= Ax || (nsub > 1 && slot != sub)) continue;
                uint32_t cand = start_candidates_wide(rows[i], y > 0 ? rows[i - 1] : 0u, wmask);
                while (cand != 0u) {
                    const int x = __ffs((int)cand) - 1;
                    cand &= cand - 1u;
                    const int at = atomicAdd(&cnt[0], 1);
                    if (at < WIDE_CLIST) clist[at] = (uint16_t)(x | (y << 5) | (slot << 10));
                }
            }
            __syncthreads();
            if (nsub == 1 && cnt[0] > WIDE_CLIST) {                        // (uniform) too many for one list: image by image
                nsub = nbi;
                sub = -1;
                continue;
            }
            const int total = cnt[0];
            if (sub <= 0) { t1 = io.phase_cycles ? (long long)clock64() : 0; t_img += t1 - t0; }
            if (tl >= 0)
                for (int ci = tl; ci < total; ci += WIDE_TL) {
                    const int e = clist[ci], x0 = e & 31, y0 = (e >> 5) & 31, slot = e >> 10;
                    const int r = imglist[base + slot] >> 8;
                    const int n = trace_border_wide_runs<5>(rows + slot * 32, cols + slot * 32, Ay, Ax, x0, y0, tpts, WIDE_LCAP);
                    bool redo = false;
                    if (n < 0) raise_error(S, IRBPP_DEVERR_TRACE_GUARD);
                    else if (n > WIDE_LCAP) redo = true;
                    else if (n > 0) redo = !approx_and_convex_t<uint16_t, 5>(tpts, n, tdst, tstk, WIDE_LCAP, vmask + r * VR);
                    if (redo) {                                            // outgrew the slot: thread 0, in global scratch, below
                        const int at = atomicAdd(&cnt[1], 1);
                        if (at < 64) redo_list[at] = (uint16_t)e;
                    }
                }
            __syncthreads();
            if (cnt[1] > 0 && tid == 0) {
                uint16_t* bp = (uint16_t*)big;
                uint16_t* bd = bp + WIDE_BIG;
                uint32_t* bs = (uint32_t*)(bd + WIDE_BIG);
                // (more long borders than the redo list holds: every candidate of the list once more -- marking a vertex twice is harmless)
                const bool all = cnt[1] > 64;
                const int nredo = all ? total : cnt[1];
                for (int k = 0; k < nredo; ++k) {
                    const int e = all ? clist[k] : redo_list[k], slot = e >> 10;
                    const int r = imglist[base + slot] >> 8;
                    const int n = trace_border_wide_runs<5>(rows + slot * 32, cols + slot * 32, Ay, Ax, e & 31, (e >> 5) & 31, bp, WIDE_BIG);
                    if (n < 0 || n > WIDE_BIG || (n > 0 && !approx_and_convex_t<uint16_t, 5>(bp, n, bd, bs, WIDE_BIG, vmask + r * VR)))
                        raise_error(S, IRBPP_DEVERR_TRACE_GUARD);
                }
            }
            __syncthreads();
        }
        if (io.phase_cycles) t_trace += (long long)clock64() - t1;
    }
    stamp(io, b, 3);
    if (io.phase_cycles && tid == 0) {
        io.phase_cycles[(size_t)b * PHASE_ROW + 5] = t_img;
        io.phase_cycles[(size_t)b * PHASE_ROW + 6] = t_trace;
        io.phase_cycles[(size_t)b * PHASE_ROW + 7] = nimg;
    }

    // ---- cur_observation's candidate block (binPhy.py:204-227): rows per rotation ordered by (col, row) (np.unique, cvTools.py:101)
    if (tid == 0) ((unsigned long long*)redd)[7] = ~0ull;
    const int prev_rows = io.obs_rows != nullptr ? io.obs_rows[b] : -1;
    uint32_t* const okey = keys + R * AC;
    int n = 0;
    if (tid < 64) {
        const int cx = lane & 31;
        for (int r0 = 0; r0 < R; r0 += 2) {
            const int r = r0 + (lane >> 5);
            uint32_t col = 0u;
            if (r < R && cx < Ay)
                for (int cy = 0; cy < Ax; ++cy) col |= ((vmask[r * VR + cy] >> cx) & 1u) << cy;
            const int c = __popc(col), incl = wave_inclusive_sum(c);
            int at = n + incl - c;
            while (col != 0u) {
                const int cy = __ffs((int)col) - 1;
                col &= col - 1u;
                keys[at++] = ((uint32_t)r << 16) | ((uint32_t)cy << 8) | (uint32_t)cx;
            }
            n += __builtin_amdgcn_readlane(incl, 63);
        }
        if (tid == 0) redi[34] = n;
    }
    __syncthreads();
    n = redi[34];
    int nrows;
    bool fallback = false;
    const uint32_t* rws;
    const uint32_t* const gvalid = vbits;
    if (n > 0 && n <= P.S) {
        nrows = n;
        rws = keys;
    } else if (n > P.S) {
        wide_select(P, zdst, nullptr, n, P.S, [&](int e) { return keys[e]; }, okey, keys, hist, skl, redi);
        nrows = P.S;
        rws = okey;
    } else {
        fallback = true;
        const int total_cells = R * AC;
        const int want = total_cells < P.S ? total_cells : P.S;
        auto cell_key = [&](int e) {
            const int r = fdiv(e, AC, P.mg_ac), c = e - r * AC, x = fdiv(c, Ay, P.mg_ay);
            return ((uint32_t)r << 16) | ((uint32_t)x << 8) | (uint32_t)(c - x * Ay);
        };
        if (nvalid == 0) {
            for (int e = tid; e < want; e += BLOCK) okey[e] = cell_key(e);
            __syncthreads();
        } else {
            wide_select(P, zdst, gvalid, total_cells, want, cell_key, okey, keys, hist, skl, redi);
        }
        nrows = want;
        rws = okey;
    }
    __syncthreads();
    int write_rows = P.S;
    if (io.obs_rows != nullptr) {
        if (prev_rows >= 0) write_rows = prev_rows > nrows ? prev_rows : nrows;
        if (tid == 0) io.obs_rows[b] = nrows;
    }
    float best = INFINITY;
    int bi = 0x7fffffff;
    for (int row = tid; row < write_rows; row += BLOCK) {
        float v0 = 0.0f, v1 = 0.0f, v2 = 0.0f, v3 = 0.0f, v4 = 0.0f;
        if (row < nrows) {
            const uint32_t k = rws[row];
            float rv;
            if (fallback) rv = (float)((gvalid[(k >> 16) * VR + ((k >> 8) & 255u)] >> (k & 255u)) & 1u);
            else rv = (float)zdst[(k >> 16) * AC + ((k >> 8) & 255u) * Ay + (k & 255u)];
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
    if (tid == 0) {
        S.bs[b].cur_item = item;
        S.bs[b].nvalid = nvalid;
        S.bs[b].nrows = nrows;
    }
    if (io.auto_action != nullptr) {                                     // the scripted MINZ policy on the rows just written
        for (int o = 32; o > 0; o >>= 1) {
            const float ob = __shfl_xor(best, o);
            const int oi = __shfl_xor(bi, o);
            if (ob < best || (ob == best && oi < bi)) { best = ob; bi = oi; }
        }
        if (lane == 0) {
            const uint32_t fb = (uint32_t)__float_as_int(best + 0.0f);
            const uint32_t su = fb ^ ((fb >> 31) != 0u ? 0xFFFFFFFFu : 0x80000000u);
            atomicMin((unsigned long long*)redd + 7, ((unsigned long long)su << 32) | (uint32_t)bi);
        }
        __syncthreads();
        if (tid == 0) {
            const int won = (int)(uint32_t)(((const unsigned long long*)redd)[7] & 0xFFFFFFFFull);
            io.auto_action[b] = won == 0x7fffffff ? 0 : won;
        }
    }
    stamp(io, b, 4);
    if (blockIdx.x == 0 && tid == 0 && io.err_out != nullptr) atomicOr(io.err_out, *S.err);
}

}  // namespace irbpp
