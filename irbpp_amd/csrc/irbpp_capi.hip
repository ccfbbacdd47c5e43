// irbpp_capi.hip -- host side of libirbpp_hip.so: the C ABI declared in include/irbpp.h.
//
// Owns the per-device state in HBM (heightmaps, item queues, candidate keys, counters), packs
// the shotInfo tables and trajectories once, and launches the transition kernel on the
// caller's stream.  No torch types, no exceptions across the boundary.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <new>
#include <utility>
#include <vector>

#include "../../include/irbpp.h"
#include "irbpp_device.h"
#include "irbpp_kernels.hip"      // single translation unit: kernels + host ABI
#include "irbpp_wide.hip"         // action grids of 17 .. 32 cells a side: the capacity path
#include "irbpp_replay.hip"
#include "irbpp_itemgen.h"

using namespace irbpp;

struct irbpp_env {
    irbpp_config cfg;
    Params P;
    Tables T;
    State S;
    bool shapes_loaded = false, seq_loaded = false, was_reset = false;
    bool item_order = false;               // launch slots grouped by observed item, one contiguous range per XCD (generic path)
    long long* phase_cycles = nullptr;
    int32_t* auto_actions = nullptr;       // irbpp_set_auto_policy
    int heavy_turn = 0;                    // State::w_heavy list of the next observing launch
    int32_t* err_mirror = nullptr;         // irbpp_step_out::err_dev of the last step: every error bit is ORed into it as it is raised
    std::vector<std::pair<const float*, int32_t*>> obs_buffers;   // irbpp_register_obs_buffer: buffer -> rows per bin
    std::vector<hipEvent_t> timing;        // tooling: event pairs around irbpp_env_kernel (ring)
    size_t timing_next = 0, timing_used = 0;
    int timing_every = 1, timing_phase = 0;   // events go around every timing_every-th transition only
    std::vector<void*> allocs;
    char kernel_names[256] = {0};          // irbpp_debug_kernel_info
    // small launches replayed as HIP graphs (launch_env): the launches of a transition, captured once per distinct
    // argument set on a stream of the library's own and replayed on the caller's
    struct GraphEntry { std::vector<uint8_t> key; hipGraphExec_t exec; hipGraph_t graph; uint64_t used; };
    std::vector<GraphEntry> graphs;
    hipStream_t cap_stream = nullptr;
    uint64_t graph_clock = 0, graph_replays = 0;
    int obs_epoch = 0;                     // bumped by every (un)registration of an observation buffer: part of a graph's key
};

#define HIP_TRY(expr)                                   \
    do {                                                \
        hipError_t _e = (expr);                         \
        if (_e != hipSuccess) return IRBPP_ERR_HIP;     \
    } while (0)

namespace {

template <typename T>
int dev_alloc(irbpp_env* env, T** out, size_t count) {
    void* p = nullptr;
    if (hipMalloc(&p, count * sizeof(T) > 0 ? count * sizeof(T) : sizeof(T)) != hipSuccess) return IRBPP_ERR_NOMEM;
    if (hipMemset(p, 0, count * sizeof(T)) != hipSuccess) { hipFree(p); return IRBPP_ERR_HIP; }
    env->allocs.push_back(p);
    *out = (T*)p;
    return IRBPP_OK;
}

template <typename T>
int dev_upload(irbpp_env* env, const T** out, const T* host, size_t count) {
    T* p = nullptr;
    int rc = dev_alloc(env, &p, count);
    if (rc != IRBPP_OK) return rc;
    if (count && hipMemcpy(p, host, count * sizeof(T), hipMemcpyHostToDevice) != hipSuccess) return IRBPP_ERR_HIP;
    *out = p;
    return IRBPP_OK;
}

inline double round6_host(double x) { return nearbyint(x * 1e6) / 1e6; }   // np.round(x, 6)
constexpr int TRACE_SMALL_GRID = 8192;      // waves of a trace launch over few bins (16 or 32 candidates per wave)
constexpr int TRACE_CPW16_BINS = 0;         // launches over at most this many bins trace 16 candidates per wave ...
constexpr int TRACE_CPW32_BINS = 1024;      // ... 32 per wave (profiles/r04 session 41: +1.5 ... 1.9 % at 512 / 1024 bins, -0.3 % at 2048; 16 per wave loses everywhere)
// dynamic-LDS carve-up of the transition kernel: irbpp::layout_lds (irbpp_device.h, shared with the specialised builds)
void layout_lds_host(Params& P) {
    int pad = 0;
#ifdef IRBPP_ABLATE
    if (const char* e = getenv("IRBPP_LDS_PAD")) pad = atoi(e);   // tooling build only: caps workgroups per CU
#endif
    irbpp::layout_lds(P, pad);
}

// The dynamic-LDS limit is an attribute of the kernel on the device, not of an environment: always raise it to the
// CU's 160 KiB, so that a later, smaller environment cannot lower it under one that is still alive.
int raise_lds_limits() {
    const void* kernels[] = {(const void*)irbpp_env_kernel_wide, (const void*)irbpp_env_kernel, (const void*)irbpp_env_kernel_box,
                             (const void*)irbpp_env_kernel_box8, (const void*)irbpp_env_kernel_generic,
                             (const void*)irbpp_env_kernel_generic8, (const void*)irbpp_env_kernel_mixed8, (const void*)irbpp_hull_kernel,
                             (const void*)irbpp_env_kernel_chain, (const void*)irbpp_wide_kernel,
#if !defined(IRBPP_NO_SPEC)
                             (const void*)irbpp_env_kernel_s1, (const void*)irbpp_env_kernel_s2, (const void*)irbpp_env_kernel_s3,
                             (const void*)irbpp_env_kernel_s4, (const void*)irbpp_env_kernel_s5, (const void*)irbpp_emit_kernel_s5,
                             (const void*)irbpp_env_kernel_chain_s1, (const void*)wg128::irbpp_env_kernel_s1_w128,
                             (const void*)irbpp_emit_kernel_s1, (const void*)irbpp_emit_kernel_s2,
                             (const void*)irbpp_emit_kernel_s3, (const void*)irbpp_emit_kernel_s4,
                             (const void*)irbpp_emit_wave_kernel_s1, (const void*)irbpp_emit_wave_kernel_s2, (const void*)irbpp_emit_wave_kernel_s5,
                             (const void*)wg512::irbpp_env_kernel_s4_w512, (const void*)wg512::irbpp_env_kernel_s4_w512c,
#endif
                             (const void*)wg512::irbpp_env_kernel_generic_w512,
                             (const void*)irbpp_emit_wave_kernel,
                             (const void*)irbpp_emit_kernel, (const void*)irbpp_heuristic_kernel};
    for (const void* k : kernels)
        if (hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return IRBPP_ERR_HIP;
    return IRBPP_OK;
}

}  // namespace

// lattice data through and through (every rotation on the block path) or box data: what the wave-per-bin emit kernel and the
// early split of the apply phase are for; a data set with list rotations (PATH_MIXED) is treated like free-form data there
// radix counters / sort keys of the emit routine's > S selection, behind the transition kernel's carve-up (CHAIN builds): the
// emit kernel's own e_hist region (irbpp_device.h: layout_lds)
static int chain_extra_lds(const Params& P) {
    int npad = 64;
    while (npad < P.S) npad <<= 1;
    return align16(10 * npad > 1024 ? 10 * npad : 1024);
}
static bool all_block(const Params& P) { return P.block_b > 0 && P.block_rots == (1 << P.R) - 1; }
static bool lattice_or_box(const Params& P) { return all_block(P) || P.box != 0; }
// ... except in what its level images look like: unions of rectangles with a few dozen candidates per bin, practically never more
// than S of them -- the wave-per-bin emit kernel's case, not the speckled free-form images the heavy-first list is for
static bool lattice_images(const Params& P) { return P.block_b > 0 || P.box != 0; }

// waves of the largest trace grid a launch over this environment's bins can ask for (16 candidates per wave: four waves per
// bin), at most TRACE_SMALL_GRID of them beyond one per bin: State::w_big holds one scratch per wave of the grid (9 KB each:
// a 1-bin probe environment allocates 37 KB, not 76 MB)
static int trace_grid_cap(int N) {
    const int small = 4 * N < TRACE_SMALL_GRID ? 4 * N : TRACE_SMALL_GRID;
    return N > small ? N : small;
}

extern "C" {

const char* irbpp_status_string(int status) {
    switch (status) {
        case IRBPP_OK: return "ok";
        case IRBPP_ERR_ARG: return "bad argument or unsupported configuration";
        case IRBPP_ERR_HIP: return "HIP runtime error";
        case IRBPP_ERR_STATE: return "call out of order (load shapes and sequences, reset, then step)";
        case IRBPP_ERR_DEVICE: return "device-side error word set";
        case IRBPP_ERR_NOMEM: return "out of device memory";
        default: return "unknown status";
    }
}

int irbpp_version(void) { return 600; }      // 3xx: irbpp_config::tuning / item_stream, unregister / invalidate_obs_buffer, stream ring, itemgen; 5xx: source hash, overlap path, specialised builds

#ifndef IRBPP_SOURCE_HASH
#define IRBPP_SOURCE_HASH "unstamped"
#endif
// (the marker lets build.py find the stamp in the file without loading the library)
const char* irbpp_source_hash(void) { static const char stamp[] = "irbpp-source-hash:" IRBPP_SOURCE_HASH; return stamp + 18; }

int irbpp_create(const irbpp_config* cfg, irbpp_env** out) {
    if (!cfg || !out) return IRBPP_ERR_ARG;
    if (cfg->num_bins > MAX_BINS) return IRBPP_ERR_ARG;
    if (cfg->num_bins < 1 || cfg->n_rot < 1 || cfg->n_rot > 8 || cfg->selected < 1 || cfg->selected > 1024 ||
        cfg->buffer_size < 1 || cfg->buffer_size > 16)
        return IRBPP_ERR_ARG;
    if (!(cfg->resolution_a > 0) || !(cfg->resolution_h > 0) || !(cfg->resolution_z > 0)) return IRBPP_ERR_ARG;
    irbpp_env* env = new (std::nothrow) irbpp_env();
    if (!env) return IRBPP_ERR_NOMEM;
    env->cfg = *cfg;
    memset(&env->T, 0, sizeof(Tables));
    memset(&env->S, 0, sizeof(State));
    Params& P = env->P;
    memset(&P, 0, sizeof(Params));
    P.N = cfg->num_bins;
    P.R = cfg->n_rot;
    P.S = cfg->selected;
    P.K = cfg->buffer_size;
    P.res_a = cfg->resolution_a;
    P.res_h = cfg->resolution_h;
    P.res_z = cfg->resolution_z;
    P.inv_res_z = 1.0 / cfg->resolution_z;
    for (int k = 0; k < 32; ++k) P.txs[k] = round6_host((double)k * cfg->resolution_a);
    P.bin_x = cfg->bin[0];
    P.bin_y = cfg->bin[1];
    P.bin_z = cfg->bin[2];
    P.bin_vol = cfg->bin[0] * cfg->bin[1] * cfg->bin[2];             // np.prod(bin_dimension)
    P.scale_z = cfg->scale_z;
    P.ibin_z = round6_host(cfg->bin[2] * cfg->scale_z);              // Interface.py:39-40
    // Space.__init__ (space.py:19-24)
    P.step = (int)(cfg->resolution_a / cfg->resolution_h);
    if ((double)P.step != cfg->resolution_a / cfg->resolution_h || P.step < 1) { delete env; return IRBPP_ERR_ARG; }
    P.Hx = (int)ceil(cfg->bin[0] / cfg->resolution_h);
    P.Hy = (int)ceil(cfg->bin[1] / cfg->resolution_h);
    P.Ax = (int)ceil(cfg->bin[0] / cfg->resolution_a);
    P.Ay = (int)ceil(cfg->bin[1] / cfg->resolution_a);
    P.Hc = P.Hx * P.Hy;
    P.AC = P.Ax * P.Ay;
    if (P.Ax > 32 || P.Ay > 32 || P.Ax < 1 || P.Ay < 1 || P.Hc > 128 * 128) { delete env; return IRBPP_ERR_ARG; }
    // 17 .. 32 action cells a side (resolutionA = 0.01): the capacity path of irbpp_wide.hip -- one kernel per observation, every
    // stage in the bin's workgroup; no stability proxy, no item streams' extras are affected, the stage-level tooling entry points
    // (possible_position, heuristic_action, convex_hull_actions) answer IRBPP_ERR_ARG
    P.wide = (P.Ax > 16 || P.Ay > 16) ? 1 : 0;
    P.vrow = P.wide ? WIDE_VROW : 16;
    if (P.wide && (P.Hc > 64 * 64 || cfg->stability != 0)) { delete env; return IRBPP_ERR_ARG; }
    if (P.Hx != P.Ax * P.step || P.Hy != P.Ay * P.step) { delete env; return IRBPP_ERR_ARG; }   // phase-plane tile layout
    // height levels are coded in 6 bits (level + 32): a placement height never exceeds bin_z, so 31 levels must cover it
    if (floor(cfg->bin[2] / cfg->resolution_z + 1e-9) > 31.0) { delete env; return IRBPP_ERR_ARG; }
    P.traj_start = cfg->traj_start;
    P.goff = cfg->global_offset;
    P.gbins = cfg->global_bins > 0 ? cfg->global_bins : cfg->num_bins;
    P.obs_len1 = 5 * P.S + 9 + P.Hc;
    P.obs_len0 = P.K > 1 ? P.K + P.Hc : P.obs_len1;
#ifdef IRBPP_ABLATE
    if (const char* rep = getenv("IRBPP_DEBUG_REPEAT")) P.dbg_repeat = atoi(rep);   // tooling build only (tools/build_variant.sh)
#endif
    P.split = 1;                                            // transition -> trace -> emit kernels
    P.stability = cfg->stability < 0 ? 0 : (cfg->stability > 2 ? 2 : cfg->stability);
    P.rect = (cfg->tuning & IRBPP_TUNE_RECT) ? 1 : 0;
    P.wimg = P.R * 64;
    P.seg_cap = P.wide ? 64 : 2 * ((P.N + NXCD - 1) / NXCD) * P.R * P.AC;      // (the wide path hands nothing over between kernels)
    P.round_cap = P.wide ? 16 : (P.N / NXCD + 64) * 16;
    layout_lds_host(P);                 // redone by irbpp_load_shapes if the block path applies
    if (P.lds_bytes_full > 160 * 1024 || (P.wide && wide_layout(P).bytes > 160 * 1024)) { delete env; return IRBPP_ERR_ARG; }

    if (hipSetDevice(cfg->device) != hipSuccess) { delete env; return IRBPP_ERR_HIP; }
    if (raise_lds_limits() != IRBPP_OK) { delete env; return IRBPP_ERR_HIP; }
    State& S = env->S;
    const size_t N = (size_t)P.N;
    int rc = IRBPP_OK;
#define ALLOC(field, count) if (rc == IRBPP_OK) rc = dev_alloc(env, &S.field, (count))
    ALLOC(hm, N * P.Hc);
    ALLOC(queue, N * P.K);
    ALLOC(cand, N * P.S);
    ALLOC(bs, N);
    ALLOC(totals, N * 4);
    ALLOC(order, N);
    ALLOC(err, 1);
    ALLOC(w_posz, N * P.R * P.AC);
    ALLOC(w_valid, N * P.R * P.vrow);
    ALLOC(w_vmask, N * P.R * 16);
    ALLOC(w_meta, N * WMETA);
    ALLOC(w_img, P.wide ? 16 : N * P.wimg * 16);
    ALLOC(w_imgrot, P.wide ? 16 : N * P.wimg);
    ALLOC(w_cand, (size_t)NXCD * P.seg_cap);
    ALLOC(w_big, P.wide ? N * wide_scratch_bytes(P) : (size_t)trace_grid_cap(P.N) * TRACE_WAVE_BYTES);   // one scratch per wave of the trace grid (wide: per bin)
    ALLOC(w_total, NXCD * XCD_STRIDE);
    ALLOC(w_nround, NXCD * XCD_STRIDE);
    ALLOC(w_round, (size_t)NXCD * P.round_cap * ROUND_BYTES);
    P.heavy_cap = P.N >= 64 ? P.N / 8 : 0;                 // expensive bins the emit kernel serves first (generic data only, see launch_group)
    P.heavy_thr = (P.S * 3) / 5;                          // (with the run-level start filter a > S bin has >= 0.70 S starts + isolated pixels, 99 % of the others < 0.65 S)
    ALLOC(w_heavy, 2 * (size_t)(XCD_STRIDE + P.heavy_cap));
#undef ALLOC
    if (rc != IRBPP_OK) { irbpp_destroy(env); return rc; }
    {   // identity launch order until an ordering pass writes another one
        std::vector<int32_t> ident(N);
        for (size_t i = 0; i < N; ++i) ident[i] = (int32_t)i;
        if (hipMemcpy(S.order, ident.data(), N * sizeof(int32_t), hipMemcpyHostToDevice) != hipSuccess) { irbpp_destroy(env); return IRBPP_ERR_HIP; }
    }
    *out = env;
    return IRBPP_OK;
}

int irbpp_destroy(irbpp_env* env) {
    if (!env) return IRBPP_OK;
    hipSetDevice(env->cfg.device);
    for (void* p : env->allocs) hipFree(p);
    for (hipEvent_t e : env->timing) hipEventDestroy(e);
    for (auto& g : env->graphs) { if (g.exec) hipGraphExecDestroy(g.exec); if (g.graph) hipGraphDestroy(g.graph); }
    if (env->cap_stream) hipStreamDestroy(env->cap_stream);
    delete env;
    return IRBPP_OK;
}

int irbpp_load_shapes(irbpp_env* env, int32_t n_shapes, const double* extents, const double* volumes,
                      const int32_t* dims, const int64_t* offsets, int64_t pool_len,
                      const double* height_top, const double* height_bottom,
                      const double* mask_top, const double* mask_bottom) {
    if (!env || n_shapes < 1 || !extents || !volumes || !dims || !offsets || pool_len < 1 || !height_top ||
        !height_bottom || !mask_top || !mask_bottom)
        return IRBPP_ERR_ARG;
    if (env->shapes_loaded) return IRBPP_ERR_STATE;
    const Params& P = env->P;
    HIP_TRY(hipSetDevice(env->cfg.device));
    const int R = P.R;
    std::vector<ShapeRot> sr((size_t)n_shapes * R);
    std::vector<Cell> bcell, tcell, blkcell;
    std::vector<GCell> gcell;
    for (int64_t i = 0; i < (int64_t)n_shapes * R; ++i) {               // table shapes and offsets first: the scans below trust them
        const int64_t fx = dims[i * 2], fy = dims[i * 2 + 1];
        if (fx < 1 || fy < 1 || fx > 4096 || fy > 4096 || offsets[i] < 0 || offsets[i] + fx * fy > pool_len) return IRBPP_ERR_ARG;
    }
    // Block path: the largest b (multiple of step, <= 8) such that every footprint of the dataset is a
    // union of b x b tiles that are fully masked out or fully masked in with one bottom height -- decided per ROTATION:
    // if every rotation qualifies the data set is pure lattice data (BlockOut at R = 4); if only some do (BlockOut at the
    // README's eight rotations: the four lattice rotations, not the 45-degree ones) those take the block loop and the
    // others their cell lists, in one kernel (PATH_MIXED); b = the largest size with the most rotations.
    int block_b = 0, block_rots = 0;
    if (!(env->cfg.tuning & IRBPP_TUNE_NO_BLOCK_PATH) && !P.wide) {
        int best_count = 0;
        for (int b = 8; b >= 2; --b) {
            if (b % P.step != 0 || (P.Hx - b) % P.step != 0 || (P.Hy - b) % P.step != 0) continue;
            int mask = 0;
            for (int r = 0; r < R; ++r) {
                bool ok = true;
                for (int64_t k = 0; k < n_shapes && ok; ++k) {
                    const int64_t i = k * R + r;
                    const int fx = dims[i * 2], fy = dims[i * 2 + 1];
                    if (fx % b != 0 || fy % b != 0) { ok = false; break; }
                    for (int ti = 0; ti < fx / b && ok; ++ti)
                        for (int tj = 0; tj < fy / b && ok; ++tj) {
                            const int64_t e0 = offsets[i] + (int64_t)(ti * b) * fy + tj * b;
                            for (int u = 0; u < b && ok; ++u)
                                for (int v = 0; v < b && ok; ++v) {
                                    const int64_t e = offsets[i] + (int64_t)(ti * b + u) * fy + tj * b + v;
                                    if (mask_bottom[e] != mask_bottom[e0] ||
                                        (mask_bottom[e0] != 0.0 && height_bottom[e] != height_bottom[e0]))
                                        ok = false;
                                }
                        }
                }
                if (ok) mask |= 1 << r;
            }
            const int count = __builtin_popcount((unsigned)mask);
            if (count > best_count) { best_count = count; block_b = b; block_rots = mask; }
            if (count == R) break;
        }
        // (a single lattice rotation among many list rotations is not worth the block-max grid; IRBPP_TUNE_NO_MIXED_PATH: A/B)
        if (block_rots != (1 << R) - 1 && (2 * best_count < R || (env->cfg.tuning & IRBPP_TUNE_NO_MIXED_PATH))) { block_b = 0; block_rots = 0; }
    }
    const bool mixed = block_b > 0 && block_rots != (1 << R) - 1;
    // Box path: every footprint of the dataset is a solid box -- maskB the rectangle [0,bx) x [0,by), one bottom height
    // over it (the Cube dataset: bottom 0, the ceil-fuzz row of space.py:105 masked out).  max over the window of (H - c)
    // is (max H) - c exactly, and the max over a rectangle is separable: row maxima first, then column maxima.
    bool box = block_b == 0 && !(env->cfg.tuning & IRBPP_TUNE_NO_BOX_PATH) && !P.wide;
    for (int64_t i = 0; i < (int64_t)n_shapes * R && box; ++i) {
        const int fx = dims[i * 2], fy = dims[i * 2 + 1];
        const double* mb = mask_bottom + offsets[i];
        const double* hb = height_bottom + offsets[i];
        int bx = 0, by = 0;
        while (bx < fx && mb[(int64_t)bx * fy] != 0.0) ++bx;
        while (by < fy && mb[by] != 0.0) ++by;
        if (bx == 0 || by == 0) { box = false; break; }
        for (int ci = 0; ci < fx && box; ++ci)
            for (int cj = 0; cj < fy && box; ++cj) {
                const bool in = ci < bx && cj < by;
                if ((mb[(int64_t)ci * fy + cj] != 0.0) != in || (in && hb[(int64_t)ci * fy + cj] != hb[0])) box = false;
            }
    }
    const int mb_h = block_b ? (P.Hx - block_b) / P.step + 1 : 0, mb_w = block_b ? (P.Hy - block_b) / P.step + 1 : 0;
    for (int k = 0; k < n_shapes; ++k) {
        for (int r = 0; r < R; ++r) {
            const size_t i = (size_t)k * R + r;
            ShapeRot& s = sr[i];
            memset(&s, 0, sizeof(s));
            s.ext_x = extents[i * 3 + 0];
            s.ext_y = extents[i * 3 + 1];
            s.ext_z = extents[i * 3 + 2];
            const double bx = round6_host(s.ext_x), by = round6_host(s.ext_y);     // space.py:104
            s.ext_z_r = round6_host(s.ext_z);
            s.fx = (int32_t)ceil(bx / P.res_h);                                   // space.py:105
            s.fy = (int32_t)ceil(by / P.res_h);
            s.ax = (int32_t)ceil(bx / P.res_a);                                   // space.py:106
            s.ay = (int32_t)ceil(by / P.res_a);
            const int64_t off = offsets[i];
            if (s.fx != dims[i * 2] || s.fy != dims[i * 2 + 1]) return IRBPP_ERR_ARG;   // table shape must match
            if (s.fx < 1 || s.fy < 1 || s.fx > s.ax * P.step || s.fy > s.ay * P.step) return IRBPP_ERR_ARG;
            if (off < 0 || off + (int64_t)s.fx * s.fy > pool_len) return IRBPP_ERR_ARG;
            // compact lists of the masked-in cells, row-major
            s.ob = (int32_t)bcell.size();
            s.ot = (int32_t)tcell.size();
            for (int ci = 0; ci < s.fx; ++ci) {
                for (int cj = 0; cj < s.fy; ++cj) {
                    const int64_t e = off + (int64_t)ci * s.fy + cj;
                    const double mt = mask_top[e], mb = mask_bottom[e];
                    if ((mt != 0.0 && mt != 1.0) || (mb != 0.0 && mb != 1.0)) return IRBPP_ERR_ARG;
                    const int32_t ij = ci | (cj << 16);
                    if (mb != 0.0) bcell.push_back(Cell{height_bottom[e], ij, ci * s.fy + cj});
                    else s.has_out = 1;
                    if (mt != 0.0) tcell.push_back(Cell{height_top[e], ij, ci * s.fy + cj});
                }
            }
            s.nb = (int32_t)bcell.size() - s.ob;
            s.nt = (int32_t)tcell.size() - s.ot;
            {   // centre of mass of the solid between bottom and top table (columns hit from both sides)
                double mass = 0.0, mx = 0.0, my = 0.0;
                for (int ci = 0; ci < s.fx; ++ci)
                    for (int cj = 0; cj < s.fy; ++cj) {
                        const int64_t e = off + (int64_t)ci * s.fy + cj;
                        if (mask_top[e] != 0.0 && mask_bottom[e] != 0.0) {
                            const double w = height_top[e] - height_bottom[e] > 0.0 ? height_top[e] - height_bottom[e] : 0.0;
                            mass += w; mx += w * (ci + 0.5); my += w * (cj + 0.5);
                        }
                    }
                s.com_x = mass > 0.0 ? mx / mass : 0.5 * s.fx;
                s.com_y = mass > 0.0 ? my / mass : 0.5 * s.fy;
            }
            s.oblk = (int32_t)blkcell.size();
            if (block_b && ((block_rots >> r) & 1))
                for (int ti = 0; ti < s.fx / block_b; ++ti)
                    for (int tj = 0; tj < s.fy / block_b; ++tj) {
                        const int64_t e0 = off + (int64_t)(ti * block_b) * s.fy + tj * block_b;
                        if (mask_bottom[e0] != 0.0)
                            blkcell.push_back(Cell{height_bottom[e0], (ti * block_b / P.step) * mb_w + tj * block_b / P.step, 0});
                    }
            s.nblk = (int32_t)blkcell.size() - s.oblk;
            // generic path: the masked-in bottom cells again, as (height, byte offset in the LDS tile relative to the
            // action cell's own entry): cell (ci, cj) of an item on action cell (X, Y) is heightmap cell
            // (X*step + ci, Y*step + cj) = plane (ci % step, cj % step), entry (X + ci / step) * Ay + Y + cj / step
            if (!box && (!block_b || mixed))
                for (int e = 0; e < s.nb; ++e) {
                    const Cell& c = bcell[s.ob + e];
                    const int ci = c.ij & 0xFFFF, cj = c.ij >> 16;
                    const int off = ((ci % P.step) * P.step + cj % P.step) * P.AC + (ci / P.step) * P.Ay + cj / P.step;
                    gcell.push_back(GCell{c.v, off * 8, 0});
                }
            if (box) {
                while (s.bx < s.fx && mask_bottom[off + (int64_t)s.bx * s.fy] != 0.0) ++s.bx;
                while (s.by < s.fy && mask_bottom[off + s.by] != 0.0) ++s.by;
                s.bc = height_bottom[off];
            }
            if (s.nb == 0 && !s.has_out) return IRBPP_ERR_ARG;
        }
    }
    if (bcell.empty()) bcell.push_back(Cell{0.0, 0, 0});
    if (tcell.empty()) tcell.push_back(Cell{0.0, 0, 0});
    if (blkcell.empty()) blkcell.push_back(Cell{0.0, 0, 0});
    // gcell mirrors bcell index for index; on the block / box paths nobody reads it
    if (gcell.empty()) gcell.push_back(GCell{0.0, 0, 0});
    // (the walk requests the next four cells while it works on the current four: up to seven cells past a list's end
    // are loaded and ignored)
    for (int i = 0; i < 8; ++i) gcell.push_back(GCell{0.0, 0, 0});
    Tables& T = env->T;
    int rc = dev_upload(env, &T.sr, sr.data(), sr.size());
    if (rc == IRBPP_OK) rc = dev_upload(env, &T.bcell, (const Cell*)bcell.data(), bcell.size());
    if (rc == IRBPP_OK) rc = dev_upload(env, &T.tcell, (const Cell*)tcell.data(), tcell.size());
    if (rc == IRBPP_OK) rc = dev_upload(env, &T.blkcell, (const Cell*)blkcell.data(), blkcell.size());
    if (rc == IRBPP_OK) rc = dev_upload(env, &T.gcell, (const GCell*)gcell.data(), gcell.size());
    if (rc == IRBPP_OK) rc = dev_upload(env, &T.volume, volumes, (size_t)n_shapes);
    if (rc != IRBPP_OK) return rc;
    T.n_shapes = n_shapes;
    if (block_b || box) {                        // switch the overlap test to the block / box path
        env->P.block_b = block_b;
        env->P.block_rots = block_rots;
        env->P.mb_h = mb_h;
        env->P.mb_w = mb_w;
        env->P.box = box ? 1 : 0;
        layout_lds_host(env->P);
        if (env->P.lds_bytes_full > 160 * 1024) return IRBPP_ERR_ARG;
        if (raise_lds_limits() != IRBPP_OK) return IRBPP_ERR_HIP;
    }
    // generic path on a data set whose footprint lists do not fit a die's L2: online steps launch the bins grouped by
    // observed item per die (irbpp_item_order_kernel)
    env->item_order = (!block_b || mixed) && !box && gcell.size() * sizeof(GCell) > (size_t)8 << 20 && env->P.N % NXCD == 0 &&
                      env->P.N >= 64 * NXCD && env->P.K == 1 && !(env->cfg.tuning & IRBPP_TUNE_NO_ITEM_ORDER);
    env->shapes_loaded = true;
    return IRBPP_OK;
}

int irbpp_load_sequences(irbpp_env* env, const int32_t* ids, int32_t n_traj, int32_t length) {
    if (!env || !ids || n_traj < 1 || length < 1) return IRBPP_ERR_ARG;
    if (env->seq_loaded) return IRBPP_ERR_STATE;
    HIP_TRY(hipSetDevice(env->cfg.device));
    int rc = dev_upload(env, &env->T.seq, ids, (size_t)n_traj * length);
    if (rc != IRBPP_OK) return rc;
    env->T.n_traj = n_traj;
    env->T.seq_len = length;
    env->T.stream = env->cfg.item_stream ? 1 : 0;
    env->seq_loaded = true;
    return IRBPP_OK;
}

int irbpp_stream_cursors(irbpp_env* env, int32_t* cursors_dev, int32_t set, void* stream) {
    if (!env || !cursors_dev) return IRBPP_ERR_ARG;
    if (!env->cfg.item_stream || !env->seq_loaded) return IRBPP_ERR_STATE;
    hipLaunchKernelGGL(irbpp_stream_cursor_kernel, dim3((env->P.N + 255) / 256), dim3(256), 0, (hipStream_t)stream, env->S.bs,
                       cursors_dev, env->P.N, set);
    return hipGetLastError() == hipSuccess ? IRBPP_OK : IRBPP_ERR_HIP;
}

int irbpp_stream_write(irbpp_env* env, const int32_t* ids_dev, const int32_t* first_dev, const int32_t* count_dev, int32_t width,
                       void* stream) {
    if (!env || !ids_dev || !first_dev || !count_dev || width < 0) return IRBPP_ERR_ARG;
    if (!env->cfg.item_stream || !env->seq_loaded) return IRBPP_ERR_STATE;
    if (width > env->T.seq_len) return IRBPP_ERR_ARG;
    if (width == 0) return IRBPP_OK;
    env->err_mirror = nullptr;                    // the caller may have cleared a STREAM_DRY condition: seed the next step's word anew
    const long long n = (long long)env->T.n_traj * width;
    hipLaunchKernelGGL(irbpp_stream_write_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       const_cast<int32_t*>(env->T.seq), env->T.n_traj, env->T.seq_len, ids_dev, first_dev, count_dev, width);
    return hipGetLastError() == hipSuccess ? IRBPP_OK : IRBPP_ERR_HIP;
}

int irbpp_obs_len(const irbpp_env* env, int32_t which) {
    if (!env) return IRBPP_ERR_ARG;
    return which == 0 ? env->P.obs_len0 : env->P.obs_len1;
}

// The transition kernel is compiled once per overlap path (lattice blocks, solid boxes, generic cell lists), with and
// without the 64-VGPR cap that makes eight workgroups per CU resident, plus one build that decides at run time.
typedef void (*env_kernel_fn)(const Params, const Tables, const State, const StepIO, const int);
struct EnvKernel { env_kernel_fn fn; const char* name; int threads = 256; };
// Specialised builds (irbpp_device.h): SPEC index whose compile-time constants equal this environment's Params, or 0.
static int pick_spec(const irbpp_env* env) {
#if defined(IRBPP_NO_SPEC) || defined(IRBPP_ABLATE)
    return 0;
#else
    if (env->cfg.tuning & (IRBPP_TUNE_NO_SPECIALISED | IRBPP_TUNE_WIDE_KERNEL | IRBPP_TUNE_NARROW_KERNEL)) return 0;
    static const Params spec[N_SPECS] = {Params{}, spec_params(SPEC_KEYS[1]), spec_params(SPEC_KEYS[2]), spec_params(SPEC_KEYS[3]),
                                         spec_params(SPEC_KEYS[4]), spec_params(SPEC_KEYS[5])};
    static_assert(N_SPECS == 6, "one table entry and one kernel per SPEC_KEYS row");
    for (int i = 1; i < N_SPECS; ++i)
        if (spec_matches(env->P, spec[i])) return i;
    return 0;
#endif
}

// step(): the actions are applied by irbpp_apply_kernel (a wave per bin) and the transition kernel only observes
// (MODE_OBSERVE), unless the stability proxy is on (it rates the placement on the LDS tile) or the caller asks for the fused form
// -- from the launch size on at which that pays.  The apply kernel costs a launch and one pass of its dependent reads
// (~9 us at any size, 15 us with free-form footprints); inside the transition kernel the same chain is paid once per ROUND
// of workgroups (eight per CU), hidden in part behind the other workgroups' arithmetic.  Measured on the specialised builds
// (profiles/r05/s6, placement-steps/s split vs fused): BlockOut 2048 / 4096 / 6144 / 8192 / 16384 bins -4 % / 0 / +1.7 /
// +3.0 / +6.1 %; cube 4096 / 8192: 0 / +2.7 %; free-form solids at R = 8: 4096 -1.1 %, 8192 +0.3 % (BlockOut at R = 8: 0 /
// +1.6 %); the 64 x 64 heightmap (four workgroups per CU, footprints of up to 1600 cells): -2 % at two and at four rounds;
// a buffered step (K > 1: the apply phase and a float32 copy of the tile) with a WAVE per bin: -22 % / -52 % (one wave takes
// 16 dependent round trips to copy the tile) -- with a WORKGROUP per bin (irbpp_apply_wg_kernel: wave 0 applies, all four
// waves copy; no LDS tile, no overlap-test code in the kernel) it wins at every size, see profiles/r05/s25.  With a second
// group of bins on another stream the split pays a round earlier (BlockOut as two groups of 4096: 59.1 -> 60.5 M,
// profiles/r05/s10).  Hence, for online steps: lattice and box data from two rounds of workgroups on, cell lists from four
// rounds on where eight workgroups share a CU.
static bool split_apply(const irbpp_env* env, int n) {
    if (env->P.stability != 0 || (env->cfg.tuning & IRBPP_TUNE_FUSED_APPLY)) return false;
    if (env->cfg.tuning & IRBPP_TUNE_SPLIT_APPLY) return true;
    if (env->P.K > 1) return true;                 // buffered: the workgroup-per-bin form (apply + order observation), at every size
    const int per_cu = (160 * 1024) / (env->P.lds_bytes > 0 ? env->P.lds_bytes : 1);
    if (per_cu < 8) return false;
    const bool lists = !lattice_or_box(env->P);
    return n >= (lists ? 4 : 2) * 256 * 8;
}

static EnvKernel pick_env_kernel(const irbpp_env* env) {
    const Params& P = env->P;
    const int t = env->cfg.tuning;
    const bool lds_allows_8 = 8 * P.lds_bytes <= 160 * 1024;
    // Generic path where the tile is so large that at most four 256-thread workgroups fit a CU's LDS (the 64 x 64 heightmap:
    // 40 KB per bin): 512-thread workgroups, eight waves on one tile (irbpp::wg512, the second pass of irbpp_kernels.hip)
    const bool generic = P.block_b == 0 && !P.box;
    const bool mixed = P.block_b > 0 && !all_block(P);
#if !defined(IRBPP_NO_SPEC) && !defined(IRBPP_ABLATE)
    if (generic && (t & IRBPP_TUNE_NARROW_KERNEL) && (t & IRBPP_TUNE_WG512) && spec_matches(P, spec_params(SPEC_KEYS[4])))
        return {wg512::irbpp_env_kernel_s4_w512c, "irbpp_env_kernel_s4_w512c", 512};      // (A/B: under the 64-VGPR cap)
#endif
    const bool wg512 = generic && !(t & (IRBPP_TUNE_NO_WG512 | IRBPP_TUNE_WIDE_KERNEL | IRBPP_TUNE_NARROW_KERNEL)) &&
                       ((t & IRBPP_TUNE_WG512) || 5 * P.lds_bytes > 160 * 1024);
#if !defined(IRBPP_NO_SPEC) && !defined(IRBPP_ABLATE)
    // (under the 64-VGPR cap four such workgroups share a CU instead of three: level at 2048 bins, +10 % at 8192, profiles/r05/s30)
    if (wg512 && pick_spec(env) == 4)
        return P.N >= 4096 ? EnvKernel{wg512::irbpp_env_kernel_s4_w512c, "irbpp_env_kernel_s4_w512c", 512}
                           : EnvKernel{wg512::irbpp_env_kernel_s4_w512, "irbpp_env_kernel_s4_w512", 512};
#endif
    if (wg512 && (pick_spec(env) == 0 || (t & IRBPP_TUNE_WG512)))
        return {wg512::irbpp_env_kernel_generic_w512, "irbpp_env_kernel_generic_w512", 512};
#if !defined(IRBPP_NO_SPEC) && !defined(IRBPP_ABLATE)
    if ((t & IRBPP_TUNE_WG128) && pick_spec(env) == 1)
        return {wg128::irbpp_env_kernel_s1_w128, "irbpp_env_kernel_s1_w128", 128};        // (A/B: two waves per bin)
    switch (pick_spec(env)) {            // (a key fixes the overlap path: block_b and box are pinned fields)
        case 1: return {irbpp_env_kernel_s1, "irbpp_env_kernel_s1"};
        case 2: return {irbpp_env_kernel_s2, "irbpp_env_kernel_s2"};
        case 3: return {irbpp_env_kernel_s3, "irbpp_env_kernel_s3"};
        case 4: return {irbpp_env_kernel_s4, "irbpp_env_kernel_s4"};
        case 5: return {irbpp_env_kernel_s5, "irbpp_env_kernel_s5"};
        default: break;
    }
#endif
    if (mixed) return (t & IRBPP_TUNE_WIDE_KERNEL) ? EnvKernel{irbpp_env_kernel_wide, "irbpp_env_kernel_wide"} : EnvKernel{irbpp_env_kernel_mixed8, "irbpp_env_kernel_mixed8"};
    if (P.block_b > 0) {
        if ((t & IRBPP_TUNE_WIDE_KERNEL) || 6 * P.lds_bytes > 150 * 1024) return {irbpp_env_kernel_wide, "irbpp_env_kernel_wide"};
        return {irbpp_env_kernel, "irbpp_env_kernel"};
    }
    if (P.box) {
        if (t & IRBPP_TUNE_WIDE_KERNEL) return {irbpp_env_kernel_box, "irbpp_env_kernel_box"};
        if ((t & IRBPP_TUNE_NARROW_KERNEL) || lds_allows_8) return {irbpp_env_kernel_box8, "irbpp_env_kernel_box8"};
        return {irbpp_env_kernel_box, "irbpp_env_kernel_box"};
    }
    // generic path: the build under the 64-VGPR cap where the LDS lets an eighth workgroup onto the CU (general 15.1 vs
    // 14.9 M steps/s, blockout at R = 8 20.4 vs 20.1 M with the blocked and pipelined loop; before it the seven-wave
    // build was ahead, 12.9 vs 12.2 M); a 64 x 64 heightmap (40 KB, four workgroups) gains nothing from the cap
    if (t & IRBPP_TUNE_WIDE_KERNEL) return {irbpp_env_kernel_generic, "irbpp_env_kernel_generic"};
    if ((t & IRBPP_TUNE_NARROW_KERNEL) || lds_allows_8) return {irbpp_env_kernel_generic8, "irbpp_env_kernel_generic8"};
    return {irbpp_env_kernel_generic, "irbpp_env_kernel_generic"};
}

// Border following: candidates per wave and grid of a launch over n bins.  A bin averages a few dozen candidate starts; at
// full width (thousands of bins) 64 per wave fill every SIMD and fewer, shorter-lived waves only add scheduling overhead
// (measured at 4096 bins: 28.3 / 26.8 / 24.3 M steps/s for 64 / 32 / 16); a launch over few bins leaves SIMDs idle, and a
// wave lasts as long as the longest of its borders, so there the candidates are spread over more waves.
static int pick_trace_cpw(const irbpp_env* env, int n) {
    const int t = env->cfg.tuning;
    if (t & IRBPP_TUNE_TRACE_CPW64) return 64;
    if (t & IRBPP_TUNE_TRACE_CPW32) return 32;
    if (t & IRBPP_TUNE_TRACE_CPW16) return 16;
    if (t & IRBPP_TUNE_TRACE_REFILL) return TRACE_REFILL_BATCH;
    // (lane refill -- a wave owns a batch of 128 candidates and hands a lane the next one as borders close, trace_refill_body --
    // is OPT-IN: measured slower at every size, profiles/r06/LOG.md session 2: 8192 BlockOut bins 54.9 -> 51.4 M as one group,
    // 63.8 -> 59.0 M as two, the kernel 36.2 -> 46.0 us.  The chip has as many lane slots as a launch has candidates, so a
    // refilled lane's work is taken from another wave, not from idleness, and one wave then pays every refill's latencies in turn)
    return n <= TRACE_CPW16_BINS ? 16 : (n <= TRACE_CPW32_BINS ? 32 : 64);
}

// One kernel per observation (OPT-IN, IRBPP_TUNE_CHAIN): the bin's own workgroup finishes its observation (CHAIN builds of the
// transition kernel: contour stage and candidate rows in LDS, no trace / polygon / emit launches).  Built for launches of up to
// ~2048 bins, where a step is a chain of launch and drain latencies whatever the number of bins, and measured SLOWER there
// (profiles/r06/LOG.md session 6, placement-steps/s one kernel vs four): a buffered placement at 512 / 1024 / 2048 bins 8.1 vs
// 9.0 / 12.8 vs 16.3 / 16.5 vs 25.9 M, BlockOut online at 1024 / 2048 bins 14.4 vs 18.7 / 18.0 vs 29.7 M, free-form solids at
// 1024 bins 3.2 vs 8.7 M; only the Cube set gains (26.7 vs 24.4 M at 1024 bins).  A bin's observation is ~25 borders to follow
// and approximate: inside its own workgroup that is one serial latency chain per bin on a CU with nobody else to issue,
// while the split kernels spread the borders of ALL bins over every SIMD of the chip; the launch boundaries they pay
// (~3 us each) are the smaller price.
static bool chain_launch(const irbpp_env* env, int n) {
    const int t = env->cfg.tuning;
    (void)n;
    if (!(t & IRBPP_TUNE_CHAIN) || env->P.stability != 0) return false;
    if (t & (IRBPP_TUNE_TRACE_CPW64 | IRBPP_TUNE_TRACE_CPW32 | IRBPP_TUNE_TRACE_CPW16 | IRBPP_TUNE_TRACE_REFILL | IRBPP_TUNE_INLINE_POLYGON |
             IRBPP_TUNE_BLOCK_EMIT | IRBPP_TUNE_WAVE_EMIT | IRBPP_TUNE_SPLIT_APPLY | IRBPP_TUNE_GRAPH | IRBPP_TUNE_WG512 | IRBPP_TUNE_NARROW_KERNEL))
        return false;                                          // (a caller that forces a shape of the split pipeline gets the split pipeline)
    if (env->P.lds_bytes > 32 * 1024) return false;            // (the 64 x 64 heightmap: 512-thread workgroups on a 40 KB tile, not this)
    return true;
}

// One launch group: the launch slots [first, first + n) of a transition -- order (for step / candidates), the
// transition kernel and, in the split pipeline, trace and emit -- on one stream.
static void launch_group(irbpp_env* env, StepIO io, int mode, hipStream_t st, int first, int n) {
    io.block_off = first;
    io.n_slots = n;
    io.auto_action = env->auto_actions;
    io.use_order = 0;
    if (mode == MODE_STEP && env->item_order && first == 0 && n == env->P.N) {
        hipLaunchKernelGGL(irbpp_item_order_kernel, dim3(1), dim3(1024), 0, st, env->T, env->S, n);
        io.use_order = 1;
    }
    io.obs_rows = nullptr;
    const bool observes = mode == MODE_CANDS || ((mode == MODE_RESET || mode == MODE_STEP) && env->P.K == 1);
    for (auto& rb : env->obs_buffers)
        if (rb.first == io.obs) {
            // reset_specific writes a row per LISTED bin and a buffered environment's step / reset write the order
            // observation: through a registered pointer either leaves the per-bin row counts meaningless
            if (observes && !(mode == MODE_RESET && io.bin_list != nullptr)) io.obs_rows = rb.second;
            else hipMemsetAsync(rb.second, 0xFF, (size_t)env->P.N * sizeof(int32_t), st);
        }
    // split pipeline: a location observation is finished by the trace kernel (one wave per 64 candidate starts of
    // the launch's flat list) and the emit kernel (one workgroup per bin), on the same stream
    if (env->P.wide) {                     // irbpp_wide.hip: [the geometry-free apply kernel,] then ONE kernel per observation
        int wmode = mode;
        if (mode == MODE_STEP) {
            if (env->P.K > 1 && n < 2048) hipLaunchKernelGGL(irbpp_apply_wg_kernel, dim3(n), dim3(256), 0, st, env->P, env->T, env->S, io, mode);
            else hipLaunchKernelGGL(irbpp_apply_kernel, dim3((n + 3) / 4), dim3(256), 0, st, env->P, env->T, env->S, io, mode);
            if (env->P.K > 1) return;
            wmode = MODE_OBSERVE;
        }
        hipLaunchKernelGGL(irbpp_wide_kernel, dim3(n), dim3(256), wide_layout(env->P).bytes, st, env->P, env->T, env->S, io, wmode);
        return;
    }
    const bool chain = observes && mode != MODE_POSSIBLE && chain_launch(env, n);
    const bool split = env->P.split && observes && !chain;
    // expensive bins first in the emit kernel: free-form level images only (lattice and box data never get there), not for a
    // listed reset (its observation rows go by list position)
    const bool heavy_first = split && env->P.heavy_cap > 0 && !lattice_images(env->P) && io.bin_list == nullptr &&
                             !(env->cfg.tuning & IRBPP_TUNE_NO_HEAVY_FIRST);
    io.heavy_turn = heavy_first ? env->heavy_turn : -1;
    if (heavy_first) env->heavy_turn ^= 1;
    int env_mode = mode;
    if (chain && !(mode == MODE_STEP && env->P.K > 1)) {
        // (a buffered step is the apply kernel below: it observes nothing)
        const bool s1 = pick_spec(env) == 1;
        hipLaunchKernelGGL(s1 ? irbpp_env_kernel_chain_s1 : irbpp_env_kernel_chain, dim3(n), dim3(256), env->P.lds_bytes + chain_extra_lds(env->P), st,
                           env->P, env->T, env->S, io, mode);
        return;
    }
    if (mode == MODE_STEP && split_apply(env, n)) {
        // a buffered step: a workgroup per bin at launches of fewer than 2048 bins (a wave per bin leaves most of the chip
        // to one dependent chain per CU there: 15.7 vs 15.3 M at 1024 bins), a wave per bin from there on (every bin resident
        // at once: 8192 bins as two groups 50.6 -> 55.2 M, 4096 bins 40.0 -> 41.6 M; profiles/r05/s27)
        if (env->P.K > 1 && n < 2048) hipLaunchKernelGGL(irbpp_apply_wg_kernel, dim3(n), dim3(256), 0, st, env->P, env->T, env->S, io, mode);
        else hipLaunchKernelGGL(irbpp_apply_kernel, dim3((n + 3) / 4), dim3(256), 0, st, env->P, env->T, env->S, io, mode);
        env_mode = MODE_OBSERVE;         // (a buffered step ends with the apply kernel: it wrote the order observation)
    }
    if (!(env_mode == MODE_OBSERVE && env->P.K > 1)) {
        const EnvKernel ek = pick_env_kernel(env);
        hipLaunchKernelGGL(ek.fn, dim3(n), dim3(ek.threads), env->P.lds_bytes, st, env->P, env->T, env->S, io, env_mode);
    }
    if (split) {
        // the grid covers an average of up to 64 candidates per bin and strides over the chunks beyond that
        // one trace wave per 64 candidates a bin may average, two polygon waves per bin; the kernels stride over anything
        // beyond (half / a third of either grid with striding measured -4 ... -9 %)
        const int cpw = pick_trace_cpw(env, n), pgrid = 2 * n;
        // (IRBPP_TUNE_INLINE_POLYGON: every trace wave runs approxPolyDP on the borders it followed itself -- the path a full
        // record list takes -- and no polygon kernel is launched.  Measured at 1024 / 2048 / 4096 bins: 11.8 / 20.2 / 28.8 M
        // steps/s against 13.9 / - / 31.1 M: the approximation stretches the slowest trace waves.  For the parity tests.)
        const int tune = env->cfg.tuning;
        const bool inline_polygon = (tune & IRBPP_TUNE_INLINE_POLYGON) != 0;
        Params Pt = env->P;
        if (inline_polygon) Pt.round_cap = 0;
        int tgrid = cpw > 64 ? (n * 64 + cpw - 1) / cpw : n * (64 / cpw);
        if (tgrid > trace_grid_cap(env->P.N)) tgrid = trace_grid_cap(env->P.N);      // (w_big holds one scratch per wave of the grid)
        auto trace_fn = cpw > 64 ? irbpp_trace_kernel_refill : cpw == 64 ? irbpp_trace_kernel : (cpw == 32 ? irbpp_trace_kernel_c32 : irbpp_trace_kernel_c16);
        hipLaunchKernelGGL(trace_fn, dim3(tgrid), dim3(64), 0, st, Pt, env->S, env->phase_cycles);
#ifdef IRBPP_AB_POLY_ACCOUNT
        if (!inline_polygon) hipLaunchKernelGGL(irbpp_polygon_kernel, dim3(pgrid), dim3(64), 0, st, env->P, env->S, env->phase_cycles);
#else
        if (!inline_polygon) hipLaunchKernelGGL(irbpp_polygon_kernel, dim3(pgrid), dim3(64), 0, st, env->P, env->S);
#endif
        // lattice and box data (practically never more than S candidates per bin): a wave per bin, four bins per workgroup
        // -- from 2048 bins on: a launch over 1024 bins is one such workgroup per CU, 2.6 % slower than a workgroup per bin (profiles/r05/s7)
        const bool wave_emit = lattice_images(env->P) && !heavy_first && !(env->cfg.tuning & IRBPP_TUNE_BLOCK_EMIT) &&
                               (n >= 2048 || (env->cfg.tuning & IRBPP_TUNE_WAVE_EMIT));
        env_kernel_fn emit_fn = wave_emit ? irbpp_emit_wave_kernel : irbpp_emit_kernel;
#if !defined(IRBPP_NO_SPEC) && !defined(IRBPP_ABLATE)
        switch (pick_spec(env)) {
            case 1: emit_fn = wave_emit ? irbpp_emit_wave_kernel_s1 : irbpp_emit_kernel_s1; break;
            case 2: emit_fn = wave_emit ? irbpp_emit_wave_kernel_s2 : irbpp_emit_kernel_s2; break;
            case 3: emit_fn = irbpp_emit_kernel_s3; break;
            case 4: emit_fn = irbpp_emit_kernel_s4; break;
            case 5: emit_fn = wave_emit ? irbpp_emit_wave_kernel_s5 : irbpp_emit_kernel_s5; break;
            default: break;
        }
#endif
        const int egrid = wave_emit ? (n + 3) / 4 : n + (heavy_first ? env->P.heavy_cap : 0);
        hipLaunchKernelGGL(emit_fn, dim3(egrid), dim3(256), env->P.emit_lds_bytes, st, env->P, env->T, env->S, io, mode);
    }
}

// A transition of `grid` launch slots (all bins, or the listed ones) on the caller's stream.  (Cutting it into
// groups on library-owned streams, forked from and joined to the caller's stream with events, was measured and
// dropped: the eight cross-queue dependencies per step cost more than the overlapped launch tails give back,
// 18.5 -> 11.8 M steps/s.  Overlap across sub-batches is offered one level up instead, where no join is needed:
// vec_env.GroupedPackingEnv steps independent groups of bins on their own streams.)
// Transitions as HIP graphs -- OPT-IN (IRBPP_TUNE_GRAPH), because on this runtime (ROCm 7.2) it LOSES at every size measured:
// a buffered placement at 1024 bins 15.3 -> 13.1 M steps/s, BlockOut online 1024 / 2048 bins 17.8 -> 16.4 / 28.4 -> 26.2 M,
// the 64 x 64 heightmap at 2048 bins 6.34 -> 6.22 M, 8192 BlockOut bins as two groups 60.8 -> 58.8 M (profiles/r05/s14): what
// hipGraphLaunch puts between the kernel nodes of a linear chain costs more than the host's launch calls and the dispatch
// gaps it removes (~7 us of a 66 us placement).  The mechanism stays for runtimes where that changes: the chain of a
// transition is captured ONCE per distinct argument set (mode, every pointer and flag of StepIO, the host-side state the
// launch code branches on) on a stream of the library's own and replayed on the caller's stream with one hipGraphLaunch.
// An argument set is captured when it is seen the SECOND time: a caller that hands over a fresh observation tensor every
// step never repeats one and is launched directly.  Everything issued while the caller's stream is itself being captured
// is launched directly too (into the caller's capture).
constexpr int GRAPH_CACHE = 24;
// A captured graph holds Params / Tables / State BY VALUE in its kernel nodes: every setter that changes one of them after
// graphs may exist (the placement log's pointers, a tooling switch) drops the cache; the next launches capture again.
static void drop_graphs(irbpp_env* env) {
    for (auto& g : env->graphs) { if (g.exec) hipGraphExecDestroy(g.exec); if (g.graph) hipGraphDestroy(g.graph); }
    env->graphs.clear();
}
static bool graph_wanted(const irbpp_env* env, int) { return (env->cfg.tuning & IRBPP_TUNE_GRAPH) != 0; }

static int launch_env(irbpp_env* env, StepIO io, int mode, void* stream, int grid = 0) {
    if (grid <= 0) grid = env->P.N;
    io.phase_cycles = env->phase_cycles;
    hipStream_t st = (hipStream_t)stream;
    size_t pairs = env->timing.size() / 2;
    const size_t slot = env->timing_next;
    if (pairs && (env->timing_phase++ % env->timing_every) != 0) pairs = 0;       // not a sampled launch
    if (pairs) hipEventRecord(env->timing[2 * slot], st);
    bool launched = false;
    if (graph_wanted(env, grid) && (mode == MODE_STEP || mode == MODE_CANDS)) {
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(st, &cs) != hipSuccess) cs = hipStreamCaptureStatusActive;
        if (cs == hipStreamCaptureStatusNone) {
            // everything launch_group reads: the arguments and the host-side state it branches on
            struct Key { StepIO io; int mode, grid, heavy_turn, obs_epoch; int32_t* auto_actions; } k;
            memset(&k, 0, sizeof k);
            k.io = io; k.mode = mode; k.grid = grid; k.heavy_turn = env->heavy_turn; k.obs_epoch = env->obs_epoch;
            k.auto_actions = env->auto_actions;
            const uint8_t* kb = (const uint8_t*)&k;
            irbpp_env::GraphEntry* hit = nullptr;
            for (auto& g : env->graphs)
                if (g.key.size() == sizeof k && memcmp(g.key.data(), kb, sizeof k) == 0) { hit = &g; break; }
            if (hit == nullptr) {                        // first sight: remember it, launch directly
                if (env->graphs.size() >= (size_t)GRAPH_CACHE) {       // (the least recently used entry makes room)
                    size_t lru = 0;
                    for (size_t i = 1; i < env->graphs.size(); ++i) if (env->graphs[i].used < env->graphs[lru].used) lru = i;
                    if (env->graphs[lru].exec) hipGraphExecDestroy(env->graphs[lru].exec);
                    if (env->graphs[lru].graph) hipGraphDestroy(env->graphs[lru].graph);
                    env->graphs.erase(env->graphs.begin() + lru);
                }
                env->graphs.push_back({std::vector<uint8_t>(kb, kb + sizeof k), nullptr, nullptr, ++env->graph_clock});
            } else {
                hit->used = ++env->graph_clock;
                const int turn_before = env->heavy_turn;
                if (hit->exec == nullptr) {              // second sight: capture the chain on the library's own stream
                    if (env->cap_stream == nullptr && hipStreamCreateWithFlags(&env->cap_stream, hipStreamNonBlocking) != hipSuccess)
                        env->cap_stream = nullptr;
                    if (env->cap_stream != nullptr && hipStreamBeginCapture(env->cap_stream, hipStreamCaptureModeRelaxed) == hipSuccess) {
                        launch_group(env, io, mode, env->cap_stream, 0, grid);
                        hipGraph_t graph = nullptr;
                        if (hipStreamEndCapture(env->cap_stream, &graph) == hipSuccess && graph != nullptr &&
                            hipGraphInstantiate(&hit->exec, graph, nullptr, nullptr, 0) == hipSuccess) {
                            hit->graph = graph;
                        } else {
                            if (graph) hipGraphDestroy(graph);
                            hit->exec = nullptr;
                        }
                        env->heavy_turn = turn_before;   // (the capture ran the host code of a launch; the launch itself follows)
                        (void)hipGetLastError();
                    }
                }
                if (hit->exec != nullptr && hipGraphLaunch(hit->exec, st) == hipSuccess) {
                    launched = true;
                    ++env->graph_replays;
                    // the host-side state a direct launch would have advanced
                    const bool observes = mode == MODE_CANDS || (mode == MODE_STEP && env->P.K == 1);
                    const bool heavy_first = env->P.split && observes && env->P.heavy_cap > 0 && !lattice_images(env->P) &&
                                             io.bin_list == nullptr && !(env->cfg.tuning & IRBPP_TUNE_NO_HEAVY_FIRST);
                    if (heavy_first) env->heavy_turn ^= 1;
                }
            }
        }
    }
    if (!launched) launch_group(env, io, mode, st, 0, grid);
    if (pairs) {
        hipEventRecord(env->timing[2 * slot + 1], st);
        env->timing_next = (slot + 1) % pairs;
        if (env->timing_used < pairs) env->timing_used++;
    }
    return hipGetLastError() == hipSuccess ? IRBPP_OK : IRBPP_ERR_HIP;
}

int irbpp_reset(irbpp_env* env, float* obs_dev, void* stream) {
    if (!env || !obs_dev) return IRBPP_ERR_ARG;
    if (!env->shapes_loaded || !env->seq_loaded) return IRBPP_ERR_STATE;
    StepIO io;
    memset(&io, 0, sizeof(io));
    io.obs = obs_dev;
    io.obs_stride = env->P.obs_len0;
    io.reset_next = env->was_reset ? 1 : 0;       // a later reset() moves every bin on to its next trajectory (IRcreator.py:86-92)
    env->err_mirror = nullptr;                    // bits a reset raises reach S.err only: the next step seeds its error word again
    const int rc = launch_env(env, io, MODE_RESET, stream);
    if (rc == IRBPP_OK) env->was_reset = true;
    return rc;
}

int irbpp_reset_bins(irbpp_env* env, const int32_t* bins_dev, int32_t count, float* obs_dev, void* stream) {
    if (!env || count < 0 || count > env->P.N || (count > 0 && (!bins_dev || !obs_dev))) return IRBPP_ERR_ARG;
    if (!env->was_reset) return IRBPP_ERR_STATE;
    if (count == 0) return IRBPP_OK;
    StepIO io;
    memset(&io, 0, sizeof(io));
    io.obs = obs_dev;
    io.obs_stride = env->P.obs_len0;
    io.bin_list = bins_dev;
    env->err_mirror = nullptr;                    // (as irbpp_reset)
    return launch_env(env, io, MODE_RESET, stream, count);
}

int irbpp_step(irbpp_env* env, const int32_t* actions_dev, float* obs_dev, const irbpp_step_out* out, void* stream) {
    if (!env || !actions_dev || !obs_dev) return IRBPP_ERR_ARG;
    if (!env->was_reset) return IRBPP_ERR_STATE;
    StepIO io;
    memset(&io, 0, sizeof(io));
    io.actions = actions_dev;
    io.obs = obs_dev;
    io.obs_stride = env->P.obs_len0;
    if (out) {
        io.reward = out->reward_dev;
        io.done = out->done_dev;
        io.counter = out->counter_dev;
        io.ratio = out->ratio_dev;
        io.ep_reward = out->ep_reward_dev;
        io.ep_len = out->ep_len_dev;
        io.stable = out->stable_dev;
        // The error word of the outputs.  Online (K == 1): stored by the emit kernel, the step's last one.  Buffered (K > 1):
        // the transition kernel is the last one and may raise bits itself, so every kernel ORs a bit into S.err AND into
        // this word as it raises it (raise_error), and the emit kernel of get_action_candidates stores S.err into it; the
        // word only needs seeding when the caller hands over a new one.  (A 4-byte device-to-device copy per step -- a
        // 5 us kernel to move one int -- did this before: 3.3 % of the GPU time of a buffered step at 1024 bins.)
        io.err_out = out->err_dev;
        if (out->err_dev != env->err_mirror) {
            if (out->err_dev != nullptr)
                HIP_TRY(hipMemcpyAsync(out->err_dev, env->S.err, sizeof(int32_t), hipMemcpyDeviceToDevice, (hipStream_t)stream));
            env->err_mirror = out->err_dev;
        }
    } else {
        env->err_mirror = nullptr;
    }
    return launch_env(env, io, MODE_STEP, stream);
}

int irbpp_get_action_candidates(irbpp_env* env, const int32_t* order_actions_dev, float* loc_obs_dev, void* stream) {
    if (!env || !order_actions_dev || !loc_obs_dev) return IRBPP_ERR_ARG;
    if (!env->was_reset) return IRBPP_ERR_STATE;
    if (env->P.K < 2) return IRBPP_ERR_ARG;
    StepIO io;
    memset(&io, 0, sizeof(io));
    io.actions = order_actions_dev;
    io.obs = loc_obs_dev;
    io.obs_stride = env->P.obs_len1;
    io.err_out = env->err_mirror;          // the error word of the step outputs follows S.err through this call too
    return launch_env(env, io, MODE_CANDS, stream);
}

int irbpp_get_all_possible_observation(irbpp_env* env, float* loc_obs_dev, void* stream) {
    if (!env || !loc_obs_dev) return IRBPP_ERR_ARG;
    if (!env->was_reset) return IRBPP_ERR_STATE;
    if (env->P.K < 2) return IRBPP_ERR_ARG;
    // one full-width transition per buffer slot, all on the caller's stream: slot j of every bin is observed into row
    // [b][j] of the [N][k][obs_len(1)] block (row stride k * obs_len(1)); like the reference's loop the last slot's
    // candidates are the ones a following step would index, and the chosen slot (orderAction) is not touched
    for (int j = 0; j < env->P.K; ++j) {
        StepIO io;
        memset(&io, 0, sizeof(io));
        io.fixed_slot = j + 1;
        io.obs = loc_obs_dev + (size_t)j * env->P.obs_len1;
        io.obs_stride = env->P.K * env->P.obs_len1;
        io.err_out = env->err_mirror;
        const int rc = launch_env(env, io, MODE_CANDS, stream);
        if (rc != IRBPP_OK) return rc;
    }
    return IRBPP_OK;
}

int irbpp_policy_minz(irbpp_env* env, const float* loc_obs_dev, int32_t obs_stride, int32_t* actions_dev, void* stream) {
    if (!env || !loc_obs_dev || !actions_dev || obs_stride < 5 * env->P.S) return IRBPP_ERR_ARG;
    const int waves_per_block = 4;
    const int grid = (env->P.N + waves_per_block - 1) / waves_per_block;
    hipLaunchKernelGGL(irbpp_policy_minz_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, loc_obs_dev,
                       obs_stride, env->P.S, env->P.N, actions_dev);
    return hipGetLastError() == hipSuccess ? IRBPP_OK : IRBPP_ERR_HIP;
}

int irbpp_set_auto_policy(irbpp_env* env, int32_t* actions_dev) {
    if (!env) return IRBPP_ERR_ARG;
    env->auto_actions = actions_dev;
    return IRBPP_OK;
}

int irbpp_register_obs_buffer(irbpp_env* env, float* obs_dev) {
    if (!env || !obs_dev) return IRBPP_ERR_ARG;
    HIP_TRY(hipSetDevice(env->cfg.device));
    env->obs_epoch++;
    for (auto& rb : env->obs_buffers)
        if (rb.first == obs_dev) {           // registered again (e.g. a new allocation at an old address): contents unknown
            HIP_TRY(hipMemset(rb.second, 0xFF, (size_t)env->P.N * sizeof(int32_t)));
            return IRBPP_OK;
        }
    int32_t* rows = nullptr;
    for (auto& rb : env->obs_buffers)        // a slot freed by irbpp_unregister_obs_buffer keeps its row counts' memory
        if (rb.first == nullptr && rows == nullptr) { rows = rb.second; rb.first = obs_dev; }
    if (rows == nullptr) {
        if (env->obs_buffers.size() >= 8) return IRBPP_ERR_ARG;
        int rc = dev_alloc(env, &rows, (size_t)env->P.N);
        if (rc != IRBPP_OK) return rc;
        env->obs_buffers.emplace_back(obs_dev, rows);
    }
    HIP_TRY(hipMemset(rows, 0xFF, (size_t)env->P.N * sizeof(int32_t)));      // -1: contents unknown, write everything once
    return IRBPP_OK;
}

int irbpp_unregister_obs_buffer(irbpp_env* env, float* obs_dev) {
    if (!env || !obs_dev) return IRBPP_ERR_ARG;
    env->obs_epoch++;
    for (auto& rb : env->obs_buffers)
        if (rb.first == obs_dev) { rb.first = nullptr; return IRBPP_OK; }
    return IRBPP_ERR_ARG;
}

int irbpp_invalidate_obs_buffer(irbpp_env* env, float* obs_dev, void* stream) {
    if (!env) return IRBPP_ERR_ARG;
    bool found = obs_dev == nullptr;
    for (auto& rb : env->obs_buffers)
        if (rb.first != nullptr && (obs_dev == nullptr || rb.first == obs_dev)) {
            HIP_TRY(hipMemsetAsync(rb.second, 0xFF, (size_t)env->P.N * sizeof(int32_t), (hipStream_t)stream));
            found = true;
        }
    return found ? IRBPP_OK : IRBPP_ERR_ARG;
}

int irbpp_possible_position(irbpp_env* env, const int32_t* item_ids_dev, double* posz_dev, uint8_t* mask_dev, void* stream) {
    if (!env || !item_ids_dev || !posz_dev || !mask_dev || env->P.wide) return IRBPP_ERR_ARG;
    if (!env->shapes_loaded) return IRBPP_ERR_STATE;
    StepIO io;
    memset(&io, 0, sizeof(io));
    io.actions = item_ids_dev;
    io.posz_out = posz_dev;
    io.mask_out = mask_dev;
    return launch_env(env, io, MODE_POSSIBLE, stream);
}

int irbpp_heuristic_action(irbpp_env* env, int32_t method, int32_t dir_idx, int32_t* out_dev, void* stream) {
    if (!env || !out_dev || method < 1 || method > 4 || dir_idx < 0 || dir_idx > 3 || env->P.wide) return IRBPP_ERR_ARG;
    if (!env->was_reset) return IRBPP_ERR_STATE;
    StepIO io;
    memset(&io, 0, sizeof(io));
    io.heur_out = out_dev;
    io.heur_method = method;
    io.heur_dir = dir_idx;
    hipLaunchKernelGGL(irbpp_heuristic_kernel, dim3(env->P.N), dim3(256), env->P.lds_bytes_full, (hipStream_t)stream, env->P,
                       env->T, env->S, io);
    return hipGetLastError() == hipSuccess ? IRBPP_OK : IRBPP_ERR_HIP;
}

int irbpp_shot_item(const double* verts_dev, const int32_t* faces_dev, int32_t n_faces, int32_t fx, int32_t fy,
                    double resolution_h, double shift, double extent_z, double* top_dev, double* bottom_dev,
                    double* mask_top_dev, double* mask_bottom_dev, int32_t* scratch_dev, void* stream) {
    if (!verts_dev || !faces_dev || n_faces < 1 || fx < 1 || fy < 1 || !top_dev || !bottom_dev || !mask_top_dev ||
        !mask_bottom_dev || !scratch_dev)
        return IRBPP_ERR_ARG;
    HIP_TRY(hipMemsetAsync(scratch_dev, 0, sizeof(int32_t), (hipStream_t)stream));
    const int n = fx * fy, grid = (n + 255) / 256;
    hipLaunchKernelGGL(irbpp_shot_item_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, verts_dev, faces_dev,
                       n_faces, fx, fy, resolution_h, shift, top_dev, bottom_dev, mask_top_dev, mask_bottom_dev,
                       scratch_dev);
    hipLaunchKernelGGL(irbpp_shot_item_fallback_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, n, extent_z,
                       top_dev, bottom_dev, mask_top_dev, mask_bottom_dev, scratch_dev);
    return hipGetLastError() == hipSuccess ? IRBPP_OK : IRBPP_ERR_HIP;
}

int irbpp_convex_hull_actions(irbpp_env* env, int32_t n_grids, const double* posz_valid_dev, const uint8_t* mask_dev,
                              uint32_t* vertex_rows_dev, void* stream) {
    if (!env || n_grids < 1 || !posz_valid_dev || !mask_dev || !vertex_rows_dev || env->P.wide) return IRBPP_ERR_ARG;
    hipLaunchKernelGGL(irbpp_hull_kernel, dim3(n_grids), dim3(256), env->P.lds_bytes, (hipStream_t)stream, env->P,
                       env->S, posz_valid_dev, mask_dev, vertex_rows_dev);
    return hipGetLastError() == hipSuccess ? IRBPP_OK : IRBPP_ERR_HIP;
}

int irbpp_get_heightmaps(irbpp_env* env, double* hm_dev, void* stream) {
    if (!env || !hm_dev) return IRBPP_ERR_ARG;
    HIP_TRY(hipMemcpyAsync(hm_dev, env->S.hm, (size_t)env->P.N * env->P.Hc * sizeof(double), hipMemcpyDeviceToDevice,
                           (hipStream_t)stream));
    return IRBPP_OK;
}

int irbpp_set_heightmaps(irbpp_env* env, const double* hm_dev, void* stream) {
    if (!env || !hm_dev) return IRBPP_ERR_ARG;
    HIP_TRY(hipMemcpyAsync(env->S.hm, hm_dev, (size_t)env->P.N * env->P.Hc * sizeof(double), hipMemcpyDeviceToDevice,
                           (hipStream_t)stream));
    // the drop heights of the last observation (w_posz, marked by w_valid) belong to the OLD maps: a step that follows
    // without a new observation recomputes its drop height from the footprint's bottom cells instead
    HIP_TRY(hipMemsetAsync(env->S.w_valid, 0, (size_t)env->P.N * env->P.R * env->P.vrow * sizeof(uint32_t), (hipStream_t)stream));
    return IRBPP_OK;
}

int irbpp_episode_totals(irbpp_env* env, double* out_dev, void* stream) {
    if (!env || !out_dev) return IRBPP_ERR_ARG;
    hipLaunchKernelGGL(irbpp_totals_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, env->S.totals, env->P.N, out_dev);
    return hipGetLastError() == hipSuccess ? IRBPP_OK : IRBPP_ERR_HIP;
}

int irbpp_set_placement_log(irbpp_env* env, uint32_t* meta_dev, double* z_dev, int32_t capacity) {
    if (!env || capacity < 0 || ((meta_dev == nullptr) != (z_dev == nullptr))) return IRBPP_ERR_ARG;
    drop_graphs(env);                      // (captured kernel nodes carry State by value)
    env->S.log_meta = meta_dev;
    env->S.log_z = z_dev;
    env->S.log_cap = meta_dev ? capacity : 0;
    return IRBPP_OK;
}

int irbpp_sumtree_find(const float* tree_dev, int32_t n_env, int32_t capacity, const float* values_dev, int32_t draws,
                       float* prob_dev, int64_t* data_idx_dev, int64_t* tree_idx_dev, void* stream) {
    if (!tree_dev || !values_dev || !prob_dev || !data_idx_dev || !tree_idx_dev || n_env < 1 || capacity < 1 || draws < 1)
        return IRBPP_ERR_ARG;
    const int n = n_env * draws;
    hipLaunchKernelGGL(irbpp_sumtree_find_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, tree_dev, n_env,
                       capacity, values_dev, draws, prob_dev, data_idx_dev, tree_idx_dev);
    return hipGetLastError() == hipSuccess ? IRBPP_OK : IRBPP_ERR_HIP;
}

int irbpp_sumtree_sample(const float* tree_dev, const int64_t* index_dev, int32_t n_env, int32_t capacity, int32_t draws,
                         int32_t n_step, uint64_t seed, int32_t max_tries, float* prob_dev, int64_t* data_idx_dev,
                         int64_t* tree_idx_dev, int32_t* failed_dev, void* stream) {
    if (!tree_dev || !index_dev || !prob_dev || !data_idx_dev || !tree_idx_dev || !failed_dev || n_env < 1 || capacity < 1 ||
        draws < 1 || n_step < 0 || max_tries < 1)
        return IRBPP_ERR_ARG;
    const int n = n_env * draws;
    hipLaunchKernelGGL(irbpp_sumtree_sample_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, tree_dev, index_dev,
                       n_env, capacity, draws, n_step, seed, max_tries, prob_dev, data_idx_dev, tree_idx_dev, failed_dev);
    return hipGetLastError() == hipSuccess ? IRBPP_OK : IRBPP_ERR_HIP;
}

int irbpp_sumtree_update(float* tree_dev, float* max_dev, int32_t n_env, int32_t capacity, const int64_t* tree_idx_dev,
                         const float* priority_dev, int32_t leaves, const uint8_t* env_mask_dev, void* stream) {
    if (!tree_dev || !max_dev || !tree_idx_dev || !priority_dev || n_env < 1 || capacity < 1 || leaves < 1) return IRBPP_ERR_ARG;
    if (2 * capacity - 1 > SUMTREE_LDS) return IRBPP_ERR_ARG;        // caller keeps its host-side path for longer rows
    hipLaunchKernelGGL(irbpp_sumtree_update_kernel, dim3(n_env), dim3(64), (size_t)(2 * capacity - 1) * sizeof(float), (hipStream_t)stream, tree_dev, max_dev, capacity,
                       tree_idx_dev, priority_dev, leaves, env_mask_dev);
    return hipGetLastError() == hipSuccess ? IRBPP_OK : IRBPP_ERR_HIP;
}

int irbpp_masked_argmax(const float* q_dev, int32_t q_stride, const float* obs_dev, int32_t obs_stride, int32_t selected,
                        int32_t n_env, int64_t* action_dev, void* stream) {
    if (!q_dev || !obs_dev || !action_dev || n_env < 1 || selected < 1 || q_stride < selected || obs_stride < 5 * selected)
        return IRBPP_ERR_ARG;
    hipLaunchKernelGGL(irbpp_masked_argmax_kernel, dim3((n_env + 3) / 4), dim3(256), 0, (hipStream_t)stream, q_dev, q_stride,
                       obs_dev, obs_stride, selected, n_env, action_dev);
    return hipGetLastError() == hipSuccess ? IRBPP_OK : IRBPP_ERR_HIP;
}

int irbpp_replay_gather(const irbpp_replay_view* v, int32_t draws, float beta, const int64_t* data_idx_dev, const float* prob_dev,
                        float* state_dev, int64_t* action_dev, float* return_dev, float* next_state_dev, float* nonterminal_dev,
                        float* weight_dev, void* stream) {
    if (!v || !v->states_dev || !v->actions_dev || !v->rewards_dev || !v->nonterminals_dev || !v->tree_dev || !v->index_dev ||
        !v->full_dev || !v->scaling_dev || v->n_env < 1 || v->capacity < 1 || v->obs_len < 1 || v->n_step < 1 || draws < 1 ||
        draws > 256 || !data_idx_dev || !prob_dev || !state_dev || !action_dev || !return_dev || !next_state_dev ||
        !nonterminal_dev || !weight_dev)
        return IRBPP_ERR_ARG;
    hipLaunchKernelGGL(irbpp_replay_gather_kernel, dim3(v->n_env), dim3(256), 0, (hipStream_t)stream, v->states_dev, v->actions_dev,
                       v->rewards_dev, v->nonterminals_dev, v->tree_dev, v->index_dev, v->full_dev, v->scaling_dev, v->capacity,
                       v->obs_len, v->n_step, draws, beta, data_idx_dev, prob_dev, state_dev, action_dev, return_dev,
                       next_state_dev, nonterminal_dev, weight_dev);
    return hipGetLastError() == hipSuccess ? IRBPP_OK : IRBPP_ERR_HIP;
}

int irbpp_replay_append(const irbpp_replay_store* m, const float* state_dev, int64_t state_stride, const void* action_dev,
                        int32_t action_bytes, const void* reward_dev, int32_t reward_bytes, const uint8_t* terminal_dev,
                        const uint8_t* valid_dev, void* stream) {
    if (!m || !m->states_dev || !m->actions_dev || !m->rewards_dev || !m->nonterminals_dev || !m->timesteps_dev || !m->tree_dev ||
        !m->max_dev || !m->index_dev || !m->full_dev || !m->t_dev || m->n_env < 1 || m->capacity < 1 || m->obs_len < 1 ||
        !state_dev || state_stride < m->obs_len || !action_dev || (action_bytes != 4 && action_bytes != 8) || !reward_dev ||
        (reward_bytes != 4 && reward_bytes != 8) || !terminal_dev)
        return IRBPP_ERR_ARG;
    hipLaunchKernelGGL(irbpp_replay_append_kernel, dim3(m->n_env), dim3(256), 0, (hipStream_t)stream, m->states_dev, m->actions_dev,
                       m->rewards_dev, m->nonterminals_dev, m->timesteps_dev, m->tree_dev, m->max_dev, m->index_dev, m->full_dev,
                       m->t_dev, m->capacity, m->obs_len, state_dev, (long long)state_stride, action_dev, action_bytes, reward_dev,
                       reward_bytes, terminal_dev, valid_dev);
    return hipGetLastError() == hipSuccess ? IRBPP_OK : IRBPP_ERR_HIP;
}

int irbpp_debug_phase_cycles(irbpp_env* env, int64_t* cycles_dev) {
    if (!env) return IRBPP_ERR_ARG;
    drop_graphs(env);
    env->phase_cycles = (long long*)cycles_dev;
    return IRBPP_OK;
}

int irbpp_debug_kernel_info(const irbpp_env* env, int32_t* lds_bytes, const char** kernel_name) {
    if (!env || !lds_bytes || !kernel_name) return IRBPP_ERR_ARG;
    *lds_bytes = env->P.lds_bytes;
    // the kernels of a step over all bins of this environment, in launch order behind the transition kernel's build
    const int n = env->P.N, spec = pick_spec(env);
    const bool lattice = lattice_images(env->P);
    const bool wave_emit = lattice && !(env->cfg.tuning & IRBPP_TUNE_BLOCK_EMIT) && (n >= 2048 || (env->cfg.tuning & IRBPP_TUNE_WAVE_EMIT));
    const int cpw = pick_trace_cpw(env, n);
    char emit[48];
    snprintf(emit, sizeof emit, "%s%s", wave_emit ? "irbpp_emit_wave_kernel" : "irbpp_emit_kernel",
             spec == 1 ? "_s1" : spec == 2 ? "_s2" : (spec == 3 && !wave_emit) ? "_s3" : (spec == 4 && !wave_emit) ? "_s4" : spec == 5 ? "_s5" : "");
    snprintf(const_cast<irbpp_env*>(env)->kernel_names, sizeof env->kernel_names, "%s + irbpp_trace_kernel%s + irbpp_polygon_kernel + %s%s",
             pick_env_kernel(env).name, cpw > 64 ? "_refill" : cpw == 64 ? "" : (cpw == 32 ? "_c32" : "_c16"), emit,
             !split_apply(env, n) ? "" : (env->P.K > 1 ? (n < 2048 ? " (step: irbpp_apply_wg_kernel alone)" : " (step: irbpp_apply_kernel alone)")
                                                       : " (step: irbpp_apply_kernel in front, transition kernel in MODE_OBSERVE)"));
    if (env->P.wide)
        snprintf(const_cast<irbpp_env*>(env)->kernel_names, sizeof env->kernel_names, "irbpp_wide_kernel alone (action grid of %d x %d cells)%s", env->P.Ax, env->P.Ay,
                 env->P.K > 1 ? " (step: the apply kernel alone)" : " (step: irbpp_apply_kernel in front)");
    else if (chain_launch(env, n)) {
        const bool s1 = spec == 1;
        snprintf(const_cast<irbpp_env*>(env)->kernel_names, sizeof env->kernel_names, "%s alone (observation finished in the bin's workgroup)%s",
                 s1 ? "irbpp_env_kernel_chain_s1" : "irbpp_env_kernel_chain", env->P.K > 1 ? " (step: irbpp_apply_wg_kernel alone)" : "");
    }
    *kernel_name = env->kernel_names;
    return IRBPP_OK;
}

int irbpp_overlap_path(const irbpp_env* env) {
    if (!env) return IRBPP_ERR_ARG;
    if (!env->shapes_loaded) return IRBPP_ERR_STATE;
    return env->P.block_b > 0 ? (all_block(env->P) ? 1 : 4) : (env->P.box ? 2 : 3);
}

int irbpp_debug_kernel_timing_every(irbpp_env* env, int32_t every) {
    if (!env || every < 1) return IRBPP_ERR_ARG;
    env->timing_every = every;
    env->timing_phase = 0;
    return IRBPP_OK;
}

int irbpp_debug_kernel_timing(irbpp_env* env, int32_t capacity) {
    if (!env || capacity < 0) return IRBPP_ERR_ARG;
    HIP_TRY(hipSetDevice(env->cfg.device));
    for (hipEvent_t e : env->timing) hipEventDestroy(e);
    env->timing.clear();
    env->timing_next = env->timing_used = 0;
    for (int i = 0; i < 2 * capacity; ++i) {
        hipEvent_t e;
        HIP_TRY(hipEventCreate(&e));
        env->timing.push_back(e);
    }
    return IRBPP_OK;
}

int irbpp_debug_kernel_times(irbpp_env* env, float* ms_host, int32_t max_count, int32_t* count) {
    if (!env || !ms_host || !count || max_count < 0) return IRBPP_ERR_ARG;
    const size_t pairs = env->timing.size() / 2;
    size_t n = env->timing_used < (size_t)max_count ? env->timing_used : (size_t)max_count;
    for (size_t i = 0; i < n; ++i) {                   // the n latest launches, oldest first
        const size_t slot = (env->timing_next + pairs - n + i) % pairs;
        HIP_TRY(hipEventSynchronize(env->timing[2 * slot + 1]));
        HIP_TRY(hipEventElapsedTime(&ms_host[i], env->timing[2 * slot], env->timing[2 * slot + 1]));
    }
    *count = (int32_t)n;
    env->timing_next = env->timing_used = 0;
    return IRBPP_OK;
}

int irbpp_device_error(irbpp_env* env, void* stream, int32_t* flags_out) {
    if (!env || !flags_out) return IRBPP_ERR_ARG;
    HIP_TRY(hipMemcpyAsync(flags_out, env->S.err, sizeof(int32_t), hipMemcpyDeviceToHost, (hipStream_t)stream));
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    return *flags_out ? IRBPP_ERR_DEVICE : IRBPP_OK;
}

}  // extern "C"
