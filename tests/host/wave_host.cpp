// Host harness (test infrastructure): the wave-cooperative Douglas-Peucker of
// irbpp_amd/csrc/contours_device.h run on the CPU by 64 threads in lockstep.  Every cross-lane
// operation the routine uses (v_readlane, v_readfirstlane, ds_bpermute via __shfl, and the DPP
// controls of its max-reduction) is an exchange through a shared array between two barriers, so the
// routine's wave-uniform control flow is executed exactly as a wave64 would.
#include <pthread.h>
#include <stdint.h>
#include <string.h>

#include <thread>
#include <vector>

#define __device__
#define __forceinline__ inline
static inline int __ffs(int v) { return __builtin_ffs(v); }
static inline int __clz(int v) { return v ? __builtin_clz((unsigned)v) : 32; }
static inline unsigned __brev(unsigned v) { unsigned r = 0; for (int i = 0; i < 32; ++i) r |= ((v >> i) & 1u) << (31 - i); return r; }
static inline unsigned atomicOr(uint32_t* p, uint32_t v) { return __atomic_fetch_or(p, v, __ATOMIC_RELAXED); }

static thread_local struct { unsigned x; } threadIdx;
static pthread_barrier_t g_bar;
static int g_x[64];

static inline int xchg(int v, int src) {                  // value of lane `src` (per-lane src allowed)
    g_x[threadIdx.x & 63] = v;
    pthread_barrier_wait(&g_bar);
    const int r = g_x[src & 63];
    pthread_barrier_wait(&g_bar);
    return r;
}
// v_mov_b32_dpp with the controls wave_max_u32 uses (CDNA ISA: quad_perm 0x00-0xFF, row_mirror 0x140,
// row_half_mirror 0x141, row_bcast15 0x142, row_bcast31 0x143); rows outside row_mask keep `old`
static inline int emu_dpp(int old, int v, int ctrl, int row_mask) {
    const int lane = threadIdx.x & 63, row = lane >> 4;
    int src = lane;
    bool has_src = true;
    if (ctrl <= 0xFF) src = (lane & ~3) | ((ctrl >> (2 * (lane & 3))) & 3);
    else if (ctrl == 0x140) src = (lane & ~15) | (15 - (lane & 15));
    else if (ctrl == 0x141) src = (lane & ~7) | (7 - (lane & 7));
    else if (ctrl == 0x142) { has_src = row >= 1; src = row * 16 - 1; }
    else if (ctrl == 0x143) { has_src = row >= 2; src = 31; }
    const int got = xchg(v, has_src ? src : lane);
    return (((row_mask >> row) & 1) && has_src) ? got : old;
}
#define __builtin_amdgcn_readlane(v, l) xchg((v), (l))
#define __builtin_amdgcn_readfirstlane(v) xchg((v), 0)
#define __builtin_amdgcn_update_dpp(old, v, ctrl, rm, bm, bc) emu_dpp((old), (v), (ctrl), (rm))
#define __shfl(v, l) xchg((v), (l))
// a wave executes one instruction for all lanes before the next: where lanes talk through LDS the
// device code has a scheduling-only wave barrier, which here is a real one
#define IRBPP_WAVE_SYNC() pthread_barrier_wait(&g_bar)
static unsigned long long g_bal;
static inline unsigned long long __ballot(bool p) {
    if ((threadIdx.x & 63) == 0) g_bal = 0ull;
    pthread_barrier_wait(&g_bar);
    if (p) __atomic_fetch_or(&g_bal, 1ull << (threadIdx.x & 63), __ATOMIC_RELAXED);
    pthread_barrier_wait(&g_bar);
    const unsigned long long r = g_bal;
    pthread_barrier_wait(&g_bar);
    return r;
}
static inline unsigned atomicMax(uint32_t* p, uint32_t v) {
    uint32_t o = __atomic_load_n(p, __ATOMIC_RELAXED);
    while (o < v && !__atomic_compare_exchange_n(p, &o, v, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
    return o;
}
static inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
static inline int __ffsll(long long v) { return __builtin_ffsll(v); }
static inline int __clzll(long long v) { return v ? __builtin_clzll((unsigned long long)v) : 64; }

#include "../../irbpp_amd/csrc/contours_device.h"

// Several borders packed back to back over the 64 * P positions of a wave (position q = u * 64 + lane),
// exactly as the kernels pack them: counts[b] points each (sum <= 64 * P), pts = the borders' point lists one
// after the other.  vrows[b*16 .. b*16+15] receives the vertex bits of border b (border b plays 'rotation' b).
template <int P>
static int run_segmented(const uint8_t* pts, const int* counts, int n_borders, uint32_t* vrows) {
    memset(vrows, 0, (size_t)n_borders * 16 * sizeof(uint32_t));
    pthread_barrier_init(&g_bar, nullptr, 64);
    static uint32_t slots[64 * 4];
    static uint8_t scratch[64 * 4];
    std::vector<std::thread> lanes;
    for (int l = 0; l < 64; ++l)
        lanes.emplace_back([&, l] {
            threadIdx.x = (unsigned)l;
            bool live[P];
            int pv[P], j[P], n[P], sb[P], rot[P];
            const uint8_t* pp[P];
            for (int u = 0; u < P; ++u) {
                const int q = u * 64 + l;
                int off = 0, mine = -1, sbb = 0, nn = 1;
                for (int b = 0; b < n_borders; ++b) {
                    if (q >= off && q < off + counts[b]) { mine = b; sbb = off; nn = counts[b]; }
                    off += counts[b];
                }
                live[u] = mine >= 0;
                pp[u] = pts + sbb;
                j[u] = live[u] ? q - sbb : 0;
                n[u] = nn;
                sb[u] = live[u] ? sbb : 0;
                rot[u] = live[u] ? mine : 0;
                pv[u] = live[u] ? pp[u][j[u]] : 0;
            }
            irbpp::approx_convex_segmented<P>(l, live, pv, j, n, sb, pp, rot, slots, scratch, vrows);
        });
    for (auto& t : lanes) t.join();
    pthread_barrier_destroy(&g_bar);
    return 0;
}

extern "C" int host_approx_convex_segmented(const uint8_t* pts, const int* counts, int n_borders, uint32_t* vrows, int points_per_lane) {
    if (points_per_lane == 1) return run_segmented<1>(pts, counts, n_borders, vrows);
    if (points_per_lane == 2) return run_segmented<2>(pts, counts, n_borders, vrows);
    if (points_per_lane == 4) return run_segmented<4>(pts, counts, n_borders, vrows);
    return -1;
}

// The clean-up pass + convexity of one polygon (vertex i in lane i, cnt <= 64) three ways: the loop-free routine
// (returns 1 if it settled the polygon, 0 if it hands over to the sequential one), the wave-uniform sequential
// routine, and the one-lane serial routine on a byte array.
extern "C" int host_cleanup_parallel(const uint8_t* poly, int cnt, uint32_t* vrows) {
    memset(vrows, 0, 16 * sizeof(uint32_t));
    pthread_barrier_init(&g_bar, nullptr, 64);
    int handled[64];
    std::vector<std::thread> lanes;
    for (int l = 0; l < 64; ++l)
        lanes.emplace_back([&, l] {
            threadIdx.x = (unsigned)l;
            handled[l] = irbpp::cleanup_convex_parallel(l, l < cnt ? (int)poly[l] : 0, cnt, vrows) ? 1 : 0;
        });
    for (auto& t : lanes) t.join();
    pthread_barrier_destroy(&g_bar);
    for (int l = 1; l < 64; ++l) if (handled[l] != handled[0]) return -1;       // the verdict is uniform
    return handled[0];
}
extern "C" void host_cleanup_wave(const uint8_t* poly, int cnt, uint32_t* vrows) {
    memset(vrows, 0, 16 * sizeof(uint32_t));
    pthread_barrier_init(&g_bar, nullptr, 64);
    std::vector<std::thread> lanes;
    for (int l = 0; l < 64; ++l)
        lanes.emplace_back([&, l] {
            threadIdx.x = (unsigned)l;
            irbpp::cleanup_convex_wave(l, l < cnt ? (int)poly[l] : 0, cnt, vrows);
        });
    for (auto& t : lanes) t.join();
    pthread_barrier_destroy(&g_bar);
}
extern "C" void host_cleanup_serial(const uint8_t* poly, int cnt, uint32_t* vrows) {
    memset(vrows, 0, 16 * sizeof(uint32_t));
    uint8_t buf[256];
    memcpy(buf, poly, (size_t)cnt);
    irbpp::cleanup_convex_serial(buf, cnt, vrows);
}
