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

#include "../../irbpp_amd/csrc/contours_device.h"

extern "C" int host_approx_and_convex_wave(const uint8_t* pts, int count, uint32_t* vrows) {
    memset(vrows, 0, 16 * sizeof(uint32_t));
    pthread_barrier_init(&g_bar, nullptr, 64);
    int ok[64];
    std::vector<std::thread> lanes;
    for (int l = 0; l < 64; ++l)
        lanes.emplace_back([&, l] {
            threadIdx.x = (unsigned)l;
            ok[l] = irbpp::approx_and_convex_wave(pts, count, vrows) ? 1 : 0;
        });
    for (auto& t : lanes) t.join();
    pthread_barrier_destroy(&g_bar);
    for (int l = 1; l < 64; ++l)
        if (ok[l] != ok[0]) return -1;                     // the result must be wave-uniform
    return ok[0];
}
