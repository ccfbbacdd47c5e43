// irbpp_itemgen.h -- host side: the item streams of the reference's random item creators, bit for bit.
//
// RandomItemCreator / RandomInstanceCreator / RandomCateCreator (IRcreator.py:26-72) draw every item with
// np.random.choice on the process-global legacy generator, which envs.py:41 seeds with seed + rank
// (PackingGame.seed, binPhy.py:118-123).  Nothing else on the no-physics path consumes that generator, so the
// items an environment sees are a pure function of (seed + rank, the creator's lists).  np.random.choice(a) on a
// 1-D list is a[RandomState.randint(0, len(a))], and the legacy randint draws MT19937 32-bit words, masks them
// with the smallest 2^k - 1 >= len(a) - 1 and rejects values above it (numpy/random/src/distributions:
// buffered_bounded_masked_uint32; a one-element list consumes nothing).  numpy is the third-party piece here: its
// legacy stream is frozen by NEP 19, and tests/test_itemgen.py pins this file against numpy itself and against
// goldens drawn by the reference's own classes (tests/golden/make_golden.py).
#pragma once
#include <stdint.h>

#include <new>
#include <vector>

#include "../../include/irbpp.h"

struct irbpp_itemgen {
    uint32_t key[624];
    int pos;
    std::vector<int32_t> offsets;    // [n_groups + 1] into members (two-stage creators); empty = one stage over members
    std::vector<int32_t> members;
};

namespace irbpp_host {

inline void mt_seed(irbpp_itemgen* g, uint32_t seed) {            // np.random.seed(int) -> mt19937_seed (init_genrand)
    for (int i = 0; i < 624; ++i) {
        g->key[i] = seed;
        seed = 1812433253u * (seed ^ (seed >> 30)) + (uint32_t)i + 1u;
    }
    g->pos = 624;
}

inline uint32_t mt_next(irbpp_itemgen* g) {                        // mt19937_next: regenerate 624 words, temper one
    if (g->pos == 624) {
        uint32_t* k = g->key;
        int i = 0;
        for (; i < 624 - 397; ++i) {
            const uint32_t y = (k[i] & 0x80000000u) | (k[i + 1] & 0x7fffffffu);
            k[i] = k[i + 397] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
        }
        for (; i < 623; ++i) {
            const uint32_t y = (k[i] & 0x80000000u) | (k[i + 1] & 0x7fffffffu);
            k[i] = k[i + (397 - 624)] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
        }
        const uint32_t y = (k[623] & 0x80000000u) | (k[0] & 0x7fffffffu);
        k[623] = k[396] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
        g->pos = 0;
    }
    uint32_t y = g->key[g->pos++];
    y ^= y >> 11;
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= y >> 18;
    return y;
}

inline uint32_t choice_index(irbpp_itemgen* g, uint32_t n) {       // RandomState.randint(0, n), n >= 1
    const uint32_t rng = n - 1u;
    if (rng == 0u) return 0u;                                      // no draw at all
    uint32_t mask = rng;
    mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4; mask |= mask >> 8; mask |= mask >> 16;
    uint32_t v;
    while ((v = mt_next(g) & mask) > rng) {}
    return v;
}

}  // namespace irbpp_host

extern "C" {

int irbpp_itemgen_create(uint32_t seed, int32_t n_groups, const int32_t* group_offsets, const int32_t* members,
                         int32_t n_members, irbpp_itemgen** out) {
    if (!out || !members || n_members < 1 || n_groups < 0 || (n_groups > 0 && !group_offsets)) return IRBPP_ERR_ARG;
    if (n_groups > 0) {
        if (group_offsets[0] != 0 || group_offsets[n_groups] != n_members) return IRBPP_ERR_ARG;
        for (int i = 0; i < n_groups; ++i)
            if (group_offsets[i + 1] <= group_offsets[i]) return IRBPP_ERR_ARG;     // np.random.choice of an empty list raises
    }
    irbpp_itemgen* g = new (std::nothrow) irbpp_itemgen();
    if (!g) return IRBPP_ERR_NOMEM;
    irbpp_host::mt_seed(g, seed);
    if (n_groups > 0) g->offsets.assign(group_offsets, group_offsets + n_groups + 1);
    g->members.assign(members, members + n_members);
    *out = g;
    return IRBPP_OK;
}

int irbpp_itemgen_draw(irbpp_itemgen* g, int32_t count, int32_t* out_host) {
    if (!g || count < 0 || (count > 0 && !out_host)) return IRBPP_ERR_ARG;
    const int n_groups = (int)g->offsets.size() - 1;
    for (int i = 0; i < count; ++i) {
        if (n_groups > 0) {                  // name = choice(names); item = choice(lists[name])   (IRcreator.py:49-51, 70-72)
            const uint32_t name = irbpp_host::choice_index(g, (uint32_t)n_groups);
            const int lo = g->offsets[name], len = g->offsets[name + 1] - lo;
            out_host[i] = g->members[lo + (int)irbpp_host::choice_index(g, (uint32_t)len)];
        } else {                             // choice(item_set)   (IRcreator.py:32-33)
            out_host[i] = g->members[irbpp_host::choice_index(g, (uint32_t)g->members.size())];
        }
    }
    return IRBPP_OK;
}

int irbpp_itemgen_destroy(irbpp_itemgen* g) {
    delete g;
    return IRBPP_OK;
}

}  // extern "C"
