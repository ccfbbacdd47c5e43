// irbpp_replay.hip -- caller-side kernels: the sum trees of the N per-env prioritised replay
// memories (memory.py:15-93) and the masked greedy action of Agent.act (agent.py:51-58) for all envs at once.
// The reference walks one recursive Python tree per env on the CPU; replay.py already holds the N trees as one
// [N][2*cap-1] float32 tensor -- these kernels replace its chains of small torch kernels (~80 launches for a
// sample) by one launch each.  float32 arithmetic in the reference's order: node = left + right, descent by
// `value <= left ? left : (value - left, right)`.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

namespace irbpp {

// SegmentTree.find / _retrieve (memory.py:72-86): one thread per (env, draw).
extern "C" __global__ void __launch_bounds__(256)
irbpp_sumtree_find_kernel(const float* __restrict__ tree, int n_env, int cap, const float* __restrict__ values, int b,
                          float* __restrict__ prob, int64_t* __restrict__ data_idx, int64_t* __restrict__ tree_idx) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= n_env * b) return;
    const float* row = tree + (size_t)(t / b) * (2 * cap - 1);
    const int len = 2 * cap - 1;
    int idx = 0;
    float v = values[t];
    for (;;) {
        const int left = 2 * idx + 1;
        if (left >= len) break;                          // a leaf
        const float lv = row[left];
        if (v <= lv) idx = left;
        else { v = v - lv; idx = left + 1; }
    }
    prob[t] = row[idx];
    data_idx[t] = idx - (cap - 1);
    tree_idx[t] = idx;
}

// ReplayMemory._get_samples_from_segments (memory.py:161-176) for every (env, segment) at once: draw a position
// uniformly in the segment, walk the tree, and redraw while the draw straddles the write index or has probability 0
// (the reference's rejection loop) -- all inside one kernel instead of a host loop of launch / test / sync rounds.
// The uniform numbers come from a counter-based generator (splitmix64 of seed, env, segment, try): a different
// stream from numpy's, as any vectorised sampler's must be.  failed[0] is set if some draw found no valid position
// within max_tries (too few transitions appended).  One thread per (env, segment).
__device__ __forceinline__ float uniform01(uint64_t seed, uint32_t env, uint32_t j, uint32_t attempt) {
    uint64_t z = seed + 0x9E3779B97F4A7C15ull * (((uint64_t)env << 32) ^ ((uint64_t)j << 8) ^ (uint64_t)attempt ^ 0xD1B54A32D192ED03ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z = z ^ (z >> 31);
    return (float)(z >> 40) * (1.0f / 16777216.0f);              // 24 random bits -> [0, 1)
}
extern "C" __global__ void __launch_bounds__(256)
irbpp_sumtree_sample_kernel(const float* __restrict__ tree, const int64_t* __restrict__ index, int n_env, int cap, int b, int n_step,
                            uint64_t seed, int max_tries, float* __restrict__ prob, int64_t* __restrict__ data_idx,
                            int64_t* __restrict__ tree_idx, int32_t* __restrict__ failed) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= n_env * b) return;
    const int env = t / b, j = t - env * b;
    const int len = 2 * cap - 1;
    const float* row = tree + (size_t)env * len;
    const float segment = row[0] / (float)b;                      // p_total / batch_size (memory.py:195)
    const float lo = (float)j * segment;
    const int w = (int)index[env];
    float p = 0.0f;
    int idx = 0;
    bool ok = false;
    for (int attempt = 0; attempt < max_tries && !ok; ++attempt) {
        float v = lo + uniform01(seed, (uint32_t)env, (uint32_t)j, (uint32_t)attempt) * segment;
        idx = 0;
        for (;;) {
            const int left = 2 * idx + 1;
            if (left >= len) break;
            const float lv = row[left];
            if (v <= lv) idx = left;
            else { v = v - lv; idx = left + 1; }
        }
        p = row[idx];
        const int d = idx - (cap - 1);
        int a = (w - d) % cap, c = (d - w) % cap;                 // Python's % : non-negative
        if (a < 0) a += cap;
        if (c < 0) c += cap;
        ok = a > n_step && c >= 1 && p != 0.0f;                   // memory.py:175
    }
    if (!ok) atomicOr(failed, 1);
    prob[t] = p;
    data_idx[t] = idx - (cap - 1);
    tree_idx[t] = idx;
}

// SegmentTree.update (memory.py:55-58) for b leaves per env, applied in list order (a leaf listed twice keeps its
// last value, as in the reference's sequential loop), then every ancestor recomputed as left + right from its
// final children -- which is what the sequence of _propagate calls leaves behind.  One wave per env, the tree
// row staged in LDS; rows longer than the LDS buffer take the host-side path.
constexpr int SUMTREE_LDS = 16384;                       // floats: capacities up to 8192 transitions per env
extern "C" __global__ void __launch_bounds__(64)
irbpp_sumtree_update_kernel(float* __restrict__ tree, float* __restrict__ maxp, int cap, const int64_t* __restrict__ tree_idx,
                            const float* __restrict__ prio, int b, const uint8_t* __restrict__ row_mask) {
    extern __shared__ float row[];                       // 2*cap - 1 floats (dynamic: a 64-transition ring needs 508 bytes, not 64 KB)
    const int env = blockIdx.x, lane = threadIdx.x;
    if (row_mask && !row_mask[env]) return;              // append() of a subset of the envs
    const int len = 2 * cap - 1;
    float* g = tree + (size_t)env * len;
    for (int i = lane; i < len; i += 64) row[i] = g[i];
    __builtin_amdgcn_wave_barrier();
    if (lane == 0) {
        float m = maxp[env];
        for (int j = 0; j < b; ++j) {
            const float v = prio[(size_t)env * b + j];
            const int64_t ti = tree_idx[(size_t)env * b + j];
            if (ti < cap - 1 || ti >= len) continue;     // not a leaf of this row: never write outside the staged tree
            row[ti] = v;
            m = fmaxf(m, v);                             // self.max = max(value, self.max)
        }
        maxp[env] = m;
    }
    __builtin_amdgcn_wave_barrier();
    // internal nodes are 0 .. cap-2; depth d occupies [2^d - 1, 2^(d+1) - 1): bottom-up, one depth at a time
    int depth = 0;
    while ((2 << depth) - 1 <= cap - 2) ++depth;
    for (int d = depth; d >= 0; --d) {
        const int lo = (1 << d) - 1, hi = (2 << d) - 1 < cap - 1 ? (2 << d) - 1 : cap - 1;
        for (int i = lo + lane; i < hi; i += 64) row[i] = row[2 * i + 1] + row[2 * i + 2];
        __builtin_amdgcn_wave_barrier();
    }
    for (int i = lane; i < len; i += 64) g[i] = row[i];
}

// ReplayMemory.append (memory.py:117-121) + SegmentTree.append (:60-70) for every env (or the envs of `valid`) in ONE launch:
// the transition enters the env's ring at its write index with the env's maximum priority, the leaf's ancestors are
// recomputed as left + right in float32 (_propagate, :47-52), index / full / the episode's timestep counter move on.
// One workgroup per env: 256 threads copy the observation row (the launch's bytes: obs_len floats in, obs_len out), thread 0
// does the scalars and walks the leaf's ancestors -- the sibling comes from memory, the node just written stays in a
// register, so nothing is read back.  action int32 or int64, reward float32 or float64 (what the environment hands out:
// no conversion launches in front), terminal / valid one byte per env.
extern "C" __global__ void __launch_bounds__(256)
irbpp_replay_append_kernel(float* __restrict__ states, int64_t* __restrict__ actions, float* __restrict__ rewards,
                           uint8_t* __restrict__ nonterminals, int32_t* __restrict__ timesteps, float* __restrict__ tree,
                           float* __restrict__ maxp, int64_t* __restrict__ index, uint8_t* __restrict__ full,
                           int32_t* __restrict__ tcount, int cap, int obs_len, const float* __restrict__ state, long long state_stride,
                           const void* __restrict__ action, int action_bytes, const void* __restrict__ reward, int reward_bytes,
                           const uint8_t* __restrict__ terminal, const uint8_t* __restrict__ valid) {
    const int env = blockIdx.x, tid = threadIdx.x;
    if (valid && !valid[env]) return;                            // trainer.py:184-186: only Valid samples are stored
    const int pos = (int)index[env];
    const float* src = state + (size_t)env * state_stride;
    float* dst = states + ((size_t)env * cap + pos) * obs_len;
    for (int i = tid; i < obs_len; i += 256) dst[i] = src[i];
    __syncthreads();                                             // every wave has read index[env]
    if (tid != 0) return;
    const size_t slot = (size_t)env * cap + pos;
    const bool term = terminal[env] != 0;
    const int t = tcount[env];
    timesteps[slot] = t;
    actions[slot] = action_bytes == 8 ? ((const int64_t*)action)[env] : (int64_t)((const int32_t*)action)[env];
    rewards[slot] = reward_bytes == 8 ? (float)((const double*)reward)[env] : ((const float*)reward)[env];
    nonterminals[slot] = term ? 0 : 1;
    float* row = tree + (size_t)env * (2 * cap - 1);
    float v = maxp[env];                                         // (self.max = max(value, self.max) leaves it as it is)
    int i = pos + cap - 1;
    row[i] = v;
    while (i > 0) {
        const int parent = (i - 1) >> 1;
        v = (i & 1) ? v + row[i + 1] : row[i - 1] + v;           // left + right: i odd is the left child
        row[parent] = v;
        i = parent;
    }
    const int nxt = pos + 1 == cap ? 0 : pos + 1;
    index[env] = nxt;
    if (nxt == 0) full[env] = 1;
    tcount[env] = term ? 0 : t + 1;
}

// Agent.act (agent.py:51-58) after the network: sum_q[(1 - mask).bool()] = -inf; argmax(1), with the mask read
// straight from the observation (get_mask_from_state, tools.py:298-299: column 4 of the [S][5] candidate block).
// One wave per env; the first maximum wins.
extern "C" __global__ void __launch_bounds__(256)
irbpp_masked_argmax_kernel(const float* __restrict__ q, int q_stride, const float* __restrict__ obs, int obs_stride, int s_rows,
                           int n_env, int64_t* __restrict__ action) {
    const int env = (blockIdx.x * 256 + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    if (env >= n_env) return;
    const float* qr = q + (size_t)env * q_stride;
    const float* c = obs + (size_t)env * obs_stride;
    float best = -INFINITY;
    int bi = 0x7fffffff;
    for (int i = lane; i < s_rows; i += 64) {
        const float v = c[i * 5 + 4] != 0.0f ? qr[i] : -INFINITY;
        if (v > best || (v == best && i < bi)) { best = v; bi = i; }       // all -inf: the lowest index, like torch.argmax
    }
    for (int o = 32; o > 0; o >>= 1) {
        const float ob = __shfl_xor(best, o);
        const int oi = __shfl_xor(bi, o);
        if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
    }
    if (lane == 0) action[env] = bi;
}

// ReplayMemory._get_transition_new + the batch assembly of ReplayMemory.sample (memory.py:123-139,178-204) for the
// `b` sampled positions of every env: the n+1 consecutive transitions from data_idx (ring order), blanked from the
// first one that follows a terminal transition; state at t, state at t+n (zeros if blanked), action at t, the n-step
// return sum_k gamma^k r_{t+k+1}, the non-terminal flag at t+n, and the importance weights
// (filled * prob / p_total)^-beta normalised by their maximum per env.  One workgroup per env; rows are written
// env-major ([env * b + j]) exactly as Agent.learn concatenates the per-env batches (agent.py:69-84).
extern "C" __global__ void __launch_bounds__(256)
irbpp_replay_gather_kernel(const float* __restrict__ states, const int64_t* __restrict__ actions, const float* __restrict__ rewards,
                           const uint8_t* __restrict__ nonterminals, const float* __restrict__ tree, const int64_t* __restrict__ index,
                           const uint8_t* __restrict__ full, const float* __restrict__ scaling, int cap, int obs_len, int n_step,
                           int b, float beta, const int64_t* __restrict__ data_idx, const float* __restrict__ prob,
                           float* __restrict__ out_state, int64_t* __restrict__ out_action, float* __restrict__ out_return,
                           float* __restrict__ out_next, float* __restrict__ out_nonterminal, float* __restrict__ out_weight) {
    __shared__ float wmax[4];
    __shared__ float wbuf[256];
    const int env = blockIdx.x, tid = threadIdx.x;
    const float p_total = tree[(size_t)env * (2 * cap - 1)];
    const float filled = full[env] ? (float)cap : (float)index[env];
    // weights: thread j < b owns sample j (b <= 256)
    float w = 0.0f;
    if (tid < b) {
        const float probs = prob[(size_t)env * b + tid] / p_total;
        w = powf(filled * probs, -beta);
    }
    float m = w;
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if ((tid & 63) == 0) wmax[tid >> 6] = m;
    __syncthreads();
    m = fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]));
    if (tid < b) out_weight[(size_t)env * b + tid] = w / m;
    for (int j = 0; j < b; ++j) {
        const size_t row = (size_t)env * b + j;
        const int d0 = (int)(data_idx[row] % cap);
        // alive chain (wave-uniform scalar work, repeated by every thread: n_step is tiny)
        bool alive = true;
        float ret = 0.0f;
        int pos = d0;
        for (int t = 0; t < n_step; ++t) {
            const float r = alive ? rewards[(size_t)env * cap + pos] : 0.0f;
            ret = ret + r * scaling[t];
            alive = alive && nonterminals[(size_t)env * cap + pos] != 0;
            pos = pos + 1 == cap ? 0 : pos + 1;
        }
        const int dn = pos;                                       // (d0 + n_step) % cap
        const float* s0 = states + ((size_t)env * cap + d0) * obs_len;
        const float* sn = states + ((size_t)env * cap + dn) * obs_len;
        float* o0 = out_state + row * obs_len;
        float* o1 = out_next + row * obs_len;
        for (int i = tid; i < obs_len; i += 256) {
            o0[i] = s0[i];
            o1[i] = alive ? sn[i] : 0.0f;
        }
        if (tid == 0) {
            out_action[row] = actions[(size_t)env * cap + d0];
            out_return[row] = ret;
            out_nonterminal[row] = (alive && nonterminals[(size_t)env * cap + dn] != 0) ? 1.0f : 0.0f;
        }
    }
}

}  // namespace irbpp
