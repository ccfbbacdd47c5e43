"""ORACLE (test infrastructure) -- ctypes front end of the plain-C restatement (oracle/c/).

``COracleVecEnv`` has the interface of ``oracle.packing.OracleVecEnv`` and must agree with it
bit for bit (tests/test_c_oracle.py); it is ~100x faster, which makes large parity runs and a
fair CPU baseline possible.  Build with ``make -C oracle`` (done by __graft_entry__.build()).
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "_ref", "libirbpp_oracle.so")
_lib = None
_f64 = C.POINTER(C.c_double)
_i32 = C.POINTER(C.c_int)


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(os.path.join(HERE, "c", "irbpp_oracle.c")):
            subprocess.run(["make", "-C", HERE, "-s"], check=True)
        lib = C.CDLL(LIB)
        lib.orc_create.restype = C.c_void_p
        lib.orc_create.argtypes = [C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, C.c_double, _f64, C.c_double,
                                   C.c_int, _f64, _f64, _i32, C.POINTER(C.c_int64), C.c_int64, _f64, _f64, _f64, _f64,
                                   _i32, C.c_int, C.c_int, C.c_int, C.c_int]
        lib.orc_create_borrowed.restype = C.c_void_p
        lib.orc_create_borrowed.argtypes = lib.orc_create.argtypes
        lib.orc_destroy.argtypes = [C.c_void_p]
        _hp = C.POINTER(C.c_void_p)
        lib.orc_reset_many.argtypes = [_hp, C.c_int, C.c_int, _f64, C.c_long]
        lib.orc_get_action_candidates_many.argtypes = [_hp, C.c_int, C.c_int, _i32, _f64, C.c_long]
        lib.orc_step_many.argtypes = [_hp, C.c_int, C.c_int, _i32, _f64, C.c_long, _f64, _i32, _i32, _f64]
        lib.orc_obs_len.argtypes = [C.c_void_p, C.c_int]
        lib.orc_reset.argtypes = [C.c_void_p, _f64]
        lib.orc_get_action_candidates.argtypes = [C.c_void_p, C.c_int, _f64]
        lib.orc_step.argtypes = [C.c_void_p, C.c_int, _f64, _f64, _i32, _f64]
        lib.orc_step.restype = C.c_int
        lib.orc_get_grids.argtypes = [C.c_void_p, _f64, _f64]
        lib.orc_get_heightmap.argtypes = [C.c_void_p, _f64]
        lib.orc_set_heightmap.argtypes = [C.c_void_p, _f64]
        _lib = lib
    return _lib


def _pack(shapes):
    n, R = shapes.n_shapes, shapes.n_rot
    dims = np.zeros((n, R, 2), dtype=np.int32)
    offs = np.zeros((n, R), dtype=np.int64)
    pools = [[], [], [], []]
    pos = 0
    for k in range(n):
        for r in range(R):
            dims[k, r] = shapes.tables[k][r][0].shape
            offs[k, r] = pos
            pos += shapes.tables[k][r][0].size
            for pool, arr in zip(pools, shapes.tables[k][r]):
                pool.append(np.ascontiguousarray(arr, dtype=np.float64).reshape(-1))
    return dims, offs, pos, [np.concatenate(p) for p in pools]


class SharedTables(object):
    """One packed copy of a data set's tables and trajectories for any number of CPackingGame handles of this process
    (orc_create_borrowed): the handles keep a reference to it, so it outlives them."""

    def __init__(self, shapes, sequences):
        self.dims, self.offs, self.plen, (self.T, self.B, self.mH, self.mB) = _pack(shapes)
        self.ext = np.ascontiguousarray(shapes.extents, dtype=np.float64)
        self.vol = np.ascontiguousarray(shapes.volumes, dtype=np.float64)
        self.seq = np.ascontiguousarray(sequences, dtype=np.int32)


class CPackingGame(object):
    def __init__(self, shapes, sequences, resolutionA=0.02, resolutionH=0.01, resolutionZ=0.01,
                 bin_dimension=(0.32, 0.32, 0.30), selectedAction=500, bufferSize=1, scale_z=100.0,
                 first_traj=1, traj_stride=1, shared=None):
        self.lib = load()
        self.shared = shared                     # (borrowed tables must outlive the handle)
        st = shared if shared is not None else SharedTables(shapes, sequences)
        dims, offs, plen, T, B, mH, mB, ext, vol, seq = st.dims, st.offs, st.plen, st.T, st.B, st.mH, st.mB, st.ext, st.vol, st.seq
        binr = np.round(np.asarray(bin_dimension, dtype=np.float64), 6)
        p = lambda a: a.ctypes.data_as(_f64)   # noqa: E731
        create = self.lib.orc_create if shared is None else self.lib.orc_create_borrowed
        self.h = C.c_void_p(create(
            shapes.n_rot, selectedAction, bufferSize, resolutionA, resolutionH, resolutionZ, p(binr), scale_z,
            shapes.n_shapes, p(ext), p(vol), dims.ctypes.data_as(_i32), offs.ctypes.data_as(C.POINTER(C.c_int64)),
            plen, p(T), p(B), p(mH), p(mB), seq.ctypes.data_as(_i32), seq.shape[0], seq.shape[1], first_traj, traj_stride))
        self.obs_len = self.lib.orc_obs_len(self.h, 0)
        self.loc_obs_len = self.lib.orc_obs_len(self.h, 1)
        self.n_rot = shapes.n_rot
        self.Hx = int(np.ceil(binr[0] / resolutionH)); self.Hy = int(np.ceil(binr[1] / resolutionH))
        self.Ax = int(np.ceil(binr[0] / resolutionA)); self.Ay = int(np.ceil(binr[1] / resolutionA))

    def reset(self):
        obs = np.empty(self.obs_len)
        self.lib.orc_reset(self.h, obs.ctypes.data_as(_f64))
        return obs

    def get_action_candidates(self, order_action):
        obs = np.empty(self.loc_obs_len)
        self.lib.orc_get_action_candidates(self.h, int(order_action), obs.ctypes.data_as(_f64))
        return obs

    def step(self, action):
        obs = np.empty(self.obs_len)
        rew, ratio, counter = C.c_double(), C.c_double(), C.c_int()
        done = self.lib.orc_step(self.h, int(action), obs.ctypes.data_as(_f64), C.byref(rew), C.byref(counter), C.byref(ratio))
        info = {"Valid": True}
        if done:
            info = {"counter": counter.value, "ratio": ratio.value, "Valid": True}
        return obs, rew.value, bool(done), info

    def grids(self):
        n = self.n_rot * self.Ax * self.Ay
        pz, mk = np.empty(n), np.empty(n)
        self.lib.orc_get_grids(self.h, pz.ctypes.data_as(_f64), mk.ctypes.data_as(_f64))
        return pz.reshape(self.n_rot, self.Ax, self.Ay), mk.reshape(self.n_rot, self.Ax, self.Ay)

    def heightmap(self):
        hm = np.empty(self.Hx * self.Hy)
        self.lib.orc_get_heightmap(self.h, hm.ctypes.data_as(_f64))
        return hm.reshape(self.Hx, self.Hy)

    def set_heightmap(self, hm):
        hm = np.ascontiguousarray(hm, dtype=np.float64)
        self.lib.orc_set_heightmap(self.h, hm.ctypes.data_as(_f64))

    def __del__(self):
        try:
            self.lib.orc_destroy(self.h)
        except Exception:
            pass


class COracleVecEnv(object):
    """Same protocol as oracle.packing.OracleVecEnv (auto-reset + Monitor episode info)."""

    def __init__(self, num_envs, shapes, sequences, traj_start=1, global_offset=0, global_num=None, threads=1, **kw):
        """threads > 1: reset / get_action_candidates / step deal the bins to that many host threads inside the C library
        (orc_*_many); results do not depend on it."""
        global_num = num_envs if global_num is None else global_num
        self.threads = max(1, min(int(threads), num_envs))
        self.shared = SharedTables(shapes, sequences)         # one copy of the tables for all bins of this environment
        self.envs = [CPackingGame(shapes, sequences, first_traj=traj_start + global_offset + g,
                                  traj_stride=global_num, shared=self.shared, **kw) for g in range(num_envs)]
        self.num_envs = num_envs
        self.obs_len = self.envs[0].obs_len
        self.rewards = [[] for _ in range(num_envs)]

    def _handles(self):
        if getattr(self, "_harr", None) is None:
            self._harr = (C.c_void_p * self.num_envs)(*[e.h.value for e in self.envs])
            self.lib = self.envs[0].lib
            self.loc_obs_len = self.envs[0].loc_obs_len
        return self._harr

    def reset(self):
        self.rewards = [[] for _ in range(self.num_envs)]
        obs = np.empty((self.num_envs, self.obs_len))
        h = self._handles()
        self.lib.orc_reset_many(h, self.num_envs, self.threads, obs.ctypes.data_as(_f64), obs.shape[1])
        return obs

    def get_action_candidates(self, order_actions):
        h = self._handles()
        obs = np.empty((self.num_envs, self.loc_obs_len))
        oa = np.ascontiguousarray(order_actions, dtype=np.int32)
        self.lib.orc_get_action_candidates_many(h, self.num_envs, self.threads, oa.ctypes.data_as(_i32), obs.ctypes.data_as(_f64),
                                                obs.shape[1])
        return obs

    def reset_specific(self, indexs):
        """shmem_vec_env.py:113-117: per-env reset of the listed envs (next trajectory, no episode statistics)."""
        out = []
        for i in indexs:
            self.rewards[i] = []
            out.append(self.envs[i].reset())
        return np.array(out).reshape(len(out), self.obs_len)

    def step(self, actions):
        h = self._handles()
        n = self.num_envs
        obs = np.empty((n, self.obs_len))
        rew, ratio = np.empty(n), np.empty(n)
        done, counter = np.empty(n, dtype=np.int32), np.empty(n, dtype=np.int32)
        act = np.ascontiguousarray(actions, dtype=np.int32)
        self.lib.orc_step_many(h, n, self.threads, act.ctypes.data_as(_i32), obs.ctypes.data_as(_f64), obs.shape[1],
                               rew.ctypes.data_as(_f64), done.ctypes.data_as(_i32), counter.ctypes.data_as(_i32),
                               ratio.ctypes.data_as(_f64))
        infos = []
        for i in range(n):
            self.rewards[i].append(float(rew[i]))
            if done[i]:
                infos.append({"counter": int(counter[i]), "ratio": float(ratio[i]), "Valid": True,
                              "episode": {"r": round(sum(self.rewards[i]), 6), "l": len(self.rewards[i])}})
                self.rewards[i] = []
            else:
                infos.append({"Valid": True})
        return obs, rew, done.astype(bool), infos
