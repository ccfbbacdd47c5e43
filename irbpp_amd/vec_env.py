"""Host-side mirror of the reference's vectorised-environment surface, backed by the HIP library.

    GpuPackingEnv   device-tensor API straight over the C ABI (include/irbpp.h)
    GpuVecEnv       the ShmemVecEnv/VecPyTorch protocol the trainer calls
                    (wrapper/vec_env.py:29-138, wrapper/shmem_vec_env.py:20-117,
                    envs.py:142-165): num_envs, observation_space, action_space, reset(),
                    step_async()/step_wait()/step(), get_action_candidates(), close()
    make_vec_envs   same return triple as envs.make_vec_envs (envs.py:67-99)

PyTorch is plumbing only (device memory, streams); all environment arithmetic runs in
libirbpp_hip.so.  Without that library this module raises -- there is no CPU path.
"""
from __future__ import annotations

import ctypes as C
import os
import time
from collections.abc import Sequence
from typing import Optional

import numpy as np
import torch

from . import _lib
from .shapes import ShapeSet

BIN_DIMENSION = (0.32, 0.32, 0.30)      # arguments.py:115


class Box(object):
    """Stand-in for gym.spaces.Box (binPhy.py:100-101); gym is not a dependency here."""

    def __init__(self, low, high, shape, dtype=np.float32):
        self.low, self.high, self.shape, self.dtype = low, high, tuple(shape), dtype


class Discrete(object):
    """Stand-in for gym.spaces.Discrete (binPhy.py:102)."""

    def __init__(self, n):
        self.n = int(n)


def _ptr(t: Optional[torch.Tensor]):
    return C.c_void_p(0 if t is None else t.data_ptr())


class GpuPackingEnv(object):
    """N independent bins on one MI355X.  All tensors live on ``device``; nothing here syncs."""

    def __init__(self, shapes: ShapeSet, sequences: np.ndarray, num_bins: int, *,
                 resolutionA: float = 0.02, resolutionH: float = 0.01, resolutionZ: float = 0.01,
                 bin_dimension=BIN_DIMENSION, selectedAction: int = 500, bufferSize: int = 1,
                 scale_z: float = 100.0, traj_start: int = 1, global_offset: int = 0,
                 global_bins: Optional[int] = None, device="cuda:0", stability: int = 0, tuning: int = 0,
                 item_stream: int = 0):
        self.lib = _lib.load()
        if not torch.cuda.is_available():
            raise RuntimeError("GpuPackingEnv needs a HIP device; irbpp_amd has no CPU fallback")
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("GpuPackingEnv runs on a HIP device only")
        dev_index = self.device.index if self.device.index is not None else torch.cuda.current_device()
        self.num_bins = int(num_bins)
        self.n_rot = shapes.n_rot
        self.S = int(selectedAction)
        self.K = int(bufferSize)
        shapes.validate(resolutionH, resolutionA)
        bin_r = np.round(np.asarray(bin_dimension, dtype=np.float64), decimals=6)
        cfg = _lib.IrbppConfig(
            num_bins=self.num_bins, n_rot=self.n_rot, selected=self.S, buffer_size=self.K,
            resolution_a=resolutionA, resolution_h=resolutionH, resolution_z=resolutionZ,
            bin=(C.c_double * 3)(*bin_r), scale_z=scale_z, traj_start=traj_start,
            global_offset=global_offset, global_bins=self.num_bins if global_bins is None else global_bins,
            device=dev_index, stability=int(stability), tuning=int(tuning), item_stream=int(item_stream))
        self._h = C.c_void_p()
        torch.cuda.set_device(self.device)
        _lib.check(self.lib.irbpp_create(C.byref(cfg), C.byref(self._h)), "irbpp_create")
        self.resolutionA, self.resolutionH, self.resolutionZ = float(resolutionA), float(resolutionH), float(resolutionZ)
        self.bin_dimension = tuple(float(x) for x in bin_r)
        self.Hx = int(np.ceil(bin_r[0] / resolutionH))
        self.Hy = int(np.ceil(bin_r[1] / resolutionH))
        self.Ax = int(np.ceil(bin_r[0] / resolutionA))
        self.Ay = int(np.ceil(bin_r[1] / resolutionA))
        self.obs_len = self.lib.irbpp_obs_len(self._h, 0)
        self.loc_obs_len = self.lib.irbpp_obs_len(self._h, 1)
        self._load_shapes(shapes)
        seq = np.ascontiguousarray(sequences, dtype=np.int32)
        assert seq.ndim == 2
        _lib.check(self.lib.irbpp_load_sequences(self._h, seq.ctypes.data_as(_lib.c_i32_p), seq.shape[0], seq.shape[1]),
                   "irbpp_load_sequences")
        # ONE contiguous block for the small per-step outputs and the error word -> one pinned D2H copy
        n = self.num_bins
        off_err = (34 * n + 3) & ~3
        self._out = torch.zeros((off_err + 4,), dtype=torch.uint8, device=self.device)
        # the pinned landing buffer of that copy; step_info_host hands out an owned copy of it (34 bytes per bin), like
        # the fresh arrays ShmemVecEnv.step_wait builds every step (shmem_vec_env.py:76-81)
        self._out_host = torch.empty((off_err + 4,), dtype=torch.uint8, pin_memory=True)
        self._out_f64 = self._out[:24 * n].view(torch.float64).view(3, n)           # reward, ratio, ep_reward
        self._out_i32 = self._out[24 * n:32 * n].view(torch.int32).view(2, n)       # counter, ep_len
        self._out_done = self._out[32 * n:33 * n]
        self._out_stable = self._out[33 * n:34 * n]           # stability proxy verdict (stability >= 1)
        self._out_err = self._out[off_err:off_err + 4].view(torch.int32)
        self._step_out = _lib.IrbppStepOut(
            reward_dev=self._out_f64[0].data_ptr(), ratio_dev=self._out_f64[1].data_ptr(),
            ep_reward_dev=self._out_f64[2].data_ptr(), counter_dev=self._out_i32[0].data_ptr(),
            ep_len_dev=self._out_i32[1].data_ptr(), done_dev=self._out_done.data_ptr(), stable_dev=self._out_stable.data_ptr(),
            err_dev=self._out_err.data_ptr())

    # -- set-up ------------------------------------------------------------------------------
    def _load_shapes(self, shapes: ShapeSet):
        n, R = shapes.n_shapes, shapes.n_rot
        dims = np.zeros((n, R, 2), dtype=np.int32)
        offs = np.zeros((n, R), dtype=np.int64)
        pools = [[], [], [], []]
        pos = 0
        for k in range(n):
            for r in range(R):
                T, B, mH, mB = shapes.tables[k][r]
                dims[k, r] = T.shape
                offs[k, r] = pos
                pos += T.size
                for pool, arr in zip(pools, (T, B, mH, mB)):
                    pool.append(np.ascontiguousarray(arr, dtype=np.float64).reshape(-1))
        T, B, mH, mB = (np.concatenate(p) for p in pools)
        ext = np.ascontiguousarray(shapes.extents, dtype=np.float64)
        vol = np.ascontiguousarray(shapes.volumes, dtype=np.float64)
        f64 = lambda a: a.ctypes.data_as(_lib.c_f64_p)   # noqa: E731
        _lib.check(self.lib.irbpp_load_shapes(
            self._h, n, f64(ext), f64(vol), dims.ctypes.data_as(_lib.c_i32_p),
            offs.ctypes.data_as(C.POINTER(C.c_int64)), pos, f64(T), f64(B), f64(mH), f64(mB)), "irbpp_load_shapes")

    def _stream(self, stream=None):
        """The HIP stream a call is issued on: ``stream`` (a torch.cuda.Stream) or the current one."""
        return C.c_void_p((stream if stream is not None else torch.cuda.current_stream(self.device)).cuda_stream)

    # -- transitions -------------------------------------------------------------------------
    def reset(self) -> torch.Tensor:
        obs = torch.empty((self.num_bins, self.obs_len), dtype=torch.float32, device=self.device)
        _lib.check(self.lib.irbpp_reset(self._h, _ptr(obs), self._stream()), "irbpp_reset")
        return obs

    def reset_bins(self, bins: torch.Tensor) -> torch.Tensor:
        """PackingGame.reset of the listed bins only (distinct int32 indices on the device):
        float32[len(bins), obs_len], row i = reset observation of bin ``bins[i]``."""
        assert bins.dtype == torch.int32 and bins.is_cuda and bins.dim() == 1
        obs = torch.empty((bins.numel(), self.obs_len), dtype=torch.float32, device=self.device)
        _lib.check(self.lib.irbpp_reset_bins(self._h, _ptr(bins), int(bins.numel()), _ptr(obs), self._stream()),
                   "irbpp_reset_bins")
        return obs

    def step(self, actions: torch.Tensor, obs_out: Optional[torch.Tensor] = None, stream=None):
        """actions: int32[N] on the device.  Returns (obs, reward f64[N], done u8[N]) device tensors;
        the reward/done tensors are views of buffers overwritten by the next step.  ``stream``: issue on this
        torch.cuda.Stream instead of the current one (only together with ``obs_out``: nothing is allocated then)."""
        assert actions.dtype == torch.int32 and actions.is_cuda and actions.numel() == self.num_bins
        obs = obs_out if obs_out is not None else \
            torch.empty((self.num_bins, self.obs_len), dtype=torch.float32, device=self.device)
        _lib.check(self.lib.irbpp_step(self._h, _ptr(actions), _ptr(obs), C.byref(self._step_out), self._stream(stream)),
                   "irbpp_step")
        return obs, self._out_f64[0], self._out_done

    def get_action_candidates(self, order_actions: torch.Tensor, obs_out: Optional[torch.Tensor] = None, stream=None) -> torch.Tensor:
        assert order_actions.dtype == torch.int32 and order_actions.is_cuda
        obs = obs_out if obs_out is not None else \
            torch.empty((self.num_bins, self.loc_obs_len), dtype=torch.float32, device=self.device)
        _lib.check(self.lib.irbpp_get_action_candidates(self._h, _ptr(order_actions), _ptr(obs), self._stream(stream)),
                   "irbpp_get_action_candidates")
        return obs

    def get_all_possible_observation(self, obs_out: Optional[torch.Tensor] = None, stream=None) -> torch.Tensor:
        """PackingGame.get_all_possible_observation (binPhy.py:171-180) of every bin: float32[N, k, loc_obs_len], the
        location observation of each buffer slot on the current heightmap (row b reshaped to -1 is the reference's
        concatenation).  The candidate rows a following step indexes are the last slot's, the chosen slot stays."""
        obs = obs_out if obs_out is not None else \
            torch.empty((self.num_bins, self.K, self.loc_obs_len), dtype=torch.float32, device=self.device)
        assert obs.is_contiguous() and obs.shape == (self.num_bins, self.K, self.loc_obs_len)
        _lib.check(self.lib.irbpp_get_all_possible_observation(self._h, _ptr(obs), self._stream(stream)),
                   "irbpp_get_all_possible_observation")
        return obs

    def policy_minz(self, loc_obs: torch.Tensor, actions_out: Optional[torch.Tensor] = None) -> torch.Tensor:
        act = actions_out if actions_out is not None else \
            torch.empty((self.num_bins,), dtype=torch.int32, device=self.device)
        _lib.check(self.lib.irbpp_policy_minz(self._h, _ptr(loc_obs), loc_obs.stride(0), _ptr(act), self._stream()),
                   "irbpp_policy_minz")
        return act

    def set_auto_policy(self, actions: Optional[torch.Tensor]) -> None:
        """Fuse the scripted MINZ policy into the observation: while ``actions`` (int32[N] on the device) is set,
        reset / reset_bins / step (online) / get_action_candidates also write the action ``policy_minz`` would pick
        on the observation they emit; ``None`` switches it off.  The buffer may be the one passed to the next step()."""
        if actions is not None:
            assert actions.dtype == torch.int32 and actions.is_cuda and actions.numel() == self.num_bins and actions.is_contiguous()
        self._auto_actions = actions                # keep the buffer alive
        _lib.check(self.lib.irbpp_set_auto_policy(self._h, _ptr(actions)), "irbpp_set_auto_policy")

    def register_obs_buffer(self, obs: torch.Tensor) -> torch.Tensor:
        """Hand a location-observation buffer ([N, loc_obs_len] float32, contiguous) over to the library for good:
        calls that get it as ``obs_out`` then store only the candidate rows that exist and clear the ones that
        existed before instead of rewriting the zero tail (irbpp_register_obs_buffer).  Nobody else may write it."""
        assert obs.dtype == torch.float32 and obs.is_cuda and obs.is_contiguous() and obs.shape == (self.num_bins, self.loc_obs_len)
        if not hasattr(self, "_obs_buffers"):
            self._obs_buffers = []
        self._obs_buffers.append(obs)                # keep it alive
        _lib.check(self.lib.irbpp_register_obs_buffer(self._h, _ptr(obs)), "irbpp_register_obs_buffer")
        return obs

    def unregister_obs_buffer(self, obs: torch.Tensor) -> None:
        """Give a registered buffer back (before it is freed: the registration is keyed by its address)."""
        _lib.check(self.lib.irbpp_unregister_obs_buffer(self._h, _ptr(obs)), "irbpp_unregister_obs_buffer")
        self._obs_buffers = [t for t in getattr(self, "_obs_buffers", []) if t.data_ptr() != obs.data_ptr()]

    def invalidate_obs_buffers(self, obs: Optional[torch.Tensor] = None) -> None:
        """After the caller wrote into a registered buffer itself (all of them with ``None``): the next emit through it
        rewrites every row."""
        if getattr(self, "_obs_buffers", None):
            _lib.check(self.lib.irbpp_invalidate_obs_buffer(self._h, _ptr(obs), self._stream()), "irbpp_invalidate_obs_buffer")

    # -- stage-level access (tests, tooling) ---------------------------------------------------
    def possible_position(self, item_ids: torch.Tensor):
        posz = torch.empty((self.num_bins, self.n_rot, self.Ax, self.Ay), dtype=torch.float64, device=self.device)
        mask = torch.empty((self.num_bins, self.n_rot, self.Ax, self.Ay), dtype=torch.uint8, device=self.device)
        _lib.check(self.lib.irbpp_possible_position(self._h, _ptr(item_ids), _ptr(posz), _ptr(mask), self._stream()),
                   "irbpp_possible_position")
        return posz, mask

    HEURISTICS = {"MINZ": 1, "DBLF": 2, "FIRSTFIT": 3, "HM": 4}

    def heuristic_action(self, method: str, dir_idx: int = 0) -> torch.Tensor:
        """Space.get_heuristic_action (space.py:162-218) per bin -> int32[N,3] = (rot, lx, ly)."""
        out = torch.empty((self.num_bins, 3), dtype=torch.int32, device=self.device)
        _lib.check(self.lib.irbpp_heuristic_action(self._h, self.HEURISTICS[method], dir_idx, _ptr(out), self._stream()),
                   "irbpp_heuristic_action")
        return out

    def convex_hull_actions(self, posz_valid: torch.Tensor, mask: torch.Tensor) -> torch.Tensor:
        """[G,R,Ax,Ay] float64 / uint8 -> uint32-as-int32 [G,R,16]: word ``row`` has bit ``col`` set."""
        g = posz_valid.shape[0]
        assert posz_valid.shape == (g, self.n_rot, self.Ax, self.Ay) and mask.shape == posz_valid.shape
        out = torch.empty((g, self.n_rot, 16), dtype=torch.int32, device=self.device)
        _lib.check(self.lib.irbpp_convex_hull_actions(self._h, g, _ptr(posz_valid.contiguous()),
                                                      _ptr(mask.contiguous()), _ptr(out), self._stream()),
                   "irbpp_convex_hull_actions")
        return out

    def get_heightmaps(self) -> torch.Tensor:
        hm = torch.empty((self.num_bins, self.Hx, self.Hy), dtype=torch.float64, device=self.device)
        _lib.check(self.lib.irbpp_get_heightmaps(self._h, _ptr(hm), self._stream()), "irbpp_get_heightmaps")
        return hm

    def set_heightmaps(self, hm: torch.Tensor) -> None:
        assert hm.dtype == torch.float64 and hm.shape == (self.num_bins, self.Hx, self.Hy)
        _lib.check(self.lib.irbpp_set_heightmaps(self._h, _ptr(hm.contiguous()), self._stream()), "irbpp_set_heightmaps")

    def episode_totals(self) -> torch.Tensor:
        """float64[4] on device: finished episodes, sum ratio, sum counter, sum reward."""
        out = torch.empty((4,), dtype=torch.float64, device=self.device)
        _lib.check(self.lib.irbpp_episode_totals(self._h, _ptr(out), self._stream()), "irbpp_episode_totals")
        return out

    def enable_placement_log(self, capacity: int = 256):
        """Device-side PackingGame.packed: (meta uint32-as-int32 [N,cap], z float64 [N,cap])."""
        self._log_meta = torch.zeros((self.num_bins, capacity), dtype=torch.int32, device=self.device)
        self._log_z = torch.zeros((self.num_bins, capacity), dtype=torch.float64, device=self.device)
        _lib.check(self.lib.irbpp_set_placement_log(self._h, _ptr(self._log_meta), _ptr(self._log_z), capacity),
                   "irbpp_set_placement_log")
        return self._log_meta, self._log_z

    def enable_phase_cycles(self, on: bool = True) -> Optional[torch.Tensor]:
        """Tooling: int64[N,16] shader-clock stamps written by every later transition launch."""
        self._cycles = torch.zeros((self.num_bins, 16), dtype=torch.int64, device=self.device) if on else None
        _lib.check(self.lib.irbpp_debug_phase_cycles(self._h, _ptr(self._cycles)), "irbpp_debug_phase_cycles")
        return self._cycles

    def kernel_info(self):
        """Tooling: (LDS bytes per workgroup, name of the transition-kernel build that launches)."""
        lds, name = C.c_int32(0), C.c_char_p()
        _lib.check(self.lib.irbpp_debug_kernel_info(self._h, C.byref(lds), C.byref(name)), "irbpp_debug_kernel_info")
        return lds.value, name.value.decode()

    def enable_kernel_timing(self, capacity: int, every: int = 1) -> None:
        """Tooling: bracket the kernels of the next transitions with HIP events on their stream, ``capacity`` pairs
        in a ring (0 switches it off), around every ``every``-th transition only."""
        _lib.check(self.lib.irbpp_debug_kernel_timing_every(self._h, int(every)), "irbpp_debug_kernel_timing_every")
        _lib.check(self.lib.irbpp_debug_kernel_timing(self._h, int(capacity)), "irbpp_debug_kernel_timing")
        self._timing_cap = int(capacity)

    def kernel_times_ms(self) -> np.ndarray:
        """Durations (ms) of the transition kernels recorded since the last call, oldest first."""
        cap = getattr(self, "_timing_cap", 0)
        buf = (C.c_float * max(cap, 1))()
        n = C.c_int32(0)
        _lib.check(self.lib.irbpp_debug_kernel_times(self._h, buf, cap, C.byref(n)), "irbpp_debug_kernel_times")
        return np.array(buf[:n.value], dtype=np.float64)

    def check_device_error(self) -> None:
        flags = C.c_int32(0)
        _lib.check(self.lib.irbpp_device_error(self._h, self._stream(), C.byref(flags)),
                   f"device error flags={flags.value} ({_lib.deverr_names(flags.value)})")

    def step_info_host(self):
        """ONE pinned asynchronous D2H copy of the small per-step outputs + the device error word, then one
        stream synchronisation -> dict of numpy arrays that belong to the caller (a copy of the pinned landing buffer:
        a rollout may keep `done` / `infos` of step t for as long as it likes, as with the reference's fresh arrays).
        Raises if a kernel raised its error word."""
        st = torch.cuda.current_stream(self.device)
        self._info_copy_async()
        st.synchronize()
        return self._info_parse()

    def _info_copy_async(self) -> None:
        """the copy alone, on the current stream (GroupedPackingEnv issues every group's on the group's own stream before it
        waits for any of them)"""
        self._out_host.copy_(self._out, non_blocking=True)

    def _info_parse(self):
        n = self.num_bins
        h = self._out_host.numpy().copy()
        err = int(h[-4:].view(np.int32)[0])
        if err:
            raise _lib.IrbppError(f"device error flags={err} ({_lib.deverr_names(err)}): " + _lib.load().irbpp_status_string(-4).decode())
        f64 = h[:24 * n].view(np.float64).reshape(3, n)
        i32 = h[24 * n:32 * n].view(np.int32).reshape(2, n)
        return dict(reward=f64[0], ratio=f64[1], ep_reward=f64[2], counter=i32[0], ep_len=i32[1],
                    done=h[32 * n:33 * n].view(np.bool_), stable=h[33 * n:34 * n].view(np.bool_))

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self.lib.irbpp_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class GroupedPackingEnv(object):
    """``num_bins`` bins as ``num_groups`` independent groups, each a GpuPackingEnv on its own HIP stream.

    Bins never interact, so a batch can be stepped as several sub-batches that do not wait for each other:
    the straggler workgroups at the end of one group's kernels overlap the next group's work, which a single
    launch over all bins cannot do (and joining the groups after every step costs more than it gives back,
    see irbpp_capi.hip).  This is the shape of an actor loop with double-buffered environment groups: act on
    group g while the other groups step.  Group g owns the global bins [g*n/G, (g+1)*n/G) and row block g of
    every [num_bins, ...] tensor; trajectories are assigned by global bin index, so results are the same as
    one GpuPackingEnv over all bins (tests/test_grouped.py).  Nothing here synchronises except
    ``synchronize()`` and ``reset()``.

    Stream discipline: ``step`` / ``get_action_candidates`` make every group's stream wait for the caller's current
    stream (the actions may have been computed there) and tell the caching allocator that the action and
    observation tensors are in use on the group streams (``record_stream``), so a temporary the caller drops right
    after the call is not handed out again before the kernels have read it.  The ``*_group`` methods do the same
    for their one stream."""

    def __init__(self, shapes: ShapeSet, sequences: np.ndarray, num_bins: int, num_groups: int = 4, *, device="cuda:0",
                 global_offset: int = 0, global_bins: Optional[int] = None, **kw):
        if num_groups < 1 or num_bins % num_groups != 0:
            raise ValueError("num_bins must be a multiple of num_groups")
        self.device = torch.device(device)
        self.num_bins, self.num_groups, self.per = int(num_bins), int(num_groups), num_bins // num_groups
        total = num_bins if global_bins is None else global_bins
        # item streams (item_stream=1): row b of the table is bin b's own ring, so group g gets rows [g*per, (g+1)*per)
        seq_of = (lambda g: sequences[g * self.per:(g + 1) * self.per]) if kw.get("item_stream") else (lambda g: sequences)
        self.groups = [GpuPackingEnv(shapes, seq_of(g), self.per, device=device, global_offset=global_offset + g * self.per,
                                     global_bins=total, **kw) for g in range(num_groups)]
        # one group: the caller's current stream; two: the process's pair of streams that was CHECKED to run side by side
        # (group_stream_pair); more: a stream each from torch's pool, whose mapping onto the runtime's hardware queues is
        # the runtime's (see groups_for)
        if num_groups == 1:
            self.streams = [torch.cuda.current_stream(self.device)]
        else:
            self.streams = group_streams(self.device, num_groups)[0]
        e = self.groups[0]
        self.obs_len, self.loc_obs_len, self.K, self.S, self.n_rot = e.obs_len, e.loc_obs_len, e.K, e.S, e.n_rot
        self.Hx, self.Hy, self.Ax, self.Ay = e.Hx, e.Hy, e.Ax, e.Ay

    def rows(self, g: int) -> slice:
        return slice(g * self.per, (g + 1) * self.per)

    def reset(self) -> torch.Tensor:
        """All groups; the result is complete when this returns to the current stream."""
        obs = torch.empty((self.num_bins, self.obs_len), dtype=torch.float32, device=self.device)
        cur = torch.cuda.current_stream(self.device)
        for g, (e, st) in enumerate(zip(self.groups, self.streams)):
            st.wait_stream(cur)
            with torch.cuda.stream(st):
                _lib.check(e.lib.irbpp_reset(e._h, _ptr(obs[self.rows(g)]), e._stream()), "irbpp_reset")
            cur.wait_stream(st)
        return obs

    def _enter(self, g: int, *tensors, wait: bool = True) -> None:
        """Group g's stream is about to read/write ``tensors`` that live on (and may be produced by) the current stream."""
        st = self.streams[g]
        cur = torch.cuda.current_stream(self.device)
        if wait and st != cur:
            st.wait_stream(cur)
            for t in tensors:
                if t is not None:
                    t.record_stream(st)

    def step_group(self, g: int, actions: torch.Tensor, obs_out: Optional[torch.Tensor] = None, wait: bool = True):
        """Group g alone, on its stream: (obs, reward, done) of its ``per`` bins.  ``wait=False``: the caller vouches
        that the tensors are long-lived and were produced on group g's own stream (the fused policy,
        ``policy_minz_group``): no cross-stream dependency is inserted."""
        self._enter(g, actions, obs_out, wait=wait)
        if obs_out is not None:              # nothing to allocate: straight onto the group's stream (the context manager
            return self.groups[g].step(actions, obs_out=obs_out, stream=self.streams[g])     # costs the host ~10 us per call)
        with torch.cuda.stream(self.streams[g]):
            return self.groups[g].step(actions, obs_out=obs_out)

    def policy_minz_group(self, g: int, loc_obs: torch.Tensor, actions_out: Optional[torch.Tensor] = None, wait: bool = True):
        self._enter(g, loc_obs, actions_out, wait=wait)
        with torch.cuda.stream(self.streams[g]):
            return self.groups[g].policy_minz(loc_obs, actions_out=actions_out)

    def policy_minz(self, loc_obs: torch.Tensor, actions_out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """The scripted MINZ policy over all bins, on the caller's current stream (as GpuPackingEnv.policy_minz: the
        observation is the caller's, e.g. what a synchronous step() returned)."""
        act = actions_out if actions_out is not None else \
            torch.empty((self.num_bins,), dtype=torch.int32, device=self.device)
        for g, e in enumerate(self.groups):
            e.policy_minz(loc_obs[self.rows(g)], actions_out=act[self.rows(g)])
        return act

    def get_action_candidates_group(self, g: int, order_actions: torch.Tensor, obs_out: Optional[torch.Tensor] = None,
                                    wait: bool = True):
        self._enter(g, order_actions, obs_out, wait=wait)
        if obs_out is not None:
            return self.groups[g].get_action_candidates(order_actions, obs_out=obs_out, stream=self.streams[g])
        with torch.cuda.stream(self.streams[g]):
            return self.groups[g].get_action_candidates(order_actions, obs_out=obs_out)

    def step(self, actions: torch.Tensor, obs_out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Every group on its own stream (no join): row block g of the result belongs to stream g until
        ``synchronize()``.  ``actions`` may still be in flight on the current stream."""
        obs = obs_out if obs_out is not None else \
            torch.empty((self.num_bins, self.obs_len), dtype=torch.float32, device=self.device)
        for g in range(self.num_groups):
            self.step_group(g, actions[self.rows(g)], obs_out=obs[self.rows(g)])
        return obs

    def get_action_candidates(self, order_actions: torch.Tensor, obs_out: Optional[torch.Tensor] = None) -> torch.Tensor:
        obs = obs_out if obs_out is not None else \
            torch.empty((self.num_bins, self.loc_obs_len), dtype=torch.float32, device=self.device)
        for g in range(self.num_groups):
            self.get_action_candidates_group(g, order_actions[self.rows(g)], obs_out=obs[self.rows(g)])
        return obs

    def get_all_possible_observation(self) -> torch.Tensor:
        """float32[N, k, loc_obs_len]: every buffer slot of every bin (binPhy.py:171-180), each group on its own stream."""
        k = self.groups[0].K
        obs = torch.empty((self.num_bins, k, self.loc_obs_len), dtype=torch.float32, device=self.device)
        for g in range(self.num_groups):
            self._enter(g, obs)
            self.groups[g].get_all_possible_observation(obs_out=obs[self.rows(g)], stream=self.streams[g])
        return obs

    def reset_bins(self, bins: torch.Tensor) -> torch.Tensor:
        """PackingGame.reset of the listed bins (distinct global-to-this-env indices), rows in list order."""
        self.synchronize()
        idx = bins.cpu().numpy()
        out = torch.empty((len(idx), self.obs_len), dtype=torch.float32, device=self.device)
        for g in range(self.num_groups):
            sel = np.nonzero(idx // self.per == g)[0]
            if len(sel):
                local = torch.from_numpy((idx[sel] - g * self.per).astype(np.int32)).to(self.device)
                out[torch.from_numpy(sel).to(self.device)] = self.groups[g].reset_bins(local)
        torch.cuda.synchronize(self.device)
        return out

    def synchronize(self) -> None:
        for st in self.streams:
            st.synchronize()

    def episode_totals(self) -> torch.Tensor:
        self.synchronize()
        return torch.stack([e.episode_totals() for e in self.groups]).sum(0)

    def get_heightmaps(self) -> torch.Tensor:
        self.synchronize()
        return torch.cat([e.get_heightmaps() for e in self.groups])

    def check_device_error(self) -> None:
        for e in self.groups:
            e.check_device_error()

    def step_info_host(self):
        """Every group's copy goes out on the group's own stream, behind its kernels, before the first wait: a group's
        D2H overlaps the other groups' kernels, and the step has one synchronisation per stream (until round 5 session 34:
        all streams joined first, then copy + wait group by group on the caller's stream)."""
        for e, st in zip(self.groups, self.streams):
            with torch.cuda.stream(st):
                e._info_copy_async()
        parts = []
        for e, st in zip(self.groups, self.streams):
            st.synchronize()
            parts.append(e._info_parse())
        return {k: np.concatenate([p[k] for p in parts]) for k in parts[0]}

    def close(self):
        for e in self.groups:
            e.close()


_STREAM_PAIRS = {}
_STREAM_PROBE = {}           # what the probe saw (pair / single time of every candidate tried): group_stream_report()


def group_stream_report(device) -> dict:
    """The outcome of this process's stream probe on ``device`` (bench.py prints it next to `value`): streams checked to overlap,
    candidates tried, the pair / single spin-kernel time ratios measured (< 1.5 = side by side)."""
    device = torch.device(device)
    key = (device.type, device.index if device.index is not None else torch.cuda.current_device())
    chosen, tried = _STREAM_PAIRS.get(key, ([], [0]))
    return {"overlapping_streams": len(chosen), "candidates_tried": tried[0], "pair_over_single": list(_STREAM_PROBE.get("ratios", []))}


def _run_side_by_side(a, b, device, cycles=1_500_000) -> bool:
    """Do kernels on streams ``a`` and ``b`` overlap?  A spin kernel (torch.cuda._sleep) on one stream, then one on each:
    the pair takes about as long as the single one if the streams have hardware queues of their own, twice as long if they
    share one."""
    import time
    if not hasattr(torch.cuda, "_sleep"):              # (no spin kernel in this torch: nothing can be checked)
        return False

    def timed(streams):
        torch.cuda.synchronize(device)
        t0 = time.perf_counter()
        for st in streams:
            with torch.cuda.stream(st):
                torch.cuda._sleep(cycles)
        for st in streams:
            st.synchronize()
        return time.perf_counter() - t0

    timed([a, b])                                      # (first launches on a stream pay for its set-up)
    ratio = 0.0
    for _ in range(3):                                 # a busy device can make one measurement look shared: ask again before rejecting
        single = min(timed([a]) for _ in range(2))
        pair = min(timed([a, b]) for _ in range(2))
        ratio = pair / single if single > 0 else 2.0
        if ratio < 1.5:
            break
    _STREAM_PROBE.setdefault("ratios", []).append(round(ratio, 3))
    return ratio < 1.5


def group_streams(device, count: int = 2):
    """``count`` HIP streams this process steps that many groups of bins on, and whether they were all seen to overlap with
    each other.

    How the runtime maps streams onto its hardware queues depends on what the process has created before: two streams
    fresh from torch's pool ran two groups of a 4096-bin BlockOut environment at 44.6 M steps/s in one instance and at
    15.9 - 28 M in the next (a shared queue: every kernel of one group then waits for the other group's, profiles/r05/s10,
    s11 -- with streams of different priority likewise).  So the streams are chosen ONCE per process and device: candidates
    from the pool are tried against the ones already chosen until a spin-kernel probe shows them running side by side with
    every one of those, and every GroupedPackingEnv uses the first ``count`` of that list.  The runtime's default of four
    hardware queues bounds the list at four.  -> (streams, overlap_seen)"""
    device = torch.device(device)
    key = (device.type, device.index if device.index is not None else torch.cuda.current_device())
    chosen, tried = _STREAM_PAIRS.setdefault(key, ([], [0]))
    if not chosen:
        chosen.append(torch.cuda.Stream(device=device))
    while len(chosen) < count and tried[0] < 24:
        cand = torch.cuda.Stream(device=device)
        tried[0] += 1
        if all(_run_side_by_side(st, cand, device) for st in chosen):
            chosen.append(cand)
    if len(chosen) >= count:
        return list(chosen[:count]), True
    extra = [torch.cuda.Stream(device=device) for _ in range(count - len(chosen))]      # (no guarantee for these)
    return list(chosen) + extra, False


def group_stream_pair(device):
    """group_streams(device, 2) as ((stream0, stream1), overlap_seen)."""
    streams, ok = group_streams(device, 2)
    return (streams[0], streams[1]), ok


def groups_for(workload_kind: str, num_bins: int, buffered: bool = False, device=None) -> int:
    """How many independent groups of bins (GroupedPackingEnv / GpuVecEnv(num_groups=...)) a caller without a preference
    should step ``num_bins`` bins as: **two, or one**.

    A step is a chain of four or five kernels, each with a ramp and a tail, some bound by instruction issue (transition,
    polygon), some by latency at three waves per SIMD (trace): two groups on two streams fill one group's idle issue slots
    with the other's work -- IF the two streams have hardware queues of their own, which ``group_stream_pair`` checks once
    per process (on a shared queue two groups run at a THIRD of one group's rate).  On the checked pair the figures repeat
    from instance to instance and from process to process (profiles/r05/s12: 4 instances x 2 processes each; one -> two
    groups): BlockOut 8192 bins 52.6 -> 59.3 M steps/s, 4096 bins 40.8 -> 44.6 M, 2048 bins 28.5 -> 31.5 M, free-form solids
    4096 bins 19.5 -> 22.4 M, the 64 x 64 heightmap at 2048 bins 6.2 -> 7.16 M, a buffered environment (k = 10) at 8192 bins
    42.4 -> 48.4 M, at 4096 bins 34.1 -> 37.8 M and at 2048 bins 24.5 -> 26.6 M -- but at 1024 bins (BASELINE config 4 per GPU) level at best (15.3 M; figures of session 12, before the buffered step became an apply kernel): a
    chain of short latency-bound launches whose length does not depend on the number of bins.  More than two groups are not
    recommended: their streams come from torch's pool and share hardware queues as the runtime sees fit (BlockOut 8192 bins
    as four groups: 43.7 / 30 M).  A caller with a preference passes ``num_groups`` itself (GpuVecEnv, GroupedPackingEnv,
    ``bench.py --groups``).

    ``workload_kind``: "general" / "abc_fine" (free-form cell lists, the generic overlap path) or anything else (lattice /
    box data); ``buffered``: bufferSize > 1 (also recognised from a kind that ends in "_k<digits>"); ``device``: if given,
    the answer is 1 unless this process's pair of streams on that device was seen to overlap."""
    import re
    if num_bins % 2 != 0:
        return 1
    if buffered or re.search(r"_k\d+$", workload_kind):
        want = 2 if num_bins >= 2048 else 1
    elif workload_kind in ("general", "abc_fine"):
        want = 2 if num_bins >= 1024 else 1
    else:
        want = 2 if num_bins >= 2048 else 1
    if want == 2 and device is not None and not group_stream_pair(device)[1]:
        return 1
    return want


class _Infos(Sequence):
    """The ``infos`` sequence of step_wait, materialised lazily: N dicts per step would cost more
    host time than the whole GPU step.  infos[i] -> {'Valid': True} plus, where done,
    'counter', 'ratio' (binPhy.py:306-309) and 'episode': {'r','l','t'} (monitor.py:58-75)."""

    def __init__(self, h, t):
        self._h, self._t = h, t

    def __len__(self):
        return len(self._h["done"])

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self[j] for j in range(*i.indices(len(self)))]
        h = self._h
        if not h["done"][i]:
            return {"Valid": True}
        return {"counter": int(h["counter"][i]), "ratio": float(h["ratio"][i]), "Valid": True,
                "episode": {"r": round(float(h["ep_reward"][i]), 6), "l": int(h["ep_len"][i]), "t": self._t}}


class GpuVecEnv(object):
    """Drop-in for ``VecPyTorch(ShmemVecEnv(...))`` as the trainer uses it (trainer.py:148,165,267,281).

    ``obs_ring`` (default 3; ``make_vec_envs`` -- the drop-in constructor -- passes 0 unless ``args.obs_ring`` says
    otherwise): observations are written into a ring of that many buffers owned by the environment
    and, for online environments, registered with the library (``irbpp_register_obs_buffer``: only the candidate
    rows that exist are stored), instead of a fresh allocation with a full rewrite per step.  The tensor a call
    returns is therefore overwritten ``obs_ring`` observation-producing calls later -- the trainer keeps ``state``
    across exactly one step (trainer.py:184-186,213) and the replay memory copies what it stores, so 3 leaves a
    spare; a caller that holds observations for longer passes ``obs_ring=0`` (a fresh tensor per call).  Do not
    write into a returned tensor except through ``reset_specific``'s documented pattern (it invalidates the ring)."""

    closed = False

    def __init__(self, shapes: ShapeSet, sequences: np.ndarray, num_envs: int, device="cuda:0",
                 allow_early_resets: bool = True, num_groups: int = 1, obs_ring: int = 3, feeder=None, **env_kw):
        """``num_groups`` > 1: the envs are stepped as that many independent groups on their own HIP streams
        (GroupedPackingEnv); 0: as many as ``groups_for`` recommends for this data set and size; ``step()`` still covers all envs, and ``step_async(actions, group=g)`` /
        ``step_wait(group=g)`` let an actor loop work on one group while the others step.  (A caller that only ever calls the
        synchronous ``step()`` is best served by one group: the groups' launches are host time on its one dependent chain.)
        ``feeder``: an ``itemgen.StreamFeeder`` for environments created with ``item_stream=1``."""
        self.num_groups = int(num_groups)
        if self.num_groups == 0:            # the library's own choice (groups_for): which overlap path does this data set take?
            probe = GpuPackingEnv(shapes, sequences[:1], 1, device=device, **{k: v for k, v in env_kw.items() if k != "item_stream"})
            generic = probe.lib.irbpp_overlap_path(probe._h) in (3, 4)        # cell lists for all or some rotations
            fine = probe.Hx * probe.Hy > 32 * 32
            probe.close()
            self.num_groups = groups_for(("abc_fine" if fine else "general") if generic else "lattice", num_envs,
                                         buffered=int(env_kw.get("bufferSize", 1)) > 1, device=device)
        if self.num_groups > 1:
            self.env = GroupedPackingEnv(shapes, sequences, num_envs, self.num_groups, device=device, **env_kw)
        else:
            self.env = GpuPackingEnv(shapes, sequences, num_envs, device=device, **env_kw)
        self.allow_early_resets = allow_early_resets                       # Monitor's flag (monitor.py:44-48)
        self.num_envs = num_envs
        self.device = self.env.device
        self.obs_len = self.env.obs_len
        self.observation_space = Box(low=0.0, high=float(env_kw.get("bin_dimension", BIN_DIMENSION)[2]),
                                     shape=(self.obs_len,))
        self.action_space = Discrete(self.env.K if self.env.K > 1 else self.env.S)
        self.waiting_step = False
        self._pending = None
        self._group_pending = {}
        self.tstart = time.time()
        self.feeder = feeder
        if feeder is not None:
            feeder.attach(self.env)                                          # one part per group, each fed on its group's stream
        # persistent action buffers: pinned staging for host actions (no pageable bounce) and one device tensor, a pair
        # for step() and a pair for get_action_candidates() -- the order actions may still be read by the candidate
        # kernels (group streams, no synchronisation with candidates_on_device) when step() stages its own
        self._staging = {}
        for kind in ("step", "cands"):
            self._staging[kind] = dict(host=torch.empty((num_envs,), dtype=torch.int32, pin_memory=True),
                                       dev=torch.empty((num_envs,), dtype=torch.int32, device=self.device), busy=[])
        self._act_host, self._act_dev = self._staging["step"]["host"], self._staging["step"]["dev"]
        self._ring, self._loc_ring, self._turn, self._loc_turn = [], [], 0, 0
        for _ in range(max(0, int(obs_ring))):
            self._ring.append(self._new_buffer(self.obs_len, register=self.env.K == 1))
            if self.env.K > 1:
                self._loc_ring.append(self._new_buffer(self.env.loc_obs_len, register=True))

    def _envs_and_rows(self):
        if self.num_groups > 1:
            return [(e, self.env.rows(g)) for g, e in enumerate(self.env.groups)]
        return [(self.env, slice(0, self.num_envs))]

    def _new_buffer(self, width: int, register: bool) -> torch.Tensor:
        t = torch.zeros((self.num_envs, width), dtype=torch.float32, device=self.device)
        if register:
            for e, rows in self._envs_and_rows():
                e.register_obs_buffer(t[rows])
        return t

    def _next_obs(self) -> Optional[torch.Tensor]:
        if not self._ring:
            return None
        t = self._ring[self._turn]
        self._turn = (self._turn + 1) % len(self._ring)
        return t

    def _staging_idle(self, kind: str) -> dict:
        """The staging pair of `kind`, once nothing launched earlier still reads it: the previous H2D copy out of the
        pinned buffer and the kernels (on whatever streams) that consumed the device tensor have recorded events."""
        sg = self._staging[kind]
        for ev in sg["busy"]:
            ev.synchronize()                                                # complete long ago in a stepping loop
        sg["busy"] = []
        return sg

    def _staging_release(self, kind: str) -> None:
        """Call after the consumers of the staging pair have been launched: one event per stream that reads it."""
        streams = self.env.streams if self.num_groups > 1 else [torch.cuda.current_stream(self.device)]
        sg = self._staging[kind]
        for st in streams:
            ev = torch.cuda.Event()
            ev.record(st)
            sg["busy"].append(ev)

    def _actions_to_device(self, actions, kind: str = "step") -> torch.Tensor:
        """int32[N] on the device, in flight on the current stream.  Host actions go through the pinned staging buffer
        (one asynchronous H2D); a device tensor is converted in place of a copy when it already is int32."""
        if isinstance(actions, torch.Tensor) and actions.is_cuda:
            a = actions.reshape(-1)
            assert a.numel() == self.num_envs
            if a.dtype == torch.int32 and a.is_contiguous():
                return a
            sg = self._staging_idle(kind)
            sg["dev"].copy_(a)
            return sg["dev"]
        a = actions.reshape(-1) if isinstance(actions, torch.Tensor) else torch.from_numpy(np.asarray(actions).reshape(-1))
        assert a.numel() == self.num_envs
        sg = self._staging_idle(kind)
        sg["host"].copy_(a)                                                  # dtype conversion on the host
        sg["dev"].copy_(sg["host"], non_blocking=True)
        return sg["dev"]

    def _group_staging(self, group: int) -> dict:
        """Per-group pinned + device staging of step_async(group=g) (idle: step_wait(group=g) synchronised its stream)."""
        if not hasattr(self, "_gstaging"):
            self._gstaging = {}
        if group not in self._gstaging:
            per = self.env.per
            self._gstaging[group] = dict(host=torch.empty((per,), dtype=torch.int32, pin_memory=True),
                                         dev=torch.empty((per,), dtype=torch.int32, device=self.device))
        return self._gstaging[group]

    def reset(self) -> torch.Tensor:
        if self.waiting_step:
            self.step_wait()
        self.tstart = time.time()
        out = self._next_obs()
        if self.feeder is not None:
            self.feeder.tick()                                               # a reset draws a whole new queue per bin
        if out is None:
            return self.env.reset()
        if self.num_groups > 1:
            cur = torch.cuda.current_stream(self.device)
            for g, (e, st) in enumerate(zip(self.env.groups, self.env.streams)):
                st.wait_stream(cur)
                with torch.cuda.stream(st):
                    _lib.check(e.lib.irbpp_reset(e._h, _ptr(out[self.env.rows(g)]), e._stream()), "irbpp_reset")
                cur.wait_stream(st)
        else:
            e = self.env
            _lib.check(e.lib.irbpp_reset(e._h, _ptr(out), e._stream()), "irbpp_reset")
        return out

    def step_async(self, actions, group: Optional[int] = None) -> None:
        """All envs, or (grouped envs only) the envs of one group: ``actions`` then has that group's length."""
        if group is not None:
            if self._group_pending.get(group) is not None:
                raise RuntimeError("already running an async step")
            per = self.env.per
            if isinstance(actions, torch.Tensor) and actions.is_cuda:
                a = actions.reshape(-1).to(dtype=torch.int32)
            else:                                                           # this group's slice of the staging buffers
                a = actions.reshape(-1) if isinstance(actions, torch.Tensor) else torch.from_numpy(np.asarray(actions).reshape(-1))
                rows = self.env.rows(group)
                sg = self._group_staging(group)
                sg["host"].copy_(a)
                sg["dev"].copy_(sg["host"], non_blocking=True)
                a = sg["dev"]
            assert a.numel() == per
            # the group's stream waits for whatever produced `a` on the current stream, and the allocator learns that
            # `a` (possibly a temporary of the caller) is read there
            obs, _, _ = self.env.step_group(group, a)
            self._group_pending[group] = obs
            return                                                          # (step_wait(group) synchronises the group's stream)
        if self.waiting_step:
            raise RuntimeError("already running an async step")          # vec_env.py:7-16
        res = self.env.step(self._actions_to_device(actions), obs_out=self._next_obs())
        self._pending = res if isinstance(res, torch.Tensor) else res[0]
        self.waiting_step = True                                            # (step_wait synchronises: the staging pair is idle after it)

    def step_wait(self, group: Optional[int] = None):
        if group is not None:
            obs = self._group_pending.get(group)
            if obs is None:
                raise RuntimeError("not running an async step")
            with torch.cuda.stream(self.env.streams[group]):
                h = self.env.groups[group].step_info_host()
            self._group_pending[group] = None
            if self.feeder is not None:
                self.feeder.tick()          # (conservative: a caller may step one group more often than the others)
            reward = torch.from_numpy(h["reward"]).unsqueeze(dim=1).float()
            return obs, reward, h["done"], _Infos(h, round(time.time() - self.tstart, 6))
        if not self.waiting_step:
            raise RuntimeError("not running an async step")              # vec_env.py:19-27
        obs = self._pending
        # ONE pinned D2H copy per (group of) envs, error word included, and the step's only synchronisation.  It cannot
        # be deferred: the reward comes back as a CPU tensor (envs.py:164) that the trainer clips right away
        # (trainer.py:180-181), and it travels in the same copy as done / infos.
        h = self.env.step_info_host()
        self._pending = None
        self.waiting_step = False
        if self.feeder is not None:
            self.feeder.tick()
        reward = torch.from_numpy(h["reward"]).unsqueeze(dim=1).float()   # envs.py:164
        return obs, reward, h["done"], _Infos(h, round(time.time() - self.tstart, 6))

    def step(self, actions):
        self.step_async(actions)
        return self.step_wait()

    # False (default): get_action_candidates returns a host float32 array, which the unmodified trainer can wrap
    # with ``torch.from_numpy(np.array(..)).float().to(device)`` (trainer.py:267-268) -- at the price of a PCIe
    # round trip of N x 14 KB.  True: the device tensor itself (callers that accept tensors skip the trip).
    candidates_on_device = False

    def get_action_candidates(self, order_actions):
        """Location observations [N, 5S+9+Hc] of the chosen buffer slots (shmem_vec_env.py:99-102): a host float32
        array by default, the device tensor with ``candidates_on_device``."""
        out = None
        if self._loc_ring:
            out = self._loc_ring[self._loc_turn]
            self._loc_turn = (self._loc_turn + 1) % len(self._loc_ring)
        loc = self.env.get_action_candidates(self._actions_to_device(order_actions, "cands"), obs_out=out)
        self._staging_release("cands")
        if self.candidates_on_device:
            return loc
        if self.num_groups > 1:
            self.env.synchronize()
        return loc.cpu().numpy()

    def get_all_possible_observation(self):
        """PackingGame.get_all_possible_observation (binPhy.py:171-180) of every env: [N, k * (5S+9+Hc)] -- per env the
        concatenation of its k location observations -- as a host float32 array, or the device tensor with
        ``candidates_on_device``.  (The reference has no caller for it and ShmemVecEnv no command: this is the method a
        maintainer would forward.)"""
        loc = self.env.get_all_possible_observation()
        loc = loc.reshape(self.num_envs, -1)
        if self.candidates_on_device:
            return loc
        if self.num_groups > 1:
            self.env.synchronize()
        return loc.cpu().numpy()

    def reset_specific(self, indexs) -> torch.Tensor:
        """shmem_vec_env.py:113-117: reset the listed envs only; their observations in list order (a fresh tensor).
        A caller that copies these rows into the state tensor it holds (a ring buffer of this class) may do so: the
        ring's registrations are invalidated here, so the next observation through each buffer is written in full."""
        if not self.allow_early_resets:                                    # every env is mid-episode: auto-reset
            raise RuntimeError("Tried to reset an environment before done. If you want to allow early "
                               "resets, pass allow_early_resets=True")     # monitor.py:45-46
        if self.waiting_step:
            self.step_wait()
        idx = np.asarray(list(indexs), dtype=np.int32).reshape(-1)
        if len(np.unique(idx)) != len(idx) or (len(idx) and (idx.min() < 0 or idx.max() >= self.num_envs)):
            raise ValueError("reset_specific needs distinct env indices in [0, num_envs)")
        if self.feeder is not None:
            self.feeder.tick()
        obs = self.env.reset_bins(torch.from_numpy(idx).to(self.device))
        for e, _ in self._envs_and_rows():
            e.invalidate_obs_buffers()
        self.env.check_device_error()
        return obs

    def close(self):
        if not self.closed:
            self.env.close()
            self.closed = True

    def get_images(self):
        raise NotImplementedError("rendering belongs to the pybullet path, which is out of scope")


def make_vec_envs(args, log_dir=None, allow_early_resets=False):
    """Same triple as envs.make_vec_envs (envs.py:67-99): (envs, [obs_space, act_space], obs_len).

    ``args`` is the reference's namespace.  Shapes: ``args.shapes`` (a ShapeSet) or the reference's ``args.shotInfo`` /
    ``args.infoDict``.  Items, in this order:
      * ``args.sequences`` (pre-drawn item ids, int32[n_traj][L]) if present -- trajectories, as in evaluation;
      * ``args.evaluate`` with ``args.test_name``: the trajectories of ``test_sequence.pt`` (LoadItemCreator,
        binPhy.py:58-59);
      * otherwise the training-time creators of binPhy.py:60-67 (``args.dataSample`` over ``args.dicPath``), every
        environment on its own np.random stream seeded ``args.seed + rank`` exactly like envs.py:41: the items each
        environment sees are the ones the reference's worker of that rank would have drawn (itemgen.py).
    """
    shapes = getattr(args, "shapes", None)
    if shapes is None:
        shapes = shape_set_from_reference(args.shotInfo, args.infoDict)
    dev = args.device if isinstance(args.device, (str, torch.device)) else f"cuda:{int(args.device)}"
    kw = dict(resolutionA=args.resolutionA, resolutionH=args.resolutionH,
              resolutionZ=getattr(args, "resolutionZ", 0.01),
              bin_dimension=tuple(getattr(args, "bin_dimension", BIN_DIMENSION)),
              selectedAction=args.selectedAction, bufferSize=args.bufferSize,
              scale_z=float(getattr(args, "scale", [100, 100, 100])[2]))
    sequences, feeder = getattr(args, "sequences", None), None
    if sequences is None and getattr(args, "evaluate", False) and getattr(args, "test_name", None):
        seqs = torch.load(args.test_name, weights_only=False)
        length = max(len(t) for t in seqs)
        sequences = np.full((len(seqs), length), -1, dtype=np.int32)
        for i, t in enumerate(seqs):
            sequences[i, :len(t)] = [(-1 if v is None else int(v)) for v in t]
    if sequences is None:
        from . import itemgen
        feeder = itemgen.StreamFeeder(itemgen.streams_for_args(args, args.num_processes),
                                      ring_len=int(getattr(args, "item_ring", 4096)), buffer_size=args.bufferSize)
        sequences, kw["item_stream"] = feeder.initial, 1
    if getattr(args, "tuning", 0):           # (not a reference argument: irbpp_config::tuning, A/B runs and tests)
        kw["tuning"] = int(args.tuning)
    # (args.num_groups is not a reference argument: > 1 steps the envs as that many independent groups on their own HIP
    # streams, item streams included -- GroupedPackingEnv; 0 = the library's own choice for PIPELINED stepping, groups_for.
    # Default 1: the reference's trainer calls envs.step() for all environments and waits (trainer.py:165) -- one dependent
    # chain per step either way, and the second group's launches are host time on that chain: 4096 BlockOut environments
    # 21.3 M steps/s as one group, 18.6 M as two; 8192: 29.4 / 27.5; 2048: 12.9 / 10.7, profiles/r05/s35.  Groups pay for
    # callers that keep the device busy across steps: step_async(group=g) / step_wait(group=g) actor loops, device-resident
    # policies)
    # args.obs_ring (not a reference argument either): 0 (default here) = every call returns a fresh observation tensor, the
    # reference's behaviour; 3 = the ring of library-registered buffers GpuVecEnv uses by default (+10 % step rate; an
    # observation is overwritten three calls later, which the reference's trainer never notices)
    envs = GpuVecEnv(shapes, sequences, args.num_processes, device=dev, allow_early_resets=allow_early_resets,
                     feeder=feeder, num_groups=int(getattr(args, "num_groups", 1)), obs_ring=int(getattr(args, "obs_ring", 0)), **kw)
    return envs, [envs.observation_space, envs.action_space], envs.obs_len


def shape_set_from_reference(shotInfo, infoDict) -> ShapeSet:
    """Build a ShapeSet from the reference's in-memory containers (tools.py:227-279)."""
    ids = sorted(shotInfo.keys())
    assert ids == list(range(len(ids))), "shape ids must be 0..n-1"
    extents = np.array([[np.asarray(infoDict[k][r]["extents"], dtype=np.float64) for r in range(len(shotInfo[k]))]
                        for k in ids])
    volumes = np.array([float(infoDict[k][0]["volume"]) for k in ids])
    tables = [[tuple(np.asarray(a, dtype=np.float64) for a in shotInfo[k][r]) for r in range(len(shotInfo[k]))]
              for k in ids]
    return ShapeSet(extents, volumes, tables, name="reference")
