"""ctypes binding of libirbpp_hip.so (include/irbpp.h).  There is no CPU fallback: if the
library is missing or a symbol is absent, importing the binding raises."""
from __future__ import annotations

import ctypes as C
import os

from .build import LIB_PATH

c_i32_p = C.POINTER(C.c_int32)
c_f64_p = C.POINTER(C.c_double)


class IrbppConfig(C.Structure):
    _fields_ = [
        ("num_bins", C.c_int32), ("n_rot", C.c_int32), ("selected", C.c_int32), ("buffer_size", C.c_int32),
        ("resolution_a", C.c_double), ("resolution_h", C.c_double), ("resolution_z", C.c_double),
        ("bin", C.c_double * 3), ("scale_z", C.c_double),
        ("traj_start", C.c_int32), ("global_offset", C.c_int32), ("global_bins", C.c_int32),
        ("device", C.c_int32), ("stability", C.c_int32), ("tuning", C.c_int32), ("item_stream", C.c_int32),
    ]


TUNE_NO_BLOCK_PATH, TUNE_WIDE_KERNEL, TUNE_NARROW_KERNEL, TUNE_NO_BOX_PATH, TUNE_NO_ITEM_ORDER = 1, 2, 4, 8, 16
TUNE_TRACE_CPW64, TUNE_TRACE_CPW32, TUNE_TRACE_CPW16, TUNE_INLINE_POLYGON, TUNE_NO_HEAVY_FIRST = 32, 64, 128, 256, 512
TUNE_NO_SPECIALISED, TUNE_FUSED_APPLY, TUNE_SPLIT_APPLY, TUNE_BLOCK_EMIT, TUNE_WAVE_EMIT = 1024, 2048, 4096, 8192, 16384
TUNE_GRAPH, TUNE_WG512, TUNE_NO_WG512, TUNE_TRACE_REFILL, TUNE_NO_MIXED_PATH = 32768, 65536, 131072, 262144, 524288
TUNE_CHAIN, TUNE_WG128, TUNE_RECT = 1048576, 2097152, 4194304


class IrbppReplayView(C.Structure):
    _fields_ = [("states_dev", C.c_void_p), ("actions_dev", C.c_void_p), ("rewards_dev", C.c_void_p),
                ("nonterminals_dev", C.c_void_p), ("tree_dev", C.c_void_p), ("index_dev", C.c_void_p),
                ("full_dev", C.c_void_p), ("scaling_dev", C.c_void_p),
                ("n_env", C.c_int32), ("capacity", C.c_int32), ("obs_len", C.c_int32), ("n_step", C.c_int32)]


class IrbppReplayStore(C.Structure):
    _fields_ = [("states_dev", C.c_void_p), ("actions_dev", C.c_void_p), ("rewards_dev", C.c_void_p),
                ("nonterminals_dev", C.c_void_p), ("timesteps_dev", C.c_void_p), ("tree_dev", C.c_void_p),
                ("max_dev", C.c_void_p), ("index_dev", C.c_void_p), ("full_dev", C.c_void_p), ("t_dev", C.c_void_p),
                ("n_env", C.c_int32), ("capacity", C.c_int32), ("obs_len", C.c_int32)]


class IrbppStepOut(C.Structure):
    _fields_ = [("reward_dev", C.c_void_p), ("done_dev", C.c_void_p), ("counter_dev", C.c_void_p),
                ("ratio_dev", C.c_void_p), ("ep_reward_dev", C.c_void_p), ("ep_len_dev", C.c_void_p),
                ("stable_dev", C.c_void_p), ("err_dev", C.c_void_p)]


# every entry point include/irbpp.h declares: name -> (restype, argtypes)
SIGNATURES = {
    "irbpp_status_string": (C.c_char_p, [C.c_int]),
    "irbpp_version": (C.c_int, []),
    "irbpp_source_hash": (C.c_char_p, []),
    "irbpp_overlap_path": (C.c_int, [C.c_void_p]),
    "irbpp_create": (C.c_int, [C.POINTER(IrbppConfig), C.POINTER(C.c_void_p)]),
    "irbpp_destroy": (C.c_int, [C.c_void_p]),
    "irbpp_load_shapes": (C.c_int, [C.c_void_p, C.c_int32, c_f64_p, c_f64_p, c_i32_p, C.POINTER(C.c_int64),
                                    C.c_int64, c_f64_p, c_f64_p, c_f64_p, c_f64_p]),
    "irbpp_load_sequences": (C.c_int, [C.c_void_p, c_i32_p, C.c_int32, C.c_int32]),
    "irbpp_obs_len": (C.c_int, [C.c_void_p, C.c_int32]),
    "irbpp_reset": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "irbpp_reset_bins": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]),
    "irbpp_step": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(IrbppStepOut), C.c_void_p]),
    "irbpp_get_action_candidates": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "irbpp_get_all_possible_observation": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "irbpp_policy_minz": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]),
    "irbpp_set_auto_policy": (C.c_int, [C.c_void_p, C.c_void_p]),
    "irbpp_register_obs_buffer": (C.c_int, [C.c_void_p, C.c_void_p]),
    "irbpp_unregister_obs_buffer": (C.c_int, [C.c_void_p, C.c_void_p]),
    "irbpp_invalidate_obs_buffer": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "irbpp_stream_cursors": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]),
    "irbpp_stream_write": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]),
    "irbpp_itemgen_create": (C.c_int, [C.c_uint32, C.c_int32, c_i32_p, c_i32_p, C.c_int32, C.POINTER(C.c_void_p)]),
    "irbpp_itemgen_draw": (C.c_int, [C.c_void_p, C.c_int32, c_i32_p]),
    "irbpp_itemgen_destroy": (C.c_int, [C.c_void_p]),
    "irbpp_possible_position": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "irbpp_heuristic_action": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    "irbpp_shot_item": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_double, C.c_double,
                                  C.c_double, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "irbpp_convex_hull_actions": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "irbpp_get_heightmaps": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "irbpp_set_heightmaps": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "irbpp_episode_totals": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "irbpp_set_placement_log": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32]),
    "irbpp_sumtree_sample": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_uint64, C.c_int32,
                                       C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "irbpp_sumtree_find": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p,
                                     C.c_void_p, C.c_void_p]),
    "irbpp_sumtree_update": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32,
                                       C.c_void_p, C.c_void_p]),
    "irbpp_replay_gather": (C.c_int, [C.POINTER(IrbppReplayView), C.c_int32, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p,
                                      C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "irbpp_replay_append": (C.c_int, [C.POINTER(IrbppReplayStore), C.c_void_p, C.c_int64, C.c_void_p, C.c_int32, C.c_void_p,
                                      C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "irbpp_masked_argmax": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p,
                                      C.c_void_p]),
    "irbpp_debug_phase_cycles": (C.c_int, [C.c_void_p, C.c_void_p]),
    "irbpp_debug_kernel_info": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_char_p)]),
    "irbpp_debug_kernel_timing": (C.c_int, [C.c_void_p, C.c_int32]),
    "irbpp_debug_kernel_timing_every": (C.c_int, [C.c_void_p, C.c_int32]),
    "irbpp_debug_kernel_times": (C.c_int, [C.c_void_p, C.POINTER(C.c_float), C.c_int32, C.POINTER(C.c_int32)]),
    "irbpp_device_error": (C.c_int, [C.c_void_p, C.c_void_p, c_i32_p]),
}

_lib = None


def load(path: str = "") -> C.CDLL:
    """dlopen the HIP library and bind every declared symbol (raises if anything is missing).
    IRBPP_LIBRARY selects another build of the same sources (the tooling variant of build.py)."""
    global _lib
    if _lib is not None:
        return _lib
    explicit = bool(path) or bool(os.environ.get("IRBPP_LIBRARY"))     # a library the caller names is the caller's business
    path = path or os.environ.get("IRBPP_LIBRARY", "") or LIB_PATH
    if not os.path.exists(path):
        raise RuntimeError(
            f"{path} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). irbpp_amd has no CPU fallback.")
    # torch first: it ships its own HIP runtime (torch/lib/libamdhip64.so), and a process must hold exactly one.
    # Loaded before torch, this library would pull in /opt/rocm's copy and every later call would talk to a
    # runtime that knows nothing of torch's device context (irbpp_create then fails with a HIP error).
    import torch  # noqa: F401
    lib = C.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    # A binary built from other sources than the ones lying here is refused: every number and every test would be about
    # code nobody is looking at.  (IRBPP_LIBRARY names an A/B variant built with extra flags by tools/build_variant.sh: its
    # stamp is its own business.)
    if not explicit and not os.environ.get("IRBPP_ALLOW_STALE_LIBRARY"):
        from . import build
        have = lib.irbpp_source_hash().decode()
        try:
            want = build.source_hash()
        except OSError:                     # an installed copy without csrc/ next to it: nothing to compare the stamp with
            want = have
        if have != want:
            raise RuntimeError(
                f"{path} was built from sources {have}, the sources in {build.CSRC} hash to {want}: rebuild with "
                "`python -c 'import __graft_entry__ as g; g.build()'` (IRBPP_ALLOW_STALE_LIBRARY=1 overrides)")
    _lib = lib
    return lib


class IrbppError(RuntimeError):
    pass


DEVERR_BITS = {1: "LEVEL_RANGE", 2: "TRACE_GUARD", 4: "BAD_ITEM", 8: "BAD_BIN", 16: "CAPACITY", 32: "STREAM_DRY", 64: "BAD_ACTION"}


def deverr_names(flags: int) -> str:
    """IRBPP_DEVERR_* names of a device error word (include/irbpp.h)."""
    return "|".join(name for bit, name in DEVERR_BITS.items() if flags & bit) or "0"


def check(status: int, what: str = "") -> None:
    if status != 0:
        msg = load().irbpp_status_string(status).decode()
        raise IrbppError(f"{what}: {msg} (status {status})")
