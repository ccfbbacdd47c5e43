"""irbpp_amd -- MI355X-native batched step() for the IR-BPP packing environment.

Only the hot path of alexfrom0815/IR-BPP lives here: the
per-bin overlap test, candidate generation, heightmap update, reward/termination
and observation assembly of ``environment/physics0``, run for thousands of bins
at once by hand-written HIP kernels (``csrc/``) behind a C ABI (``include/irbpp.h``),
and re-exposed through the reference's VecEnv surface (``vec_env.GpuVecEnv``).
"""
import os

from .shapes import ShapeSet  # noqa: F401
from . import synthetic  # noqa: F401

__all__ = ["ShapeSet", "synthetic", "use_hardware_queues"]


def use_hardware_queues(n: int = 8) -> None:
    """For callers that step groups of bins on several HIP streams (vec_env.GroupedPackingEnv / GpuVecEnv(num_groups > 1)):
    ask the ROCm runtime for ``n`` hardware queues instead of its default of four, with which the group streams share
    queues (four groups of a 4096-bin BlockOut environment: 27 M steps/s with four queues, 35 - 38 M with eight;
    profiles/r04/s42).  The runtime reads ``GPU_MAX_HW_QUEUES`` when the process initialises HIP, so this has to be
    called before the first use of the GPU (a value already in the environment is left alone); it raises if that is over."""
    import torch
    if torch.cuda.is_initialized():
        raise RuntimeError("use_hardware_queues() has to be called before the process initialises HIP (first torch.cuda use)")
    os.environ.setdefault("GPU_MAX_HW_QUEUES", str(int(n)))
