"""irbpp_amd -- MI355X-native batched step() for the IR-BPP packing environment.

Only the hot path of alexfrom0815/IR-BPP lives here: the
per-bin overlap test, candidate generation, heightmap update, reward/termination
and observation assembly of ``environment/physics0``, run for thousands of bins
at once by hand-written HIP kernels (``csrc/``) behind a C ABI (``include/irbpp.h``),
and re-exposed through the reference's VecEnv surface (``vec_env.GpuVecEnv``).
"""
from .shapes import ShapeSet  # noqa: F401
from . import synthetic  # noqa: F401

__all__ = ["ShapeSet", "synthetic"]
