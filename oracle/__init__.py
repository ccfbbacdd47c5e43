"""CPU oracle for the IR-BPP packing-environment hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is product code: it may be
imported by ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg
of ``bench.py`` -- as the checker, never as the thing measured or shipped.  The
product (``irbpp_amd/``) never imports it and fails loudly without its HIP
library.

It restates, line by line, the reference's no-physics branch of
``environment/physics0`` (file:line citations in each module).  Pinning status:
the Python glue is checked against the reference's own modules executed in the
build container (tests/golden/make_golden.py -> tests/golden/*.npz); the two
OpenCV calls and the pybullet/trimesh kinematics have no golden data anywhere
and are "parity unpinned" (see oracle/contours.py, oracle/packing.py).
"""
