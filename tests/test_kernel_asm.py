"""The shipped code object obeys the scalar-load rule the generic overlap loop relies on (irbpp_amd/asmcheck.py):
no use of a scalar load's destination before lgkmcnt(0) -- checked on the disassembly of the in-tree library, so it
holds for the hipcc that built it; plus a self-test of the checker on an injected hazard."""
import os

import pytest

import irbpp_amd  # noqa: F401
from irbpp_amd import asmcheck, build


@pytest.fixture(scope="module")
def disassembly():
    path = build.build(force=False)
    assert os.path.exists(path)
    return asmcheck.disassemble(path)


def _problems(text):
    out = []
    for name, ins in asmcheck.parse(text).items():
        out += asmcheck.check_function(name, ins)[1]
    return out


def test_no_use_of_scalar_load_destinations_before_the_wait(disassembly):
    funcs = asmcheck.parse(disassembly)
    for kernel in ("irbpp_env_kernel", "irbpp_env_kernel_generic8", "irbpp_env_kernel_generic", "irbpp_env_kernel_wide",
                   "irbpp_trace_kernel", "irbpp_polygon_kernel", "irbpp_emit_kernel", "irbpp_heuristic_kernel"):
        assert kernel in funcs and len(funcs[kernel]) > 100
    # the pipelined 64-byte list loads are where they are expected (three walk depths x two address forms x three quads)
    for kernel in ("irbpp_env_kernel_generic8", "irbpp_env_kernel_generic", "irbpp_env_kernel_wide", "irbpp_env_kernel_s3", "irbpp_env_kernel_s4",
                   "irbpp_env_kernel_generic_w512", "irbpp_env_kernel_s4_w512", "irbpp_env_kernel_s4_w512c"):
        n16 = sum(1 for _, m, o, _ in funcs[kernel] if m == "s_load_dwordx16" and o.rstrip().endswith("0x0"))
        assert n16 >= 6, (kernel, n16)
    assert _problems(disassembly) == []


def test_checker_sees_an_injected_hazard(disassembly):
    out, done = [], 0
    for line in disassembly.splitlines():
        out.append(line)
        if done == 0 and "s_load_dwordx16 s[" in line and "0x0 " in line:
            first = int(line.split("s[")[1].split(":")[0])
            out.append(f"\ts_mov_b32 s100, s{first + 3}                                   // 000000000004: BE800000")
            done = 1
    assert done == 1
    probs = _problems("\n".join(out))
    assert len(probs) == 1 and "destination touched before lgkmcnt(0)" in probs[0]


def test_scratch_of_the_step_kernels_stays_where_it_was_measured():
    """Bytes of scratch per lane, from the code object's metadata.  The 64-VGPR / 96-SGPR cap of the eight-wave builds makes
    the register allocation touchy (round 4: one variant of the emit kernel spilled four VGPRs and cost the BlockOut step
    1.1 us): a build that spills more than the state the profiles were taken on fails here instead of shipping."""
    sizes = asmcheck.scratch_sizes(build.build(force=False))
    limits = {"irbpp_emit_kernel": 0, "irbpp_trace_kernel": 0, "irbpp_trace_kernel_c32": 0, "irbpp_trace_kernel_c16": 0,
              "irbpp_polygon_kernel": 0, "irbpp_env_kernel": 12, "irbpp_env_kernel_box8": 0, "irbpp_env_kernel_generic8": 48, "irbpp_apply_kernel": 0,
              "irbpp_env_kernel_generic": 0, "irbpp_env_kernel_box": 0,
              # round 5: the builds BASELINE.json's geometries run (sizes as compile-time constants, irbpp_device.h SPEC_KEYS);
              "irbpp_env_kernel_s1": 0, "irbpp_env_kernel_s2": 0, "irbpp_env_kernel_s3": 16, "irbpp_env_kernel_s4": 0,
              "irbpp_emit_kernel_s1": 0, "irbpp_emit_kernel_s2": 0, "irbpp_emit_kernel_s3": 0, "irbpp_emit_kernel_s4": 0,
              # wave-per-bin emit kernel: its ordinary path (a wave's own bin) touches no scratch; the spills sit in the path behind
              # its early return (bins that need the workgroup: more than S candidates), around the loop over those bins
              "irbpp_emit_wave_kernel": 216, "irbpp_emit_wave_kernel_s1": 152, "irbpp_emit_wave_kernel_s2": 108,
              "irbpp_env_kernel_generic_w512": 0, "irbpp_env_kernel_s4_w512": 0, "irbpp_env_kernel_s4_w512c": 16,
              "irbpp_heuristic_kernel": 32,          # (784 until round 5 session 37: the recursion of numpy's pairwise sum)
              "irbpp_env_kernel_wide": 36}           # (_wide: the A/B build that decides the overlap path at run time; not a default)
    for kernel, limit in limits.items():
        assert kernel in sizes, kernel
        assert sizes[kernel] <= limit, (kernel, sizes[kernel], limit)
