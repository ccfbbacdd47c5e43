"""Build libirbpp_hip.so in-tree with hipcc for gfx950 (no JIT cache: the .so travels with the repo)."""
from __future__ import annotations

import os
import shutil
import subprocess

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG_DIR, "csrc")
LIB_PATH = os.path.join(PKG_DIR, "libirbpp_hip.so")
SOURCES = ["irbpp_capi.hip", "irbpp_kernels.hip", "irbpp_wide.hip", "irbpp_replay.hip", "irbpp_device.h", "contours_device.h", "irbpp_itemgen.h",
           os.path.join("..", "..", "include", "irbpp.h")]
# -ffp-contract=off: the float64 results must equal numpy's, so no FMA contraction anywhere
# -fno-honor-nans: no NaN ever enters the path, so fmax needs no canonicalising v_max(x,x) per use
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-honor-nans",
               "-shared", "-fPIC", "-Wno-unused-value"]


def source_hash() -> str:
    """Short hash of everything the library is compiled from (+ flags): profiles/ stamps its counter passes
    with it and bench.py only quotes a pass whose stamp matches the sources it runs."""
    import hashlib
    h = hashlib.sha256(" ".join(HIPCC_FLAGS).encode())
    for name in sorted(SOURCES):
        with open(os.path.join(CSRC, name), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def _hipcc() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: the HIP library cannot be built on this machine")


def built_hash(path: str = LIB_PATH) -> str:
    """The stamp inside a built library (irbpp_source_hash), read without loading it into this process: the string
    follows the marker the C side puts in front of it."""
    try:
        with open(path, "rb") as f:
            blob = f.read()
    except OSError:
        return ""
    i = blob.find(b"irbpp-source-hash:")
    return blob[i + 18:i + 34].decode("ascii", "replace") if i >= 0 else ""


def needs_build(path: str = LIB_PATH) -> bool:
    """True unless `path` carries the stamp of the present sources (not a question of file times: a checkout, a copy to
    the GPU box or an editor can leave any order of mtimes behind)."""
    return built_hash(path) != source_hash()


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile csrc/irbpp_capi.hip (which includes the kernels) into libirbpp_hip.so.  (A/B builds under another
    name: tools/build_variant.sh, selected with IRBPP_LIBRARY.)"""
    out = LIB_PATH
    if not force and not needs_build(out):
        return out
    cmd = [_hipcc()] + HIPCC_FLAGS + [f'-DIRBPP_SOURCE_HASH="{source_hash()}"', os.path.join(CSRC, "irbpp_capi.hip"), "-o", out]
    if verbose:
        print(" ".join(cmd))
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError("hipcc failed:\n" + res.stdout + res.stderr)
    # The generic overlap loop waits for its pipelined scalar loads in a second asm statement: refuse a library in
    # which the compiler put a use of the loaded registers in between (asmcheck.py; disassembly of what was just built)
    from . import asmcheck
    try:
        chk = asmcheck.check_library(out)
    except (RuntimeError, OSError, subprocess.CalledProcessError) as e:      # no llvm-objdump & co on this machine: the CPU suite's
        if verbose:                                                        # test_kernel_asm.py is where the check is mandatory
            print(f"asmcheck skipped: {e}")
        return out
    if verbose:
        print(f"asmcheck: {chk['scalar_loads']} scalar loads in {chk['functions']} kernels, {len(chk['problems'])} hazards")
    if chk["problems"]:
        os.replace(out, out + ".rejected")
        raise RuntimeError("scalar-load hazard in the built library (kept as " + out + ".rejected):\n" + "\n".join(chk["problems"]))
    return out
