#!/usr/bin/env python
"""Tooling: static instruction counts of one kernel by source line.  Builds the library with -gline-tables-only (no effect
on the generated code), disassembles the gfx950 code object with `llvm-objdump -d -l` and counts vector (v_*) and scalar
(s_*, without waits / nops / branches) instructions per file:line; prints the fattest lines with their source text.

    python tools/kernel_line_profile.py [kernel name, default irbpp_env_kernel] [lines to print, default 60] [-DFLAG ...]

Static counts are not dynamic ones -- a loop body counts once, a branch not taken counts too -- but a short loop that the
compiler has blown up shows at once: round 4's session 40 started here (fifteen workgroup-strided loops of one or two trips,
each unrolled sixteen-fold: ~100 vector instructions per loop; `#pragma unroll 1` took the transition kernel's block and
box builds from 63 VGPRs and 12 bytes of scratch to 58 and none)."""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("-D")]
    flags = [a for a in sys.argv[1:] if a.startswith("-D")]
    kernel = args[0] if args else "irbpp_env_kernel"
    top = int(args[1]) if len(args) > 1 else 60
    with tempfile.TemporaryDirectory() as tmp:
        so, fat, co = (os.path.join(tmp, n) for n in ("lib.so", "fat.bin", "dev.co"))
        subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-honor-nans", "-shared", "-fPIC",
                        "-Wno-unused-value", "-gline-tables-only", *flags, os.path.join(ROOT, "irbpp_amd/csrc/irbpp_capi.hip"), "-o", so],
                       check=True, capture_output=True)
        subprocess.run([f"{LLVM}/llvm-objcopy", "--dump-section", f".hip_fatbin={fat}", so, os.path.join(tmp, "stripped.so")], check=True, capture_output=True)
        subprocess.run([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={fat}", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                        f"--output={co}"], check=True, capture_output=True)
        text = subprocess.run([f"{LLVM}/llvm-objdump", "-d", "-l", co], check=True, capture_output=True, text=True).stdout.splitlines()
    start = next(i for i, l in enumerate(text) if l.endswith(f"<{kernel}>:"))
    end = next((i for i in range(start + 1, len(text)) if re.match(r"^[0-9a-f]+ <\w+>:$", text[i])), len(text))
    valu, salu, cur = collections.Counter(), collections.Counter(), None
    for l in text[start:end]:
        m = re.match(r"^; (/[^:]+):(\d+)", l)
        if m:
            cur = (m.group(1), int(m.group(2)))
            continue
        t = l.strip()
        if t.startswith("v_"):
            valu[cur] += 1
        elif t.startswith("s_") and not t.startswith(("s_waitcnt", "s_nop", "s_endpgm", "s_branch", "s_cbranch")):
            salu[cur] += 1
    sources = {}

    def line_of(key):
        if key is None:
            return ""
        f, n = key
        if f not in sources:
            try:
                sources[f] = open(f).read().splitlines()
            except OSError:
                sources[f] = []
        return sources[f][n - 1].strip()[:110] if 0 < n <= len(sources[f]) else ""
    print(f"{kernel}: {sum(valu.values())} vector and {sum(salu.values())} scalar instructions (static)")
    print("vector scalar  file:line  source")
    for key, c in sorted(valu.items(), key=lambda kv: -kv[1])[:top]:
        where = f"{os.path.basename(key[0])}:{key[1]}" if key else "?"
        print(f"{c:6d} {salu.get(key, 0):6d}  {where:28s} {line_of(key)}")


if __name__ == "__main__":
    main()
