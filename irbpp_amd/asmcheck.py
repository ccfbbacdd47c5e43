#!/usr/bin/env python
"""Build-time check of the SHIPPED code object: no instruction touches the destination registers of a scalar load
before a `s_waitcnt lgkmcnt(0)` has been executed on every path from the load.

Why: the generic overlap loop issues its `s_load_dwordx16` from one inline-asm statement and waits for it in another
(csrc/irbpp_kernels.hip, gcell_request / gcell_await), so that the load's latency runs beside the trip's LDS and VALU
work.  Between the two statements the compiler believes the 16-SGPR tuple is ready: a copy, split or spill of it there
(register pressure under the 96-SGPR cap) would read registers the load has not written yet -- silently wrong overlap
heights.  Scalar loads return out of order, so only lgkmcnt(0) proves arrival.  The rule is checked for EVERY scalar
load of EVERY kernel in the library (compiler-issued ones satisfy it by construction; ours must too), on the
disassembly of the .so that ships, so it holds for whatever hipcc built it.

    python -m irbpp_amd.asmcheck [path/to/libirbpp_hip.so]      exit code 0 = clean

irbpp_amd/build.py runs this after every build and refuses a library that fails; tests/test_kernel_asm.py runs it on
the in-tree library.
"""
import os
import re
import shutil
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
PKG_DIR = os.path.dirname(os.path.abspath(__file__))


def _tool(name):
    for cand in (os.path.join(LLVM, name), shutil.which(name)):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError(f"{name} not found")


def scratch_sizes(so_path):
    """{kernel: bytes of scratch (private segment) per lane} from the code object's metadata notes."""
    with tempfile.TemporaryDirectory() as tmp:
        fat, co = os.path.join(tmp, "fat.bin"), os.path.join(tmp, "dev.co")
        subprocess.run([_tool("llvm-objcopy"), "--dump-section", f".hip_fatbin={fat}", so_path, os.path.join(tmp, "copy.so")],
                       check=True, capture_output=True)
        subprocess.run([_tool("clang-offload-bundler"), "--unbundle", "--type=o", f"--input={fat}",
                        "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"], check=True, capture_output=True)
        notes = subprocess.run([_tool("llvm-readelf"), "--notes", co], check=True, capture_output=True, text=True).stdout
    out, name = {}, None
    for line in notes.splitlines():
        line = line.strip()
        if line.startswith(".name:"):
            name = line.split(":", 1)[1].strip()
        elif line.startswith(".private_segment_fixed_size:") and name is not None:
            out[name] = int(line.split(":", 1)[1])
        elif line.startswith("- .") or line.startswith("- .agpr_count"):
            pass
    return out


def disassemble(so_path):
    """llvm-objdump -d of the gfx950 code object bundled in the library's .hip_fatbin section."""
    with tempfile.TemporaryDirectory() as tmp:
        fat, co = os.path.join(tmp, "fat.bin"), os.path.join(tmp, "dev.co")
        subprocess.run([_tool("llvm-objcopy"), "--dump-section", f".hip_fatbin={fat}", so_path, os.path.join(tmp, "copy.so")],
                       check=True, capture_output=True)
        subprocess.run([_tool("clang-offload-bundler"), "--unbundle", "--type=o", f"--input={fat}",
                        "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"], check=True, capture_output=True)
        return subprocess.run([_tool("llvm-objdump"), "-d", co], check=True, capture_output=True, text=True).stdout


FUNC_RE = re.compile(r"^([0-9a-f]+) <([^>]+)>:")
INS_RE = re.compile(r"^\s+(\S+)\s*(.*?)\s*//\s*([0-9A-Fa-f]+):\s*[0-9A-Fa-f ]+?(?:<([^>+]+)\+0x([0-9a-fA-F]+)>|<([^>+]+)>)?\s*$")
SREG_RE = re.compile(r"\bs\[(\d+):(\d+)\]|\bs(\d+)\b")


def parse(text):
    """-> {function: [(addr, mnemonic, operands, branch target addr or None)]}"""
    funcs, cur, base = {}, None, {}
    for line in text.splitlines():
        m = FUNC_RE.match(line)
        if m:
            cur = m.group(2)
            funcs[cur] = []
            base[cur] = int(m.group(1), 16)
            continue
        if cur is None:
            continue
        m = INS_RE.match(line)
        if not m:
            continue
        mnem, ops, addr = m.group(1), m.group(2), int(m.group(3), 16)
        target = None
        if mnem.startswith(("s_cbranch", "s_branch")):
            if m.group(4) is not None:
                target = base.get(m.group(4), base[cur]) + int(m.group(5), 16)
            elif m.group(6) is not None:
                target = base.get(m.group(6), base[cur])
        funcs[cur].append((addr, mnem, ops, target))
    return funcs


def sregs(ops):
    out = set()
    for m in SREG_RE.finditer(ops):
        if m.group(3) is not None:
            out.add(int(m.group(3)))
        else:
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
    return out


def check_function(name, ins):
    index = {a: i for i, (a, _, _, _) in enumerate(ins)}
    problems, n_loads = [], 0
    for i, (addr, mnem, ops, _) in enumerate(ins):
        if not (mnem.startswith("s_load_dword") or mnem.startswith("s_buffer_load_dword")):
            continue
        n_loads += 1
        dest = sregs(ops.split(",")[0])
        seen, stack = set(), [i + 1]
        while stack:
            j = stack.pop()
            while j < len(ins) and j not in seen:
                seen.add(j)
                a, mn, op, tgt = ins[j]
                if mn == "s_waitcnt" and "lgkmcnt(0)" in op:
                    break
                if mn == "s_endpgm":
                    break
                if sregs(op) & dest:
                    problems.append(f"{name}: {mnem} {ops} at {addr:#x}: destination touched before lgkmcnt(0) by "
                                    f"`{mn} {op}` at {a:#x}")
                    stack = []
                    break
                if mn.startswith("s_cbranch") or mn == "s_branch":
                    if tgt is None or tgt not in index:
                        problems.append(f"{name}: branch at {a:#x} with an unresolved target while {mnem} at {addr:#x} is in flight")
                        stack = []
                        break
                    stack.append(index[tgt])
                    if mn == "s_branch":
                        break
                j += 1
    return n_loads, problems


def check_library(so_path):
    funcs = parse(disassemble(so_path))
    total, problems, wide = 0, [], {}
    for name, ins in funcs.items():
        n, p = check_function(name, ins)
        total += n
        problems += p
        wide[name] = sum(1 for _, m, o, _ in ins if m == "s_load_dwordx16" and o.rstrip().endswith("0x0"))
    return {"functions": len(funcs), "scalar_loads": total, "problems": problems, "x16_loads": wide}


def main():
    so = sys.argv[1] if len(sys.argv) > 1 else os.path.join(PKG_DIR, "libirbpp_hip.so")
    res = check_library(so)
    for p in res["problems"]:
        print("HAZARD", p)
    generic = {k: v for k, v in res["x16_loads"].items() if v and "env_kernel" in k}
    print(f"{so}: {res['functions']} kernels, {res['scalar_loads']} scalar loads checked, "
          f"{len(res['problems'])} hazards; pipelined s_load_dwordx16 per transition build: {generic}")
    return 1 if res["problems"] else 0


if __name__ == "__main__":
    sys.exit(main())
