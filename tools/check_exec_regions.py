#!/usr/bin/env python3
"""Build-time check of the hand-written exec masking of march_p2_kernel (csrc/vr_p2.h: p2_request).

The idle lanes of a packet are switched off for the eight corner loads by hand -- `s_and_b64 exec, exec, <keep>` ... the loads
... `s_mov_b64 exec, <saved>` in inline asm, with scheduler fences on both sides.  Scheduler fences do not fence the register
allocator: a copy, a spill or a rematerialised vector instruction placed between the two exec writes would run with the idle
lanes off and silently corrupt what the live lanes read later.  This script disassembles the gfx950 code objects inside
libvr_hip.so and asserts that, in every march_p2_kernel, nothing but buffer loads (and the s_nop hazard padding between
them) sits between an `s_and_b64 exec, exec, sN` and the next write of exec.

    python tools/check_exec_regions.py [path/to/libvr_hip.so]      exit status 0 = clean
"""
import os
import re
import struct
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def code_objects(path):
    """The gfx950 device ELFs of every clang offload bundle in the file."""
    blob = open(path, "rb").read()
    out, pos = [], 0
    while True:
        i = blob.find(MAGIC, pos)
        if i < 0:
            break
        n = struct.unpack_from("<Q", blob, i + len(MAGIC))[0]
        p = i + len(MAGIC) + 8
        for _ in range(n):
            off, size, tlen = struct.unpack_from("<QQQ", blob, p)
            triple = blob[p + 24:p + 24 + tlen].decode(errors="replace")
            p += 24 + tlen
            if "gfx950" in triple and size:
                out.append(blob[i + off:i + off + size])
        pos = i + len(MAGIC)
    return out


def disassemble(elf_bytes):
    with tempfile.NamedTemporaryFile(suffix=".co", delete=False) as f:
        f.write(elf_bytes)
        name = f.name
    try:
        return subprocess.run([OBJDUMP, "-d", "--mcpu=gfx950", name], check=True, capture_output=True, text=True).stdout
    finally:
        os.unlink(name)


def check(text, kernel_pat="march_p2_kernel"):
    """Returns (regions checked, list of offending (kernel, instruction))."""
    regions, bad = 0, []
    kernel, inside = None, False
    for line in text.split("\n"):
        m = re.match(r"^[0-9a-f]+ <([^>]+)>:", line)
        if m:
            kernel = m.group(1) if kernel_pat in m.group(1) else None
            inside = False
            continue
        if kernel is None:
            continue
        ins = line.strip().split("//")[0].strip()
        if not ins:
            continue
        op = ins.split()[0]
        if not inside:
            if re.match(r"s_and_b64\s+exec,\s*exec,\s*s\[", ins):
                inside = True
                regions += 1
            continue
        if re.match(r"s_mov_b64\s+exec,", ins):
            inside = False
        elif op.startswith("buffer_load_dword") or op == "s_nop":
            pass
        else:
            bad.append((kernel, ins))
            if "exec" in ins:  # (some other exec write ended the region)
                inside = False
    return regions, bad


def main():
    path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "volumerendering_amd", "libvr_hip.so")
    objs = code_objects(path)
    if not objs:
        print("no gfx950 code object found in", path)
        return 2
    total, bad = 0, []
    for o in objs:
        r, b = check(disassemble(o))
        total += r
        bad += b
    print(f"{len(objs)} code object(s), {total} hand-masked load regions in march_p2_kernel, {len(bad)} foreign instruction(s)")
    for k, ins in bad[:20]:
        print("  ", k[:80], "::", ins)
    return 1 if bad or total == 0 else 0


if __name__ == "__main__":
    sys.exit(main())
