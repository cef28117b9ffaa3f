"""Test helper: the gfx950 code objects inside the shipped libsimt_hip.so, disassembled.  The library's .hip_fatbin section is a sequence
of clang offload bundles (one per translation unit); each bundle lists (offset, size, target triple) entries.  No GPU needed."""
import os
import re
import struct
import subprocess
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def code_objects(lib_path):
    with tempfile.TemporaryDirectory() as td:
        fat = os.path.join(td, "fat.bin")
        subprocess.check_call([os.path.join(LLVM, "llvm-objcopy"), "--dump-section", f".hip_fatbin={fat}", lib_path, os.path.join(td, "x")],
                              stderr=subprocess.DEVNULL)
        blob = open(fat, "rb").read()
    out, pos = [], 0
    while True:
        pos = blob.find(MAGIC, pos)
        if pos < 0:
            break
        p = pos + len(MAGIC)
        (n,) = struct.unpack_from("<Q", blob, p)
        p += 8
        for _ in range(n):
            off, size, tlen = struct.unpack_from("<QQQ", blob, p)
            p += 24
            triple = blob[p:p + tlen].decode()
            p += tlen
            if "gfx950" in triple and size:
                out.append(blob[pos + off:pos + off + size])
        pos = p
    return out


def kernels_with_scratch(lib_path):
    """-> {demangled kernel name: number of scratch_* instructions} over every gfx950 kernel of the library."""
    res = {}
    with tempfile.TemporaryDirectory() as td:
        for i, co in enumerate(code_objects(lib_path)):
            f = os.path.join(td, f"co{i}.o")
            open(f, "wb").write(co)
            dis = subprocess.check_output([os.path.join(LLVM, "llvm-objdump"), "-d", "--demangle", f]).decode(errors="replace")
            cur = None
            for ln in dis.split("\n"):
                m = re.match(r"^[0-9a-f]+ <(.*)>:$", ln)
                if m:
                    cur = m.group(1)
                    res.setdefault(cur, 0)
                elif cur and "scratch_" in ln:
                    res[cur] += 1
    return res
