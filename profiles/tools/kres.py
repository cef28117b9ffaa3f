"""kres.py <lib.so> [name filter]: LDS / VGPR / SGPR / scratch of every gfx950 kernel in a built library (code-object metadata; no GPU).
What may share a CU with what: a SIMD has 512 VGPRs (granule 8 per wave), a CU 160 KB of LDS."""
import os
import re
import subprocess
import sys
import tempfile

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
from _codeobj import code_objects, LLVM      # noqa: E402


def kernel_resources(lib):
    out = {}
    with tempfile.TemporaryDirectory() as td:
        for i, co in enumerate(code_objects(lib)):
            f = os.path.join(td, f"co{i}.o")
            open(f, "wb").write(co)
            notes = subprocess.check_output([LLVM + "/llvm-readelf", "--notes", f]).decode(errors="replace")
            for b in notes.split("- .agpr_count")[1:]:
                g = lambda k: (re.search(r"\." + k + r":\s+(\S+)", b) or [None, "?"])[1]      # noqa: E731
                name = subprocess.check_output(["c++filt", g("symbol").replace(".kd", "")]).decode().strip()
                ag = re.match(r":\s+(\d+)", b)
                out[name] = dict(lds=int(g("group_segment_fixed_size")), vgpr=int(g("vgpr_count")), agpr=int(ag.group(1)) if ag else 0,
                                 sgpr=int(g("sgpr_count")), wg=int(g("max_flat_workgroup_size")), scratch=int(g("private_segment_fixed_size")))
    return out


if __name__ == "__main__":
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    for n, r in kernel_resources(sys.argv[1]).items():
        if flt in n:
            print(f"{n[:100]:100s} lds={r['lds']:6d} vgpr={r['vgpr']:4d} agpr={r['agpr']:3d} sgpr={r['sgpr']:4d} wg={r['wg']:5d} scratch={r['scratch']}")
