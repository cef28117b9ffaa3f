"""CPU-only checks of the host side: the C-ABI library loads and exports every symbol include/simt_hip.h declares,
ctypes struct layouts match the header (sizes via a tiny C probe compiled with gcc), plan geometry, optimiser listing.
No compute call is made (no GPU here)."""
import ctypes as C
import os
import re
import subprocess
import sys
import tempfile

import pytest
import torch

from oracle import simt_oracle as so

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_functions():
    src = open(os.path.join(ROOT, "include", "simt_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(simt_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    import __graft_entry__ as ge
    from simt_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        ge.build()
    lib = _lib.load()
    names = _header_functions()
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/simt_hip.h but not exported"
        assert n in _lib.SIGNATURES, f"{n} has no ctypes signature"
    assert lib.simt_abi_version() == 1


def test_ctypes_struct_sizes_match_header():
    from simt_amd import _lib
    structs = {"simt_conv_desc": _lib.ConvDesc, "simt_wgrad_desc": _lib.WgradDesc, "simt_bn_bwd_desc": _lib.BnBwdDesc,
               "simt_head_desc": _lib.HeadDesc, "simt_ntm_inner_desc": _lib.NtmInnerDesc,
               "simt_ntm_post_desc": _lib.NtmPostDesc, "simt_sgd_desc": _lib.SgdDesc, "simt_tap_desc": _lib.TapDesc}
    prog = '#include <stdio.h>\n#include "simt_hip.h"\nint main(){' + "".join(
        f'printf("{n} %zu\\n", sizeof({n}));' for n in structs) + "return 0;}"
    with tempfile.TemporaryDirectory() as td:
        cpath, exe = os.path.join(td, "p.c"), os.path.join(td, "p")
        open(cpath, "w").write(prog)
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), cpath, "-o", exe])
        out = subprocess.check_output([exe]).decode().split()
    sizes = dict(zip(out[::2], map(int, out[1::2])))
    for n, cls in structs.items():
        assert C.sizeof(cls) == sizes[n], f"{n}: ctypes {C.sizeof(cls)} vs C {sizes[n]}"


def test_missing_library_fails_loudly(monkeypatch):
    from simt_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libsimt_hip.so")
    with pytest.raises(_lib.SimtHipError):
        _lib.load()


def test_ops_refuse_cpu_tensors():
    from simt_amd import ops
    with pytest.raises(AssertionError):
        ops.softmax_rows(torch.zeros(4, 32), 32, torch.zeros(4, 32), 32, 4, 19)


def test_trunk_geometry_matches_torch():
    import torch.nn.functional as F
    from simt_amd.engine import trunk_geometry
    for H, W in [(65, 65), (97, 129), (512, 512), (768, 768), (512, 1024), (33, 47)]:
        x = torch.zeros(1, 1, H, W)
        y0 = F.conv2d(x, torch.zeros(1, 1, 7, 7), stride=2, padding=3)
        yp = F.max_pool2d(y0, 3, 2, 1, ceil_mode=True)
        y2 = F.conv2d(yp, torch.zeros(1, 1, 1, 1), stride=2)
        g = trunk_geometry(H, W)
        assert g[0] == tuple(y0.shape[2:]) and g[1] == tuple(yp.shape[2:]) and g[2] == tuple(y2.shape[2:])


def test_optim_listing_matches_reference_multiplicity():
    from simt_amd.step import optim_listing
    names = [k for k in so.state_shapes(19, 3, True) if k.endswith("weight") or k.endswith("bias")]
    g0, g1 = optim_listing(names)
    og0, og1 = so.optim_param_names(so.state_shapes(19, 3, True))
    for n, m in g0.items():
        assert og0.count(n) == m, n
    assert set(g0) == set(og0) and set(g1) == set(og1)


def test_lr_poly_matches_golden():
    import numpy as np
    from simt_amd.step import lr_poly
    e = np.load(os.path.join(ROOT, "tests", "golden", "g10_lr_poly.npz"))
    for i, lr in zip(e["it"], e["lr"]):
        assert lr_poly(6e-4, int(i), 250000, 0.9) == lr
