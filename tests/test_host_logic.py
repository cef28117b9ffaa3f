"""CPU-only checks of the host side: the C-ABI library loads and exports every symbol include/simt_hip.h declares,
ctypes struct layouts match the header (sizes via a tiny C probe compiled with gcc), plan geometry, optimiser listing.
No compute call is made (no GPU here)."""
import ctypes as C
import os
import re
import subprocess
import sys
import tempfile

import numpy as np
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
    assert lib.simt_abi_version() == _lib.ABI_VERSION == 2          # bumped with the trailing descriptor fields of rounds 5-6 (include/simt_hip.h)


def test_ctypes_struct_sizes_match_header():
    from simt_amd import _lib
    structs = {"simt_conv_desc": _lib.ConvDesc, "simt_wgrad_desc": _lib.WgradDesc, "simt_bn_bwd_desc": _lib.BnBwdDesc,
               "simt_head_desc": _lib.HeadDesc, "simt_ntm_inner_desc": _lib.NtmInnerDesc,
               "simt_ntm_post_desc": _lib.NtmPostDesc, "simt_sgd_desc": _lib.SgdDesc, "simt_tap_desc": _lib.TapDesc,
               "simt_wgrad_reduce_job": _lib.WgradReduceJob, "simt_fbn_desc": _lib.FbnDesc, "simt_stem_desc": _lib.StemDesc}
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
    assert _lib.FBN_BAR_WORDS == int(re.search(r"#define SIMT_FBN_BAR_WORDS (\d+)", open(os.path.join(ROOT, "include", "simt_hip.h")).read()).group(1))


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


def test_bench_gpus_flag_spawns_n_ranks_and_fails_loudly_without_gpu():
    """`python bench.py --gpus 2` (no torch.distributed environment) must become a 2-rank torch.distributed.run job by itself -- the
    driver's `--gpus N` is not advisory -- and, on a box without a GPU, every rank must refuse to run (no CPU fallback)."""
    if torch.cuda.is_available():
        pytest.skip("GPU present: covered by tests/test_gpu_bench.py")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert p.returncode != 0
    err = p.stderr + p.stdout
    # at least one rank got as far as the GPU check and refused; the launcher terminates its sibling as soon as the first rank exits
    # (SIGTERM: the sibling may or may not have printed its own refusal yet), and its report names both ranks
    assert err.count("bench.py needs a GPU") >= 1, err[-2000:]
    assert "local_rank: 0" in err and "local_rank: 1" in err, err[-2000:]


def test_bench_rejects_world_size_mismatch():
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="3", MASTER_ADDR="127.0.0.1", MASTER_PORT="29999")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env, capture_output=True, text=True, timeout=120)
    assert p.returncode != 0 and "WORLD_SIZE=3" in (p.stderr + p.stdout)


def test_restore_prefix_required_and_not_restore_last(tmp_path):
    """Checkpoint restore of the tools: key filter + shape filter (trainV2_simt.py:248-255), the warm-up stage's 6-character prefix
    strip (trainV1_warmup.py:177), --not-restore-last, and loud failure on a missing file / zero matches."""
    from simt_amd.tools.trainV2_simt import restore
    state = {"conv1.weight": torch.zeros(4, 3), "layer5.conv2d_list.0.weight": torch.zeros(2, 2), "bn1.bias": torch.zeros(4)}
    ck = {"Scale.conv1.weight": torch.ones(4, 3), "Scale.layer5.conv2d_list.0.weight": torch.ones(2, 2), "Scale.bn1.bias": torch.ones(5),
          "Scale.fc.weight": torch.ones(3)}
    path = str(tmp_path / "ck.pth")
    torch.save(ck, path)
    st = dict(state)
    assert restore(st, path) == 0                                            # verbatim keys: nothing matches
    with pytest.raises(RuntimeError):
        restore(dict(state), path, required=True)
    st = dict(state)
    assert restore(st, path, strip_prefix=6) == 2 and st["conv1.weight"].sum() == 12 and st["bn1.bias"].sum() == 0     # shape-filtered
    st = dict(state)
    assert restore(st, path, strip_prefix=6, not_restore_last=True) == 1 and st["layer5.conv2d_list.0.weight"].sum() == 0
    with pytest.raises(FileNotFoundError):
        restore(dict(state), str(tmp_path / "missing.pth"), required=True)
    assert restore(dict(state), str(tmp_path / "missing.pth")) == 0


def test_one_instruction_exp_error_bound():
    """csrc/head_loss.hip exp_le0: e^x for x <= 0 as exp2(fl32(x * log2 e)).  Emulated in numpy (v_exp_f32 taken as correctly rounded):
    the absolute error of a softmax term stays below half an ulp of the sum it is added to (>= 1), which is the claim the kernel's
    comment and DESIGN.md section 5 make for replacing expf."""
    x = np.linspace(-60.0, 0.0, 600001).astype(np.float32)
    arg = (x * np.float32(1.4426950408889634)).astype(np.float32)
    fast = np.exp2(arg.astype(np.float64)).astype(np.float32).astype(np.float64)
    err = np.abs(fast - np.exp(x.astype(np.float64)))
    assert err.max() < 2.0 ** -24          # half an ulp of 1.0f
    rel = err[x > -20] / np.exp(x[x > -20].astype(np.float64))
    assert rel.max() < 2.5e-6              # and the gradient terms q_j keep six digits down to e^-20


def test_wgrad_tile_rule_matches_header():
    """The planner's copy of the weight-gradient tile rule (ops.wgrad_tile_co / WGRAD3_MIN_PIXELS) and the library's
    (include/simt_hip.h SIMT_WGRAD3_MIN_PIXELS, simt_conv_wgrad_tile_co) must agree: the split count that fills the chip depends on it."""
    import re
    from simt_amd import _lib, ops
    hdr = open(os.path.join(ROOT, "include", "simt_hip.h")).read()
    m = re.search(r"#define\s+SIMT_WGRAD3_MIN_PIXELS\s+(\d+)", hdr)
    assert m and int(m.group(1)) == ops.WGRAD3_MIN_PIXELS
    m = re.search(r"#define\s+SIMT_WGRAD_MULTI_MAX\s+(\d+)", hdr)
    assert m and int(m.group(1)) == 16
    lib = _lib.load()
    for (M, Cd, cin, ntaps) in ((37636, 256, 256, 9), (37636, 1024, 256, 1), (37636, 128, 512, 1), (8192, 256, 256, 9), (16384, 512, 64, 1),
                                (40000, 432, 2048, 1)):
        d = _lib.WgradDesc()
        d.B, d.Ho, d.Wo, d.H, d.W = 1, 1, M, 1, M
        d.Cd, d.Cin, d.ntaps, d.stride, d.dtype = Cd, cin, ntaps, 1, 1
        assert lib.simt_conv_wgrad_tile_co(C.byref(d)) == ops.wgrad_tile_co(M, Cd, ntaps * cin), (M, Cd, cin, ntaps)
    # a grouped launch's split count: whole 256-CU rounds
    assert ops.wgrad_group_nsplit(37636, 51) == 5 and ops.wgrad_group_nsplit(37636, 34) == 7


@pytest.mark.parametrize("tool", ["trainV2_simt", "trainV1_warmup"])
def test_cli_accepts_every_reference_flag(tool):
    """Golden g15 (oracle/gen_golden_flags.py): every flag of the reference's argparse (trainV2_simt.py:72-157, trainV1_warmup.py:60-150)
    parses with a value of the reference's type -- a reference command line that spells them all out must not die in argparse."""
    import importlib
    import json
    flags = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "g15_cli_flags.json")))[tool]
    mod = importlib.import_module(f"simt_amd.tools.{tool}")
    argv = []
    for name, kind in flags:
        argv.append(name)
        if kind != "store_true":
            argv.append({"int": "3", "float": "0.25", "str": "x,y"}[kind])
    ns = vars(mod.get_arguments(argv))
    for name, kind in flags:
        v = ns[name[2:].replace("-", "_")]
        assert v == {"int": 3, "float": 0.25, "str": "x,y", "store_true": True}[kind], (name, v)


def test_snapshot_keeper_rotation_and_atomic_save(tmp_path):
    """trainV2_simt.py:452-464 / trainV1_warmup.py:243-256: one best-mIoU file with the reference's names; the old file goes only after
    the new one is complete (a failed save must not leave the run without a snapshot)."""
    import torch
    from simt_amd.tools import trainV2_simt as tool
    k = tool.SnapshotKeeper(str(tmp_path), "GTA5_BAPA_warmup_iter")
    sd = {"a": torch.zeros(3)}
    assert k.best(sd, 1000, 31.25) and not k.best(sd, 2000, 30.0) and k.best(sd, 3000, 40.5)
    assert sorted(os.listdir(tmp_path)) == ["GTA5_BAPA_warmup_iter3000_mIoU40.5.pth"]
    k.rolling(sd, 4000)
    k.rolling(sd, 5000)
    assert sorted(os.listdir(tmp_path)) == ["GTA5_BAPA_warmup_iter3000_mIoU40.5.pth", "GTA5_BAPA_warmup_iter5000.pth"]
    real_save = torch.save

    def failing(obj, path, *a, **kw):
        real_save(obj, path, *a, **kw)
        raise OSError("disk full")
    torch.save = failing
    try:
        with pytest.raises(OSError):
            k.rolling(sd, 6000)
    finally:
        torch.save = real_save
    assert "GTA5_BAPA_warmup_iter5000.pth" in os.listdir(tmp_path)            # the previous snapshot survived the failed save


def test_no_scratch_in_production_kernels():
    """no kernel a production plan launches may use scratch memory (spilled VGPRs or stack arrays).  The shipped
    library's gfx950 code objects are disassembled (tests/_codeobj.py) and every kernel is checked for scratch_* instructions; the only
    ones allowed to have any are the generic FALLBACK instantiations no BASELINE configuration runs: the head kernels for class counts
    other than (Q, C) = (22, 19) / (25, 19), and the run-time-flag epilogue of the 2-slot short-K conv (training plans launch its
    compile-time flavours 4-7: asserted by launch tag in tests/test_gpu_prod_shapes.py)."""
    import __graft_entry__ as ge
    import _codeobj
    from simt_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        ge.build()
    res = _codeobj.kernels_with_scratch(_lib.LIB_PATH)
    assert len(res) > 100 and any("conv_igemm2_kernel<256, 5, 3, 0, 1>" in k for k in res) and any("head_pass1_kernel<24, 22, 19>" in k for k in res)
    allowed = ("conv_igemm2_kernel<128, 5, 2, 0, 0>", "conv_igemm2_kernel<128, 4, 2, 0, 0>",
               "head_pass1_kernel<24, 0, 0>", "head_pass1_kernel<40, 0, 0>", "head_pass2_kernel<40, 0, 0>")
    bad = {k: v for k, v in res.items() if v and not any(a in k for a in allowed)}
    assert not bad, f"kernels with scratch_* instructions: {bad}"


def test_bench_power_watch_is_silent_without_a_gpu():
    """bench.py's power / clock sampler (sysfs of the process's GPU) must never break a run: no GPU (here) or unreadable sysfs -> no object in
    the JSON line; with readable files it reports their means (a fake card directory)."""
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    with bench.PowerWatch(0) as pw:
        pass
    assert pw.report() is None or isinstance(pw.report(), dict)
    with tempfile.TemporaryDirectory() as td:
        os.makedirs(os.path.join(td, "hwmon", "hwmon3"))
        open(os.path.join(td, "hwmon", "hwmon3", "power1_average"), "w").write("1218000000\n")
        open(os.path.join(td, "pp_dpm_sclk"), "w").write("0: 132Mhz\n1: 2166Mhz *\n2: 2400Mhz\n")
        pw = bench.PowerWatch.__new__(bench.PowerWatch)
        pw.dir, pw.pfile, pw.samples = td, os.path.join(td, "hwmon", "hwmon3", "power1_average"), []
        pw.samples.append(pw._read())
        r = pw.report()
        assert r["avg_w"] == 1218.0 and r["sclk_mhz_avg"] == 2166.0 and r["sclk_mhz_top_level"] == 2400.0 and r["samples"] == 1


def test_measurement_tools_compile():
    """profiles/tools/*.py, profiles/*.py and bench.py are run only on the GPU box: at least their syntax is checked here."""
    import glob
    import py_compile
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    files = sorted(glob.glob(os.path.join(root, "profiles", "tools", "*.py")) + glob.glob(os.path.join(root, "profiles", "*.py"))) + [os.path.join(root, "bench.py")]
    assert len(files) > 20
    with tempfile.TemporaryDirectory() as td:
        for f in files:
            py_compile.compile(f, cfile=os.path.join(td, "x.pyc"), doraise=True)
