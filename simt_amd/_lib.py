"""ctypes binding of libsimt_hip.so (the C ABI declared in include/simt_hip.h).

The product path has no CPU fallback: if the shared library is missing or a call fails, this raises.
Build it with `python -c "import __graft_entry__ as g; g.build()"` or `simt_amd/csrc/build.sh`.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SIMT_LIB_PATH") or os.path.join(_HERE, "libsimt_hip.so")      # SIMT_LIB_PATH: A/B a scratch build (measurements only)

SIMT_F32, SIMT_BF16 = 0, 1
MAX_TAPS = 36
ABI_VERSION = 2          # include/simt_hip.h SIMT_ABI_VERSION
QMAX = 40

c_p = C.c_void_p
i32 = C.c_int32
f32 = C.c_float


class ConvDesc(C.Structure):
    _fields_ = [("x", c_p), ("w", c_p), ("y", c_p), ("bias", c_p), ("res", c_p), ("stats", c_p),
                ("B", i32), ("H", i32), ("W", i32), ("Cin", i32), ("Ho", i32), ("Wo", i32), ("Cout", i32),
                ("Npad", i32), ("Nstore", i32), ("ldy", i32), ("ldr", i32), ("stride", i32), ("ntaps", i32),
                ("relu", i32), ("dtype_in", i32), ("dtype_out", i32), ("tile_n", i32),
                ("dy", C.c_int16 * MAX_TAPS), ("dx", C.c_int16 * MAX_TAPS), ("mask", c_p), ("ldm", i32),
                ("res_bits", c_p), ("bnr_y", c_p), ("bnr_mean", c_p), ("bnr_rstd", c_p), ("bnr_scale", c_p), ("bnr_shift", c_p),
                ("bnr_bits", c_p), ("bnr_part", c_p), ("bnr_mode", i32), ("bnr_ld", i32), ("w_frag", c_p), ("fbn", c_p),
                ("cu_budget", i32), ("reserved_", i32), ("in_scale", c_p), ("in_shift", c_p), ("in_out", c_p)]


FBN_BAR_WORDS = 144
FBN_ERR_WORD = 136


class FbnDesc(C.Structure):
    """simt_fbn_desc (include/simt_hip.h): the train-mode BatchNorm behind a conv, fused into the producing launch."""
    _fields_ = [("mode", i32), ("ldo", i32), ("out", c_p), ("work", c_p), ("gamma", c_p), ("beta", c_p),
                ("running_mean", c_p), ("running_var", c_p), ("momentum", f32), ("eps", f32),
                ("mean", c_p), ("rstd", c_p), ("scale", c_p), ("shift", c_p), ("coef", c_p), ("dgamma", c_p), ("dbeta", c_p),
                ("err", c_p)]


class WgradDesc(C.Structure):
    _fields_ = [("dy", c_p), ("x", c_p), ("slab", c_p),
                ("B", i32), ("H", i32), ("W", i32), ("Cin", i32), ("Ho", i32), ("Wo", i32), ("Cd", i32), ("ldd", i32),
                ("stride", i32), ("ntaps", i32), ("nsplit", i32), ("dtype", i32),
                ("dy_", C.c_int16 * MAX_TAPS), ("dx_", C.c_int16 * MAX_TAPS)]


class WgradReduceJob(C.Structure):
    """simt_wgrad_reduce_job (include/simt_hip.h)."""
    _fields_ = [("slab", c_p), ("dst", c_p), ("nsplit", i32), ("Cd", i32), ("Ktot", i32), ("Cin", i32), ("co_off", i32),
                ("tap_off", i32), ("Cout", i32), ("RS", i32), ("accumulate", i32), ("block0", i32)]


class BnBwdDesc(C.Structure):
    _fields_ = [("dz", c_p), ("z", c_p), ("y", c_p), ("mean", c_p), ("rstd", c_p), ("scale", c_p), ("shift", c_p),
                ("y2", c_p), ("mean2", c_p), ("rstd2", c_p), ("scale2", c_p), ("part", c_p), ("coef", c_p),
                ("dy", c_p), ("dy2", c_p), ("gout", c_p), ("M", C.c_int64), ("C", i32), ("mask_mode", i32),
                ("dtype", i32), ("dgamma", c_p), ("dbeta", c_p), ("dgamma2", c_p), ("dbeta2", c_p), ("reduce_done_nblk", i32)]


class HeadDesc(C.Structure):
    _fields_ = [("pred1", c_p), ("pred2", c_p), ("fixp", c_p), ("label", c_p), ("T1", c_p), ("T2", c_p),
                ("part", c_p), ("keys", c_p), ("hout", c_p), ("g1", c_p), ("dpred1_f32", c_p), ("dpred2_f32", c_p),
                ("dpred1_t", c_p), ("dpred2_t", c_p),
                ("B", i32), ("h", i32), ("w", i32), ("H", i32), ("W", i32), ("C", i32), ("Q", i32), ("ldp", i32),
                ("ldf", i32), ("QP", i32), ("ld_f32", i32), ("ld_t", i32), ("grad_dtype", i32),
                ("th_high", f32), ("th_low", f32), ("lambda_seg", f32), ("lambda_place", f32), ("gscale", f32), ("mode", i32),
                ("single", i32), ("up_half_pixel", i32), ("fix_logits", i32), ("conf_out", c_p), ("label_ws", c_p)]


class StemDesc(C.Structure):
    """simt_stem_desc: the direct 7x7 stem convolution for up to two weight sets."""
    _fields_ = [("x", c_p), ("B", i32), ("H", i32), ("W", i32), ("Ho", i32), ("Wo", i32), ("nsets", i32),
                ("w", c_p * 2), ("y", c_p * 2), ("bias", c_p * 2), ("relu", i32 * 2), ("stats", c_p * 2)]


class NtmInnerDesc(C.Structure):
    _fields_ = [("ntm", c_p * 2), ("w", c_p * 2), ("ntm_grad", c_p * 2), ("w_m", c_p * 2), ("w_v", c_p * 2),
                ("T_out", c_p * 2), ("class_dist", c_p),
                ("Q", i32), ("C", i32), ("steps", i32), ("step0", i32),
                ("lr", f32), ("beta1", f32), ("beta2", f32), ("eps", f32), ("single", i32), ("skip_if", c_p)]


class NtmPostDesc(C.Structure):
    _fields_ = [("ntm", c_p * 2), ("w", c_p * 2), ("ntm_grad", c_p * 2), ("class_dist", c_p), ("hout", c_p),
                ("lout", c_p), ("Q", i32), ("C", i32),
                ("lambda_seg", f32), ("lambda_convex", f32), ("lambda_volume", f32), ("lambda_anchor", f32),
                ("gscale", f32), ("single", i32)]


class TapDesc(C.Structure):
    _fields_ = [("src", c_p), ("bias", c_p), ("dst", c_p),
                ("B", i32), ("H", i32), ("W", i32), ("Q", i32), ("QP", i32), ("lds", i32), ("ldd", i32), ("ntaps", i32),
                ("dy", C.c_int16 * MAX_TAPS), ("dx", C.c_int16 * MAX_TAPS)]


class SgdDesc(C.Structure):
    _fields_ = [("segs", c_p), ("chunks", c_p), ("nchunks", i32), ("chunk", i32), ("lr", f32 * 4), ("wd", f32 * 4),
                ("momentum", f32), ("dampening", f32), ("first_step", i32), ("skip_if", c_p)]


# name -> (restype, argtypes); every symbol include/simt_hip.h declares
_L = C.c_long
_I = C.c_int
SIGNATURES = {
    "simt_last_error": (C.c_char_p, []),
    "simt_abi_version": (_I, []),
    "simt_conv_fprop": (_I, [C.POINTER(ConvDesc), c_p]),
    "simt_stem7_tiles": (_I, [_I, _I, _I]),
    "simt_stem7_pack": (_I, [c_p, c_p, c_p, c_p]),
    "simt_stem7_fwd": (_I, [C.POINTER(StemDesc), c_p]),
    "simt_stem7_wgrad_workgroups": (_I, [_I, _I, _I]),
    "simt_stem7_wgrad": (_I, [c_p, c_p, c_p, c_p, _I, _I, _I, _I, _I, c_p]),
    "simt_conv_mtiles": (_I, [C.POINTER(ConvDesc)]),
    "simt_conv_inbn_ok": (_I, [C.POINTER(ConvDesc)]),
    "simt_conv_fprop_pair": (_I, [C.POINTER(ConvDesc), C.POINTER(ConvDesc), c_p]),
    "simt_conv_pair_fused": (_I, [C.POINTER(ConvDesc), C.POINTER(ConvDesc)]),
    "simt_conv_fbn_ok": (_I, [C.POINTER(ConvDesc)]),
    "simt_conv_epilogue_flavour": (_I, [C.POINTER(ConvDesc)]),
    "simt_conv_fbn_words": (_L, [C.POINTER(ConvDesc)]),
    "simt_conv_wants_frag": (_I, [C.POINTER(ConvDesc)]),
    "simt_conv_variant": (_I, [C.POINTER(ConvDesc), C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "simt_conv_wgrad": (_I, [C.POINTER(WgradDesc), c_p]),
    "simt_wgrad_reduce": (_I, [c_p, c_p, _I, _I, _I, _I, _I, _I, _I, _I, _I, c_p]),
    "simt_conv_wgrad_multi_ok": (_I, [C.POINTER(WgradDesc)]),
    "simt_conv_wgrad_multi_bytes": (_I, []),
    "simt_conv_wgrad_multi_prepare": (_I, [C.POINTER(WgradDesc), _I, c_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "simt_conv_wgrad_multi": (_I, [c_p, _I, _I, _I, _I, c_p]),
    "simt_conv_wgrad_tile_co": (_I, [C.POINTER(WgradDesc)]),
    "simt_wgrad_reduce_multi": (_I, [c_p, _I, _I, c_p]),
    "simt_wgrad_reduce_blocks": (_I, [_I, _I, _I]),
    "simt_pack_weight": (_I, [c_p, c_p, _I, _I, _I, _I, _I, _L, _I, _I, c_p, _I, c_p]),
    "simt_pack_weight_multi": (_I, [c_p, c_p, _I, _I, c_p]),
    "simt_bn_fold": (_I, [c_p, c_p, c_p, c_p, f32, c_p, c_p, _I, c_p]),
    "simt_bn_finalize": (_I, [c_p, _I, _I, _L, c_p, c_p, c_p, c_p, f32, f32, c_p, c_p, c_p, c_p, c_p]),
    "simt_bn_apply": (_I, [c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, _L, _I, _I, _I, c_p]),
    "simt_bn_apply_bits": (_I, [c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, _L, _I, _I, _I, c_p]),
    "simt_bn_bwd_nblk": (_I, [_L, _I]),
    "simt_bn_bwd": (_I, [C.POINTER(BnBwdDesc), c_p]),
    "simt_im2col_stem": (_I, [c_p, c_p, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, c_p]),
    "simt_bn_relu_maxpool": (_I, [c_p, c_p, c_p, c_p, c_p, _I, _I, _I, _I, _I, _I, _I, c_p]),
    "simt_maxpool_bwd": (_I, [c_p, c_p, c_p, _I, _I, _I, _I, _I, _I, _I, c_p]),
    "simt_scatter_stride": (_I, [c_p, c_p, _I, _I, _I, _I, _I, _I, _I, _I, c_p]),
    "simt_colsum": (_I, [c_p, c_p, _L, _I, _I, _I, _I, c_p]),
    "simt_maxpool2": (_I, [c_p, c_p, c_p, _I, _I, _I, _I, _I, c_p]),
    "simt_maxpool2_bwd": (_I, [c_p, c_p, c_p, c_p, _I, _I, _I, _I, _I, c_p]),
    "simt_colsum_wide": (_I, [c_p, c_p, _L, _I, _I, _I, c_p]),
    "simt_head_nblk": (_I, [_I, _I, _I]),
    "simt_head_part_floats": (_I, [_I, _I]),
    "simt_head_hout_floats": (_I, [_I, _I]),
    "simt_head_keys_count": (_I, []),
    "simt_softmax_rows": (_I, [c_p, _I, c_p, _I, _L, _I, c_p]),
    "simt_head_loss": (_I, [C.POINTER(HeadDesc), c_p]),
    "simt_head_grad": (_I, [C.POINTER(HeadDesc), c_p]),
    "simt_ntm_inner_loop": (_I, [C.POINTER(NtmInnerDesc), c_p]),
    "simt_ntm_post": (_I, [C.POINTER(NtmPostDesc), c_p]),
    "simt_sig_ntm": (_I, [c_p, c_p, c_p, c_p, c_p, _I, _I, c_p]),
    "simt_sig_w": (_I, [c_p, c_p, c_p, c_p, _I, c_p]),
    "simt_adam_step": (_I, [c_p, c_p, c_p, c_p, _L, f32, f32, f32, f32, _I, c_p]),
    "simt_adam_step_guarded": (_I, [c_p, c_p, c_p, c_p, _L, f32, f32, f32, f32, _I, c_p, c_p]),
    "simt_sgd_multi": (_I, [C.POINTER(SgdDesc), c_p]),
    "simt_event_create": (_I, [C.POINTER(c_p), _I]),
    "simt_event_destroy": (_I, [c_p]),
    "simt_event_record": (_I, [c_p, c_p]),
    "simt_stream_wait_event": (_I, [c_p, c_p]),
    "simt_vec_acc": (_I, [c_p, c_p, _I, _I, c_p]),
    "simt_tap_gather_sum": (_I, [C.POINTER(TapDesc), c_p]),
    "simt_tap_scatter": (_I, [C.POINTER(TapDesc), c_p]),
    "simt_wgrad_reduce_exp": (_I, [c_p, c_p, _I, _I, _I, _I, _I, _I, _I, _I, c_p]),
    "simt_upsample_sum_argmax": (_I, [c_p, _I, _I, _I, c_p, _I, _I, _I, _I, _I, _I, _I, c_p, c_p]),
    "simt_confusion_hist": (_I, [c_p, c_p, _L, _I, c_p, c_p]),
    "simt_upsample_nchw": (_I, [c_p, _I, _I, _I, _I, _I, _I, _I, _I, c_p, c_p]),
    "simt_upsample_nchw_bwd": (_I, [c_p, _I, _I, _I, _I, _I, _I, _I, _I, c_p, _I, c_p, c_p]),
    "simt_loss_ws_bytes": (_I, []),
    "simt_ce2d_fwd": (_I, [c_p, c_p, c_p, _I, _I, _I, _I, _I, _I, c_p, c_p, c_p]),
    "simt_ce2d_bwd": (_I, [c_p, c_p, c_p, _I, _I, _I, _I, _I, _I, c_p, c_p, c_p, c_p]),
    "simt_entropy2d": (_I, [c_p, _I, _I, _I, _I, c_p, c_p, c_p, c_p, c_p]),
    "simt_hist2d_u8": (_I, [c_p, c_p, _L, _I, _I, c_p, c_p, c_p]),
    "simt_resample_u8": (_I, [c_p, c_p, _I, _I, _I, _I, _I, _I, c_p, c_p, _I, c_p]),
    "simt_image_to_input": (_I, [c_p, c_p, _I, _I, _I, f32, f32, f32, _I, c_p]),
    "simt_label_nearest": (_I, [c_p, c_p, _I, _I, _I, _I, _I, c_p, c_p, _I, c_p]),
}

_lib = None


class SimtHipError(RuntimeError):
    pass


def load():
    """Load libsimt_hip.so and bind every declared symbol.  Raises if the library is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise SimtHipError(
            f"{LIB_PATH} not found: the HIP extension is required (no CPU fallback). "
            "Run `python -c 'import __graft_entry__ as g; g.build()'`.")
    # PyTorch-ROCm ships its own libamdhip64 (same soname); it must be mapped first so that this library binds to
    # the SAME HIP runtime instance that owns torch's streams and allocations (two runtimes in one process cannot
    # share stream handles: launches fail with "no ROCm-capable device").
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the .so does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    if lib.simt_abi_version() != ABI_VERSION:      # a stale libsimt_hip.so reads past the end of the shorter descriptors it was built for
        raise SimtHipError(f"{LIB_PATH} has ABI version {lib.simt_abi_version()}, these bindings need {ABI_VERSION}: rebuild it "
                           "(python -c 'import __graft_entry__ as g; g.build()')")
    _lib = lib
    return lib


def check(rc):
    if rc != 0:
        msg = load().simt_last_error()
        raise SimtHipError(f"libsimt_hip call failed (code {rc}): {msg.decode() if msg else '?'}")


def call(name, *args):
    check(getattr(load(), name)(*args))
