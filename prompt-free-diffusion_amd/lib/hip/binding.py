"""ctypes binding of libpfd_hip.so (C ABI declared in include/pfd_hip.h).

This is the only place the Python host touches native code.  There is NO fallback: if the
library is missing, was built for another ABI version, or a call returns a negative code,
a RuntimeError is raised -- the product path never silently runs torch ops instead.

The reference has no FFI of its own (its operator API is a Python class registry over stock
torch ops, lib/model_zoo/common/get_model.py:54-124); INTEGRATION.md shows how a reference
maintainer would bind the same entry points.
"""
import ctypes as C
import os

_LIB = None
_LIB_PATH = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "libpfd_hip.so"))

ABI_VERSION = 9

ACT_NONE, ACT_GELU, ACT_RELU, ACT_SILU, ACT_GEGLU = 0, 1, 2, 3, 4

_i32, _i64, _f32, _vp, _sz = C.c_int32, C.c_int64, C.c_float, C.c_void_p, C.c_size_t


class PfdGemmDesc(C.Structure):
    _fields_ = [
        ("A", _vp), ("W", _vp), ("bias", _vp), ("rowvec", _vp), ("R", _vp), ("C", _vp),
        ("lda", _i64), ("ldw", _i64), ("ldr", _i64), ("ldc", _i64), ("ldrv", _i64),
        ("M", _i32), ("N", _i32), ("K", _i32),
        ("rows_per_rv", _i32), ("act", _i32), ("bias_per_row", _i32),
        ("ksize", _i32), ("stride", _i32), ("pad", _i32), ("ups", _i32),
        ("B", _i32), ("H", _i32), ("Wd", _i32), ("Cin", _i32), ("Ho", _i32), ("Wo", _i32),
        ("ws", _vp), ("ws_bytes", _sz),
        ("Ct", _vp), ("ldct", _i64), ("n_split", _i32), ("w_tiled", _i32),
        ("gn_table", _vp), ("A2", _vp), ("lda2", _i64), ("gn_c1", _i32), ("gn_act", _i32),
        ("ln_stats", _vp), ("ln_colsum", _vp), ("ln_parts", _i32), ("ln_eps", _f32), ("ln_out", _vp),
        ("k_split", _i32), ("zero_rows", _i32), ("gn_out", _vp),
        ("gnf_gamma", _vp), ("gnf_beta", _vp), ("gnf_y", _vp), ("gnf_ldy", _i64), ("gnf_eps", _f32), ("gnf_act", _i32),
        ("gnf_rows", _i32), ("gnf_skip_raw", _i32), ("res_rows", _i32),
    ]


class PfdAttnDesc(C.Structure):
    _fields_ = [
        ("Q", _vp), ("K", _vp), ("Vt", _vp), ("O", _vp),
        ("ldq", _i64), ("ldk", _i64), ("ldvt", _i64), ("ldo", _i64),
        ("q_bs", _i64), ("k_bs", _i64), ("vt_bs", _i64), ("o_bs", _i64),
        ("B", _i32), ("H", _i32), ("Nq", _i32), ("Nk", _i32), ("D", _i32),
        ("scale", _f32),
    ]


class PfdSwinAttnDesc(C.Structure):
    _fields_ = [
        ("qkv", _vp), ("qkv_bias", _vp), ("rpb", _vp), ("out", _vp),
        ("B", _i32), ("H", _i32), ("W", _i32), ("C", _i32), ("nH", _i32), ("ws", _i32), ("shift", _i32),
        ("scale", _f32),
    ]


# name -> (restype, argtypes); this table is also what tests/test_host.py (test_cabi_exports_every_declared_symbol) checks against the header
SIGNATURES = {
    "pfd_abi_version": (_i32, []),
    "pfd_last_error": (C.c_char_p, []),
    "pfd_gemm_f16": (_i32, [C.POINTER(PfdGemmDesc), _vp]),
    "pfd_gemm_f16_ex": (_i32, [C.POINTER(PfdGemmDesc), _i32, _vp]),
    "pfd_gemm_geglu_group": (_i32, [_i32]),
    "pfd_attention_f16": (_i32, [C.POINTER(PfdAttnDesc), _vp]),
    "pfd_swin_window_attention_f16": (_i32, [C.POINTER(PfdSwinAttnDesc), _vp]),
    "pfd_groupnorm_ws_bytes": (_sz, [_i32, _i32, _i32]),
    "pfd_groupnorm_table_f16": (_i32, [_vp, _i32, _i64, _vp, _i32, _i64, _vp, _vp, _vp, _i32, _i32, _i32, _f32, _vp, _sz,
                                       _vp]),
    "pfd_groupnorm_f16": (_i32, [_vp, _i32, _i64, _vp, _i32, _i64, _vp, _vp, _vp, _i64, _i32, _i32, _i32, _f32,
                                 _i32, _vp, _sz, _vp]),
    "pfd_groupnorm_takes_pstats": (_i32, [_i32, _i32, _i32, _i32, _i32]),
    "pfd_groupnorm_pstats_f16": (_i32, [_vp, _i32, _i64, _vp, _vp, _i32, _i64, _vp, _vp, _vp, _vp, _i64, _i32, _i32, _i32,
                                        _f32, _i32, _vp]),
    "pfd_layernorm_f16": (_i32, [_vp, _i64, _vp, _vp, _vp, _i64, _i32, _i32, _f32, _i32, _i32, _i32, _i32, _vp]),
    "pfd_ln_rowstats_f16": (_i32, [_vp, _i64, _i32, _i32, _vp, _vp]),
    "pfd_softmax_rows_f16": (_i32, [_vp, _i64, _vp, _i64, _i32, _i32, _f32, _vp]),
    "pfd_nchw_to_nhwc_f16": (_i32, [_vp, _i32, _vp, _i32, _i32, _i32, _i32, _f32, _f32, _i32, _vp]),
    "pfd_nhwc_to_nchw": (_i32, [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _f32, _f32, _f32, _f32, _vp]),
    "pfd_im2col_f16": (_i32, [_vp, _i64, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp]),
    "pfd_timestep_embedding_f16": (_i32, [_vp, _vp, _i32, _i32, _f32, _vp]),
    "pfd_cfg_ddim_step": (_i32, [_vp, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp]),
    "pfd_add_f16": (_i32, [_vp, _vp, _vp, _i64, _vp]),
    "pfd_axpby_f16": (_i32, [_vp, _f32, _vp, _f32, _vp, _i64, _vp]),
    "pfd_add_rowvec_f16": (_i32, [_vp, _i64, _vp, _vp, _i64, _i32, _i32, _vp]),
    "pfd_add_rowvec_lnstats_f16": (_i32, [_vp, _i64, _vp, _vp, _i64, _i32, _i32, _vp, _vp]),
    "pfd_act_f16": (_i32, [_vp, _vp, _i64, _i32, _vp]),
    "pfd_image_u8_f16": (_i32, [_vp, _vp, _i64, _f32, _f32, _i32, _vp]),
    "pfd_prof_enable": (_i32, [_i32]),
    "pfd_prof_read": (_i32, [_i32, C.POINTER(C.c_double), C.POINTER(_i64), C.POINTER(C.c_double),
                             C.POINTER(C.c_double)]),
    "pfd_prof_bucket_name": (C.c_char_p, [_i32]),
    "pfd_prof_num_buckets": (_i32, []),
}


def lib_path():
    return os.environ.get("PFD_HIP_LIB", _LIB_PATH)


def load():
    """dlopen libpfd_hip.so once and type every entry point.  Raises if anything is missing."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = lib_path()
    if not os.path.exists(path):
        raise RuntimeError(
            f"libpfd_hip.so not found at {path}: build it with `python __graft_entry__.py` "
            "(or `make -C prompt-free-diffusion_amd/csrc`).  There is no CPU/torch fallback.")
    lib = C.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name, None)
        if fn is None:
            raise RuntimeError(f"{path} does not export {name}")
        fn.restype = res
        fn.argtypes = args
    v = lib.pfd_abi_version()
    if v != ABI_VERSION:
        raise RuntimeError(f"libpfd_hip.so ABI {v} != binding ABI {ABI_VERSION}")
    _LIB = lib
    return lib


class PfdError(RuntimeError):
    pass


PFD_EINVAL, PFD_ESHAPE, PFD_ELAUNCH = -1, -2, -3
_ERR = {PFD_EINVAL: "PFD_EINVAL", PFD_ESHAPE: "PFD_ESHAPE", PFD_ELAUNCH: "PFD_ELAUNCH"}


def check(rc, what):
    if rc != 0:
        msg = load().pfd_last_error().decode(errors="replace")
        raise PfdError(f"{what} failed: {_ERR.get(rc, rc)} {msg}")


def prof_enable(on=True):
    check(load().pfd_prof_enable(1 if on else 0), "pfd_prof_enable")


def prof_read():
    """[{name, ms, launches, flops, bytes}] for every bucket that saw a launch"""
    lib = load()
    out = []
    for b in range(lib.pfd_prof_num_buckets()):
        ms, n, fl, by = C.c_double(), _i64(), C.c_double(), C.c_double()
        check(lib.pfd_prof_read(b, C.byref(ms), C.byref(n), C.byref(fl), C.byref(by)), "pfd_prof_read")
        if n.value:
            out.append(dict(bucket=b, name=lib.pfd_prof_bucket_name(b).decode(), ms=ms.value, launches=n.value,
                            flops=fl.value, bytes=by.value))
    return out
