"""ctypes binding of libssqp_hip.so (include/ssqp_hip.h).

The library is the product: there is no Python/CPU fallback.  If it has not
been built (`python __graft_entry__.py build` or `make -C .../csrc`) importing
this module raises, and creating a Context without a GPU raises NoDeviceError.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# SSQP_HIP_LIB selects another build of the same library (e.g. the phase-profile diagnostic build)
LIB_PATH = os.environ.get("SSQP_HIP_LIB") or os.path.join(_HERE, "libssqp_hip.so")

OK, ERR_ARG, ERR_NO_DEVICE, ERR_HIP, ERR_ALLOC, ERR_UNSUPPORTED = range(6)


class NoDeviceError(RuntimeError):
    pass


class SSQPError(RuntimeError):
    pass


class CSettings(C.Structure):
    _fields_ = [("maxIter", C.c_int32), ("rule", C.c_int32), ("tol", C.c_double), ("tolG", C.c_double)]


class CStats(C.Structure):
    _fields_ = [("iters", C.c_int64), ("alg_bytes", C.c_int64), ("read_bytes", C.c_int64), ("alg_flops", C.c_int64),
                ("sum_k3", C.c_int64),
                ("max_k", C.c_int32), ("path", C.c_int32)]


class CTrace(C.Structure):
    _fields_ = [("K", C.c_int32), ("W", C.c_int32), ("kind", C.c_int32), ("id", C.c_int32)]


class CStrides(C.Structure):
    _fields_ = [(k, C.c_size_t) for k in ("V", "A", "G", "q", "b", "g", "d", "u")]


class CGenCfg(C.Structure):
    _fields_ = [("N", C.c_int32), ("M", C.c_int32), ("J", C.c_int32), ("T", C.c_int32), ("delta", C.c_double),
                ("ub", C.c_double), ("gscale", C.c_double), ("qscale", C.c_double)]


_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int32)
_lp = C.POINTER(C.c_int64)
_vp = C.c_void_p

# every symbol include/ssqp_hip.h declares: name -> (restype, argtypes)
SIGNATURES = {
    "ssqp_ctx_create": (C.c_int, [C.c_int, C.POINTER(_vp)]),
    "ssqp_ctx_destroy": (C.c_int, [_vp]),
    "ssqp_last_error": (C.c_char_p, [_vp]),
    "ssqp_ctx_set_option": (C.c_int, [_vp, C.c_char_p, C.c_int]),
    "ssqp_ctx_get_option": (C.c_int, [_vp, C.c_char_p, C.POINTER(C.c_int)]),
    "ssqp_version": (C.c_char_p, []),
    "ssqp_default_settings": (None, [C.POINTER(CSettings)]),
    "ssqp_solve_f64": (C.c_int, [_vp] + [C.c_int] * 3 + [_vp] * 8 + [_vp, _vp, _vp, C.POINTER(CSettings), _lp, _ip]),
    "ssqp_solve_full_f64": (C.c_int, [_vp] + [C.c_int] * 3 + [_vp] * 8 + [C.c_int, _vp, _vp, C.POINTER(CSettings),
                                                                          C.POINTER(CSettings), _lp, _ip]),
    "ssqp_solve_batch_f64": (C.c_int, [_vp] + [C.c_int] * 4 + [_vp] * 8 + [_vp, _vp, _vp, C.POINTER(CSettings),
                                                                          _vp, _vp, _vp, _vp, _vp]),
    "ssqp_problem_upload": (C.c_int, [_vp] + [C.c_int] * 4 + [_vp] * 8 + [C.POINTER(_vp)]),
    "ssqp_problem_set_vector": (C.c_int, [_vp, C.c_int, _vp]),
    "ssqp_problem_solve": (C.c_int, [_vp, _vp, _vp, _vp, C.POINTER(CSettings), _vp, _vp, _vp]),
    "ssqp_problem_free": (C.c_int, [_vp]),
    "ssqp_solve_batch_multi_f64": (C.c_int, [C.POINTER(_vp), C.c_int] + [C.c_int] * 4 + [_vp] * 8 +
                                   [_vp, _vp, _vp, C.POINTER(CSettings), _vp, _vp, _vp]),
    "ssqp_solve_batch_dev_f64": (C.c_int, [_vp] + [C.c_int] * 4 + [_vp] * 8 + [_vp, _vp, _vp, C.POINTER(CSettings),
                                                                              _vp, _vp, _vp, _vp, C.c_int, _vp, _vp, _vp]),
    "ssqp_solve_batch_strided_dev_f64": (C.c_int, [_vp] + [C.c_int] * 4 + [_vp] * 8 + [C.POINTER(CStrides)] +
                                         [_vp, _vp, _vp, C.POINTER(CSettings), _vp, _vp, _vp, _vp, C.c_int, _vp, _vp, _vp]),
    "ssqp_sync": (C.c_int, [_vp, _vp]),
    "ssqp_flush": (C.c_int, [_vp]),
    "ssqp_flush_to": (C.c_int, [_vp, _vp]),
    "ssqp_last_kernel_ms": (C.c_int, [_vp, C.POINTER(C.c_float)]),
    "ssqp_recent_kernel_ms": (C.c_int, [_vp, C.c_int, C.POINTER(C.c_float)]),
    "ssqp_phase1_f64": (C.c_int, [C.c_int] * 3 + [_vp] * 6 + [C.POINTER(CSettings), _vp, _vp, _ip]),
    "ssqp_phase1_batch_f64": (C.c_int, [C.c_int] * 4 + [_vp] * 6 + [C.POINTER(CSettings), _vp, _vp, _vp, C.c_int]),
    "ssqp_solve_full_batch_dev_f64": (C.c_int, [_vp] + [C.c_int] * 4 + [_vp] * 8 + [_vp, _vp, C.POINTER(CSettings),
                                                 C.POINTER(CSettings), _vp, _vp, _vp, _vp, _vp, _vp]),
    "ssqp_phase1_batch_dev_f64": (C.c_int, [_vp] + [C.c_int] * 4 + [_vp] * 6 + [C.POINTER(CSettings), _vp, _vp, _vp, _vp]),
    "ssqp_generate_problem": (C.c_int, [C.POINTER(CGenCfg), C.c_uint64] + [_vp] * 8),
    "ssqp_generate_batch": (C.c_int, [C.POINTER(CGenCfg), C.c_uint64, C.c_int] + [_vp] * 8 + [C.c_int]),
    "ssqp_generate_V_dev": (C.c_int, [_vp, C.POINTER(CGenCfg), C.c_uint64, C.c_int, _vp, _vp]),
}

_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                "libssqp_hip.so is not built (%s). Run `python __graft_entry__.py build`; "
                "there is no CPU fallback." % LIB_PATH)
        # PyTorch-ROCm bundles its own libamdhip64.so.7.  Two HIP runtimes in one process do not
        # both see the GPU, so when torch is present it is imported FIRST: the loader then binds
        # this library's NEEDED libamdhip64.so.7 to the copy torch already loaded.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        _lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(_lib, name)
            fn.restype = res
            fn.argtypes = args
    return _lib


def check(rc, ctx=None):
    if rc == OK:
        return
    msg = ""
    if ctx:
        m = lib().ssqp_last_error(ctx)
        msg = m.decode() if m else ""
    if rc == ERR_NO_DEVICE:
        raise NoDeviceError("no HIP device: the SSQP backend has no CPU fallback")
    raise SSQPError("libssqp_hip error %d %s" % (rc, msg))
