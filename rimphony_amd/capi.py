"""ctypes binding of the C ABI declared in include/rimphony_hip.h.

No CPU fallback: loading fails loudly if librimphony_hip.so is missing, and
creating a context fails loudly if there is no GPU.
"""
import ctypes
import os
from ctypes import POINTER, c_char_p, c_double, c_int, c_int32, c_size_t, c_uint32, c_uint64, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RIMPHONY_HIP_LIB") or os.path.join(_HERE, "librimphony_hip.so")   # env override: tuning builds

SYMBOLS = [
    "rimphony_dist_nparams", "rimphony_ctx_create", "rimphony_ctx_destroy", "rimphony_strerror",
    "rimphony_version", "rimphony_last_work", "rimphony_last_symphony_ms", "rimphony_last_faraday_ms", "rimphony_debug_heartbeat", "rimphony_debug_counters", "rimphony_batch_compute_device", "rimphony_batch_compute",
    "rimphony_batch_norm_device", "rimphony_bessel_batch_device", "rimphony_gamma_integrand_batch_device",
    "rimphony_gamma_integral_batch_device", "rimphony_n_integral_batch_device", "rimphony_gamma_contribution_batch_device", "rimphony_calc_f_batch_device", "rimphony_calc_f_batch", "rimphony_qag_selftest_device", "rimphony_highfreq_batch_device", "rimphony_highfreq_batch", "rimphony_detmath_batch_device", "pkgw_bessel_j", "pkgw_bessel_dj",
    "rimphony_ctx_shared_mode", "rimphony_last_error", "rimphony_batch_compute_device_ex", "rimphony_batch_compute_ex",
    "rimphony_batch_compute_multi", "rimphony_status_histogram_device",
    "rimphony_hey_element_batch_device", "rimphony_hey_outer_batch_device", "rimphony_last_tail", "rimphony_deriv_probe_batch_device",
    "rimphony_ctx_device", "rimphony_batch_compute_multi_device", "rimphony_rccl_available", "rimphony_rccl_unique_id",
    "rimphony_rccl_comm_create", "rimphony_rccl_comm_destroy", "rimphony_rccl_gather_table",
]


class RimphonyError(RuntimeError):
    pass


class Work(ctypes.Structure):
    _fields_ = [("samples", c_uint64), ("passes", c_uint64), ("inner_qags", c_uint64),
                ("faraday_samples", c_uint64), ("faraday_passes", c_uint64), ("faraday_inner_qags", c_uint64)]


_lib = None


def load():
    """Load librimphony_hip.so (raises if it has not been built)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RimphonyError(
            "librimphony_hip.so not found at %s -- run `python -c 'import __graft_entry__ as g; g.build()'`; "
            "there is no CPU fallback" % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    dp = POINTER(c_double)
    lib.rimphony_dist_nparams.restype = c_int
    lib.rimphony_dist_nparams.argtypes = [c_int]
    lib.rimphony_ctx_create.restype = c_int
    lib.rimphony_ctx_create.argtypes = [c_int, POINTER(c_void_p)]
    lib.rimphony_ctx_destroy.restype = None
    lib.rimphony_ctx_destroy.argtypes = [c_void_p]
    lib.rimphony_strerror.restype = c_char_p
    lib.rimphony_strerror.argtypes = [c_int]
    lib.rimphony_version.restype = c_char_p
    lib.rimphony_last_work.restype = c_int
    lib.rimphony_last_work.argtypes = [c_void_p, POINTER(Work)]
    lib.rimphony_last_symphony_ms.restype = c_int
    lib.rimphony_last_symphony_ms.argtypes = [c_void_p, POINTER(ctypes.c_float)]
    lib.rimphony_last_faraday_ms.restype = c_int
    lib.rimphony_last_faraday_ms.argtypes = [c_void_p, POINTER(ctypes.c_float)]
    lib.rimphony_debug_counters.restype = c_int
    lib.rimphony_debug_counters.argtypes = [c_void_p, POINTER(c_uint64)]
    lib.rimphony_debug_heartbeat.restype = c_int
    lib.rimphony_debug_heartbeat.argtypes = [c_void_p, c_uint64, POINTER(POINTER(c_uint64))]
    lib.rimphony_batch_compute_device.restype = c_int
    lib.rimphony_batch_compute_device.argtypes = [c_void_p, c_int, c_size_t, c_void_p, c_void_p, POINTER(c_void_p),
                                                  c_uint32, c_void_p, c_void_p, c_void_p]
    lib.rimphony_batch_compute.restype = c_int
    lib.rimphony_batch_compute.argtypes = [c_void_p, c_int, c_size_t, dp, dp, POINTER(dp), c_uint32, dp,
                                           POINTER(c_int32)]
    lib.rimphony_batch_norm_device.restype = c_int
    lib.rimphony_batch_norm_device.argtypes = [c_void_p, c_int, c_size_t, POINTER(c_void_p), c_void_p, c_void_p]
    lib.rimphony_bessel_batch_device.restype = c_int
    lib.rimphony_bessel_batch_device.argtypes = [c_void_p, c_size_t, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]
    lib.rimphony_gamma_integrand_batch_device.restype = c_int
    lib.rimphony_gamma_integrand_batch_device.argtypes = [c_void_p, c_int, dp, c_int, c_int, c_double, c_double,
                                                          c_size_t, c_void_p, c_void_p, c_void_p, c_void_p]
    lib.rimphony_gamma_integral_batch_device.restype = c_int
    lib.rimphony_gamma_integral_batch_device.argtypes = [c_void_p, c_int, dp, c_int, c_int, c_int, c_double, c_double,
                                                         c_size_t, c_void_p, c_void_p, c_void_p]
    for name in ("pkgw_bessel_j", "pkgw_bessel_dj"):
        getattr(lib, name).restype = c_double
        getattr(lib, name).argtypes = [c_double, c_double]
    lib.rimphony_n_integral_batch_device.restype = c_int
    lib.rimphony_n_integral_batch_device.argtypes = [c_void_p, c_int, dp, c_int, c_int, c_int, c_double, c_double, c_size_t,
                                                     c_void_p, c_void_p, c_void_p, c_void_p]
    lib.rimphony_gamma_contribution_batch_device.restype = c_int
    lib.rimphony_gamma_contribution_batch_device.argtypes = [c_void_p, c_int, dp, c_int, c_int, c_double, c_double, c_size_t,
                                                             c_void_p, c_void_p, c_void_p]
    lib.rimphony_calc_f_batch_device.restype = c_int
    lib.rimphony_calc_f_batch_device.argtypes = [c_void_p, c_int, dp, c_double, c_size_t, c_void_p, c_void_p,
                                                 c_void_p, c_void_p, c_void_p, c_void_p]
    lib.rimphony_calc_f_batch.restype = c_int
    lib.rimphony_calc_f_batch.argtypes = [c_void_p, c_int, dp, c_double, c_size_t, dp, dp, dp, dp, dp]
    lib.rimphony_detmath_batch_device.restype = c_int
    lib.rimphony_detmath_batch_device.argtypes = [c_void_p, c_int, c_size_t, c_void_p, c_void_p, c_void_p, c_void_p]
    lib.rimphony_highfreq_batch.restype = c_int
    lib.rimphony_highfreq_batch.argtypes = [c_void_p, c_int, c_size_t, dp, dp, POINTER(dp), dp]
    lib.rimphony_highfreq_batch_device.restype = c_int
    lib.rimphony_highfreq_batch_device.argtypes = [c_void_p, c_int, c_size_t, c_void_p, c_void_p, POINTER(c_void_p), c_void_p, c_void_p]
    lib.rimphony_qag_selftest_device.restype = c_int
    lib.rimphony_qag_selftest_device.argtypes = [c_void_p, c_size_t, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                                 c_double, c_double, c_int32, c_void_p, c_void_p, c_void_p, c_void_p,
                                                 c_void_p]
    lib.rimphony_ctx_shared_mode.restype = c_int
    lib.rimphony_ctx_shared_mode.argtypes = [c_void_p]
    lib.rimphony_deriv_probe_batch_device.restype = c_int
    lib.rimphony_deriv_probe_batch_device.argtypes = [c_void_p, c_int, dp, c_int, c_int, c_int, c_double, c_double, c_size_t,
                                                      c_void_p, c_void_p, c_void_p]
    lib.rimphony_last_tail.restype = c_int
    lib.rimphony_last_tail.argtypes = [c_void_p, POINTER(c_uint64)]
    lib.rimphony_last_error.restype = c_char_p
    lib.rimphony_last_error.argtypes = []
    lib.rimphony_batch_compute_device_ex.restype = c_int
    lib.rimphony_batch_compute_device_ex.argtypes = [c_void_p, c_int, c_size_t, c_void_p, c_void_p, POINTER(c_void_p),
                                                     c_uint32, c_int, c_void_p, c_void_p, c_void_p, c_void_p]
    lib.rimphony_batch_compute_ex.restype = c_int
    lib.rimphony_batch_compute_ex.argtypes = [c_void_p, c_int, c_size_t, dp, dp, POINTER(dp), c_uint32, c_int, dp,
                                              POINTER(c_int32), POINTER(c_uint64)]
    lib.rimphony_batch_compute_multi.restype = c_int
    lib.rimphony_batch_compute_multi.argtypes = [POINTER(c_void_p), c_int, c_int, c_size_t, dp, dp, POINTER(dp), c_uint32,
                                                 c_int, dp, POINTER(c_int32), POINTER(c_uint64)]
    lib.rimphony_hey_element_batch_device.restype = c_int
    lib.rimphony_hey_element_batch_device.argtypes = [c_void_p, c_int, dp, c_int, c_double, c_double, c_int, c_size_t,
                                                      c_void_p, c_void_p, c_void_p, c_void_p]
    lib.rimphony_hey_outer_batch_device.restype = c_int
    lib.rimphony_hey_outer_batch_device.argtypes = [c_void_p, c_int, dp, c_int, c_double, c_double, c_int, c_size_t,
                                                    c_void_p, c_void_p, c_void_p]
    lib.rimphony_status_histogram_device.restype = c_int
    lib.rimphony_status_histogram_device.argtypes = [c_void_p, c_size_t, c_void_p, POINTER(c_uint64), c_void_p]
    # (an older build loaded through RIMPHONY_HIP_LIB for an A/B run -- tools/ab_libs.py -- predates these entries)
    if hasattr(lib, "rimphony_ctx_device"):
        lib.rimphony_ctx_device.restype = c_int
        lib.rimphony_ctx_device.argtypes = [c_void_p, POINTER(c_int)]
        lib.rimphony_batch_compute_multi_device.restype = c_int
        lib.rimphony_batch_compute_multi_device.argtypes = [POINTER(c_void_p), c_int, c_int, POINTER(c_size_t), POINTER(c_void_p),
                                                            POINTER(c_void_p), POINTER(POINTER(c_void_p)), c_uint32, c_int,
                                                            POINTER(c_void_p), POINTER(c_void_p), POINTER(c_void_p),
                                                            POINTER(c_void_p), c_int]
        lib.rimphony_rccl_available.restype = c_int
        lib.rimphony_rccl_available.argtypes = []
        lib.rimphony_rccl_unique_id.restype = c_int
        lib.rimphony_rccl_unique_id.argtypes = [c_void_p]
        lib.rimphony_rccl_comm_create.restype = c_int
        lib.rimphony_rccl_comm_create.argtypes = [c_void_p, c_int, c_int, c_void_p, POINTER(c_void_p)]
        lib.rimphony_rccl_comm_destroy.restype = c_int
        lib.rimphony_rccl_comm_destroy.argtypes = [c_void_p]
        lib.rimphony_rccl_gather_table.restype = c_int
        lib.rimphony_rccl_gather_table.argtypes = [c_void_p, c_void_p, c_int, c_int, c_int, c_size_t, c_void_p, c_void_p,
                                                   c_void_p, c_void_p]
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        lib = load()
        msg = lib.rimphony_strerror(rc).decode()
        # the thread's last HIP failure only describes this call when this call failed in HIP (RIMPHONY_EHIP = -2)
        # (RIMPHONY_ERCCL = -7 and a librccl that could not be loaded, -6, leave their text there too)
        detail = lib.rimphony_last_error().decode() if rc in (-2, -6, -7) else ""
        raise RimphonyError("%s failed: %s (code %d)%s" % (what, msg, rc, " -- " + detail if detail else ""))
