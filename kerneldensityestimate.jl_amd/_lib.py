"""ctypes binding of libkdehip.so (the C ABI declared in include/kdehip.h).

There is no Python/CPU fallback: if the shared library is missing this module raises at import, and
every compute entry point fails with KDEHIP_ERR_NO_DEVICE when no MI355X is usable.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("KDEHIP_LIB", os.path.join(_HERE, "libkdehip.so"))  # KDEHIP_LIB: diagnostic builds

f64p = C.POINTER(C.c_double)
i64p = C.POINTER(C.c_int64)
i32p = C.POINTER(C.c_int32)
u8p = C.POINTER(C.c_uint8)

KDEHIP_OK = 0
ERR_ARG, ERR_DIM_MISMATCH, ERR_RAND_SHORT, ERR_NO_DEVICE, ERR_HIP, ERR_ALLOC, ERR_UNSUPPORTED = -1, -2, -3, -4, -5, -6, -7
MAX_DIMS, MAX_DENS = 8, 16


class CDensity(C.Structure):
    """struct kdehip_density"""
    _fields_ = [
        ("npts", C.c_int64), ("ndim", C.c_int64),
        ("means", f64p), ("bandwidth", f64p), ("weights", f64p),
        ("left_child", i64p), ("right_child", i64p), ("permutation", i64p),
    ]


class CBatchItem(C.Structure):
    """struct kdehip_batch_item"""
    _fields_ = [
        ("Ndens", C.c_int32), ("Niter", C.c_int32), ("trees", C.POINTER(C.c_void_p)), ("Np", C.c_int64),
        ("seed", C.c_uint64), ("sample_offset", C.c_int64), ("addEntropy", C.c_int32), ("reserved_", C.c_int32),
        ("partialDimMask", u8p), ("d_points", C.c_void_p), ("d_indices", C.c_void_p), ("d_labels", C.c_void_p),
    ]


class CMulItem(C.Structure):
    """struct kdehip_mul_item"""
    _fields_ = [("Ndens", C.c_int32), ("addEntropy", C.c_int32), ("trees", C.POINTER(C.c_void_p)), ("seed", C.c_uint64)]


class CProductInfo(C.Structure):
    """struct kdehip_product_info_t"""
    _fields_ = [
        ("ndens", C.c_int32), ("ndims", C.c_int32), ("nlevels", C.c_int32), ("precision", C.c_int32),
        ("nodes_per_sweep", C.c_int64), ("bytes_per_eval", C.c_int64), ("packed_bytes", C.c_int64),
        ("fast_math_path", C.c_int32), ("device", C.c_int32),
    ]


# every symbol include/kdehip.h declares: (restype, argtypes)
SIGNATURES = {
    "kdehip_version": (C.c_int, []),
    "kdehip_last_error": (C.c_char_p, []),
    "kdehip_device_count": (C.c_int, []),
    "kdehip_clear_cache": (None, []),
    "kdehip_gibbs1": (C.c_int, [C.c_int, C.POINTER(CDensity), C.c_int64, C.c_int, f64p, i64p, f64p, C.c_int64,
                                f64p, C.c_int64, C.c_int, C.c_int, u8p, C.c_int]),
    "kdehip_gibbs1_trace": (C.c_int, [C.c_int, C.POINTER(CDensity), C.c_int64, C.c_int, f64p, i64p, f64p, C.c_int64,
                                      f64p, C.c_int64, C.c_int, C.c_int, u8p, C.c_int, i32p]),
    "kdehip_gibbs1_multi": (C.c_int, [C.c_int, C.POINTER(CDensity), C.c_int64, C.c_int, f64p, i64p, f64p, C.c_int64,
                                      f64p, C.c_int64, C.c_int, C.c_int, u8p, C.c_int, C.c_int, i32p]),
    "kdehip_gibbs1_manifold": (C.c_int, [C.c_int, C.POINTER(CDensity), C.c_int64, C.c_int, f64p, i64p, f64p, C.c_int64,
                                         f64p, C.c_int64, C.c_int, C.c_int, u8p, u8p, C.c_int, i32p]),
    "kdehip_prod_philox": (C.c_int, [C.c_int, C.POINTER(CDensity), C.c_int64, C.c_int, f64p, i64p, C.c_uint64, C.c_int,
                                     C.c_int, u8p, C.c_int, C.c_int, C.c_int, i32p]),
    "kdehip_product_multi_create": (C.c_int, [C.POINTER(C.c_void_p), C.c_int, C.POINTER(CDensity), C.c_int, u8p,
                                              C.c_int, C.c_int, C.c_int]),
    "kdehip_product_multi_destroy": (None, [C.c_void_p]),
    "kdehip_product_multi_ngpus": (C.c_int, [C.c_void_p]),
    "kdehip_product_multi_plan": (C.c_void_p, [C.c_void_p, C.c_int]),
    "kdehip_product_multi_sample_philox": (C.c_int, [C.c_void_p, C.c_int64, C.c_int, C.c_uint64, C.c_int64, C.c_int,
                                                     C.POINTER(C.c_void_p), C.POINTER(C.c_void_p),
                                                     C.POINTER(C.c_void_p)]),
    "kdehip_product_multi_transfers_per_product": (C.c_int, [C.c_void_p]),
    "kdehip_product_create": (C.c_int, [C.POINTER(C.c_void_p), C.c_int, C.POINTER(CDensity), C.c_int, u8p,
                                        C.c_int, C.c_int]),
    "kdehip_product_destroy": (None, [C.c_void_p]),
    "kdehip_product_info": (C.c_int, [C.c_void_p, C.POINTER(CProductInfo)]),
    "kdehip_product_randu_per_sample": (C.c_int64, [C.c_void_p, C.c_int]),
    "kdehip_product_randn_per_sample": (C.c_int64, [C.c_void_p]),
    "kdehip_product_fallback_count": (C.c_int64, [C.c_void_p]),
    "kdehip_product_sample_streams": (C.c_int, [C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_int64,
                                                C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_void_p,
                                                C.c_void_p, C.c_void_p]),
    "kdehip_product_sample_philox": (C.c_int, [C.c_void_p, C.c_int64, C.c_int, C.c_uint64, C.c_int64, C.c_int,
                                               C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "kdehip_product_sample_philox_host": (C.c_int, [C.c_void_p, C.c_int64, C.c_int, C.c_uint64, C.c_int64,
                                                    C.c_int, f64p, i64p, i32p]),
    "kdehip_product_screen_stats": (C.c_int, [C.c_void_p, i32p, i64p, i64p]),
    "kdehip_product_set_variant": (C.c_int, [C.c_void_p, C.c_int]),
    "kdehip_product_launch_geometry": (C.c_int, [C.c_void_p, C.c_int64, i32p, i32p]),
    "kdehip_product_kernel_name": (C.c_char_p, [C.c_void_p, C.c_int64]),
    "kdehip_density_upload": (C.c_int, [C.POINTER(C.c_void_p), C.POINTER(CDensity), C.c_int]),
    "kdehip_density_free": (None, [C.c_void_p]),
    "kdehip_density_npts": (C.c_int64, [C.c_void_p]),
    "kdehip_density_ndim": (C.c_int, [C.c_void_p]),
    "kdehip_prod_philox_device": (C.c_int, [C.c_int, C.POINTER(C.c_void_p), C.c_int64, C.c_int, C.c_uint64, C.c_int64,
                                            C.c_int, u8p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "kdehip_prod_philox_resident": (C.c_int, [C.c_int, C.POINTER(C.c_void_p), C.c_int64, C.c_int, C.c_uint64, C.c_int, u8p,
                                              C.c_int, f64p, i64p]),
    "kdehip_philox_fill_uniform": (None, [C.c_uint64, C.c_int64, C.c_int64, C.c_int64, f64p]),
    "kdehip_philox_fill_normal": (None, [C.c_uint64, C.c_int64, C.c_int64, C.c_int64, f64p]),
    "kdehip_evaluate": (C.c_int, [C.POINTER(CDensity), f64p, C.c_int64, C.c_int, f64p, C.c_int]),
    "kdehip_auto_bandwidth": (C.c_int, [C.c_int64, C.c_int64, f64p, f64p, i32p, C.c_int]),
    "kdehip_make_density_device_supported": (C.c_int, [C.c_int64, C.c_int64]),
    "kdehip_make_densities_device": (C.c_int, [C.c_int, C.c_int64, i64p] + [C.POINTER(C.c_void_p)] * 2 + [C.c_int64] +
                                     [C.POINTER(C.c_void_p)] * 13 + [C.c_int]),
    "kdehip_density_set_bandwidth": (C.c_int, [C.c_int64, C.c_int64, f64p, C.c_int64, f64p, i64p, i64p, f64p, f64p, f64p, f64p]),
    "kdehip_profile_sampler": (None, [C.c_int]),
    "kdehip_profile_phase_read": (C.c_int, [C.c_int, f64p, i64p]),
    "kdehip_selftest_fp32": (C.c_int, [C.c_int, C.c_uint32, C.c_uint64, C.c_int, f64p, C.POINTER(C.c_uint32),
                                      C.POINTER(C.c_uint32)]),
    "kdehip_profile_sampler_read": (C.c_int, [C.c_int, C.c_void_p, f64p, i64p]),
    "kdehip_product_multi_timing": (C.c_int, [C.c_void_p, f64p, f64p]),
    "kdehip_density_from_device_points": (C.c_int, [C.POINTER(C.c_void_p), C.c_void_p, C.c_int64, C.c_int64, C.c_int,
                                                    C.c_void_p, f64p, i32p]),
    "kdehip_mul_device": (C.c_int, [C.POINTER(C.c_void_p), C.c_int, C.POINTER(C.c_void_p), C.c_uint64, C.c_int, f64p, i32p]),
    "kdehip_mul_device_batch": (C.c_int, [C.c_int, C.POINTER(CMulItem), C.POINTER(C.c_void_p), f64p, i32p]),
    "kdehip_density_download": (C.c_int, [C.c_void_p, f64p, f64p, f64p, i64p, i64p, i64p, i64p, i64p, f64p, f64p, f64p,
                                          f64p, f64p]),
    "kdehip_prod_philox_batch": (C.c_int, [C.c_int, C.POINTER(CBatchItem), C.c_int, C.c_void_p]),
    "kdehip_make_density_auto": (C.c_int, [C.c_int64, C.c_int64, f64p, f64p, i32p, C.c_int, f64p, f64p, f64p, i64p, i64p, i64p,
                                           i64p, i64p, f64p, f64p, f64p, f64p]),
    "kdehip_make_density": (C.c_int, [C.c_int64, C.c_int64, f64p, f64p, C.c_int64, f64p, f64p, f64p, f64p,
                                      i64p, i64p, i64p, i64p, i64p, f64p, f64p, f64p, f64p]),
}


class KdeHipError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libkdehip error {code}: {msg}")
        self.code = code


def _share_hip_runtime_with_torch():
    """PyTorch-ROCm wheels bundle their own libamdhip64.so (no SONAME), while libkdehip.so links the
    system one (libamdhip64.so.7).  Two HIP runtimes in one process fight over the device: whichever
    initialises second sees "No HIP GPUs".  Binding libkdehip's hip* symbols to torch's copy (global
    scope, loaded first) keeps ONE runtime whatever the import order.  No torch -> system runtime."""
    if os.environ.get("KDEHIP_NO_TORCH_HIP") == "1":
        return
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
        if spec is None or not spec.submodule_search_locations:
            return
        cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
        if os.path.exists(cand):
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
    except Exception:  # noqa: BLE001  (purely a courtesy; the system runtime still works on its own)
        pass


def _load():
    _share_hip_runtime_with_torch()
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  This package has no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the library does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    return lib


lib = _load()


def check(rc: int):
    if rc != KDEHIP_OK:
        msg = lib.kdehip_last_error().decode("utf-8", "replace")
        if rc == ERR_DIM_MISMATCH:
            raise ValueError(msg)  # Julia: error("kdes must have same dimension")
        if rc == ERR_RAND_SHORT:
            raise IndexError(msg)  # Julia: BoundsError
        raise KdeHipError(rc, msg)


def ptr(a, t):
    return a.ctypes.data_as(t)
