"""ctypes binding of ``libfv3hip.so`` (C ABI declared in ``include/fv3hip.h``).

The library is built in-tree by ``make -C fv3net_amd/csrc`` (or ``__graft_entry__.build()``).
If it is missing, every entry point raises :class:`ExtensionMissingError` -- the product path
never falls back to a CPU implementation.
"""
import ctypes
import os
from ctypes import POINTER, c_char_p, c_double, c_float, c_int, c_int64, c_size_t, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
# FV3HIP_LIBRARY points at an alternative build of the same ABI (diagnostic builds with stamps)
LIB_PATH = os.environ.get("FV3HIP_LIBRARY") or os.path.join(_HERE, "libfv3hip.so")

F32, F64, I32, I64 = 0, 1, 2, 3
OP_SUM, OP_MEAN, OP_MIN, OP_MAX, OP_MEDIAN, OP_MODE = range(6)
NAN_SKIP, NAN_PROPAGATE, NAN_OMIT = range(3)
LAYOUT_COL_LEVEL, LAYOUT_LEVEL_COL = 0, 1
TRANSFORM_NONE, TRANSFORM_LOG = 0, 1
ACT_LINEAR, ACT_RELU = 0, 1
ARITH_EXACT, ARITH_FAST = 0, 1
ABI_VERSION = 3

OK, EINVAL, EUNSUPPORTED, EHIP, ENOMEM = 0, -1, -2, -3, -4


class ExtensionMissingError(ImportError):
    pass


class Fv3HipError(RuntimeError):
    def __init__(self, code, message):
        super().__init__(f"libfv3hip error {code}: {message}")
        self.code = code


class DeviceInfo(ctypes.Structure):
    _fields_ = [
        ("name", ctypes.c_char * 128),
        ("arch", ctypes.c_char * 64),
        ("compute_units", c_int),
        ("wavefront_size", c_int),
        ("lds_bytes_per_cu", c_int),
        ("clock_mhz", c_int),
        ("hbm_bytes", c_size_t),
    ]


class MlpDesc(ctypes.Structure):
    _fields_ = [
        ("n_sources", c_int),
        ("n_inputs", c_int),
        ("in_source", POINTER(c_int)),
        ("in_feat_start", POINTER(c_int)),
        ("in_nfeat", POINTER(c_int)),
        ("in_transform", POINTER(c_int)),
        ("in_eps", POINTER(c_float)),
        ("in_center", POINTER(c_float)),
        ("in_scale", POINTER(c_float)),
        ("n_hidden", c_int),
        ("width", c_int),
        ("hidden_activation", c_int),
        ("hidden_kernels", POINTER(POINTER(c_float))),
        ("hidden_biases", POINTER(POINTER(c_float))),
        ("n_outputs", c_int),
        ("out_nfeat", POINTER(c_int)),
        ("out_kernel", POINTER(c_float)),
        ("out_bias", POINTER(c_float)),
        ("out_scale", POINTER(c_float)),
        ("out_center", POINTER(c_float)),
        ("out_min", POINTER(c_float)),
        ("out_max", POINTER(c_float)),
        ("out_mask", POINTER(c_float)),
        ("n_residual", c_int),
        ("res_source", POINTER(c_int)),
        ("res_output", POINTER(c_int)),
        ("hidden_output", c_int),
    ]


# name -> (restype, argtypes); every name here must be declared in include/fv3hip.h
SIGNATURES = {
    "fv3hip_last_error": (c_char_p, []),
    "fv3hip_abi_version": (c_int, []),
    "fv3hip_spin": (c_int, [c_int64, c_int, c_void_p]),
    "fv3hip_init": (c_int, [c_int]),
    "fv3hip_device_info": (c_int, [POINTER(DeviceInfo)]),
    "fv3hip_weighted_block_average": (
        c_int,
        [c_void_p, c_int, c_void_p, c_int, c_int64, c_int, c_int, c_int64, c_int, c_void_p, c_void_p],
    ),
    "fv3hip_mass_weighted_block_average": (
        c_int,
        [c_void_p, c_int, c_int, c_void_p, c_void_p, c_int, c_int64, c_int, c_int, c_int64, c_int, c_void_p, c_void_p],
    ),
    "fv3hip_edge_weighted_block_average": (
        c_int,
        [c_void_p, c_int, c_void_p, c_int, c_int64, c_int, c_int, c_int64, c_int, c_int, c_void_p, c_void_p],
    ),
    "fv3hip_block_reduce": (
        c_int,
        [c_void_p, c_int, c_int64, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p],
    ),
    "fv3hip_block_upsample": (c_int, [c_void_p, c_int, c_int64, c_int, c_int, c_int, c_void_p, c_void_p]),
    "fv3hip_zc_squash": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int64, c_double, c_int, c_void_p, c_void_p, c_void_p]),
    "fv3hip_zc_infer_cloud": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_int, c_int64, c_int, c_void_p, c_void_p]),
    "fv3hip_zc_gscond_conserve": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int, c_int, c_void_p, c_int,
                                          c_int, c_int, c_int64, c_int64, c_int, c_int, c_void_p, c_void_p, c_void_p,
                                          c_void_p]),
    "fv3hip_zc_precpd_conserve": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int,
                                          c_int64, c_int64, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "fv3hip_zc_precip_simple": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int, c_int64, c_int64,
                                        c_int, c_void_p, c_void_p]),
    "fv3hip_zc_class_zero": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int64, c_void_p, c_void_p]),
    "fv3hip_non_negative_sphum": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int64, c_double, c_int, c_void_p, c_void_p,
                                          c_void_p]),
    "fv3hip_clamp": (c_int, [c_void_p, c_int, c_int64, c_double, c_double, c_int, c_int, c_void_p, c_void_p]),
    "fv3hip_level_fill": (c_int, [c_void_p, c_int, c_void_p, c_int, c_double, c_int64, c_int64, c_int64, c_int64, c_void_p,
                                  c_void_p]),
    "fv3hip_column_sum": (c_int, [c_void_p, c_int, c_int64, c_int, c_int64, c_double, c_void_p, c_void_p]),
    "fv3hip_blend_weights": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int64, c_int, c_int64, c_void_p, c_void_p]),
    "fv3hip_hydrostatic_balance": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int64, c_int, c_int64,
                                           c_double, c_void_p, c_void_p, c_void_p]),
    "fv3hip_mappm_multi": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int, c_int64, c_int64, c_int, c_int, c_int, c_int,
                                   c_int, c_int, c_void_p, c_size_t, c_void_p]),
    "fv3hip_mappm_multi_coarse_target": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int, c_int64, c_int, c_int, c_int, c_int, c_int,
                                                 c_int, c_int, c_int, c_void_p, c_size_t, c_void_p]),
    "fv3hip_mappm_block_mean_workspace_bytes": (c_size_t, [c_int64, c_int]),
    "fv3hip_mappm_block_mean": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_int64, c_void_p, c_void_p,
                                        c_int, c_int64, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_size_t, c_void_p]),
    "fv3hip_mask_weights_coarse": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_void_p, c_int, c_int64, c_int, c_int, c_int, c_int,
                                           c_int64, c_void_p, c_void_p]),
    "fv3hip_level_scale": (c_int, [c_void_p, c_int, c_void_p, c_int64, c_int, c_int64, c_void_p, c_void_p]),
    "fv3hip_member_reduce": (c_int, [c_void_p, c_int, c_int, c_int, c_int64, c_void_p, c_void_p]),
    "fv3hip_tendency_to_flux": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int64, c_int, c_int64, c_int, c_int,
                                        c_void_p, c_void_p, c_void_p]),
    "fv3hip_flux_to_tendency": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int64, c_int, c_int64, c_void_p, c_void_p]),
    "fv3hip_minmax_score": (c_int, [c_void_p, c_int, c_int64, c_int64, c_int, c_void_p, c_void_p, c_int64, c_int, c_int, c_void_p,
                                    c_void_p, c_void_p, c_void_p]),
    "fv3hip_ocsvm_score": (c_int, [c_void_p, c_int, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_double, c_void_p, c_void_p]),
    "fv3hip_local_pack": (c_int, [c_void_p, c_int, c_int, c_int, c_double, c_void_p, c_void_p, c_int, c_int64, c_void_p, c_void_p]),
    "fv3hip_local_unpack": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int,
                                    c_double, c_void_p, c_int, c_int, c_double, c_double, c_double, c_double, c_int, c_int64,
                                    c_void_p, c_void_p, c_void_p, c_void_p]),
    "fv3hip_classify_onehot": (c_int, [c_void_p, c_int, c_int, c_int64, c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "fv3hip_interpolate_2d": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int, c_int, c_double, c_int, c_void_p,
                                      c_void_p]),
    "fv3hip_ew": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_double, c_int, c_int64, c_int64, c_int64, c_int64, c_void_p,
                          c_void_p]),
    "fv3hip_cube_edge_rows": (c_int, [c_void_p, c_int, c_int, c_int64, c_int, c_void_p, c_void_p]),
    "fv3hip_halo_pick": (c_int, [c_void_p, c_int, c_int, c_int64, c_int, POINTER(c_int), POINTER(c_int), POINTER(c_int), c_void_p, c_void_p]),
    "fv3hip_cast": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int64, c_void_p]),
    "fv3hip_cast_many": (c_int, [c_void_p, POINTER(c_int), c_void_p, c_int, POINTER(c_int64), c_int, c_void_p]),
    "fv3hip_interp_center_to_outer": (c_int, [c_void_p, c_int, c_int64, c_int, c_int, c_int, c_void_p, c_void_p,
                                              c_void_p, c_void_p]),
    "fv3hip_interp_center_to_outer_lines": (c_int, [c_void_p, c_int, c_int64, c_int, c_int, c_int, c_int, c_void_p, c_void_p,
                                                    c_void_p, c_void_p]),
    "fv3hip_weighted_window_average": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int64, c_int, c_int, c_int64, c_int, c_int,
                                               c_int, c_int, c_void_p, c_void_p]),
    "fv3hip_repeat": (c_int, [c_void_p, c_int, c_int64, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "fv3hip_pressure_at_interface": (
        c_int,
        [c_void_p, c_int, c_int64, c_int, c_int64, c_double, c_void_p, c_void_p],
    ),
    "fv3hip_pressure_at_midpoint_log": (
        c_int,
        [c_void_p, c_int, c_int64, c_int, c_int64, c_double, c_void_p, c_void_p],
    ),
    "fv3hip_mask_weights": (
        c_int,
        [c_void_p, c_int, c_void_p, c_int, c_int, c_void_p, c_int, c_int64, c_int, c_int64, c_int64, c_void_p,
         c_void_p],
    ),
    "fv3hip_mappm_workspace_bytes": (c_size_t, [c_int64, c_int]),
    "fv3hip_mappm": (
        c_int,
        [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int64, c_int64, c_int, c_int, c_int, c_int, c_int, c_int,
         c_void_p, c_size_t, c_void_p],
    ),
    "fv3hip_mlp_create": (c_int, [POINTER(MlpDesc), POINTER(c_void_p)]),
    "fv3hip_mlp_destroy": (c_int, [c_void_p]),
    "fv3hip_mlp_predict": (
        c_int,
        [c_void_p, POINTER(c_void_p), POINTER(c_int), POINTER(c_int64), POINTER(c_int64), c_int64,
         POINTER(c_void_p), c_int, POINTER(c_int64), POINTER(c_int64), c_void_p],
    ),
    "fv3hip_mlp_flops_per_sample": (c_int64, [c_void_p]),
    "fv3hip_mlp_set_small_limit": (c_int, [c_void_p, c_int64]),
    "fv3hip_mlp_last_variant": (c_char_p, [c_void_p]),
    "fv3hip_mlp3_create": (c_int, [POINTER(MlpDesc), POINTER(c_void_p)]),
    "fv3hip_mlp3_destroy": (c_int, [c_void_p]),
    "fv3hip_mlp3_flops_per_sample": (c_int64, [c_void_p]),
    "fv3hip_mlp3_predict": (c_int, [c_void_p, POINTER(c_void_p), POINTER(c_int64), c_int64, POINTER(c_void_p), POINTER(c_int64), c_void_p]),
    "fv3hip_timer_create": (c_int, [POINTER(c_void_p)]),
    "fv3hip_timer_start": (c_int, [c_void_p, c_void_p]),
    "fv3hip_timer_stop": (c_int, [c_void_p, c_void_p]),
    "fv3hip_timer_elapsed_ms": (c_int, [c_void_p, POINTER(c_float)]),
    "fv3hip_timer_destroy": (c_int, [c_void_p]),
}

_lib = None


def load():
    """Load libfv3hip.so (once) and attach the ABI signatures.  Raises if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ExtensionMissingError(
            f"{LIB_PATH} is not built.  Run `make -C fv3net_amd/csrc` (needs hipcc) or "
            "`python -c 'import __graft_entry__ as g; g.build()'`.  There is no CPU fallback."
        )
    lib = ctypes.CDLL(LIB_PATH)
    for name, (restype, argtypes) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError = the .so does not export a declared symbol
        fn.restype = restype
        fn.argtypes = argtypes
    if lib.fv3hip_abi_version() != ABI_VERSION:
        raise ExtensionMissingError(f"{LIB_PATH} has ABI version {lib.fv3hip_abi_version()}, expected {ABI_VERSION}")
    _lib = lib
    return lib


def check(code):
    if code != OK:
        raise Fv3HipError(code, load().fv3hip_last_error().decode("utf-8", "replace"))


def call(name, *args):
    """Call an int-returning entry point and raise Fv3HipError on a non-zero status."""
    check(getattr(load(), name)(*args))


_torch = None


def call_on(where, name, *args):
    """``call`` with the HIP device of ``where`` (a torch device or tensor) current for the duration of the call, and the
    caller's device restored afterwards: the library launches on, and allocates its scratch on, the current device."""
    global _torch
    if _torch is None:
        import torch as _torch_module

        _torch = _torch_module
    dev = getattr(where, "device", where)
    idx = None if dev is None else getattr(dev, "index", None)
    lib = _lib or load()
    if idx is None or idx == _torch.cuda.current_device():
        code = getattr(lib, name)(*args)
        if code != OK:
            check(code)
        return
    with _torch.cuda.device(idx):
        check(getattr(lib, name)(*args))
