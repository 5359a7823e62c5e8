"""ctypes binding of liboptex_hip.so (include/optex.h).  There is NO fallback: if the library is missing or no
MI355X is visible, every op raises."""
import ctypes
import os

import torch  # imported BEFORE the CDLL so the library binds to the HIP runtime instance torch already loaded

_HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.path.join(_HERE, "liboptex_hip.so")
ABI_VERSION = 10
CHANNEL_MAJOR, PIXEL_MAJOR = 0, 1

_c = ctypes
_P, _L, _I, _F, _SZ, _U = _c.c_void_p, _c.c_long, _c.c_int, _c.c_float, _c.c_size_t, _c.c_uint

# the `flags` word of ABI 10 (include/optex.h): per-call choices, 0 = defaults
F_DEFAULT, F_CDF_TWO_KERNEL, F_SORT_RANK4 = 0, 0x100, 0x200


def f_spare_cus(n: int) -> int:
    """OPTEX_F_SPARE_CUS(n): the persistent rotation GEMM of THIS call leaves n CUs out of its grid"""
    return (int(n) + 1) & 0xff


# name -> (restype, argtypes); mirrors include/optex.h one to one (tests/test_abi.py checks the export list)
SIGNATURES = {
    "optex_abi_version": (_I, []),
    "optex_last_error": (_c.c_char_p, []),
    "optex_device_info": (_I, [_P, _P, _P]),
    "optex_gemm_spare_cus": (_I, [_I]),
    "optex_cdf_fused": (_I, [_I]),
    "optex_gemm_tn": (_I, [_P, _L, _L, _P, _L, _L, _I, _P, _L, _L, _I, _I, _I, _L, _I, _P, _L, _P, _L, _P, _F, _U, _P]),
    "optex_col_minmax": (_I, [_P, _L, _L, _L, _I, _I, _P, _P, _P]),
    "optex_col_histc": (_I, [_P, _L, _L, _L, _I, _I, _P, _P, _P, _P]),
    "optex_interp": (_I, [_P, _L, _P, _P, _L, _P, _P]),
    "optex_cdf_ws_bytes": (_SZ, [_I, _I]),
    "optex_cdf_match": (_I, [_P, _L, _L, _L, _P, _L, _L, _L, _I, _I, _I, _P, _L, _L, _P, _SZ, _P, _U, _P]),
    "optex_cdf_bins_ws_bytes": (_SZ, [_I, _I, _I]),
    "optex_cdf_match_bins": (_I, [_P, _L, _L, _L, _P, _L, _L, _L, _I, _I, _I, _I, _P, _L, _L, _P, _SZ, _P]),
    "optex_sort_ws_bytes": (_SZ, [_L, _I, _I]),
    "optex_sort_columns": (_I, [_P, _L, _L, _L, _I, _I, _P, _P, _P, _SZ, _P]),
    "optex_sort_match_ws_bytes": (_SZ, [_L, _L, _I, _I, _I]),
    "optex_sort_match": (_I, [_P, _L, _L, _L, _P, _L, _L, _L, _I, _I, _I, _P, _L, _L, _P, _SZ, _U, _P]),
    "optex_linear_stats_ws_bytes": (_SZ, [_L, _I, _I]),
    "optex_linear_stats": (_I, [_P, _L, _L, _L, _I, _I, _I, _F, _P, _P, _P, _SZ, _P]),
    "optex_chol_ld": (_I, [_I]),
    "optex_chol_inv": (_I, [_P, _L, _I, _I, _P, _P, _P]),
    "optex_spd_sqrt_ws_bytes": (_SZ, [_I, _I]),
    "optex_spd_sqrt": (_I, [_P, _L, _I, _I, _F, _P, _P, _P, _SZ, _P]),
    "optex_transfer_operator_ws_bytes": (_SZ, [_I, _I, _I, _I]),
    "optex_transfer_operator": (_I, [_I, _P, _P, _I, _I, _I, _F, _P, _P, _SZ, _P]),
    "optex_rotation_normals": (_L, [_I]),
    "optex_rotation_ws_bytes": (_SZ, [_I, _I]),
    "optex_mt19937_state_bytes": (_SZ, []),
    "optex_mt19937_seed": (_I, [_P, _I, _c.c_uint32, _c.c_uint32, _P]),
    "optex_legacy_normals_ws_bytes": (_SZ, [_I, _L]),
    "optex_legacy_normals": (_I, [_P, _I, _L, _P, _L, _P, _SZ, _P]),
    "optex_rotations_from_normals": (_I, [_P, _I, _I, _P, _P, _P, _P, _SZ, _P]),
    "optex_ot_loop_ws_bytes": (_SZ, [_I, _L, _L, _I, _I, _I, _I, _I, _L]),
    "optex_ot_loop": (_I, [_I, _P, _L, _I, _P, _L, _I, _I, _P, _P, _L, _I, _P, _F, _I, _P, _SZ, _U, _P]),
    "optex_ot_loop_pca_ws_bytes": (_SZ, [_I, _L, _L, _I, _I, _I, _I, _I]),
    "optex_ot_loop_pca": (_I, [_I, _P, _I, _P, _P, _L, _I, _P, _L, _I, _I, _P, _P, _I, _P, _F, _P, _SZ, _U, _P]),
    "optex_vgg_glue": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    "optex_vgg_glue_layout": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    "optex_prof_enable": (_I, [_I]),
    "optex_prof_num_classes": (_I, []),
    "optex_prof_class_name": (_c.c_char_p, [_I]),
    "optex_prof_collect": (_I, [_I, _P, _P, _P, _P]),
}

_lib = None


def load():
    """dlopen the library and set prototypes (no GPU needed for this step)."""
    global _lib
    if _lib is None:
        if not os.path.exists(SO_PATH):
            raise RuntimeError(
                f"{SO_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(or `make -C optimaltextures_amd/csrc`). optimaltextures_amd has no CPU fallback.")
        lib = ctypes.CDLL(SO_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
        if lib.optex_abi_version() != ABI_VERSION:
            raise RuntimeError(f"liboptex_hip.so ABI {lib.optex_abi_version()} != expected {ABI_VERSION}; rebuild")
        _lib = lib
    return _lib


def lib():
    """The library, for compute: additionally requires a visible GPU."""
    if not torch.cuda.is_available():
        raise RuntimeError("optimaltextures_amd needs an MI355X (torch.cuda.is_available() is False); "
                           "there is no CPU fallback for the HIP path")
    return load()


def check(rc):
    if rc != 0:
        raise RuntimeError(load().optex_last_error().decode() or f"liboptex_hip error {rc}")


def stream_ptr():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def workspace(nbytes, device):
    """scratch from torch's caching allocator (stream-ordered reuse, no hipMalloc on the hot path)"""
    return torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)
