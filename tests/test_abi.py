"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports exactly what include/optex.h declares,
the ctypes prototypes cover every export, and the product path fails loudly without a GPU (no CPU fallback)."""
import os
import re

import pytest
import torch

from conftest import ROOT
from optimaltextures_amd import _lib


def header_functions():
    src = open(os.path.join(ROOT, "include", "optex.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(optex_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    names = header_functions()
    assert len(names) >= 19
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/optex.h but not exported by liboptex_hip.so"
    assert sorted(_lib.SIGNATURES) == names, "ctypes prototypes and header disagree"
    assert lib.optex_abi_version() == _lib.ABI_VERSION == 10


def test_ctypes_prototypes_have_the_headers_argument_counts():
    """every prototype of include/optex.h against _lib.SIGNATURES: the same number of arguments, `unsigned flags` (ABI 10) where
    the header has it and in the same position — a mismatch here is a silently shifted argument on the GPU box"""
    src = open(os.path.join(ROOT, "include", "optex.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    protos = dict(re.findall(r"\b(optex_[a-z0-9_]+)\s*\(([^;]*?)\)\s*;", src, flags=re.S))
    assert sorted(protos) == sorted(_lib.SIGNATURES)
    for name, params in protos.items():
        params = [q.strip() for q in params.split(",")] if params.strip() not in ("", "void") else []
        argtypes = _lib.SIGNATURES[name][1]
        assert len(params) == len(argtypes), f"{name}: header has {len(params)} parameters, ctypes {len(argtypes)}"
        for q, a in zip(params, argtypes):
            if q == "unsigned flags":
                assert a is _lib._U, f"{name}: `unsigned flags` bound as {a}"
            elif not q.startswith("uint32_t"):   # (ctypes.c_uint32 IS c_uint)
                assert a is not _lib._U, f"{name}: {q} bound as c_uint"
    flagged = sorted(n for n, q in protos.items() if "unsigned flags" in q)
    assert flagged == ["optex_cdf_match", "optex_gemm_tn", "optex_ot_loop", "optex_ot_loop_pca", "optex_sort_match"]


def test_call_flags_are_thread_local_and_nest():
    """ops.call_flags (the Python side of ABI 10's per-call flags): nested blocks combine, another thread sees nothing"""
    import threading
    from optimaltextures_amd import ops
    seen = {}
    with ops.call_flags(ops.f_spare_cus(0)):
        assert ops._flags(None).value == 1
        with ops.call_flags(ops.F_CDF_TWO_KERNEL):
            assert ops._flags(None).value == (1 | 0x100)
            with ops.call_flags(ops.f_spare_cus(3) | ops.F_SORT_RANK4):
                assert ops._flags(None).value == (4 | 0x100 | 0x200)
            t = threading.Thread(target=lambda: seen.setdefault("other", ops._flags(None).value))
            t.start()
            t.join()
        assert ops._flags(None).value == 1 and ops._flags(0x200).value == 0x200
    assert ops._flags(None).value == 0 and seen["other"] == 0


def test_size_helpers_need_no_gpu():
    lib = _lib.load()
    assert lib.optex_rotation_normals(256) == 256 * 257 // 2 - 1 == 32895
    assert lib.optex_cdf_ws_bytes(256, 4) > 4 * 256 * (4 * 4 + 2 * 256 * 4 + 3 * 256 * 4) - 1
    assert lib.optex_cdf_bins_ws_bytes(256, 4, 256) > 0 and lib.optex_cdf_bins_ws_bytes(8, 2, 5000) == 6 * 4 * 5000 * 16
    assert lib.optex_ot_loop_ws_bytes(0, 16384, 12288, 256, 2, 1, 13, 0, 0) >= (2 * 16384 + 12288) * 256 * 4


def test_argument_errors_are_reported_without_launching():
    lib = _lib.load()
    rc = lib.optex_gemm_tn(None, 0, 0, None, 0, 0, 0, None, 0, 0, 0, 4, 4, 16, 1, None, 0, None, 0, None, 0.0, 0, None)
    assert rc == -1 and b"optex_gemm_tn" in lib.optex_last_error()
    rc = lib.optex_rotations_from_normals(None, 1, 1, None, None, None, None, 0, None)
    assert rc == -1
    # glue: pool and upsample are exclusive; NHWC on both sides needs C % 4 == 0 (both rejected before any launch)
    import ctypes
    buf = (ctypes.c_float * 64)()
    ptr = ctypes.cast(buf, ctypes.c_void_p)
    assert lib.optex_vgg_glue_layout(ptr, None, ptr, 1, 4, 2, 2, 0, 1, 1, 0, 1, 1, None) == -1
    assert b"optex_vgg_glue_layout" in lib.optex_last_error()
    assert lib.optex_vgg_glue_layout(ptr, None, ptr, 1, 3, 4, 4, 0, 0, 0, 0, 1, 1, None) != 0
    assert b"C % 4" in lib.optex_last_error()


def test_round4_entry_points_check_their_arguments_without_launching():
    """optex_legacy_normals (ABI 7): the device side of numpy's gaussian stream refuses bad calls on the host; the state is
    numpy's get_state() tuple, 624 + 4 words"""
    import ctypes
    lib = _lib.load()
    assert lib.optex_mt19937_state_bytes() == (624 + 4) * 4
    buf = (ctypes.c_uint32 * 628)()
    p = ctypes.cast(buf, ctypes.c_void_p)
    assert lib.optex_legacy_normals(None, 1, 10, p, 10, p, 1 << 20, None) == -1 and b"optex_legacy_normals" in lib.optex_last_error()
    assert lib.optex_legacy_normals(p, 1, 10, p, 5, p, 1 << 20, None) == -1      # out_stride smaller than count
    assert lib.optex_legacy_normals(p, 0, 10, p, 10, p, 1 << 20, None) == -1
    assert lib.optex_legacy_normals(p, 1, 0, p, 0, None, 0, None) == 0           # nothing to draw: no launch
    need = lib.optex_legacy_normals_ws_bytes(2, 1001)
    assert need >= 2 * 501 * 24 and lib.optex_legacy_normals(p, 2, 1001, p, 1001, p, need - 1, None) == -1   # undersized scratch
    assert lib.optex_mt19937_seed(None, 1, 5, 1, None) == -1 and lib.optex_mt19937_seed(p, 0, 5, 1, None) == -1


def test_round3_entry_points_check_their_arguments_without_launching():
    """optex_cdf_match_bins (ABI 6) and the collapsed chain of optex_ot_loop (fuse_rotations = 3) refuse bad calls on the host"""
    import ctypes
    lib = _lib.load()
    buf = (ctypes.c_float * 64)()
    p = ctypes.cast(buf, ctypes.c_void_p)
    C, S, n, ns = 8, 2, 64, 48
    assert lib.optex_cdf_match_bins(p, n, C * n, n, p, ns, C * ns, ns, 1, C, S, 0, p, n, C * n, p, 1 << 20, None) == -1
    assert b"optex_cdf_match_bins" in lib.optex_last_error() and b"bins=0" in lib.optex_last_error()
    assert lib.optex_cdf_match_bins(p, n, C * n, n, p, ns, C * ns, ns, 3, C, S, 16, p, n, C * n, p, 1 << 20, None) == -1
    need = lib.optex_cdf_bins_ws_bytes(C, S, 5000)
    assert lib.optex_cdf_match_bins(p, n, C * n, n, p, ns, C * ns, ns, 1, C, S, 5000, p, n, C * n, p, need - 1, None) == -1
    assert b"scratch buffer too small" in lib.optex_last_error()
    # collapsed chain: linear modes only, no content blend
    for mode, content in ((0, None), (1, None), (2, p)):
        rc = lib.optex_ot_loop(mode, p, n, S, p, ns, 1, C, p, p, 0, 2, content, 0.1, 3, p, 1 << 30, 0, None)
        assert rc == -1 and b"collapsed chain" in lib.optex_last_error(), mode
    assert lib.optex_ot_loop(2, p, n, S, p, ns, 1, C, p, p, 0, 2, None, 0.0, 4, p, 1 << 30, 0, None) == -1
    assert lib.optex_ot_loop_ws_bytes(2, n, ns, C, S, 1, 2, 3, 0) > lib.optex_ot_loop_ws_bytes(2, n, ns, C, S, 1, 2, 1, 0)


def test_undersized_scratch_is_refused_before_any_launch():
    """ABI 3: every entry point that takes `ws` also takes `ws_bytes` and refuses a buffer smaller than its *_ws_bytes
    helper asks for (ABI 2 trusted the pointer: an undersized buffer was silent device-memory corruption)."""
    import ctypes
    lib = _lib.load()
    buf = (ctypes.c_float * 64)()
    p = ctypes.cast(buf, ctypes.c_void_p)
    C, S, n, ns = 8, 2, 64, 48
    need = lib.optex_cdf_ws_bytes(C, S)
    rc = lib.optex_cdf_match(p, n, C * n, n, p, ns, C * ns, ns, 1, C, S, p, n, C * n, p, need - 1, None, 0, None)
    assert rc == -1 and b"scratch buffer too small" in lib.optex_last_error() and b"optex_cdf_match" in lib.optex_last_error()
    need = lib.optex_sort_match_ws_bytes(n, ns, C, S, 1)
    assert lib.optex_sort_match(p, n, C * n, n, p, ns, C * ns, ns, 1, C, S, p, n, C * n, p, need - 1, 0, None) == -1
    assert b"optex_sort_match" in lib.optex_last_error()
    assert lib.optex_sort_columns(p, n, C * n, n, C, S, p, p, p, lib.optex_sort_ws_bytes(n, C, S) - 1, None) == -1
    assert lib.optex_linear_stats(p, n, C * n, n, C, S, 0, 1.0, p, p, p, lib.optex_linear_stats_ws_bytes(n, C, S) - 1, None) == -1
    assert b"optex_linear_stats" in lib.optex_last_error()
    assert lib.optex_rotations_from_normals(p, 4, 2, None, p, p, p, lib.optex_rotation_ws_bytes(4, 2) - 1, None) == -1
    for mode in (0, 1):
        need = lib.optex_ot_loop_ws_bytes(mode, n, ns, C, S, 1, 3, 0, 0)
        assert lib.optex_ot_loop(mode, p, n, S, p, ns, 1, C, p, p, 0, 3, None, 0.0, 0, p, need - 1, 0, None) == -1
        assert b"optex_ot_loop" in lib.optex_last_error()
    # the sort-mode loop scratch covers pastiche columns longer than one LDS (it was sized with nt = 0)
    big = lib.optex_ot_loop_ws_bytes(1, 65536, 49152, 64, 1, 1, 4, 0, 0)
    fixed = 4 * 64 * (65536 + 49152)
    assert big - fixed >= lib.optex_sort_match_ws_bytes(65536, 49152, 64, 1, 1)


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-GPU failure mode")
def test_product_path_fails_loudly_without_gpu():
    import optimaltextures_amd as ot
    x = torch.rand(1, 8, 8, 4)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ot.hist_match(x, x, "cdf")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ot.interp(torch.rand(4), torch.rand(4), torch.rand(4))


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "optimaltextures_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                text = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in text and "from oracle" not in text and "liboptex_oracle" not in text, f
