"""ctypes wrapper around oracle/liboptex_oracle.so — the CPU checker.  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the product
package (optimaltextures_amd) never does.  Function names follow the reference (/root/reference):
optimal_transport (optex.py:167-177), hist_match / cdf_match / interp (histmatch.py:5-92),
random_rotation (optex.py:142-149).  Arrays are numpy; "cm" means channel-major [C, N] fp32.
Parity status: pinned by tests/golden/*.npz (tests/test_oracle_golden.py).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "liboptex_oracle.so")
MODES = {"chol": 0, "pca": 1, "sym": 2}
BINS = 256


def build(force=False):
    src = os.path.join(_HERE, "optex_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "-B"] if force else ["make", "-s", "-C", _HERE])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = ctypes.CDLL(_SO)
        _lib.orc_rotation_normals.restype = ctypes.c_long
    return _lib


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def num_threads():
    return int(lib().orc_num_threads())


def set_num_threads(n):
    lib().orc_set_num_threads(ctypes.c_int(int(n)))


# ----------------------------------------------------------------------------------------------- RNG / rotation
class LegacyRNG:
    """numpy RandomState(seed).normal stream, restated in C (MT19937 + polar gauss with cache)."""

    def __init__(self, seed):
        self.state = np.zeros(lib().orc_rng_size(), dtype=np.uint8)
        lib().orc_mt_seed(_p(self.state), ctypes.c_uint32(int(seed)))

    def normal(self, n):
        out = np.empty(int(n), dtype=np.float64)
        lib().orc_normals(_p(self.state), _p(out), ctypes.c_long(int(n)))
        return out


def rotation_normals(N):
    return int(lib().orc_rotation_normals(ctypes.c_int(int(N))))


def random_rotation_from_normals(normals, N):
    """scipy special_ortho_group.rvs(N) given the N(N+1)/2-1 normals it would draw (fp64 [N,N])."""
    nn = np.array(normals, dtype=np.float64, copy=True)
    assert nn.size == rotation_normals(N)
    H = np.empty((N, N), dtype=np.float64)
    lib().orc_random_rotation(_p(nn), ctypes.c_int(int(N)), _p(H))
    return H


def random_rotation(N, rng=None):
    """optex.py:142-149.  rng=None draws from numpy's GLOBAL RandomState exactly like scipy does there."""
    if N is None or N <= 1 or N != int(N):
        raise ValueError("Dimension of rotation must be specified, and must be a scalar greater than 1.")
    k = rotation_normals(N)
    if rng is None:
        normals = np.random.normal(size=k)
    elif isinstance(rng, LegacyRNG):
        normals = rng.normal(k)
    else:
        normals = rng.normal(size=k)
    return random_rotation_from_normals(normals, N)


# ----------------------------------------------------------------------------------------------- GEMM
def gemm_tn(At, B, bsub=None, badd=None):
    """OUT[m,n] = sum_k At[k,m] * (B[k,n] - bsub[k]) + badd[m]; fp32 fma chain, k ascending."""
    At, B = _f32(At), _f32(B)
    K, M = At.shape
    K2, N = B.shape
    assert K == K2
    out = np.empty((M, N), dtype=np.float32)
    bs = _f32(bsub) if bsub is not None else None
    ba = _f32(badd) if badd is not None else None
    lib().orc_gemm_tn(_p(At), ctypes.c_long(M), _p(B), ctypes.c_long(N), _p(out), ctypes.c_long(N), ctypes.c_int(M),
                      ctypes.c_int(K), ctypes.c_long(N), _p(bs) if bs is not None else None,
                      _p(ba) if ba is not None else None)
    return out


def rotate_cm(x_cm, R):
    """channel-major form of `x @ R` (optex.py:170-171): Y[c,n] = sum_k x[k,n] R[k,c]"""
    return gemm_tn(_f32(R), x_cm)


def unrotate_cm(m_cm, R):
    """channel-major form of `m @ R.T` (optex.py:175): X[k,n] = sum_c m[c,n] R[k,c]"""
    return gemm_tn(np.ascontiguousarray(_f32(R).T), m_cm)


def content_blend(feat, content, strength):
    """optex.py:115-117 (in place on a copy)"""
    f = _f32(feat).copy()
    c = _f32(content)
    lib().orc_content_blend(_p(f), _p(c), ctypes.c_float(strength), ctypes.c_long(f.size))
    return f


# ----------------------------------------------------------------------------------------------- cdf pieces
def histc(x, lo, hi):
    x = _f32(x).ravel()
    h = np.empty(BINS, dtype=np.float32)
    lib().orc_histc(_p(x), ctypes.c_long(x.size), ctypes.c_float(lo), ctypes.c_float(hi), _p(h))
    return h


def linspace257(lo, hi):
    e = np.empty(BINS + 1, dtype=np.float32)
    lib().orc_linspace257(ctypes.c_float(lo), ctypes.c_float(hi), _p(e))
    return e


def interp(x, xp, fp):
    x, xp, fp = _f32(x), _f32(xp), _f32(fp)
    out = np.empty_like(x)
    lib().orc_interp(_p(x), ctypes.c_long(x.size), _p(xp), _p(fp), ctypes.c_long(xp.size), _p(out))
    return out


def cdf_match(target_cm, source_cm, debug=False):
    """histmatch.py:49-69 on channel-major [C,Nt] / [C,Ns]"""
    t, s = _f32(target_cm), _f32(source_cm)
    C, nt = t.shape
    ns = s.shape[1]
    out = np.empty_like(t)
    dbg = np.empty((C, 2 + 4 * BINS), dtype=np.float32) if debug else None
    lib().orc_cdf_match(_p(t), ctypes.c_long(nt), ctypes.c_long(nt), _p(s), ctypes.c_long(ns), ctypes.c_long(ns),
                        ctypes.c_int(C), _p(out), _p(dbg) if debug else None)
    if debug:
        d = dict(lo=dbg[:, 0].copy(), hi=dbg[:, 1].copy(), hist_t=dbg[:, 2:2 + BINS].copy(),
                 hist_s=dbg[:, 2 + BINS:2 + 2 * BINS].copy(), bin_edges=dbg[:, 2 + 2 * BINS:2 + 3 * BINS].copy(),
                 remapped=dbg[:, 2 + 3 * BINS:].copy())
        return out, d
    return out


# ----------------------------------------------------------------------------------------------- sort mode
def cdf_match_bins(target_cm, source_cm, bins):
    """histmatch.py:49-69 with the reference's `bins` argument"""
    t, s = _f32(target_cm), _f32(source_cm)
    C, nt = t.shape
    ns = s.shape[1]
    out = np.empty_like(t)
    lib().orc_cdf_match_bins(_p(t), ctypes.c_long(nt), ctypes.c_long(nt), _p(s), ctypes.c_long(ns), ctypes.c_long(ns),
                             ctypes.c_int(C), ctypes.c_int(int(bins)), _p(out))
    return out


def sort_columns(keys_cm):
    k = _f32(keys_cm)
    C, n = k.shape
    ok = np.empty_like(k)
    oi = np.empty((C, n), dtype=np.uint32)
    lib().orc_sort_columns(_p(k), ctypes.c_long(n), ctypes.c_long(n), ctypes.c_int(C), _p(ok), _p(oi))
    return ok, oi


def sort_match(target_cm, source_cm):
    t, s = _f32(target_cm), _f32(source_cm)
    C, nt = t.shape
    ns = s.shape[1]
    out = np.empty_like(t)
    lib().orc_sort_match(_p(t), ctypes.c_long(nt), ctypes.c_long(nt), _p(s), ctypes.c_long(ns), ctypes.c_long(ns),
                         ctypes.c_int(C), _p(out))
    return out


# ----------------------------------------------------------------------------------------------- linear modes
def linear_match(target_cm, nb, source_cm, sb, mode, eps=1.0, return_T=False):
    t, s = _f32(target_cm), _f32(source_cm)
    C, N = t.shape
    S = s.shape[1]
    assert N % nb == 0 and S % sb == 0
    out = np.empty_like(t)
    T = np.empty((C, C), dtype=np.float32)
    lib().orc_linear_match(_p(t), ctypes.c_long(N), ctypes.c_int(nb), ctypes.c_long(N // nb), _p(s), ctypes.c_long(S),
                           ctypes.c_int(sb), ctypes.c_long(S // sb), ctypes.c_int(C), ctypes.c_int(MODES[mode]),
                           ctypes.c_float(eps), _p(out), _p(T))
    return (out, T) if return_T else out


# ----------------------------------------------------------------------------------------------- reference-level API
def _to_cm(x_nhwc):
    x = _f32(x_nhwc)
    return np.ascontiguousarray(x.reshape(-1, x.shape[-1]).T)


def _from_cm(x_cm, shape):
    return np.ascontiguousarray(x_cm.T).reshape(shape)


def hist_match_cm(t_cm, nb, s_cm, sb, mode, eps=1.0):
    if mode == "cdf":
        return cdf_match(t_cm, s_cm)
    if mode == "sort":
        return sort_match(t_cm, s_cm)
    if sb != nb and sb != 1:
        raise RuntimeError(f"The size of tensor a ({nb}) must match the size of tensor b ({sb})")
    return linear_match(t_cm, nb, s_cm, sb, mode, eps)


def hist_match(target, source, mode="chol", eps=1.0):
    """histmatch.py:5-46 on NHWC arrays"""
    target, source = _f32(target), _f32(source)
    out = hist_match_cm(_to_cm(target), target.shape[0], _to_cm(source), source.shape[0], mode, eps)
    return _from_cm(out, target.shape)


def optimal_transport(pastiche_feature, style_feature, hist_mode, rotation, return_intermediates=False):
    """optex.py:167-177 with the rotation made an explicit input (fp64 or fp32 [C,C]; cast to fp32 like :168)."""
    p, s = _f32(pastiche_feature), _f32(style_feature)
    R = np.asarray(rotation).astype(np.float32)
    rp, rs = rotate_cm(_to_cm(p), R), rotate_cm(_to_cm(s), R)
    m = hist_match_cm(rp, p.shape[0], rs, s.shape[0], hist_mode)
    out = _from_cm(unrotate_cm(m, R), p.shape)
    if return_intermediates:
        return out, dict(rotated_pastiche=_from_cm(rp, p.shape), rotated_style=_from_cm(rs, s.shape),
                         matched=_from_cm(m, p.shape))
    return out
