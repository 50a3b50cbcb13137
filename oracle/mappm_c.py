"""ORACLE -- TEST INFRASTRUCTURE ONLY.  ctypes access to oracle/libfv3oracle.so (the C
restatement of mappm.f90, see mappm_oracle.c) and, when present, to oracle/_ref/libmappm_ref.so
(the reference's own Fortran compiled by `make -C oracle ref` in the build container)."""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_SO = os.path.join(_HERE, "libfv3oracle.so")
REF_SO = os.path.join(_HERE, "_ref", "libmappm_ref.so")

_oracle = None
_ref = None


def _load_oracle():
    global _oracle
    if _oracle is None:
        if not os.path.exists(ORACLE_SO):
            raise ImportError(f"{ORACLE_SO} is not built; run `make -C oracle`")
        _oracle = ctypes.CDLL(ORACLE_SO)
        _oracle.fv3_oracle_mappm.restype = ctypes.c_int
        _oracle.fv3_oracle_mappm.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_long] + [ctypes.c_int] * 4
        _oracle.fv3_oracle_ppm_profile.restype = ctypes.c_int
        _oracle.fv3_oracle_ppm_profile.argtypes = [ctypes.c_void_p] * 5 + [ctypes.c_long] + [ctypes.c_int] * 3
    return _oracle


def have_reference() -> bool:
    return os.path.exists(REF_SO)


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def mappm(p_in, f_in, p_out, iv=1, kord=1):
    """C restatement.  Arrays are [ncol, levels] of any float dtype (rounded to float32, as
    f2py does); returns float32 [ncol, kn]."""
    lib = _load_oracle()
    a = np.ascontiguousarray(p_in, dtype=np.float32)
    b = np.ascontiguousarray(f_in, dtype=np.float32)
    c = np.ascontiguousarray(p_out, dtype=np.float32)
    ncol, km = b.shape
    kn = c.shape[1] - 1
    assert a.shape == (ncol, km + 1) and c.shape[0] == ncol
    out = np.zeros((ncol, kn), dtype=np.float32)
    rc = lib.fv3_oracle_mappm(_p(a), _p(b), _p(c), _p(out), ncol, km, kn, int(iv), int(kord))
    if rc != 0:
        raise ValueError(f"fv3_oracle_mappm failed with code {rc}")
    return out


def ppm_profile(p_in, f_in, iv=1, kord=1):
    lib = _load_oracle()
    a = np.ascontiguousarray(p_in, dtype=np.float32)
    b = np.ascontiguousarray(f_in, dtype=np.float32)
    ncol, km = b.shape
    al, ar, a6 = (np.zeros((ncol, km), np.float32) for _ in range(3))
    rc = lib.fv3_oracle_ppm_profile(_p(a), _p(b), _p(al), _p(ar), _p(a6), ncol, km, int(iv), int(kord))
    if rc != 0:
        raise ValueError(f"fv3_oracle_ppm_profile failed with code {rc}")
    return al, ar, a6


def reference_mappm(p_in, f_in, p_out, iv=1, kord=1, chunk=256):
    """The reference's own mappm.f90 (compiled, never copied).  Fortran order, real*4, all
    arguments by reference; called in chunks because its work arrays live on the stack."""
    global _ref
    if _ref is None:
        if not have_reference():
            raise ImportError(f"{REF_SO} not present (only buildable where /root/reference exists)")
        _ref = ctypes.CDLL(REF_SO)
    ncol = p_in.shape[0]
    if ncol > chunk:
        return np.concatenate(
            [reference_mappm(p_in[i:i + chunk], f_in[i:i + chunk], p_out[i:i + chunk], iv, kord, chunk)
             for i in range(0, ncol, chunk)]
        )
    a = np.asfortranarray(p_in, dtype=np.float32)
    b = np.asfortranarray(f_in, dtype=np.float32)
    c = np.asfortranarray(p_out, dtype=np.float32)
    km, kn = b.shape[1], c.shape[1] - 1
    out = np.zeros((ncol, kn), dtype=np.float32, order="F")
    I = ctypes.c_int
    _ref.mappm_(
        ctypes.byref(I(km)), _p(a), _p(b), ctypes.byref(I(kn)), _p(c), _p(out), ctypes.byref(I(1)),
        ctypes.byref(I(ncol)), ctypes.byref(I(int(iv))), ctypes.byref(I(int(kord))),
        ctypes.byref(ctypes.c_float(0.0)),
    )
    return np.ascontiguousarray(out)



def interpolate_2d(xp, x, y, fill_value=np.nan):
    """C restatement of interpolate_2d.f90: [m, n] float64 arrays, returns [m, n_out]."""
    lib = _load_oracle()
    lib.fv3_oracle_interpolate_2d.restype = ctypes.c_int
    lib.fv3_oracle_interpolate_2d.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_double, ctypes.c_long, ctypes.c_int, ctypes.c_int]
    a = np.ascontiguousarray(xp, dtype=np.float64)
    b = np.ascontiguousarray(x, dtype=np.float64)
    c = np.ascontiguousarray(y, dtype=np.float64)
    m, n_in = b.shape
    n_out = a.shape[1]
    assert a.shape[0] == m and c.shape == b.shape
    out = np.empty((m, n_out), dtype=np.float64)
    lib.fv3_oracle_interpolate_2d(_p(a), _p(b), _p(c), _p(out), float(fill_value), m, n_in, n_out)
    return out


def reference_interpolate_2d(xp, x, y, fill_value=np.nan):
    """The reference's own interpolate_2d.f90 (compiled, never copied): Fortran order real(8)."""
    global _ref
    if _ref is None:
        if not have_reference():
            raise ImportError(f"{REF_SO} not present (only buildable where /root/reference exists)")
        _ref = ctypes.CDLL(REF_SO)
    a = np.asfortranarray(xp, dtype=np.float64)
    b = np.asfortranarray(x, dtype=np.float64)
    c = np.asfortranarray(y, dtype=np.float64)
    m, n_in = b.shape
    n_out = a.shape[1]
    out = np.zeros((m, n_out), dtype=np.float64, order="F")
    fill = ctypes.c_double(float(fill_value))
    mm, ni, no = ctypes.c_int(m), ctypes.c_int(n_in), ctypes.c_int(n_out)
    _ref.interpolate_2d_(_p(a), _p(b), _p(c), _p(out), ctypes.byref(fill), ctypes.byref(mm), ctypes.byref(ni), ctypes.byref(no))
    return np.ascontiguousarray(out)
