"""Python mirror of the reference's MUMPS wrapper (MUMPS/src/MUMPSfuncs.jl) over the Fortran-convention symbols
libhmcmt_hip.so exports (include/hmcmt_mumps.h): factorMUMPS / applyMUMPS / solveMUMPS / destroyMUMPS with the
reference's argument order.  Matrices are scipy CSC (the layout of Julia's SparseMatrixCSC); index arrays go over
1-based Int64 exactly as the reference's ccall passes `A.rowval`, `A.colptr`.  No CPU fallback: the symbols fail
with stat < 0 without a HIP device."""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

import numpy as np
import scipy.sparse as sp

from . import lib as L

_i64p = C.POINTER(C.c_int64)
_dp = C.POINTER(C.c_double)


def _so():
    so = L.load_library()
    if not getattr(so, "_mumps_ready", False):
        for name in ("factor_mumps_cmplx_", "factor_mumps_"):
            f = getattr(so, name)
            f.restype = C.c_int64
            f.argtypes = [_i64p, _i64p, _i64p, _dp, _i64p, _i64p, _i64p]
        for name in ("solve_mumps_cmplx_", "solve_mumps_"):
            f = getattr(so, name)
            f.restype = C.c_int64
            f.argtypes = [_i64p, _i64p, _dp, _dp, _i64p]
        for name in ("solve_mumps_cmplx_sparse_rhs_", "solve_mumps_sparse_rhs_"):
            f = getattr(so, name)
            f.restype = None
            f.argtypes = [_i64p, _i64p, _i64p, _dp, _i64p, _i64p, _dp, _i64p]
        for name in ("destroy_mumps_cmplx_", "destroy_mumps_"):
            f = getattr(so, name)
            f.restype = C.c_int64
            f.argtypes = [_i64p]
        so.hmcmt_mumps_last_solve.restype = C.c_int64
        so.hmcmt_mumps_last_solve.argtypes = [_i64p, _dp]
        so._mumps_ready = True
    return so


def _ref(v):
    return C.byref(C.c_int64(int(v)))


def _ptr(a, t):
    return a.ctypes.data_as(t)


@dataclass
class MUMPSfactorization:
    """MUMPS/src/MUMPS.jl: ptr, worker, n, real/complex marker, time."""
    ptr: int
    n: int
    cmplx: bool


def checkMUMPSerror(stat):
    """MUMPSfuncs.jl:59-73."""
    s = int(stat[0])
    if s == -10:
        raise RuntimeError("MUMPS: Numerically singular matrix.")
    if s == -13:
        raise RuntimeError("MUMPS: memory allocation error")
    if s == -40:
        raise RuntimeError("MUMPS: matrix is not positive definite")
    if s == -90:
        raise RuntimeError("MUMPS: Error in out-of-core management.Probably there is not enough disk space.")
    if s < 0:
        raise RuntimeError(f"MUMPS: error --> {s} <--. Please refer to Ch. 7 of MUMPS User's guide!")


def factorMUMPS(A, sym=0, ooc=0) -> MUMPSfactorization:
    """MUMPSfuncs.jl:24-57."""
    A = sp.csc_matrix(A)
    if A.shape[0] != A.shape[1]:
        raise ValueError("factorMUMPS: Matrix must be square!")
    A.sort_indices()
    cm = np.iscomplexobj(A.data)
    nz = np.ascontiguousarray(A.data, dtype=np.complex128 if cm else np.float64)
    rowval = np.ascontiguousarray(A.indices, dtype=np.int64) + 1
    colptr = np.ascontiguousarray(A.indptr, dtype=np.int64) + 1
    stat = np.zeros(1, dtype=np.int64)
    so = _so()
    f = so.factor_mumps_cmplx_ if cm else so.factor_mumps_
    p = f(_ref(A.shape[0]), _ref(sym), _ref(ooc), _ptr(nz, _dp), _ptr(rowval, _i64p), _ptr(colptr, _i64p),
          _ptr(stat, _i64p))
    checkMUMPSerror(stat)
    return MUMPSfactorization(int(p), A.shape[0], cm)


def applyMUMPS(factor: MUMPSfactorization, rhs, x=None, tr=0):
    """MUMPSfuncs.jl:75-146: dense (n or n x nrhs, column-major like Julia) or sparse CSC right-hand sides."""
    so = _so()
    dt = np.complex128 if factor.cmplx else np.float64
    if sp.issparse(rhs):
        R = sp.csc_matrix(rhs).astype(dt)
        R.sort_indices()
        if R.shape[0] != factor.n:
            raise ValueError("applyMUMPS: wrong size of rhs")
        nrhs = R.shape[1]
        xf = np.zeros((factor.n, nrhs), dtype=dt, order="F")
        nz = np.ascontiguousarray(R.data)
        rowval = np.ascontiguousarray(R.indices, dtype=np.int64) + 1
        colptr = np.ascontiguousarray(R.indptr, dtype=np.int64) + 1
        f = so.solve_mumps_cmplx_sparse_rhs_ if factor.cmplx else so.solve_mumps_sparse_rhs_
        f(_ref(factor.ptr), _ref(R.nnz), _ref(nrhs), _ptr(nz, _dp), _ptr(rowval, _i64p), _ptr(colptr, _i64p),
          _ptr(xf, _dp), _ref(tr))
        return xf
    rhs = np.asarray(rhs)
    if rhs.shape[0] != factor.n:
        raise ValueError(f"applyMUMPS: wrong size of rhs, size(A)={factor.n}, size(rhs)={rhs.shape}")
    if np.iscomplexobj(rhs) and not factor.cmplx:
        raise TypeError("complex right-hand side for a real factorization")
    nrhs = 1 if rhs.ndim == 1 else rhs.shape[1]
    rf = np.asfortranarray(rhs.reshape(factor.n, nrhs), dtype=dt)
    xf = np.zeros((factor.n, nrhs), dtype=dt, order="F")
    f = so.solve_mumps_cmplx_ if factor.cmplx else so.solve_mumps_
    f(_ref(factor.ptr), _ref(nrhs), _ptr(rf, _dp), _ptr(xf, _dp), _ref(tr))
    return xf[:, 0].copy() if rhs.ndim == 1 else xf


def destroyMUMPS(factor: MUMPSfactorization):
    """MUMPSfuncs.jl:148-184."""
    so = _so()
    (so.destroy_mumps_cmplx_ if factor.cmplx else so.destroy_mumps_)(_ref(factor.ptr))
    factor.ptr = -1
    factor.n = -1


def solveMUMPS(A, rhs, sym=0, ooc=0, tr=0):
    """MUMPSfuncs.jl:2-22: factor, solve, free."""
    F = factorMUMPS(A, sym, ooc)
    try:
        return applyMUMPS(F, rhs, None, tr)
    finally:
        destroyMUMPS(F)


def lastSolveStats(factor: MUMPSfactorization):
    out = np.zeros(3)
    _so().hmcmt_mumps_last_solve(_ref(factor.ptr), _ptr(out, _dp))
    return {"iterations": int(out[0]), "refinement_passes": int(out[1]), "relres": float(out[2])}
