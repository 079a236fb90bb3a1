"""ctypes front-end of the TEST-ONLY host instantiation of the kernel bodies (tests/emul/emul.cpp)."""
from __future__ import annotations

import ctypes as C
import os
import subprocess
import numpy as np

from hmcmt2d_amd.marshal import CreateArgs, CREATE_ARGTYPES, c_double_p

HERE = os.path.dirname(os.path.abspath(__file__))
# HMCMT_EMUL_SANITIZE=1: the same source under AddressSanitizer + UndefinedBehaviorSanitizer (CPU only -- SURVEY section 5;
# the Python process must then run with libasan preloaded: tests/test_kernel_math.py::test_..._under_sanitizers does that)
SANITIZE = os.environ.get("HMCMT_EMUL_SANITIZE", "") not in ("", "0")
SO = os.path.join(HERE, "_build", "libhmcmt_emul_asan.so" if SANITIZE else "libhmcmt_emul.so")


def build(force=False):
    src = os.path.join(HERE, "emul.cpp")
    hdrs = [os.path.join(HERE, "..", "..", "hmcmt2d_amd", "csrc", h)
            for h in ("hmcmt_math.h", "hmcmt_items.h", "hmcmt_host.h")]
    newest = max(os.path.getmtime(f) for f in [src] + hdrs)
    if force or not os.path.exists(SO) or os.path.getmtime(SO) < newest:
        os.makedirs(os.path.dirname(SO), exist_ok=True)
        flags = ["-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer"] if SANITIZE else ["-O2"]
        subprocess.check_call(["g++"] + flags + ["-std=c++17", "-shared", "-fPIC", src, "-o", SO])
    return SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
        _lib.emul_create.restype = C.c_void_p
        _lib.emul_create.argtypes = CREATE_ARGTYPES + [C.c_char_p, C.c_int]
        _lib.emul_destroy.argtypes = [C.c_void_p]
        _lib.emul_grad.argtypes = [C.c_void_p, c_double_p, C.c_int, C.c_int, C.c_double, C.c_int,
                                   c_double_p, c_double_p, c_double_p, C.POINTER(C.c_int)]
        _lib.emul_dims.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
        _lib.emul_apply.argtypes = [C.c_void_p, C.c_int, c_double_p, c_double_p]
        _lib.emul_get.restype = C.c_long
        _lib.emul_get.argtypes = [C.c_void_p, C.c_int, c_double_p, C.c_long]
    return _lib


class Emul:
    GET = dict(X=0, Lam=1, sigma=2, Vpad=3, lam=4, gL=5, gR=6, gMn=7, bcsL=8, bcsR=9, bcsB=10, Zrx=11,
               rxD=12, gPart=13)

    def __init__(self, mtMesh, mtData, invParam):
        self.args = CreateArgs(mtMesh, mtData, invParam)
        err = C.create_string_buffer(512)
        self.h = lib().emul_create(*self.args.as_tuple(), err, 512)
        if not self.h:
            raise RuntimeError(err.value.decode())
        d = (C.c_int * 6)()
        lib().emul_dims(self.h, d)
        self.NYP, self.NZP, self.S, self.ny, self.nz, self.zid = list(d)

    def grad(self, m, want_grad=True, precond=2, tol=1e-12, maxit=20000):
        m = np.ascontiguousarray(m, dtype=np.float64)
        pred = np.zeros(self.args.nData, dtype=np.complex128)
        grad = np.zeros(self.args.nAC)
        misfit = C.c_double(0)
        iters = (C.c_int * (2 * self.S))()
        lib().emul_grad(self.h, m.ctypes.data_as(c_double_p), int(want_grad), precond, tol, maxit,
                        pred.ctypes.data_as(c_double_p), C.byref(misfit), grad.ctypes.data_as(c_double_p), iters)
        self.iters = np.array(list(iters)).reshape(2, self.S)
        return pred, misfit.value, grad

    def get(self, name, complex_=None):
        which = self.GET[name]
        n = lib().emul_get(self.h, which, None, 0)
        out = np.zeros(n)
        lib().emul_get(self.h, which, out.ctypes.data_as(c_double_p), n)
        if complex_ is None:
            complex_ = name in ("X", "Lam", "gL", "gR", "gMn", "bcsL", "bcsR", "bcsB", "Zrx", "rxD")
        return out.view(np.complex128) if complex_ else out

    def apply(self, which, x):
        """which: 'spmv' | 'fdm' | 'jacobi' on vectors in the padded nodal layout [S][NZP][NYP]."""
        x = np.ascontiguousarray(x, dtype=np.complex128)
        y = np.empty_like(x)
        lib().emul_apply(self.h, {"spmv": 0, "fdm": 1, "jacobi": 2, "fdmj": 3}[which], x.ctypes.data_as(c_double_p),
                         y.ctypes.data_as(c_double_p))
        return y

    def close(self):
        if self.h:
            lib().emul_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
