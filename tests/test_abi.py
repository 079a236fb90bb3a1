"""The C-ABI shared library: builds for gfx950, loads, exports every symbol include/hmcmt.h declares,
and refuses to run without a HIP device (no CPU fallback).  No compute calls here."""
import ctypes
import os
import re

import pytest

from hmcmt2d_amd import lib as L
from tests.conftest import HAVE_GPU, ROOT


@pytest.fixture(scope="module")
def so():
    return ctypes.CDLL(L.build_library())


def declared_functions(header="hmcmt.h", pattern=r"\b(hmcmt_[a-z_]+)\s*\("):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(pattern, text)))


def test_header_symbols_are_exported(so):
    names = declared_functions()
    assert len(names) >= 18
    for n in names:
        assert hasattr(so, n), f"{n} declared in include/hmcmt.h but not exported"
    assert set(names) == set(L.EXPORTED_SYMBOLS)


def test_mumps_interface_symbols_are_exported(so):
    """include/hmcmt_mumps.h: the eight Fortran-convention symbols MUMPS/src/MUMPSfuncs.jl binds (:32,49,105,115,
    128,139,155,170) plus the statistics call."""
    names = declared_functions("hmcmt_mumps.h", r"\b([a-z_]+mumps[a-z_]*)\s*\(")
    assert set(names) == {"factor_mumps_cmplx_", "factor_mumps_", "solve_mumps_cmplx_", "solve_mumps_",
                          "solve_mumps_cmplx_sparse_rhs_", "solve_mumps_sparse_rhs_", "destroy_mumps_cmplx_",
                          "destroy_mumps_", "hmcmt_mumps_last_solve"}
    for n in names:
        assert hasattr(so, n), f"{n} declared in include/hmcmt_mumps.h but not exported"


def test_default_options(so):
    o = L.Options()
    L.load_library().hmcmt_default_options(ctypes.byref(o))
    assert o.precond == 2 and o.maxit >= 100 and 0 < o.tol < 1e-8 and o.check_every >= 1


@pytest.mark.skipif(HAVE_GPU, reason="checks the no-device error path")
def test_create_fails_loudly_without_a_device():
    from tests.helpers import make_problem
    mesh, data, inv, m = make_problem("tiny")
    with pytest.raises(L.HmcmtError) as e:
        L.HipContext(mesh, data, inv)
    assert e.value.code == -2 and "no host compute path" in str(e.value)


def test_create_rejects_bad_arguments_before_touching_the_device():
    from tests.helpers import make_problem
    import numpy as np
    mesh, data, inv, m = make_problem("tiny")
    data.rxLoc = data.rxLoc.copy(); data.rxLoc[:, 1] = 37.0      # no grid node at that depth
    with pytest.raises(L.HmcmtError) as e:
        L.HipContext(mesh, data, inv)
    assert e.value.code == -1 and "receiver depth" in str(e.value)
    mesh, data, inv, m = make_problem("tiny")
    data.dataType = "Rho_Pha"
    with pytest.raises(ValueError):
        L.HipContext(mesh, data, inv)


@pytest.mark.skipif(HAVE_GPU, reason="needs a GPU-less box")
def test_mumps_interface_without_a_device_fails_loudly():
    """No CPU fallback behind the MUMPS symbols either: factor reports stat < 0, which the wrapper raises
    (MUMPSfuncs.jl:59-73)."""
    import numpy as np
    import scipy.sparse as sp
    from hmcmt2d_amd import mumps as M
    A = sp.csc_matrix(np.array([[2.0, -1.0], [-1.0, 2.0]]))
    with pytest.raises(RuntimeError, match="MUMPS: error"):
        M.factorMUMPS(A, 1)
