"""Kernel arithmetic (the per-item bodies of hmcmt_items.h / hmcmt_math.h, instantiated on the host
by tests/emul) against the oracle.  No GPU involved; the same source is compiled into the HIP
kernels, which the -m gpu tests check through the C ABI."""
import os
import numpy as np
import pytest

from tests.helpers import GOLDEN, make_problem, oracle_eval, relmax
from tests.emul.emul_py import Emul


@pytest.fixture(scope="module")
def tiny():
    mesh, data, inv, m = make_problem("tiny")
    keep = {}
    pred, misfit, grad = oracle_eval(mesh, data, inv, m, dense_dbc=True, keep=keep)
    E = Emul(mesh, data, inv)
    out = E.grad(m, True, 1, 1e-13)
    return dict(mesh=mesh, data=data, inv=inv, m=m, keep=keep, oracle=(pred, misfit, grad), E=E, emul=out)


def test_fdm_eigenbasis(tiny):
    E, mesh = tiny["E"], tiny["mesh"]
    ny = mesh.gridSize[0]
    n = ny - 1
    V = E.get("Vpad").reshape(E.NYP, E.NYP)[1:ny, :n]
    lam = E.get("lam")[:n]
    y = mesh.yLen
    my = 0.5 * (y[:-1] + y[1:])
    Ty = np.diag(1 / y[:-1] + 1 / y[1:]) - np.diag(1 / y[1:-1], 1) - np.diag(1 / y[1:-1], -1)
    assert np.abs(V.T @ np.diag(my) @ V - np.eye(n)).max() < 1e-12
    assert np.abs(Ty @ V - np.diag(my) @ V @ np.diag(lam)).max() / np.abs(Ty).max() < 1e-12


def test_forward_fields_and_boundary_values(tiny):
    E, keep, mesh, data = tiny["E"], tiny["keep"], tiny["mesh"], tiny["data"]
    ny, nz = mesh.gridSize
    nF = len(data.freqs)
    X = E.get("X").reshape(E.S, E.NZP, E.NYP)[:, :, :ny + 1]
    ex = keep["exTE"].reshape(nz + 1, ny + 1, nF)
    hx = keep["hxTM"].reshape(nz + 1, ny + 1, nF)
    for f in range(nF):
        assert np.abs(X[f] - ex[:, :, f]).max() < 1e-9
        assert np.abs(X[nF + f] - hx[:, :, f]).max() < 1e-9


def test_pred_misfit_gradient(tiny):
    pred, misfit, grad = tiny["oracle"]
    p2, m2, g2 = tiny["emul"]
    assert relmax(p2, pred) < 1e-10
    assert abs(m2 - misfit) / misfit < 1e-10
    assert relmax(g2, grad) < 1e-8          # limited by the reference's unstable bottom-row dE (App. B.7)


def test_adjoint_fields_and_terms(tiny):
    E, keep, mesh, data, inv, m = (tiny[k] for k in ("E", "keep", "mesh", "data", "inv", "m"))
    ny, nz = mesh.gridSize
    nF = len(data.freqs)
    Lam = E.get("Lam").reshape(E.S, E.NZP, E.NYP)
    gP = E.get("gPart").reshape(2, -1)[:, inv.activeIdx]
    PT = [np.zeros(len(m), complex), np.zeros(len(m), complex)]
    for f in range(nF):
        for mi, md in enumerate(("TE", "TM")):
            t = keep["terms"][(md, f)]
            ev = t["eVal"].reshape(nz - 1, ny - 1)
            assert np.abs(Lam[mi * nF + f, 1:nz, 1:ny] - ev).max() / np.abs(ev).max() < 1e-8
            PT[mi] += t["PTv"] + (t["BTvii2"] if md == "TM" else 0)
    for mi in range(2):
        assert relmax(gP[mi], PT[mi].real) < 1e-8


def test_sensitivity_boundary_fields(tiny):
    """bc returned by getBCderivTM (no displacement term, mean-profile bottom) used in BTvii2."""
    E, keep, mesh, data = tiny["E"], tiny["keep"], tiny["mesh"], tiny["data"]
    ny, nz = mesh.gridSize
    nF = len(data.freqs)
    bL = E.get("bcsL").reshape(E.S, nz); bR = E.get("bcsR").reshape(E.S, nz); bB = E.get("bcsB")
    for f in range(nF):
        bc = keep["terms"][("TM", f)]["bc_sens"]
        assert np.abs(bL[nF + f] - bc[ny + 1:ny + nz + 1]).max() < 1e-9
        assert np.abs(bR[nF + f] - bc[ny + nz + 1:ny + 2 * nz + 1]).max() < 1e-9
        assert abs(bB[nF + f] - bc[-1]) < 1e-9


def test_receiver_derivatives_match_sparse_L_and_Q(tiny):
    """d0/d1/dq windows of rx_impedance_deriv against rows of the oracle's L, Q (dataFuncSens.jl)."""
    from oracle import hmcmt_oracle as O
    E, keep, mesh, data = tiny["E"], tiny["keep"], tiny["mesh"], tiny["data"]
    ny, nz = mesh.gridSize
    nF, nRx = len(data.freqs), data.rxLoc.shape[0]
    yNode = np.concatenate([[0.0], np.cumsum(mesh.yLen)]) - mesh.origin[0]
    zNode = np.concatenate([[0.0], np.cumsum(mesh.zLen)]) - mesh.origin[1]
    rx = O.preSetRxFieldSens(data.rxLoc, yNode, zNode, mesh.sigma)
    zid = rx.zid
    D = E.get("rxD").reshape(E.S, nRx, 11)
    X = E.get("X").reshape(E.S, E.NZP, E.NYP)
    for f in range(nF):
        omega = 2 * np.pi * data.freqs[f]
        for mi, fn in ((0, O.getDataFuncSensTE), (1, O.getDataFuncSensTM)):
            s = mi * nF + f
            F01 = np.stack([X[s, zid, :ny + 1], X[s, zid + 1, :ny + 1]], axis=1)
            L, Q = fn(omega, rx, F01, "Impedance")
            L, Q = L.toarray(), Q.toarray()
            for r in range(nRx):
                mine = np.zeros((nz + 1) * (ny + 1), complex)
                # recover n0 from the support of the oracle row
                nzidx = np.nonzero(L[r])[0]
                n0 = nzidx.min() % (ny + 1)
                n0 = min(n0, max(0, ny - 3))
                # find the window offset that matches (the kernel stores its own n0 separately)
                best = None
                for cand in range(max(0, n0 - 2), n0 + 1):
                    v = np.zeros_like(mine)
                    for i in range(4):
                        if cand + i <= ny:
                            v[zid * (ny + 1) + cand + i] += D[s, r, i]
                            v[(zid + 1) * (ny + 1) + cand + i] += D[s, r, 4 + i]
                    e = np.abs(v - L[r]).max() / np.abs(L[r]).max()
                    best = e if best is None else min(best, e)
                assert best < 1e-10
                q = np.abs(Q[r]).max()
                if q > 0:
                    assert abs(np.abs(D[s, r, 8:11]).max() - q) / q < 1e-10


def test_jacobi_preconditioned_path_agrees(tiny):
    E, m = tiny["E"], tiny["m"]
    p1, m1, g1 = tiny["emul"]
    p0, m0, g0 = E.grad(m, True, 0, 1e-13)
    assert relmax(p0, p1) < 1e-10 and relmax(g0, g1) < 1e-8
    assert E.iters.max() > 3 * 14                       # Jacobi needs far more iterations than FDM


def test_cfg2_against_golden():
    g = np.load(os.path.join(GOLDEN, "cfg2.npz"))
    mesh, data, inv, m = make_problem("cfg2")
    E = Emul(mesh, data, inv)
    pred, misfit, grad = E.grad(m, True, 1, 1e-12)
    assert relmax(pred, g["pred"]) < 1e-9
    assert abs(misfit - float(g["misfit"])) / float(g["misfit"]) < 1e-9
    assert relmax(grad, g["grad"]) < 1e-7
    assert E.iters.max() < 40


def test_te_only_subset_of_the_data():
    """Only ZXY data, one (freq, rx) missing: the TM polarisation is never solved and masked out."""
    from hmcmt2d_amd import synthetic as S, invsetup as I
    from hmcmt2d_amd.structs import MTData
    mesh, data, inv, m = make_problem("tiny")
    keepm = (data.dtID == 1) & ~((data.freqID == 2) & (data.rxID == 3))
    nF, nR = len(data.freqs), data.rxLoc.shape[0]
    dataID = np.zeros((nF, nR, 1), bool)
    dataID[data.freqID[keepm] - 1, data.rxID[keepm] - 1, 0] = True
    d2 = MTData(data.rxLoc, data.freqs, "Impedance", ["ZXY"], data.rxID[keepm], data.freqID[keepm],
                np.ones(keepm.sum(), np.int64), dataID.reshape(-1), True, False)
    inv2 = I.setupInverseDataModel(mesh, [S.SIG_AIR], 0, 0, inv.obsData[keepm], (1.0 / inv.dataW)[keepm])
    E = Emul(mesh, d2, inv2)
    pred, f, g = E.grad(m, True, 1, 1e-13)
    po, fo, go = oracle_eval(mesh, d2, inv2, m)
    assert relmax(pred, po) < 1e-10 and abs(f - fo) / fo < 1e-10 and relmax(g, go) < 1e-8
    assert E.iters[:, nF:].max() == 0


def test_kernel_bodies_under_address_and_undefined_behaviour_sanitizers():
    """SURVEY section 5 (sanitizers on the CPU build; never on the GPU box): the host instantiation of the kernel bodies
    (hmcmt_items.h / hmcmt_math.h / hmcmt_host.h: the source the HIP kernels inline) compiled with
    -fsanitize=address,undefined, and this file's tests run against it in a child process with libasan preloaded.
    Any out-of-bounds access, use after free, signed overflow, misaligned or null access aborts the child."""
    import subprocess
    import sys
    if os.environ.get("HMCMT_EMUL_SANITIZE"):
        pytest.skip("already inside the sanitized child")
    asan = subprocess.check_output(["gcc", "-print-file-name=libasan.so"], text=True).strip()
    if not os.path.isabs(asan):
        pytest.skip("libasan is not installed")
    env = dict(os.environ, HMCMT_EMUL_SANITIZE="1", LD_PRELOAD=asan,
               ASAN_OPTIONS="detect_leaks=0:halt_on_error=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-q", "-x", "-p", "no:cacheprovider",
                        "-k", "not sanitizers"], env=env, capture_output=True, text=True, timeout=280,
                       cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    out = r.stdout + r.stderr
    assert "AddressSanitizer" not in out and "runtime error:" not in out, out[-3000:]
    assert r.returncode == 0 and " passed" in out, out[-3000:]
