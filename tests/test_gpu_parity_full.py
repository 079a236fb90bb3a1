"""Parity at the sizes and inputs that matter: BASELINE `configs[0]` (cfg1) in full, the headline
mesh (cfg3) on a frequency subset the oracle finishes in seconds, the reference's own example directories read from
their files, and every intermediate the oracle exposes on the small config (full nodal fields, adjoint fields, the
per-system J^T v sums).  All through the C ABI.

Tolerances = the levels measured on MI355X at the default solver tolerance 1e-11 (`python -m tests.tools.gpu_parity_levels`,
log in profiles/r02_parity_levels.log) x 3-5:
  predData, misfit        1e-9 relative            (measured 1e-12 .. 3e-10)
  gradient                1e-8 of max|g| away from the deepest rows (measured 1e-10 .. 2e-9; 1.6e-8 at the true model of
                          cfg3, where max|g| is 1000x smaller: 5e-8 there); 5e-7 in the deepest rows (measured <= 6e-8) --
                          the reference's own rounding floor, tests/test_oracle_kat.py::test_reference_gradient_is_ill_
                          conditioned_in_the_deepest_rows
  nodal fields            1e-9 of max|field| (measured 1e-11); adjoint fields 1e-8 (measured 2e-9)
  true residual           1e-9 (measured 1e-11 .. 3e-10; 1.4e-8 at the half-space start model of dprism3d, see there; was 1e-8 in round 1.  The reference's own bar, 1e-14 in
                          MUMPS/test/testDivGrad.jl:19, is for a direct solver; with options.tol = 1e-12 the iterative
                          solves reach 3e-12 .. 6e-12 at one more iteration)
"""
import os
import numpy as np
import pytest

from hmcmt2d_amd import synthetic as S, invsetup as I
from hmcmt2d_amd.fileio import readstartupFile
from hmcmt2d_amd.lib import HipContext
from hmcmt2d_amd.structs import MTData
from tests.helpers import (GOLDEN, make_problem, oracle_eval, relmax, cfg3_subset_problem, gerr_split,
                           dprism_generating_problem, assert_reproduces_dprism_file)

pytestmark = pytest.mark.gpu

PRED_TOL, GRAD_TOL, GRAD_DEEP_TOL, RES_TOL = 1e-9, 1e-8, 5e-7, 1e-9


def _check(ctx, m, pred_ref, misfit_ref, grad_refs, inv, mesh, deep_rows=5, grad_tol=GRAD_TOL, misfit_tol=PRED_TOL, pred_tol=PRED_TOL,
           res_tol=RES_TOL):
    """grad_refs: one reference gradient, or several rounding-equivalent evaluations of the reference formula (the
    gradient must agree with one of them, see tests/golden/make_golden.py::make_example)."""
    pred, misfit, grad = ctx.grad(m)
    st = ctx.stats()
    assert st["status"] == 0 and st["true_res_max"] < res_tol, st
    assert relmax(pred, pred_ref) < pred_tol and abs(misfit - misfit_ref) / misfit_ref < misfit_tol
    refs = grad_refs if isinstance(grad_refs, (list, tuple)) else [grad_refs]
    errs = [gerr_split(grad, r, inv, mesh, deep_rows) for r in refs]
    shallow, deep = min(errs)
    assert shallow < grad_tol and deep < GRAD_DEEP_TOL, errs
    return pred, misfit, grad


def _ran_the_persistent_kernel(ctx, parts=None, width=None):
    """Which solver a parity test tested: the one-launch-per-solve kernel, with the expected number of column
    parts -- not, silently, the launch-per-phase loop (a context leaked by an earlier test, a lost device lock) -- and, where
    the mesh has one, the width-specialised instantiation of it (hmcmt_persist_width: 112 / 208 / 416 padded nodes per row)."""
    info = ctx.persist_info()
    assert info["usable_now"] == 1 and info["enabled"] == 1 and info["solves"] >= 2 and info["placement_fallbacks"] == 0 and info["timeouts"] == 0, info
    if parts is not None:
        assert info["column_parts"] == parts, info
    if width is not None:
        assert ctx.persist_width() == width, ctx.persist_width()


def test_cfg1_full_parity():
    """BASELINE configs[0]: the dprism example mesh (96x49 + 7 air rows), 2-layer model, 4 frequencies -- oracle live
    and the committed golden."""
    mesh, data, inv, m = make_problem("cfg1")
    g = np.load(os.path.join(GOLDEN, "cfg1.npz"))
    ctx = HipContext(mesh, data, inv, verify=True)
    _check(ctx, m, g["pred"], float(g["misfit"]), g["grad"], inv, mesh)
    po, mo, go = oracle_eval(mesh, data, inv, m)
    _check(ctx, m, po, mo, go, inv, mesh)
    ex, hx = ctx.fields()
    ny = mesh.gridSize[0]; zid = len(mesh.airLayer)
    rows = slice(zid * (ny + 1), (zid + 2) * (ny + 1))
    assert relmax(ex[rows], g["exTE_rx"]) < 1e-9 and relmax(hx[rows], g["hxTM_rx"]) < 1e-9
    _ran_the_persistent_kernel(ctx, 1, width=112)
    ctx.close()


def test_cfg3_mesh_frequency_subset_parity_and_full_run_agreement():
    """The headline mesh (200x100 cells + 7 air rows) with 4 of the 16 frequencies (100, 4.64, 0.215, 0.01 Hz), TE+TM:
    predData / misfit / gradient against the oracle's golden at the rough bench state AND at the true model (where the
    lateral structure is); then the full 16-frequency context at the same data: its systems at the subset's
    frequencies must reproduce the subset run's forward and adjoint fields (the systems of a batch are independent)."""
    g = np.load(os.path.join(GOLDEN, "cfg3s.npz"))
    mesh, data, inv, m, data16, inv16 = cfg3_subset_problem(g)
    assert np.array_equal(m, g["m"])
    ctx = HipContext(mesh, data, inv, verify=True)
    _check(ctx, m, g["pred"], float(g["misfit"]), g["grad"], inv, mesh)
    ex, hx = ctx.fields()
    ny = mesh.gridSize[0]; zid = len(mesh.airLayer)
    rows = slice(zid * (ny + 1), (zid + 2) * (ny + 1))
    assert relmax(ex[rows], g["exTE_rx"]) < 1e-9 and relmax(hx[rows], g["hxTM_rx"]) < 1e-9
    ea, ha = ctx.fields(adjoint=True)
    _, sig_true = S.make_config("cfg3")[1:]
    m_true = np.log(sig_true[inv.activeIdx])
    # (at the true model the residuals are the 3 % noise, so the misfit's relative error is ~30x the predicted data's:
    #  measured 3e-10 .. 2e-9 for predicted data good to 3e-10 .. 8e-10)
    #  The gradient there is 1000x smaller than at the rough state (the residuals it is built from are noise) with the
    #  same absolute error: 3e-7 of its own maximum (measured 2e-8 .. 1e-7), i.e. 1e-10 of the rough state's.
    _, _, g_true = _check(ctx, m_true, g["pred_true"], float(g["misfit_true"]), g["grad_true"], inv, mesh, grad_tol=3e-7, misfit_tol=2e-8)
    assert np.abs(g_true - g["grad_true"]).max() < 1e-9 * np.abs(g["grad"]).max()
    _ran_the_persistent_kernel(ctx, 1, width=208)
    ctx.close()
    # full headline problem: 16 frequencies, the subset's observations at the subset's frequencies
    ctx16 = HipContext(mesh, data16, inv16, verify=True)
    pred16, _, _ = ctx16.grad(m)
    assert ctx16.stats()["true_res_max"] < RES_TOL
    fidx = g["fidx"]
    sel = np.isin(data16.freqID - 1, fidx)
    assert relmax(pred16[sel], g["pred"]) < PRED_TOL
    ex16, hx16 = ctx16.fields()
    ea16, ha16 = ctx16.fields(adjoint=True)
    for a, b in ((ex16[:, fidx], ex), (hx16[:, fidx], hx), (ea16[:, fidx], ea), (ha16[:, fidx], ha)):
        assert np.abs(a - b).max() <= 1e-10 * np.abs(b).max()
    _ran_the_persistent_kernel(ctx16, 1, width=208)
    ctx16.close()


def test_cfg3_all_sixteen_frequencies_parity():
    """BASELINE configs[2] -- the configuration the metric is quoted on -- in full: 200x100 cells + 7 air rows, ALL 16
    frequencies, TE+TM, 1312 data: predData / misfit / gradient / receiver-row fields against the oracle's golden at
    the rough bench state and at the true model (tests/golden/make_golden.py::make_cfg3_full)."""
    g = np.load(os.path.join(GOLDEN, "cfg3.npz"))
    mesh, data, inv, m = make_problem("cfg3")
    assert np.array_equal(m, g["m"]) and len(data.freqs) == 16 and len(g["pred"]) == 2 * 16 * 41
    ctx = HipContext(mesh, data, inv, verify=True)
    _check(ctx, m, g["pred"], float(g["misfit"]), g["grad"], inv, mesh)
    ex, hx = ctx.fields()
    ny = mesh.gridSize[0]; zid = len(mesh.airLayer)
    rows = slice(zid * (ny + 1), (zid + 2) * (ny + 1))
    assert relmax(ex[rows], g["exTE_rx"]) < 1e-9 and relmax(hx[rows], g["hxTM_rx"]) < 1e-9
    _, sig_true = S.make_config("cfg3")[1:]
    m_true = np.log(sig_true[inv.activeIdx])
    # (true model: the residuals are the 3 % noise -- see the subset test above for the two tolerances)
    _, _, g_true = _check(ctx, m_true, g["pred_true"], float(g["misfit_true"]), g["grad_true"], inv, mesh, grad_tol=3e-7, misfit_tol=2e-8)
    assert np.abs(g_true - g["grad_true"]).max() < 1e-9 * np.abs(g["grad"]).max()
    # the default (non-verify, warm-started, mixed-precision) path the bench runs gives the same numbers
    ctx.set_options(verify=0)
    for _ in range(2):
        pred, misfit, grad = ctx.grad(m + 0.0)
    assert relmax(pred, g["pred"]) < PRED_TOL and gerr_split(grad, g["grad"], inv, mesh)[0] < GRAD_TOL
    info = ctx.persist_info()            # the headline shape of the kernel: 8 workgroups of 512 threads per system, 4 systems per XCD at a time
    assert info["threads_half"] == 256 and info["workgroups_per_system"] == 8 and info["slots_per_xcd"] == 4 and info["solves"] >= 6
    _ran_the_persistent_kernel(ctx, 1, width=208)
    ctx.close()


def test_cfg3_parity_of_the_four_strip_kernel(monkeypatch):
    """The headline configuration through k_cocg_persist4 (HMCMT_PERSIST_STRIPS=4: 1 024 threads per workgroup, four waves per SIMD,
    kernels_persist4.h) against the same golden: rough state and true model, verify on, then the warm-started default path."""
    monkeypatch.setenv("HMCMT_PERSIST_STRIPS", "4")
    g = np.load(os.path.join(GOLDEN, "cfg3.npz"))
    mesh, data, inv, m = make_problem("cfg3")
    ctx = HipContext(mesh, data, inv, verify=True)
    info = ctx.persist_info()
    assert info["strips"] == 4 and info["threads_half"] == 256 and info["workgroups_per_system"] == 8 and info["slots_per_xcd"] == 4
    _check(ctx, m, g["pred"], float(g["misfit"]), g["grad"], inv, mesh)
    _, sig_true = S.make_config("cfg3")[1:]
    m_true = np.log(sig_true[inv.activeIdx])
    _check(ctx, m_true, g["pred_true"], float(g["misfit_true"]), g["grad_true"], inv, mesh, grad_tol=3e-7, misfit_tol=2e-8)
    ctx.set_options(verify=0)
    for _ in range(2):
        pred, misfit, grad = ctx.grad(m + 0.0)
    assert relmax(pred, g["pred"]) < PRED_TOL and gerr_split(grad, g["grad"], inv, mesh)[0] < GRAD_TOL
    info = ctx.persist_info()
    assert info["solves"] >= 6 and info["placement_fallbacks"] == 0 and info["timeouts"] == 0 and info["enabled"] == 1
    ctx.close()


def test_cfg5_all_thirtytwo_frequencies_parity():
    """BASELINE configs[4] IN FULL: 400x200 cells + 7 air rows, ALL 32 frequencies, TE+TM, 5 184 data, 64 systems
    of 82 194 unknowns on the persistent kernel with two column parts -- predData / misfit / gradient / receiver-row fields against
    the oracle's golden at the rough state and at the true model (tests/golden/make_golden.py::make_cfg5_full, dense_dbc=False)."""
    g = np.load(os.path.join(GOLDEN, "cfg5.npz"))
    mesh, data, inv, m = make_problem("cfg5")
    assert np.array_equal(m, g["m"]) and len(data.freqs) == 32 and len(g["pred"]) == 2 * 32 * 81 and np.array_equal(inv.obsData, g["obs"])
    ctx = HipContext(mesh, data, inv, verify=True)
    _check(ctx, m, g["pred"], float(g["misfit"]), g["grad"], inv, mesh, res_tol=2e-8)
    ex, hx = ctx.fields()
    ny = mesh.gridSize[0]; zid = len(mesh.airLayer)
    rows = slice(zid * (ny + 1), (zid + 2) * (ny + 1))
    assert relmax(ex[rows], g["exTE_rx"]) < 1e-9 and relmax(hx[rows], g["hxTM_rx"]) < 1e-9
    _, sig_true = S.make_config("cfg5")[1:]
    m_true = np.log(sig_true[inv.activeIdx])
    _, _, g_true = _check(ctx, m_true, g["pred_true"], float(g["misfit_true"]), g["grad_true"], inv, mesh, grad_tol=3e-7, misfit_tol=2e-8, res_tol=2e-8)
    assert np.abs(g_true - g["grad_true"]).max() < 1e-9 * np.abs(g["grad"]).max()
    info = ctx.persist_info()
    assert info["workgroups_per_system"] == 30 and info["slots_per_xcd"] == 1 and info["slab_modes"] == 16 and info["column_parts"] == 2
    _ran_the_persistent_kernel(ctx, 2, width=416)
    ctx.close()


def test_cfg5_mesh_frequency_subset_parity_and_full_run_agreement():
    """BASELINE configs[4]'s mesh (400x200 cells + 7 air rows, 82 194 unknowns per system: since round 5 the persistent kernel
    with two column parts per row block; rounds 1-4: the wide-mesh launch-per-phase kernels) with 3 of its 32 frequencies (100, 0.59, 0.01 Hz): predData / misfit / gradient
    against the oracle's golden at the rough state and at the true model; then the full 32-frequency context must give,
    at the subset's frequencies, the subset run's predicted data."""
    g = np.load(os.path.join(GOLDEN, "cfg5s.npz"))
    mesh, data32, sig_true = S.make_config("cfg5")
    fidx = g["fidx"]
    data = S.make_data_layout(data32.freqs[fidx], data32.rxLoc[:, 0])
    from tests.helpers import start_sigma
    mesh.sigma = start_sigma(mesh)
    inv = I.setupInverseDataModel(mesh, [S.SIG_AIR], 0.0, 0.0, g["obs"], g["err"])
    m = S.rough_state(len(inv.strModel))
    assert np.array_equal(m, g["m"])
    ctx = HipContext(mesh, data, inv, verify=True)
    # (the solves stop on the ERROR estimate; the true residual that leaves is 4e-9 on this mesh -- its norm is
    #  dominated by the 1e8-weighted air rows of the TM systems, four times as many unknowns as at cfg3 -- for predicted
    #  data and gradients at the usual levels)
    _check(ctx, m, g["pred"], float(g["misfit"]), g["grad"], inv, mesh, res_tol=2e-8)
    ex, hx = ctx.fields()
    ny = mesh.gridSize[0]; zid = len(mesh.airLayer)
    rows = slice(zid * (ny + 1), (zid + 2) * (ny + 1))
    assert relmax(ex[rows], g["exTE_rx"]) < 1e-9 and relmax(hx[rows], g["hxTM_rx"]) < 1e-9
    m_true = np.log(sig_true[inv.activeIdx])
    _, _, g_true = _check(ctx, m_true, g["pred_true"], float(g["misfit_true"]), g["grad_true"], inv, mesh, grad_tol=3e-7, misfit_tol=2e-8, res_tol=2e-8)
    assert np.abs(g_true - g["grad_true"]).max() < 1e-9 * np.abs(g["grad"]).max()
    info = ctx.persist_info()            # the wide mesh runs the persistent kernel with two column parts: 15 row blocks x 2 = 30 workgroups, one system per XCD
    assert info["workgroups_per_system"] == 30 and info["slots_per_xcd"] == 1 and info["slab_modes"] == 16
    _ran_the_persistent_kernel(ctx, 2, width=416)
    ctx.close()
    # all 32 frequencies (the stress configuration itself): the subset's systems inside the full batch
    n32 = len(data32.rxID)
    obs32 = np.full(n32, 0.02 + 0.02j) * np.where(data32.dtID == 1, 1.0, -1.0)
    sel = np.isin(data32.freqID - 1, fidx)
    obs32[sel] = g["obs"]; err32 = np.full(n32, 1e-3); err32[sel] = g["err"]
    inv32 = I.setupInverseDataModel(mesh, [S.SIG_AIR], 0.0, 0.0, obs32, err32)
    ctx32 = HipContext(mesh, data32, inv32, verify=True)
    pred32, _, _ = ctx32.grad(m)
    assert ctx32.stats()["status"] == 0 and ctx32.stats()["true_res_max"] < 2e-8
    assert relmax(pred32[sel], g["pred"]) < PRED_TOL
    _ran_the_persistent_kernel(ctx32, 2, width=416)
    ctx32.close()


@pytest.mark.parametrize("name", ["dprism3d", "coprod2"])
def test_reference_example_directories(name):
    """HMCMT/examples/{dprism3d,coprod2} (startupfile + model + data files, committed unchanged as fixtures): read by
    readstartupFile, evaluated at the file's start model and at a seeded perturbation of it.  coprod2 is field data on
    a non-synthetic mesh with 470 of 480 data present (masked), dprism3d the 96x49 synthetic with 11 frequencies."""
    g = np.load(os.path.join(GOLDEN, f"example_{name}.npz"))
    mesh, data, inv, prior = readstartupFile(os.path.join(GOLDEN, "examples", name, "startupfile"))
    assert np.array_equal(inv.strModel, g["m0"])
    ctx = HipContext(mesh, data, inv, verify=True)
    # (the homogeneous start model: the reference formula's gradient is rounding-dependent there, see make_example)
    # (... and the one model where the residual norm says little: for a half-space the FDM background IS the operator, the
    # solve ends after a handful of iterations on the ERROR estimate -- predData within 5e-12 of the oracle's --
    # while the 2^-9 rounding of the eigenvectors leaves 1e-8 of residual in short-wavelength modes that carry no error)
    _check(ctx, g["m0"], g["pred0"], float(g["misfit0"]), [g["grad0"], g["grad0_alt"][0], g["grad0_alt"][1]], inv, mesh,
           pred_tol=1e-10, res_tol=5e-8 if name == "dprism3d" else RES_TOL)
    _check(ctx, g["m1"], g["pred1"], float(g["misfit1"]), g["grad1"], inv, mesh)
    ex, hx = ctx.fields()
    ny = mesh.gridSize[0]; zid = len(mesh.airLayer)
    rows = slice(zid * (ny + 1), (zid + 2) * (ny + 1))
    assert relmax(ex[rows], g["exTE_rx"]) < 1e-9 and relmax(hx[rows], g["hxTM_rx"]) < 1e-9
    _ran_the_persistent_kernel(ctx, 1)
    ctx.close()


def test_hip_forward_reproduces_the_reference_example_data_to_the_last_printed_digit():
    """The HIP path against output of the reference authors' own forward code, no oracle in between: the noise-free
    imaginary parts and the error column of examples/dprism3d/dprism2dobs.dat for the model the file was generated
    from (tests/test_oracle_kat.py, oracle/pin/recover_dprism.py).  1e-3 of a digit of slack for the iterative
    solver's 1e-10."""
    mesh, data, obs, err = dprism_generating_problem()
    sig = mesh.sigma.copy()
    inv = I.setupInverseDataModel(mesh, [S.SIG_AIR], 0.0, 0.0, obs, err)
    ctx = HipContext(mesh, data, inv, verify=True)
    pred, misfit = ctx.forward(np.log(sig[inv.activeIdx]))
    assert ctx.stats()["status"] == 0 and ctx.stats()["true_res_max"] < RES_TOL
    assert_reproduces_dprism_file(pred, obs, err, slack=1e-3)
    # the misfit at the generating model is the noise alone: chi^2 / N close to 1 on the 902 real parts
    z = (obs - pred).real / err
    assert abs(2 * misfit - np.sum(z * z)) < 1e-3 and 0.8 < 2 * misfit / len(z) < 1.0
    ctx.close()


def test_full_fields_adjoint_fields_and_per_system_terms():
    """Everything tiny.npz holds: the complete nodal fields exTE / hxTM (boundary values included), the adjoint
    fields (`eVal` of compJacTMatVec.jl:221,292 on interior nodes), and -- by masking the data down to one frequency
    and one polarisation -- the J^T v contribution of every single system, PTv + BTvii (+ BTvii2) + BTvio + QTv."""
    g = np.load(os.path.join(GOLDEN, "tiny.npz"))
    mesh, data, inv, m = make_problem("tiny")
    ny, nz = mesh.gridSize
    ctx = HipContext(mesh, data, inv, verify=True)
    ctx.grad(m)
    assert ctx.stats()["true_res_max"] < RES_TOL
    ex, hx = ctx.fields()
    nF = len(data.freqs)
    from oracle import hmcmt_oracle as O
    ii, io = O.getBoundaryIndex(ny, nz)
    for got, ref, md in ((ex, g["exTE"], "TE"), (hx, g["hxTM"], "TM")):
        for f in range(nF):
            sc = np.abs(ref[:, f]).max()
            assert np.abs(got[:, f] - ref[:, f]).max() < 1e-9 * sc
            # Dirichlet values on all four sides (bc vector in the order of getBoundaryIndex, MT2DFwdSolver.jl:232-244)
            assert np.abs(got[io, f] - g[f"{md}{f}_bc"]).max() < 1e-10 * sc
    ea, ha = ctx.fields(adjoint=True)
    for got, md in ((ea, "TE"), (ha, "TM")):
        for f in range(nF):
            ref = g[f"{md}{f}_eVal"]
            assert np.abs(got[ii, f] - ref).max() < 1e-8 * np.abs(ref).max()      # (measured 2e-9: its source inherits the forward error)
            assert np.abs(got[io, f]).max() == 0.0
    ctx.close()
    # one system at a time through the data mask
    nR = data.rxLoc.shape[0]
    em = np.exp(m)
    for f in range(nF):
        for dt, md in ((1, "TE"), (2, "TM")):
            keep = (data.freqID == f + 1) & (data.dtID == dt)
            dataID = np.zeros((nF, nR, 2), bool)
            dataID[data.freqID[keep] - 1, data.rxID[keep] - 1, dt - 1] = True
            d1 = MTData(data.rxLoc, data.freqs, "Impedance", ["ZXY", "ZYX"], data.rxID[keep], data.freqID[keep],
                        data.dtID[keep], dataID.reshape(-1), True, True)
            inv1 = I.setupInverseDataModel(mesh, [S.SIG_AIR], 0, 0, inv.obsData[keep], (1.0 / inv.dataW)[keep])
            c1 = HipContext(mesh, d1, inv1, verify=True)
            _, _, g1 = c1.grad(m)
            c1.close()
            t = {k: g[f"{md}{f}_{k}"] for k in ("PTv", "BTvio", "QTv")}
            tot = t["PTv"] + t["BTvio"] + t["QTv"] + (g[f"TE{f}_BTvii"] if md == "TE" else g[f"TM{f}_BTvii1"] + g[f"TM{f}_BTvii2"])
            ref = em * tot.real
            shallow, deep = gerr_split(g1, ref, inv, mesh, 3)
            assert shallow < GRAD_TOL and deep < GRAD_DEEP_TOL, (md, f, shallow, deep)


@pytest.mark.parametrize("columns,blocked", [("0", "0"), ("2", "0"), ("16", "0"), ("0", "4,256,8,8"), ("0", "16,128,4,3")])
def test_boundary_fields_by_every_kernel_form(columns, blocked, monkeypatch):
    """The Dirichlet values of tiny.npz through the two-kernel form (HMCMT_BC_FUSED=0, HMCMT_BC_BLOCKED=0: k_bc_layers +
    k_bc_forward), through k_bc_fused with 2 and with 16 boundary columns per workgroup -- 16 puts both edge columns of the
    13-column mesh into one workgroup (its two-slot case), which the column count chosen for real meshes never does -- and
    through the layer-blocked k_bc_blocked (the stress size's form) with 4 columns per workgroup of 256 threads (11 layers:
    one whole block of 8 and one of 3) and with 16 per workgroup of 128 (blocks of 4 bottom -> top, of 3 top -> bottom: more
    blocks than the prefetch is deep)."""
    monkeypatch.setenv("HMCMT_BC_FUSED", columns)
    monkeypatch.setenv("HMCMT_BC_BLOCKED", blocked)
    g = np.load(os.path.join(GOLDEN, "tiny.npz"))
    mesh, data, inv, m = make_problem("tiny")
    ny, nz = mesh.gridSize
    ctx = HipContext(mesh, data, inv)
    ctx.grad(m)
    ex, hx = ctx.fields()
    from oracle import hmcmt_oracle as O
    ii, io = O.getBoundaryIndex(ny, nz)
    for got, ref, md in ((ex, g["exTE"], "TE"), (hx, g["hxTM"], "TM")):
        for f in range(len(data.freqs)):
            sc = np.abs(ref[:, f]).max()
            assert np.abs(got[io, f] - g[f"{md}{f}_bc"]).max() < 1e-10 * sc
    ctx.close()


@pytest.mark.parametrize("name", ["cfg2", "cfg1"])
def test_layer_blocked_boundary_fields_equal_the_fused_ones(name, monkeypatch):
    """k_bc_blocked (layer blocks through an LDS ring, the form of meshes too deep and wide for k_bc_fused's slabs) against
    k_bc_fused on meshes with several whole layer blocks: the same Dirichlet values on the boundary nodes of every system and
    the same predicted data.  Same item functions and order of operations per column; the two kernels' unrolled steps differ
    in the last bits (1e-15 of the value down to the last rows).  In the thick padding layers at the bottom the amplitude
    recurrence (mt1DField.jl:62-83: two amplitudes that grow like e^{+a} per layer while their sum decays like e^{-a})
    amplifies such a difference up to a thousandfold per layer, in any implementation (k_bc_fused against the ORACLE shows the
    same there): 1e-12 of the field's scale in the upper three quarters of the rows, 1e-8 below (where the values are 1e-4 of
    the surface field and less)."""
    mesh, data, inv, m = make_problem(name)
    ny, nz = mesh.gridSize
    from oracle import hmcmt_oracle as O
    ii, io = O.getBoundaryIndex(ny, nz)
    io = np.asarray(io)
    upper = io[io // (ny + 1) <= (3 * nz) // 4]
    out = {}
    for form, env in (("fused", {}), ("blocked", {"HMCMT_BC_FUSED": "0", "HMCMT_BC_BLOCKED": "24,512,14,8"})):
        for k in ("HMCMT_BC_FUSED", "HMCMT_BC_BLOCKED"):
            monkeypatch.delenv(k, raising=False)
        for k, val in env.items():
            monkeypatch.setenv(k, val)
        ctx = HipContext(mesh, data, inv)
        pred, _, _ = ctx.grad(m)
        ex, hx = ctx.fields()
        out[form] = (pred, ex, hx)
        ctx.close()
    for a, b in zip(out["fused"][1:], out["blocked"][1:]):
        assert np.abs(a[upper] - b[upper]).max() <= 1e-12 * np.abs(a).max()
        assert np.abs(a[io] - b[io]).max() <= 1e-8 * np.abs(a).max()
    assert np.abs(out["fused"][0] - out["blocked"][0]).max() <= 1e-10 * np.abs(out["fused"][0]).max()


@pytest.mark.parametrize("name", ["cfg3", "cfg5"])
def test_headline_chain_trajectory_from_the_rough_state_against_the_oracle(name):
    """ONE trajectory of the chain bench.py times (`value`: started at SURVEY 8(d)'s rough state) at the headline size and at the
    stress size, against the oracle's proposeLeapfrog (HMCSampler.jl:206-269; tests/golden/make_chain_par.py <name> traj ->
    <name>_rough_traj.npz): nine evaluations of the hot path along models that move by up to 5.2 in ln sigma (clamped steps: the
    gradient is 1e5), the device-resident leapfrog, the persistent kernel with the smoother the library picks there (cfg5: two
    column parts, the layer-blocked boundary fields).  The proposal (model: exact steps of a clamped momentum; momentum: eight
    gradients of size 1e5 added up), the Hamiltonian terms and the predicted data at it."""
    import copy
    from hmcmt2d_amd import sampler, synthetic as S
    from hmcmt2d_amd.structs import initHMCParameter, HMCPrior
    from oracle import hmcmt_oracle as O
    g = np.load(os.path.join(GOLDEN, f"{name}_rough_traj.npz"))
    mesh, data, inv, _ = make_problem(name)
    n = len(inv.strModel)
    inv.refModel = np.full(n, float(g["mref"][0]))
    m0, p0 = S.rough_state(n), O.getMomentumVector(n, np.ones(n), np.random.default_rng(int(g["seed"])))
    if "m0" in g.files:
        assert np.array_equal(m0, g["m0"]) and np.array_equal(p0, g["p0"])
    L = int(g["L"])
    prior = HMCPrior(dt=float(g["dt"]), timestep=[L, L], sigBounds=list(g["bounds"]), regParam=1.0)
    ctx = HipContext(mesh, data, inv)
    hp = initHMCParameter(n); hp.invM[:] = 1.0; hp.sqrtM[:] = 1.0
    hp.rhomodel, hp.momentum = m0.copy(), p0.copy()
    ctx.set_prior(inv.refModel, inv.Wm, hp.invM)
    inv_b = copy.deepcopy(inv)
    m1, p1 = sampler.proposeLeapfrogDevice(hp, mesh, data, inv_b, prior, None, L, ctx)
    assert prior.nfevals == L + 1 and ctx.stats()["status"] == 0
    _ran_the_persistent_kernel(ctx, 1 if name == "cfg3" else 2)
    hp2 = initHMCParameter(n); hp2.invM[:] = 1.0; hp2.momentum = p1
    d, k, h, mn, pred = sampler.getHamiltonian(data, mesh, inv_b, prior, hp2, ctx)
    em, ep = relmax(m1, g["m1"]), relmax(p1, g["p1"])
    print(f"\n[headline trajectory, {name}] model {em:.1e} momentum {ep:.1e} misfit {abs(d - float(g['D'])) / float(g['D']):.1e} kinetic {abs(k - float(g['K'])) / float(g['K']):.1e} "
          f"model norm {abs(mn - float(g['M'])) / float(g['M']):.1e} predicted data {relmax(pred, g['pred']):.1e}")
    assert em < 2e-8 and ep < 1e-6
    assert abs(d - float(g["D"])) < 1e-7 * float(g["D"]) and abs(k - float(g["K"])) < 1e-7 * float(g["K"]) and abs(mn - float(g["M"])) < 1e-7 * float(g["M"])
    assert relmax(pred, g["pred"]) < 1e-8
    ctx.close()


@pytest.mark.parametrize("name", ["tiny", "cfg1"])
def test_rho_phase_data_type(name):
    """DataType Rho_Pha (apparent resistivity + phase in degrees, both polarisations, a tenth of the data masked out;
    SURVEY 8(f)4): predicted data, misfit and gradient against the oracle and its golden; then the TE-only subset
    (RhoXY + PhsXY), for which the TM systems must not iterate.  On the tiny mesh and (end of round 6) on BASELINE
    configs[0]'s: 96 x 49 cells + 7 air rows, 4 frequencies, 41 receivers -- the width-specialised persistent kernel."""
    from tests.helpers import rhophase_problem
    mesh, data, inv, m, g = rhophase_problem(name)
    ctx = HipContext(mesh, data, inv, verify=True)
    # (rho_a = |Z|^2/(w mu0) doubles the impedance's relative error: 3e-9, measured 1e-9)
    rp = dict(deep_rows=3, pred_tol=3e-9, misfit_tol=3e-9)
    pred, misfit, grad = _check(ctx, m, g["pred"], float(g["misfit"]), g["grad"], inv, mesh, **rp)
    assert pred.dtype == np.float64 and pred.shape == g["obs"].shape
    po, mo, go = oracle_eval(mesh, data, inv, m)
    _check(ctx, m, po, mo, go, inv, mesh, **rp)
    pf, mf = ctx.forward(m + 0.01)
    assert pf.dtype == np.float64 and mf > 0
    ctx.close()
    te = data.dtID <= 2
    nF, nR = len(data.freqs), data.rxLoc.shape[0]
    dataID = np.zeros((nF, nR, 2), bool)
    dataID[data.freqID[te] - 1, data.rxID[te] - 1, data.dtID[te] - 1] = True
    d2 = MTData(data.rxLoc, data.freqs, "Rho_Pha", ["RhoXY", "PhsXY"], data.rxID[te], data.freqID[te], data.dtID[te],
                dataID.reshape(-1), True, False)
    inv2 = I.setupInverseDataModel(mesh, [S.SIG_AIR], 0, 0, inv.obsData[te], (1.0 / inv.dataW)[te])
    c2 = HipContext(mesh, d2, inv2, verify=True)
    po, mo, go = oracle_eval(mesh, d2, inv2, m)
    _check(c2, m, po, mo, go, inv2, mesh, **rp)
    assert c2.iters()[:, nF:].max() == 0
    c2.close()


def test_example_directory_short_chain_matches_the_oracle_chain():
    """HMCMT/examples/dprism3d run as the reference's runHMCscript.jl runs it -- readstartupFile, runHMCSampler,
    outputHMCSamples, getPosteriorModel -- but two samples long with two leapfrog steps each: the HIP chain (host
    leapfrog loop and device trajectory) against the oracle's chain under the same random stream."""
    import copy
    from oracle import hmcmt_oracle as O
    from hmcmt2d_amd import sampler, fileio
    mesh, data, inv, prior = readstartupFile(os.path.join(GOLDEN, "examples", "dprism3d", "startupfile"))
    prior.totalsamples, prior.burninsamples, prior.timestep = 2, 0, [2, 2]
    mesh_o, inv_o, prior_o = copy.deepcopy(mesh), copy.deepcopy(inv), copy.deepcopy(prior)
    O.setupTensorMesh2D(mesh_o)
    mo, so, do = O.runHMCSampler(mesh_o, data, inv_o, prior_o, np.random.default_rng(21), dense_dbc=False)
    for dev in (False, True):
        inv_p, prior_p = copy.deepcopy(inv), copy.deepcopy(prior)
        mp, sp_, dp = sampler.runHMCSampler(copy.deepcopy(mesh), data, inv_p, prior_p, np.random.default_rng(21),
                                            device_leapfrog=dev)
        sampler.release_context(inv_p)
        assert np.array_equal(sp_.acceptstats, so["acceptstats"]) and prior_p.nfevals == prior_o.nfevals
        assert relmax(mp, mo) < 1e-7 and relmax(dp, do) < 1e-7 and relmax(sp_.hmstats, so["hmstats"]) < 1e-7
    # the reference's writers on the result (byte-compatible formats)
    import tempfile
    with tempfile.TemporaryDirectory() as td:
        fileio.outputHMCSamples(mp, sp_, dp, ichain=1, cputime=1.0, outdir=td)
        mean, std = fileio.getPosteriorModel(mp, copy.deepcopy(mesh), inv, prior, outdir=td)
        assert sorted(os.listdir(td)) == ["hmcsamples_id1.data", "hmcsamples_id1.model", "hmcstatistics_id1.log",
                                          "meanModel.model", "stdModel.model"]
        back = fileio.readEMModel2D(os.path.join(td, "meanModel.model"))
        assert back.gridSize == mesh.gridSize and np.allclose(np.log(back.sigma[inv.activeIdx]), mean, atol=6e-3)
