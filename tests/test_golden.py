"""Oracle vs the committed golden vectors (regression pin; generator: tests/golden/make_golden.py)."""
import os
import numpy as np
import pytest

from tests.helpers import GOLDEN, make_problem, oracle_eval, relmax


@pytest.mark.parametrize("name", ["tiny", "cfg2"])
def test_oracle_reproduces_golden(name):
    g = np.load(os.path.join(GOLDEN, f"{name}.npz"))
    mesh, data, inv, m = make_problem(name)
    assert np.array_equal(m, g["m"])
    keep = {}
    pred, misfit, grad = oracle_eval(mesh, data, inv, m, keep=keep)
    assert relmax(pred, g["pred"]) < 1e-12
    assert abs(misfit - float(g["misfit"])) / float(g["misfit"]) < 1e-12
    assert relmax(grad, g["grad"]) < 1e-10
    ny = mesh.gridSize[0]
    zid = len(mesh.airLayer)
    rows = slice(zid * (ny + 1), (zid + 2) * (ny + 1))
    assert relmax(keep["exTE"][rows, :], g["exTE_rx"]) < 1e-12
    assert relmax(keep["hxTM"][rows, :], g["hxTM_rx"]) < 1e-12


def test_golden_leapfrog_trajectory():
    from oracle import hmcmt_oracle as O
    from hmcmt2d_amd.structs import HMCPrior
    g = np.load(os.path.join(GOLDEN, "tiny.npz"))
    mesh, data, inv, m = make_problem("tiny")
    O.setupTensorMesh2D(mesh)
    inv.refModel = g["lf_mref"].copy()
    prior = HMCPrior(dt=float(g["lf_dt"]), timestep=[3, 3], sigBounds=list(g["lf_bounds"]), regParam=1.0)
    m1, p1 = O.proposeLeapfrog(g["lf_m0"], g["lf_p0"], np.ones(len(m)), mesh, data, inv, prior, 3, False)
    assert relmax(m1, g["lf_m1"]) < 1e-10 and relmax(p1, g["lf_p1"]) < 1e-9
    # the trajectory exercises the bound reflection
    assert np.any(g["lf_m0"] + 3 * 0.03 * 2.5 > np.log(prior.sigBounds[1]))


def test_oracle_reproduces_cfg1_golden():
    g = np.load(os.path.join(GOLDEN, "cfg1.npz"))
    mesh, data, inv, m = make_problem("cfg1")
    pred, misfit, grad = oracle_eval(mesh, data, inv, m)
    assert relmax(pred, g["pred"]) < 1e-12 and relmax(grad, g["grad"]) < 1e-10
    assert abs(misfit - float(g["misfit"])) / float(g["misfit"]) < 1e-12


def test_oracle_reproduces_cfg3_subset_golden():
    """Headline mesh, 4 of 16 frequencies (about 15 s of SuperLU on one core)."""
    from tests.helpers import cfg3_subset_problem
    g = np.load(os.path.join(GOLDEN, "cfg3s.npz"))
    mesh, data, inv, m, data16, inv16 = cfg3_subset_problem(g)
    assert np.array_equal(m, g["m"]) and len(data16.freqs) == 16 and len(inv16.obsData) == 16 * 41 * 2
    assert np.array_equal(inv16.obsData[np.isin(data16.freqID - 1, g["fidx"])], inv.obsData)
    pred, misfit, grad = oracle_eval(mesh, data, inv, m)
    assert relmax(pred, g["pred"]) < 1e-12 and relmax(grad, g["grad"]) < 1e-9


def test_oracle_reproduces_entries_of_the_full_size_goldens():
    """cfg3.npz (headline config, all 16 frequencies) and cfg5s.npz (stress mesh, 3 frequencies) take minutes of oracle
    time to regenerate; here the oracle's forward response of one frequency of each is held against the stored
    predicted data (frequencies that are NOT in the cfg3 subset golden), and the cfg3 golden against the subset golden
    where they overlap."""
    from oracle import hmcmt_oracle as O
    from hmcmt2d_amd import synthetic as S
    g = np.load(os.path.join(GOLDEN, "cfg3.npz")); gs = np.load(os.path.join(GOLDEN, "cfg3s.npz"))
    mesh, data, inv, m = make_problem("cfg3")
    assert np.array_equal(m, g["m"]) and np.array_equal(inv.obsData, gs["obs16"])
    sel = np.isin(data.freqID - 1, gs["fidx"])
    assert relmax(g["pred"][sel], gs["pred"]) < 1e-12 and relmax(g["pred_true"][sel], gs["pred_true"]) < 1e-12
    O.setupTensorMesh2D(mesh)
    sig = inv.bgModel.copy(); sig[inv.activeIdx] += np.exp(m); mesh.sigma = sig
    one = S.make_data_layout(data.freqs[[7]], data.rxLoc[:, 0])
    pred, _ = O.MT2DFwdSolver(mesh, one)
    assert relmax(pred, g["pred"][data.freqID == 8]) < 1e-12
    g5 = np.load(os.path.join(GOLDEN, "cfg5s.npz"))
    mesh5, data32, _ = S.make_config("cfg5")
    O.setupTensorMesh2D(mesh5)
    from tests.helpers import start_sigma
    from hmcmt2d_amd import invsetup as I
    mesh5.sigma = start_sigma(mesh5)
    inv5 = I.setupInverseDataModel(mesh5, [S.SIG_AIR], 0.0, 0.0, g5["obs"], g5["err"])
    sig = inv5.bgModel.copy(); sig[inv5.activeIdx] += np.exp(g5["m"]); mesh5.sigma = sig
    one = S.make_data_layout(data32.freqs[g5["fidx"][[1]]], data32.rxLoc[:, 0])
    pred, _ = O.MT2DFwdSolver(mesh5, one)
    n = len(pred)
    assert relmax(pred, g5["pred"][n:2 * n]) < 1e-12
    # cfg5.npz (round 6): the stress configuration in FULL, all 32 frequencies -- its predicted data at the subset's frequencies are the
    # subset golden's (same model, same frequencies; the observations differ: noise drawn on another array), and one more frequency
    # that is in neither is reproduced by the oracle's forward solve
    g5f = np.load(os.path.join(GOLDEN, "cfg5.npz"))
    assert np.array_equal(g5f["m"], g5["m"]) and len(g5f["pred"]) == 2 * 32 * 81
    sel = np.isin(data32.freqID - 1, g5["fidx"])
    assert relmax(g5f["pred"][sel], g5["pred"]) < 1e-12 and relmax(g5f["pred_true"][sel], g5["pred_true"]) < 1e-12
    one = S.make_data_layout(data32.freqs[[9]], data32.rxLoc[:, 0])
    pred, _ = O.MT2DFwdSolver(mesh5, one)
    assert relmax(pred, g5f["pred"][data32.freqID == 10]) < 1e-12


@pytest.mark.parametrize("name,ndata,grid,nfreq,nrx", [("dprism3d", 902, (96, 56), 11, 41), ("coprod2", 470, (76, 52), 12, 20)])
def test_reference_example_files_are_read_and_reproduced_by_the_oracle(name, ndata, grid, nfreq, nrx):
    """The reference's example directories (HMCMT/examples/<name>/{startupfile,*.mod,*.dat}, committed unchanged under
    tests/golden/examples/) through readstartupFile (readstartupFile.jl:4-103, readMT2DData.jl:14-179,
    readEMModel2D.jl:11-154), and the oracle's gradient at a seeded perturbation of the file's start model."""
    from hmcmt2d_amd.fileio import readstartupFile
    from hmcmt2d_amd.structs import HMCPrior
    from oracle import hmcmt_oracle as O
    g = np.load(os.path.join(GOLDEN, f"example_{name}.npz"))
    mesh, data, inv, prior = readstartupFile(os.path.join(GOLDEN, "examples", name, "startupfile"))
    assert mesh.gridSize == grid and len(inv.obsData) == ndata and len(data.freqs) == nfreq and data.rxLoc.shape[0] == nrx
    assert data.dataType == "Impedance" and data.dataComp == ["ZXY", "ZYX"] and data.dataID.sum() == ndata
    assert data.dataID.size == 2 * nfreq * nrx                      # coprod2: 470 of 480 present
    assert prior.totalsamples == 10000 and prior.burninsamples == 100 and prior.timestep == [6, 10]
    assert prior.dt == (0.03 if name == "dprism3d" else 0.015) and prior.regParam == 1.0
    assert np.array_equal(inv.strModel, g["m0"]) and len(mesh.airLayer) == 7
    O.setupTensorMesh2D(mesh)
    inv.strModel = g["m1"].copy()
    pred, misfit, grad = O.compDataGradient(mesh, data, inv, HMCPrior(), False)
    assert relmax(pred, g["pred1"]) < 1e-12 and relmax(grad, g["grad1"]) < 1e-9


def _read_reference_output(path):
    """tests/golden/reference_dprism3d.txt as julia/crosscheck_reference.jl writes it -> {tag: (pred, misfit, grad)}."""
    out, cur, tag = {}, None, None
    with open(path) as f:
        lines = [l.strip() for l in f if l.strip() and not l.startswith("#")]
    i = 0
    while i < len(lines):
        w = lines[i].split()
        if len(w) == 3 and w[1] == "misfit":
            tag = w[0]; out[tag] = {"misfit": float(w[2])}; i += 1
        elif len(w) == 3 and w[1] in ("pred", "grad"):
            n = int(w[2]); block = lines[i + 1:i + 1 + n]
            if w[1] == "pred":
                a = np.array([[float(x) for x in b.split()] for b in block]); out[tag]["pred"] = a[:, 0] + 1j * a[:, 1]
            else:
                out[tag]["grad"] = np.array([float(b) for b in block])
            i += 1 + n
        else:
            raise ValueError(f"unexpected line {i}: {lines[i][:60]}")
    return out


def test_reference_output_of_the_julia_crosscheck_when_present():
    """The ONLY reference-held values of the gradient half: julia/crosscheck_reference.jl runs the
    unmodified reference compDataGradient (HMCSampler.jl:277-330, compJacTMatVec.jl:8-327) on examples/dprism3d at the file's
    start model m0 and at the committed perturbation m1 and writes tests/golden/reference_dprism3d.txt.  There is no Julia in the
    build image, so the file is normally absent and this test says so; with the file present the committed golden (what the HIP
    path is held to, tests/test_gpu_parity_full.py) and a fresh oracle evaluation must reproduce it."""
    path = os.path.join(GOLDEN, "reference_dprism3d.txt")
    m1txt = np.loadtxt(os.path.join(GOLDEN, "reference_crosscheck", "dprism3d_m1.txt"))
    g = np.load(os.path.join(GOLDEN, "example_dprism3d.npz"))
    assert np.array_equal(m1txt, g["m1"])                      # the text fixture the Julia script reads IS the golden's m1
    if not os.path.exists(path):
        pytest.skip("reference output absent: run julia/crosscheck_reference.jl in HMCMT/examples/dprism3d (needs Julia >= 1.10) to pin the gradient half")
    ref = _read_reference_output(path)
    ny, nzE = 96, 49
    for tag, k in (("m0", "0"), ("m1", "1")):
        r = ref[tag]
        assert relmax(g["pred" + k], r["pred"]) < 1e-9 and abs(g["misfit" + k] - r["misfit"]) / r["misfit"] < 1e-9
        gg, gr = g["grad" + k].reshape(nzE, ny), r["grad"].reshape(nzE, ny)
        scale = np.abs(gr).max()
        # (deepest five rows: the reference formula's own gradient is rounding-dependent there, DESIGN section 2 -- held at m1 only, looser)
        assert np.abs(gg[:-5] - gr[:-5]).max() < 1e-6 * scale
        if tag == "m1":
            assert np.abs(gg[-5:] - gr[-5:]).max() < 1e-4 * scale
    from hmcmt2d_amd.fileio import readstartupFile
    from hmcmt2d_amd.structs import HMCPrior
    from oracle import hmcmt_oracle as O
    mesh, data, inv, prior = readstartupFile(os.path.join(GOLDEN, "examples", "dprism3d", "startupfile"))
    O.setupTensorMesh2D(mesh)
    inv.strModel = g["m1"].copy()
    pred, misfit, grad = O.compDataGradient(mesh, data, inv, HMCPrior(), True)      # (dense dBC: the reference's own form)
    assert relmax(pred, ref["m1"]["pred"]) < 1e-9 and np.abs(grad - ref["m1"]["grad"])[: -5 * ny].max() < 1e-6 * np.abs(ref["m1"]["grad"]).max()


def test_the_parallel_chain_generator_evaluates_what_the_serial_oracle_evaluates(monkeypatch):
    """tests/golden/make_chain_par.py (the generator of cfg3_chain.npz, cfg5_chain.npz and the *_rough_traj.npz files) hands the
    oracle's frequency loop to worker processes: every worker runs the oracle's own compDataGradient / MT2DFwdSolver on a pair of
    frequencies' data, the parent concatenates the predicted data and adds misfits and gradients.  Here, in one process on
    BASELINE configs[1]: the split evaluation against the serial oracle at the same model -- predicted data and misfit equal
    (bitwise with single-threaded BLAS on both sides; 1e-13 allows for a threaded one), the gradient to the order of one sum --,
    and the forward-only task against the same predicted data."""
    import importlib.util
    from threadpoolctl import threadpool_limits
    from oracle import hmcmt_oracle as O
    from hmcmt2d_amd.structs import HMCPrior
    monkeypatch.setenv("HMCMT_CHAIN_NAME", "cfg2")
    spec = importlib.util.spec_from_file_location("make_chain_par", os.path.join(GOLDEN, "make_chain_par.py"))
    mcp = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mcp)
    assert mcp.NAME == "cfg2"
    mesh, data, inv, m = make_problem("cfg2")
    O.setupTensorMesh2D(mesh)
    inv.strModel = m.copy()
    with threadpool_limits(limits=1):
        p0, f0, g0 = O.compDataGradient(mesh, data, inv, HMCPrior(), False)
    nW = (len(data.freqs) + mcp.PER - 1) // mcp.PER
    res = [mcp._task((w, m, True)) for w in range(nW)]
    p1, f1, g1 = np.concatenate([r[0] for r in res]), sum(r[1] for r in res), sum(r[2] for r in res)
    assert relmax(p1, p0) < 1e-13 and abs(f1 - f0) < 1e-13 * f0 and relmax(g1, g0) < 1e-12
    fw = [mcp._task((w, mesh.sigma, False)) for w in range(nW)]
    assert relmax(np.concatenate([r[0] for r in fw]), p0) < 1e-13
