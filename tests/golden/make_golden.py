"""Generates the golden vectors tests/golden/*.npz with the oracle (run in the build
container: `python tests/golden/make_golden.py [names...]`; names: tiny cfg2 cfg1 cfg3s cfg3 cfg5s cfg5 dprism3d coprod2 rhophase rhophase_cfg1).  Inputs: BASELINE.json-style synthetic configs
(hmcmt2d_amd/synthetic.py), observed data = oracle forward of the true model + 3 % seeded noise,
evaluation state m = ln(0.01) + 0.3 N(0,1) (seed 1).  Outputs: predData, misfit, gradient, the
receiver-row fields and (tiny only) every intermediate term of J^T v.

The oracle is a restatement, not the Julia reference (which cannot run here: no julia binary, MUMPS
blob stripped), so these vectors pin the HIP path and the oracle against EACH OTHER and against
regressions; the oracle itself is pinned by the known-answer tests in tests/test_oracle_kat.py.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import hmcmt_oracle as O                                   # noqa: E402
from hmcmt2d_amd import synthetic as S, invsetup as I                   # noqa: E402
from hmcmt2d_amd.structs import HMCPrior                                # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def make(name, full):
    mesh, data, sig_true = S.make_config(name)
    O.setupTensorMesh2D(mesh)
    mesh.sigma = sig_true.copy()
    pred_true, _ = O.MT2DFwdSolver(mesh, data)
    obs, err = S.noisy_observations(pred_true)
    ny, nz = mesh.gridSize
    nair = len(mesh.airLayer)
    mesh.sigma = np.concatenate([np.full(ny * nair, S.SIG_AIR), np.full(ny * (nz - nair), 0.01)])
    inv = I.setupInverseDataModel(mesh, [S.SIG_AIR], 0.0, 0.0, obs, err)
    m = S.rough_state(len(inv.strModel))
    inv.strModel = m.copy()
    keep = {}
    pred, misfit, grad = O.compDataGradient(mesh, data, inv, HMCPrior(), True, keep)
    zid = nair
    rows = slice(zid * (ny + 1), (zid + 2) * (ny + 1))
    out = dict(obs=obs, err=err, m=m, pred=pred, misfit=misfit, grad=grad,
               exTE_rx=keep["exTE"][rows, :], hxTM_rx=keep["hxTM"][rows, :])
    if full:
        out["exTE"], out["hxTM"] = keep["exTE"], keep["hxTM"]
        nF = len(data.freqs)
        for md in ("TE", "TM"):
            for f in range(nF):
                t = keep["terms"][(md, f)]
                for k, v in t.items():
                    out[f"{md}{f}_{k}"] = v
                out[f"{md}{f}_bc"] = keep["bc"][(md, data.freqs[f])]
        # one leapfrog trajectory with a bound reflection (fixed momentum draw, fixed L)
        prior = HMCPrior(dt=0.03, timestep=[3, 3], sigBounds=[1e-4, 0.0135], regParam=1.0)
        inv2 = I.setupInverseDataModel(mesh, [S.SIG_AIR], 0.0, 0.0, obs, err)
        inv2.strModel = m.copy()
        inv2.refModel = np.full(len(m), np.log(0.01))
        p0 = np.clip(np.random.default_rng(7).standard_normal(len(m)), -2.5, 2.5)
        m0 = np.minimum(m, np.log(0.0135) - 1e-3)
        m1, p1 = O.proposeLeapfrog(m0, p0, np.ones(len(m)), mesh, data, inv2, prior, 3, True)
        out.update(lf_m0=m0, lf_p0=p0, lf_m1=m1, lf_p1=p1, lf_mref=inv2.refModel,
                   lf_bounds=np.array(prior.sigBounds), lf_dt=prior.dt)
    np.savez_compressed(os.path.join(HERE, f"{name}.npz"), **out)
    print(name, "misfit", misfit, "|grad|max", np.abs(grad).max(), "file kB",
          os.path.getsize(os.path.join(HERE, f"{name}.npz")) // 1024)


def make_cfg3_subset():
    """Headline mesh (200x100 cells + 7 air rows), 4 of the 16 frequencies (100, 4.64, 0.215, 0.01 Hz), TE+TM:
    oracle pred / misfit / gradient at the rough bench state.  `obs16`/`err16` are the observations of ALL 16
    frequencies (oracle forward of the true model + seeded noise), so that a full 16-frequency HIP run sees, at the
    subset's frequencies, exactly the data of the subset run (its systems there must then agree with the subset run)."""
    mesh, data16, sig_true = S.make_config("cfg3")
    O.setupTensorMesh2D(mesh)
    mesh.sigma = sig_true.copy()
    pred16, _ = O.MT2DFwdSolver(mesh, data16)
    obs16, err16 = S.noisy_observations(pred16)
    fidx = np.array([0, 5, 10, 15])
    sel = np.isin(data16.freqID - 1, fidx)
    data = S.make_data_layout(data16.freqs[fidx], data16.rxLoc[:, 0])
    obs, err = obs16[sel], err16[sel]
    ny, nz = mesh.gridSize
    nair = len(mesh.airLayer)
    mesh.sigma = np.concatenate([np.full(ny * nair, S.SIG_AIR), np.full(ny * (nz - nair), 0.01)])
    inv = I.setupInverseDataModel(mesh, [S.SIG_AIR], 0.0, 0.0, obs, err)
    m = S.rough_state(len(inv.strModel))
    inv.strModel = m.copy()
    keep = {}
    pred, misfit, grad = O.compDataGradient(mesh, data, inv, HMCPrior(), False, keep)
    rows = slice(nair * (ny + 1), (nair + 2) * (ny + 1))
    out = dict(fidx=fidx, obs16=obs16, err16=err16, obs=obs, err=err, m=m, pred=pred, misfit=misfit, grad=grad,
               exTE_rx=keep["exTE"][rows, :], hxTM_rx=keep["hxTM"][rows, :])
    # second state: the true model (layers + block), where the lateral structure is; gradient only in float32-free form
    m_true = np.log(sig_true[inv.activeIdx])
    inv.strModel = m_true.copy()
    pred2, misfit2, grad2 = O.compDataGradient(mesh, data, inv, HMCPrior(), False)
    out.update(pred_true=pred2, misfit_true=misfit2, grad_true=grad2)
    np.savez_compressed(os.path.join(HERE, "cfg3s.npz"), **out)
    print("cfg3s misfit", misfit, misfit2, "file kB", os.path.getsize(os.path.join(HERE, "cfg3s.npz")) // 1024)


def make_cfg3_full():
    """BASELINE configs[2] in full: the headline mesh, ALL 16 frequencies, TE+TM, on the observations of cfg3s.npz
    (`obs16`/`err16`): oracle pred / misfit / gradient at the rough bench state and at the true model, plus the
    receiver-row fields of every frequency.  (About a minute of oracle time.)"""
    mesh, data, sig_true = S.make_config("cfg3")
    O.setupTensorMesh2D(mesh)
    g = np.load(os.path.join(HERE, "cfg3s.npz"))
    obs, err = g["obs16"], g["err16"]
    ny, nz = mesh.gridSize
    nair = len(mesh.airLayer)
    mesh.sigma = np.concatenate([np.full(ny * nair, S.SIG_AIR), np.full(ny * (nz - nair), 0.01)])
    inv = I.setupInverseDataModel(mesh, [S.SIG_AIR], 0.0, 0.0, obs, err)
    m = S.rough_state(len(inv.strModel))
    inv.strModel = m.copy()
    keep = {}
    pred, misfit, grad = O.compDataGradient(mesh, data, inv, HMCPrior(), False, keep)
    rows = slice(nair * (ny + 1), (nair + 2) * (ny + 1))
    out = dict(obs=obs, err=err, m=m, pred=pred, misfit=misfit, grad=grad, exTE_rx=keep["exTE"][rows, :], hxTM_rx=keep["hxTM"][rows, :])
    m_true = np.log(sig_true[inv.activeIdx])
    inv.strModel = m_true.copy()
    pred2, misfit2, grad2 = O.compDataGradient(mesh, data, inv, HMCPrior(), False)
    out.update(pred_true=pred2, misfit_true=misfit2, grad_true=grad2)
    np.savez_compressed(os.path.join(HERE, "cfg3.npz"), **out)
    print("cfg3 misfit", misfit, misfit2, "file kB", os.path.getsize(os.path.join(HERE, "cfg3.npz")) // 1024)


CFG5_FIDX = np.array([0, 16, 31])


def make_cfg5_subset():
    """BASELINE configs[4]'s mesh (400x200 cells + 7 air rows, 82 194 unknowns per system) with 3 of its 32
    frequencies (100, 0.59, 0.01 Hz), TE+TM: oracle pred / misfit / gradient at the rough state and at the true model.
    Observations = oracle forward of the true model at these frequencies + seeded noise."""
    mesh, data32, sig_true = S.make_config("cfg5")
    O.setupTensorMesh2D(mesh)
    data = S.make_data_layout(data32.freqs[CFG5_FIDX], data32.rxLoc[:, 0])
    mesh.sigma = sig_true.copy()
    pred_t, _ = O.MT2DFwdSolver(mesh, data)
    obs, err = S.noisy_observations(pred_t)
    ny, nz = mesh.gridSize
    nair = len(mesh.airLayer)
    mesh.sigma = np.concatenate([np.full(ny * nair, S.SIG_AIR), np.full(ny * (nz - nair), 0.01)])
    inv = I.setupInverseDataModel(mesh, [S.SIG_AIR], 0.0, 0.0, obs, err)
    m = S.rough_state(len(inv.strModel))
    inv.strModel = m.copy()
    keep = {}
    pred, misfit, grad = O.compDataGradient(mesh, data, inv, HMCPrior(), False, keep)
    rows = slice(nair * (ny + 1), (nair + 2) * (ny + 1))
    out = dict(fidx=CFG5_FIDX, obs=obs, err=err, m=m, pred=pred, misfit=misfit, grad=grad,
               exTE_rx=keep["exTE"][rows, :], hxTM_rx=keep["hxTM"][rows, :])
    m_true = np.log(sig_true[inv.activeIdx])
    inv.strModel = m_true.copy()
    pred2, misfit2, grad2 = O.compDataGradient(mesh, data, inv, HMCPrior(), False)
    out.update(pred_true=pred2, misfit_true=misfit2, grad_true=grad2)
    np.savez_compressed(os.path.join(HERE, "cfg5s.npz"), **out)
    print("cfg5s misfit", misfit, misfit2, "file kB", os.path.getsize(os.path.join(HERE, "cfg5s.npz")) // 1024)


def make_cfg5_full():
    """BASELINE configs[4] in full: the stress mesh (400x200 cells + 7 air rows, 82 194 unknowns per system), ALL 32 frequencies,
    TE+TM, 5 184 data (cfg5s.npz holds 3 of the 32).  Oracle pred / misfit / gradient at the rough state and
    at the true model on observations = oracle forward of the true model + seeded noise; receiver-row fields of every frequency.
    dense_dbc=False (the boundary-derivative products without the 1.6 GB dense matrix per system; the same numbers: tests/test_golden.py).
    About seven core-minutes; the file holds receiver rows only (< 3 MB)."""
    mesh, data, sig_true = S.make_config("cfg5")
    O.setupTensorMesh2D(mesh)
    mesh.sigma = sig_true.copy()
    pred_t, _ = O.MT2DFwdSolver(mesh, data)
    obs, err = S.noisy_observations(pred_t)
    ny, nz = mesh.gridSize
    nair = len(mesh.airLayer)
    mesh.sigma = np.concatenate([np.full(ny * nair, S.SIG_AIR), np.full(ny * (nz - nair), 0.01)])
    inv = I.setupInverseDataModel(mesh, [S.SIG_AIR], 0.0, 0.0, obs, err)
    m = S.rough_state(len(inv.strModel))
    inv.strModel = m.copy()
    keep = {}
    pred, misfit, grad = O.compDataGradient(mesh, data, inv, HMCPrior(), False, keep)
    rows = slice(nair * (ny + 1), (nair + 2) * (ny + 1))
    out = dict(obs=obs, err=err, m=m, pred=pred, misfit=misfit, grad=grad,
               exTE_rx=keep["exTE"][rows, :], hxTM_rx=keep["hxTM"][rows, :])
    m_true = np.log(sig_true[inv.activeIdx])
    inv.strModel = m_true.copy()
    pred2, misfit2, grad2 = O.compDataGradient(mesh, data, inv, HMCPrior(), False)
    out.update(pred_true=pred2, misfit_true=misfit2, grad_true=grad2)
    np.savez_compressed(os.path.join(HERE, "cfg5.npz"), **out)
    print("cfg5 misfit", misfit, misfit2, "file kB", os.path.getsize(os.path.join(HERE, "cfg5.npz")) // 1024)


def make_rho_phase(name="tiny"):
    """tiny config (and, end of round 6, cfg1: BASELINE configs[0]'s mesh, 96 x 49 cells + 7 air rows, 4 frequencies) with DataType
    Rho_Pha (apparent resistivity + phase of both polarisations, a tenth of the data masked out): observations = oracle response of
    the true model + 5 % / 1.5 degree noise."""
    mesh, dz, sig_true = S.make_config(name)
    data = S.make_rhophase_layout(dz.freqs, dz.rxLoc[:, 0])
    O.setupTensorMesh2D(mesh)
    mesh.sigma = sig_true.copy()
    full, _ = O.MT2DFwdSolver(mesh, data)
    rng = np.random.default_rng(20250114)
    keep = rng.random(len(full)) > 0.1
    data.dataID = keep.copy()
    data.rxID, data.freqID, data.dtID = data.rxID[keep], data.freqID[keep], data.dtID[keep]
    isrho = (data.dtID % 2) == 1
    err = np.where(isrho, 0.05 * np.abs(full[keep]), 1.5)
    obs = full[keep] + err * rng.standard_normal(keep.sum())
    ny, nz = mesh.gridSize
    nair = len(mesh.airLayer)
    mesh.sigma = np.concatenate([np.full(ny * nair, S.SIG_AIR), np.full(ny * (nz - nair), 0.01)])
    inv = I.setupInverseDataModel(mesh, [S.SIG_AIR], 0.0, 0.0, obs, err)
    m = S.rough_state(len(inv.strModel))
    inv.strModel = m.copy()
    pred, misfit, grad = O.compDataGradient(mesh, data, inv, HMCPrior(), True)
    np.savez_compressed(os.path.join(HERE, f"{name}_rhophase.npz"), keep=keep, obs=obs, err=err, m=m, pred=pred, misfit=misfit, grad=grad)
    print(f"{name}_rhophase nData", len(obs), "misfit", misfit, "|grad|max", np.abs(grad).max())


def make_example(name):
    """The reference's own example directory (data files copied as fixtures to tests/golden/examples/<name>/:
    startupfile, model file, data file -- inputs only): oracle compDataGradient at the file's start model and at a
    seeded perturbation of it, through the package's readstartupFile."""
    from hmcmt2d_amd.fileio import readstartupFile
    mesh, data, inv, prior = readstartupFile(os.path.join(HERE, "examples", name, "startupfile"))
    O.setupTensorMesh2D(mesh)
    m0 = inv.strModel.copy()
    out = dict(m0=m0)
    pred, misfit, grad = O.compDataGradient(mesh, data, inv, HMCPrior(), True)
    out.update(pred0=pred, misfit0=misfit, grad0=grad)
    # The reference formula's gradient is rounding-dependent where the 1-D recurrence's overflow cut-off fires in the
    # LAST (thick) layer: the row computed just before the cut-off is kept (MT1DSensitivity.jl:145-155 zeroes the
    # lower-right block only) and holds amplified rounding noise.  At dprism3d's homogeneous start model a relative
    # perturbation of 3e-14 of the model moves the reference's own gradient by 11.8 x max|g| (1e-14: by 1e-6).  So the
    # golden holds the evaluations at m0*(1 +- 1e-14) as well: a correct implementation agrees with ONE of the
    # rounding-equivalent evaluations of the reference formula (tests/test_gpu_parity_full.py).
    alts = []
    for eps in (1e-14, -1e-14):
        inv.strModel = m0 * (1 + eps)
        alts.append(O.compDataGradient(mesh, data, inv, HMCPrior(), False)[2])
    out["grad0_alt"] = np.stack(alts)
    m1 = m0 + 0.3 * np.random.default_rng(3).standard_normal(len(m0))
    inv.strModel = m1.copy()
    keep = {}
    pred, misfit, grad = O.compDataGradient(mesh, data, inv, HMCPrior(), False, keep)
    ny = mesh.gridSize[0]
    nair = len(mesh.airLayer)
    rows = slice(nair * (ny + 1), (nair + 2) * (ny + 1))
    out.update(m1=m1, pred1=pred, misfit1=misfit, grad1=grad, exTE_rx=keep["exTE"][rows, :], hxTM_rx=keep["hxTM"][rows, :])
    np.savez_compressed(os.path.join(HERE, f"example_{name}.npz"), **out)
    print(name, "nData", len(pred), "grid", mesh.gridSize, "misfit", out["misfit0"], misfit, "file kB",
          os.path.getsize(os.path.join(HERE, f"example_{name}.npz")) // 1024)


if __name__ == "__main__":
    names = sys.argv[1:] or ["tiny", "cfg2", "cfg1", "cfg3s", "cfg3", "cfg5s", "dprism3d", "coprod2", "rhophase"]
    for nm in names:
        if nm == "tiny":
            make("tiny", True)
        elif nm in ("cfg2", "cfg1"):
            make(nm, False)
        elif nm == "cfg3s":
            make_cfg3_subset()
        elif nm == "cfg3":
            make_cfg3_full()
        elif nm == "cfg5s":
            make_cfg5_subset()
        elif nm == "cfg5":
            make_cfg5_full()
        elif nm == "rhophase":
            make_rho_phase()
        elif nm == "rhophase_cfg1":
            make_rho_phase("cfg1")
        else:
            make_example(nm)
