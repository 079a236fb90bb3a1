"""Generates the golden vectors tests/golden/{tiny,cfg2}.npz with the oracle (run in the build
container: `python tests/golden/make_golden.py`).  Inputs: BASELINE.json-style synthetic configs
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


if __name__ == "__main__":
    make("tiny", True)
    make("cfg2", False)
