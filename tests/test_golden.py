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
