"""Shared builders for the tests: synthetic problems (BASELINE.json configs) with oracle-made data."""
import os
import numpy as np

from hmcmt2d_amd import synthetic as S, invsetup as I
from hmcmt2d_amd.structs import HMCPrior

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def start_sigma(mesh):
    ny, nz = mesh.gridSize
    nair = len(mesh.airLayer)
    return np.concatenate([np.full(ny * nair, S.SIG_AIR), np.full(ny * (nz - nair), 0.01)])


def make_problem(name, obs=None, err=None):
    """(mesh, data, inv, m_eval): obs/err default to the golden file's (oracle forward of the true
    model + seeded noise) when present, else to a smooth synthetic."""
    mesh, data, sig_true = S.make_config(name)
    if obs is None:
        path = os.path.join(GOLDEN, f"{name}.npz")
        if os.path.exists(path):
            g = np.load(path)
            obs, err = g["obs"], g["err"]
        else:
            n = len(data.rxID)
            obs = np.full(n, 0.02 + 0.02j) * np.where(data.dtID == 1, 1.0, -1.0)
            err = np.full(n, 1e-3)
    mesh.sigma = start_sigma(mesh)
    inv = I.setupInverseDataModel(mesh, [S.SIG_AIR], 0.0, 0.0, obs, err)
    m = S.rough_state(len(inv.strModel))
    return mesh, data, inv, m


def oracle_eval(mesh, data, inv, m, dense_dbc=False, keep=None):
    from oracle import hmcmt_oracle as O
    if not mesh.setup:
        O.setupTensorMesh2D(mesh)
    inv.strModel = np.asarray(m, dtype=float).copy()
    return O.compDataGradient(mesh, data, inv, HMCPrior(), dense_dbc, keep)


def relmax(a, b):
    return float(np.abs(np.asarray(a) - np.asarray(b)).max() / np.abs(np.asarray(b)).max())
