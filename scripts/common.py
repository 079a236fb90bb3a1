"""Shared set-up of the developer scripts (no oracle here: the oracle is test infrastructure and lives behind
tests/ only): a synthetic problem with placeholder observations."""
import numpy as np

from hmcmt2d_amd import synthetic as S, invsetup as I


def problem(name, _unused=False):
    mesh, data, sig_true = S.make_config(name)
    ny, nz = mesh.gridSize
    nair = len(mesh.airLayer)
    obs = np.ones(len(data.rxID), dtype=complex) * (0.02 + 0.02j)
    err = np.full(len(obs), 1e-3)
    mesh.sigma = np.concatenate([np.full(ny * nair, 1e-8), np.full(ny * (nz - nair), 0.01)])
    inv = I.setupInverseDataModel(mesh, [1e-8], 0, 0, obs, err)
    return mesh, data, inv
