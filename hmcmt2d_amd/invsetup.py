"""Host set-up of the inverse problem (runs once per run, never inside a leapfrog step).

Mirrors HMCMT/src/HMCStruct/HMCStruct.jl:99-125 (`setupInverseDataModel`),
HMCMT/src/HMCUtility/HMCUtility.jl:168-190 (`compDataWeightMat`), :217-258
(`setActiveElement`) and MTFwdSolver/MT2DOperators.jl:52-63 (`getCellGradient2D`).
"""
from __future__ import annotations

import numpy as np
import scipy.sparse as sp

from .structs import InvDataModel, TensorMesh2D


def setActiveElement(sigma, sigFix, fixIndex=None):
    """Cells whose conductivity equals (exact float compare, as the reference) one of `sigFix`
    are frozen; returns (activeIdx 0-based, bgModel)."""
    sigma = np.asarray(sigma, dtype=np.float64)
    frozen = np.zeros(len(sigma), dtype=np.int64)
    bgModel = np.zeros(len(sigma))
    for s in sigFix:
        hit = sigma == s
        if not hit.any():
            continue
        frozen += hit
        bgModel[hit] += s
    if fixIndex is not None and len(fixIndex):
        frozen[fixIndex] = 1
        bgModel[fixIndex] = sigma[fixIndex]
    return np.nonzero(frozen == 0)[0].astype(np.int64), bgModel


def compDataWeightMat(obsData, dataError):
    """Diagonal of W = diag(1/|err|)."""
    return 1.0 / np.abs(np.asarray(dataError, dtype=np.float64))


def cellGradient2D(ny, nz):
    """Unscaled first differences between neighbouring cells over ALL cells (air included):
    [kron(I_nz, ddx(ny-1)); kron(ddx(nz-1), I_ny)]."""
    def ddx(n):
        return sp.diags([-np.ones(n), np.ones(n)], [0, 1], shape=(n, n + 1), format="csr")
    G1 = sp.kron(sp.identity(nz), ddx(ny - 1))
    G2 = sp.kron(ddx(nz - 1), sp.identity(ny))
    return sp.vstack([G1, G2], format="csr")


def smoothnessMatrix(ny, nz, activeIdx):
    """Wm = (G A)^T (G A); the top earth row keeps the extra diagonal term of its removed air
    neighbour ('Prior')."""
    nCell = ny * nz
    nAC = len(activeIdx)
    A = sp.csr_matrix((np.ones(nAC), (activeIdx, np.arange(nAC))), shape=(nCell, nAC))
    GA = cellGradient2D(ny, nz) @ A
    return (GA.T @ GA).tocsr()


def setupInverseDataModel(mtMesh: TensorMesh2D, sigFix, sigLB, sigUB, obsData, dataErr,
                          fixIndex=None) -> InvDataModel:
    sigma = mtMesh.sigma
    activeIdx, bgModel = setActiveElement(sigma, sigFix, fixIndex)
    dataW = compDataWeightMat(obsData, dataErr)
    strModel = np.log(sigma[activeIdx])
    ny, nz = mtMesh.gridSize
    Wm = smoothnessMatrix(ny, nz, activeIdx)
    return InvDataModel(np.asarray(obsData, dtype=np.complex128), dataW, strModel, strModel.copy(),
                        activeIdx, bgModel, Wm)
