"""CPU oracle for the HMCMT2D hot path (TEST INFRASTRUCTURE ONLY).

This file is a numpy/scipy restatement of the reference's per-leapfrog-step
forward 2-D MT solve + adjoint gradient (`compDataGradient`,
HMCMT/src/HMCSampler/HMCSampler.jl:277-330) and of the sampler functions that call
it.  It is the checker the GPU path is compared against.  Only `tests/`,
`__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import it; the
product (`hmcmt2d_amd/`) never does.

PARITY -- the reference cannot be run here (Julia: no `julia` in this image; its
native solver binary, MUMPS, is stripped from the snapshot) and HMCMT ships no tests
or golden vectors for this path.  What pins this file:

* FORWARD HALF (m -> predData): PINNED against the reference's own example data.
  HMCMT/examples/dprism3d/dprism2dobs.dat is synthetic output of its authors'
  forward code with noise in the real parts only; for the model the file was
  generated from (two prisms, recovered by oracle/pin/recover_dprism.py) this
  restatement reproduces all 902 imaginary parts to the last of their seven printed
  digits -- TE and TM, 11 frequencies, 41 receivers, 10/100/1000 Ohm-m -- and the
  error column as 5 % of its |Z| (tests/test_oracle_kat.py, first test; the HIP
  path is held to the same file in tests/test_gpu_parity_full.py).
* ADJOINT / GRADIENT HALF (-> dataGrad): PARITY UNPINNED by reference outputs (none
  exist).  What holds it: (1) at the model the forward pin holds at, Richardson-
  extrapolated difference quotients of the pinned forward map, cell by cell on 256
  cells of the mesh core (tests/test_gradient_pin.py, tests/golden/make_fd_pin.py):
  with the Dirichlet values frozen they equal the P- and Q-terms of J^T v
  (compJacTMatVec.jl:235 / :306-307, :209 / :280) to 3e-7; with everything
  recomputed they differ from the full gradient by at most 1.6e-4, within the size
  of the reference's APPROXIMATE boundary-derivative terms,
  whose share of the gradient in those cells is <= 1.2e-3 -- that share is restated
  from compJacTMatVec.jl:237-242, 309-316 / MT1DSensitivity.jl without an independent
  check, and being approximations of the derivative no finite difference can confirm
  it; (2) identities: the adjoint against the explicit Jacobian built as
  MTSensitivity/compJacMat.jl:206-314 specifies it (1e-15), the Rho/phase chain
  rule, the MUMPS wrapper's residual bar (MUMPS/test/testDivGrad.jl:19) for the solver.

The sparse direct solve of the reference (UMFPACK `lu` via SuiteSparse_jll 7.2.1,
mt2DTE.jl:48 / MUMPS `sym=1`, mt2DTE.jl:51-53) is a third-party dependency that is
not in /root/reference; it is replaced here by scipy's SuperLU (`splu`), another
sparse direct LU with partial pivoting -- same mathematical operation, residuals
~1e-15.

All indices below are 0-based; the reference (Julia) is 1-based.  Layouts follow
SURVEY.md Appendix A: cells y-fastest `kz*ny+ky`, nodes y-fastest `iz*(ny+1)+iy`,
zLen includes air layers first (top -> down).
"""
from __future__ import annotations

from dataclasses import dataclass, field
import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla

MU0 = 4 * np.pi * 1e-7          # MT2DFwdSolver.jl:76
EPS0 = 8.85 * 1e-12             # mt1DField.jl:35


# ----------------------------------------------------------------------------
# 1-D difference / averaging operators (MT2DOperators.jl:139-201, HMCUtility.jl:86-90)
# ----------------------------------------------------------------------------
def spunit(n):
    return sp.identity(n, dtype=float, format="csr")


def sdiag(v):
    return sp.diags(np.asarray(v), 0, format="csr")


def ddx(n):
    """n x (n+1) [-1, +1] (MT2DOperators.jl:161-163)."""
    return sp.diags([-np.ones(n), np.ones(n)], [0, 1], shape=(n, n + 1), format="csr")


def av(n):
    """n x (n+1) [0.5, 0.5] (MT2DOperators.jl:172-174); `avnc` in HMCUtility.jl:86-90 is identical."""
    return sp.diags([0.5 * np.ones(n), 0.5 * np.ones(n)], [0, 1], shape=(n, n + 1), format="csr")


avnc = av


def avcn(n):
    """(n+1) x n cell->node averaging with 1.0 at both ends (MT2DOperators.jl:183-190)."""
    a = sp.diags([0.5 * np.ones(n), 0.5 * np.ones(n)], [-1, 0], shape=(n + 1, n), format="lil")
    a[0, 0] = 1.0
    a[n, n - 1] = 1.0
    return a.tocsr()


def meshGeoFace2D(d1, d2):
    """Cell areas, kron(diag(zLen), diag(yLen)) (MT2DOperators.jl:84-88)."""
    return sp.kron(sdiag(d2), sdiag(d1), format="csr")


def meshGeoEdgeInv2D(d1, d2):
    n1, n2 = len(d1), len(d2)
    L1 = sp.kron(spunit(n2 + 1), sdiag(1.0 / d1))
    L2 = sp.kron(sdiag(1.0 / d2), spunit(n1 + 1))
    return sp.block_diag([L1, L2], format="csr")


def getNodalGradient2D(d1, d2):
    """Nodal gradient: y-edges first then z-edges (MT2DOperators.jl:35-48)."""
    n1, n2 = len(d1), len(d2)
    G1 = sp.kron(spunit(n2 + 1), ddx(n1))
    G2 = sp.kron(ddx(n2), spunit(n1 + 1))
    Grad = sp.vstack([G1, G2], format="csr")
    return (meshGeoEdgeInv2D(d1, d2) @ Grad).tocsr()


def getCellGradient2D(d1, d2):
    """Unscaled cell first differences (MT2DOperators.jl:52-63)."""
    n1, n2 = len(d1), len(d2)
    G1 = sp.kron(spunit(n2), ddx(n1 - 1))
    G2 = sp.kron(ddx(n2 - 1), spunit(n1))
    return sp.vstack([G1, G2], format="csr")


def aveCell2Node2D(n):
    return sp.kron(avcn(n[1]), avcn(n[0]), format="csr")       # MT2DOperators.jl:118-122


def aveCell2Face2D(n):
    A1 = sp.kron(spunit(n[1]), avcn(n[0]))
    A2 = sp.kron(avcn(n[1]), spunit(n[0]))
    return sp.vstack([A2, A1], format="csr")                    # MT2DOperators.jl:126-130


# ----------------------------------------------------------------------------
# data structures (HMCFileIO.jl:26-60, HMCStruct.jl:18-91)
# ----------------------------------------------------------------------------
@dataclass
class TensorMesh2D:
    yLen: np.ndarray
    zLen: np.ndarray            # includes the air layers (top -> down)
    airLayer: np.ndarray        # as in the file: bottom -> up
    gridSize: tuple
    origin: np.ndarray
    sigma: np.ndarray
    Face: object = None
    Grad: object = None
    AveCN: object = None
    AveCF: object = None
    setup: bool = False


@dataclass
class MTData:
    rxLoc: np.ndarray           # (nRx, 2)
    freqs: np.ndarray
    dataType: str
    dataComp: list
    rxID: np.ndarray            # 1-based, as stored by the reference reader
    freqID: np.ndarray          # 1-based
    dtID: np.ndarray            # 1-based
    dataID: np.ndarray          # bool mask over (dt, rx, freq), dt fastest
    compTE: bool
    compTM: bool


@dataclass
class HMCPrior:                 # HMCStruct.jl:18-36, defaults :129-140
    burninsamples: int = 100
    totalsamples: int = 500
    sigBounds: list = field(default_factory=lambda: [0.01, 10.0])
    sigmastd: float = 0.05
    dt: float = 0.01
    timestep: list = field(default_factory=lambda: [10, 15])
    linearSolver: str = ""
    massType: str = "diagonal"
    regParam: float = 1.0
    nfevals: int = 0


@dataclass
class InvDataModel:             # HMCStruct.jl:75-91
    obsData: np.ndarray
    dataW: np.ndarray           # diagonal of the weighting matrix
    strModel: np.ndarray
    refModel: np.ndarray
    activeIdx: np.ndarray       # column pattern of `activeCell` (0-based cell ids)
    bgModel: np.ndarray
    Wm: object


def setupTensorMesh2D(mesh: TensorMesh2D):
    """MT2DOperators.jl:16-27."""
    mesh.Face = meshGeoFace2D(mesh.yLen, mesh.zLen)
    mesh.Grad = getNodalGradient2D(mesh.yLen, mesh.zLen)
    mesh.AveCN = aveCell2Node2D(mesh.gridSize)
    mesh.AveCF = aveCell2Face2D(mesh.gridSize)
    mesh.setup = True
    return mesh


def activeCellMatrix(activeIdx, nCell):
    nAC = len(activeIdx)
    return sp.csr_matrix((np.ones(nAC), (activeIdx, np.arange(nAC))), shape=(nCell, nAC))


def getBoundaryIndex(ny, nz):
    """Interior / boundary node split (MT2DFwdSolver.jl:227-248), 0-based."""
    idx2D = np.arange((ny + 1) * (nz + 1)).reshape(nz + 1, ny + 1)
    ii = idx2D[1:-1, 1:-1].reshape(-1)
    it = idx2D[0, :]
    il = idx2D[1:, 0]
    ir = idx2D[1:, -1]
    ib = idx2D[-1, 1:-1]
    io = np.concatenate([it, il, ir, ib])
    return ii, io


# ----------------------------------------------------------------------------
# 1-D layered-earth boundary fields (mt1DField.jl:23-98)
# ----------------------------------------------------------------------------
def mt1DAnalyticField(freq, sigma, zNode, compH=False):
    sigma = np.asarray(sigma, dtype=float)
    if len(sigma) != len(zNode) - 1:
        raise ValueError("layer's conductivity is not the same size with its depth.")
    eTop = 1.0 + 0j
    omega = 2 * np.pi * freq
    omu0 = omega * MU0
    sigma = np.concatenate([sigma, sigma[-1:]])      # half-space below (:40)
    nLayer = len(zNode)
    zLen = np.diff(zNode)

    with np.errstate(over="ignore", invalid="ignore"):
        k = np.sqrt(MU0 * EPS0 * omega ** 2 - MU0 * sigma[-1] * omega * 1j)
        ztmp = omega * MU0 / k
        for j in range(nLayer - 2, -1, -1):           # :51-55
            k = np.sqrt(MU0 * EPS0 * omega ** 2 - MU0 * sigma[j] * omega * 1j)
            zp = omega * MU0 / k
            th = np.tanh(k * zLen[j] * 1j)
            ztmp = zp * (ztmp + zp * th) / (zp + ztmp * th)
        z0 = ztmp

        eLayer = np.zeros((2, nLayer), dtype=complex)
        eLayer[0, 0] = 0.5 * eTop * (1 - omega * MU0 / (z0 * k))   # k = top layer's (:62)
        eLayer[1, 0] = 0.5 * eTop * (1 + omega * MU0 / (z0 * k))
        ka = np.sqrt(MU0 * EPS0 * omega ** 2 - MU0 * sigma * omega * 1j)

        for i in range(nLayer - 1):                   # :69-83
            kr = ka[i] / ka[i + 1]
            pInv = 0.5 * np.array([[1 + kr, 1 - kr], [1 - kr, 1 + kr]])
            eUD = np.array([[np.exp(ka[i] * zLen[i] * 1j), 0], [0, np.exp(-ka[i] * zLen[i] * 1j)]])
            eLayer[:, i + 1] = (pInv @ eUD) @ eLayer[:, i]
            e2 = abs(eLayer[0, i + 1] + eLayer[1, i + 1])
            e1 = abs(eLayer[0, i] + eLayer[1, i])
            if e2 - e1 > 0 or np.isnan(e2):
                eLayer[:, i + 1:] = 0.0
                break

    eField = eLayer.sum(axis=0)
    if compH:
        hField = (eLayer[0, :] * (-ka) + eLayer[1, :] * ka) / omu0   # :87-91
        return eField, hField
    return eField


def _getBoundaryMT2D(freq, yLen, zLen, sigma, mode):
    """getBoundaryMT2DTE (mt2DTE.jl:100-134) / getBoundaryMT2DTM (mt2DTM.jl:100-134)."""
    ny, nz = len(yLen), len(zLen)
    zNode = np.concatenate([[0.0], np.cumsum(zLen)])
    sigma2D = sigma.reshape(nz, ny)                   # [kz, ky]
    nb = 2 * (ny + nz)
    bc = np.zeros(nb, dtype=complex)
    bc[0:ny + 1] = 1.0

    def field(s1d):
        if mode == "TE":
            return mt1DAnalyticField(freq, s1d, zNode)
        return mt1DAnalyticField(freq, s1d, zNode, True)[1]

    fb = field(sigma2D[:, 0]);  fb = fb / fb[0]
    bc[ny + 1:ny + nz + 1] = fb[1:]
    fb = field(sigma2D[:, -1]); fb = fb / fb[0]
    bc[ny + nz + 1:ny + 2 * nz + 1] = fb[1:]
    for i in range(1, ny):                            # Julia i = 2:ny (node index)
        s1d = (sigma2D[:, i - 1] * yLen[i - 1] + sigma2D[:, i] * yLen[i]) / (yLen[i - 1] + yLen[i])
        fb = field(s1d)
        bc[ny + 2 * nz + i] = fb[-1] / fb[0]
    return bc


def getBoundaryMT2DTE(freq, yLen, zLen, sigma):
    return _getBoundaryMT2D(freq, yLen, zLen, sigma, "TE")


def getBoundaryMT2DTM(freq, yLen, zLen, sigma):
    return _getBoundaryMT2D(freq, yLen, zLen, sigma, "TM")


# ----------------------------------------------------------------------------
# receiver-layer functionals (mt2DTE.jl:153-259, mt2DTM.jl:152-242)
# ----------------------------------------------------------------------------
def _find_zid(zNode, zRx):
    """0-based index of the first node with |zNode - zRx| < 0.1 (mt2DTE.jl:66-67)."""
    hit = np.nonzero(np.abs(zNode - zRx) < 0.1)[0]
    if len(hit) == 0:
        raise ValueError("receiver depth does not coincide with a grid node")
    return int(hit[0])


def compFieldsAtRxTE(omega, rxLoc, yNode, zLen1, sigma1, Er01):
    yLen = np.diff(yNode)
    ny = len(yLen)
    mu = MU0 * np.ones(ny)
    nRx = rxLoc.shape[0]
    Ex0 = Er01[:, 0]
    Bz0 = (ddx(ny) @ Er01[:, 0]) / yLen / (1j * omega)
    Bz1 = (ddx(ny) @ Er01[:, 1]) / yLen / (1j * omega)
    HzQ = (0.75 * Bz0 + 0.25 * Bz1) / mu
    HyH = -(Er01[1:-1, 1] - Er01[1:-1, 0]) / zLen1 / (1j * omega * MU0)
    ExQ = 0.75 * Er01[1:-1, 0] + 0.25 * Er01[1:-1, 1]
    sigma1v = (av(ny - 1) @ (sigma1 * yLen)) / (av(ny - 1) @ yLen)
    dHzQ = (ddx(ny - 1) @ HzQ) / (av(ny - 1) @ yLen)
    Hy0 = np.zeros(ny + 1, dtype=complex)
    Hy0[1:-1] = HyH - (dHzQ - sigma1v * ExQ) * (0.5 * zLen1)
    Hy0[0] = Hy0[1]
    Hy0[-1] = Hy0[-2]
    Exr = np.zeros(nRx, dtype=complex)
    Hyr = np.zeros(nRx, dtype=complex)
    for ir in range(nRx):
        rxY = rxLoc[ir, 0]
        hit = np.nonzero(yNode > rxY)[0]
        if len(hit) == 0 or hit[0] == 0:
            raise ValueError("The receiver location seems to be out of range!")
        idn = hit[0]
        dy1 = rxY - yNode[idn - 1]
        dy2 = yNode[idn] - rxY
        Exr[ir] = Ex0[idn - 1] * dy2 + Ex0[idn] * dy1        # un-normalised weights (:203-206)
        Hyr[ir] = Hy0[idn - 1] * dy2 + Hy0[idn] * dy1
    return Exr, Hyr


def compFieldsAtRxTM(omega, rxLoc, yNode, zLen1, sigma1, Hr01):
    yLen = np.diff(yNode)
    ny = len(yLen)
    nRx = rxLoc.shape[0]
    Hx0 = Hr01[:, 0]
    Jz0 = -(ddx(ny) @ Hr01[:, 0]) / yLen
    Jz1 = -(ddx(ny) @ Hr01[:, 1]) / yLen
    EzQ = (0.75 * Jz0 + 0.25 * Jz1) / sigma1
    JyH = (Hr01[1:-1, 1] - Hr01[1:-1, 0]) / zLen1
    rho1v = (av(ny - 1) @ ((1.0 / sigma1) * yLen)) / (av(ny - 1) @ yLen)
    EyH = JyH * rho1v
    HxQ = 0.75 * Hr01[1:-1, 0] + 0.25 * Hr01[1:-1, 1]
    dEzQ = (ddx(ny - 1) @ EzQ) / (av(ny - 1) @ yLen)
    Ey0 = np.zeros(ny + 1, dtype=complex)
    Ey0[1:-1] = EyH - (dEzQ + 1j * omega * MU0 * HxQ) * (0.5 * zLen1)
    Ey0[0] = Ey0[1]
    Ey0[-1] = Ey0[-2]
    Eyr = np.zeros(nRx, dtype=complex)
    Hxr = np.zeros(nRx, dtype=complex)
    for ir in range(nRx):
        rxY = rxLoc[ir, 0]
        hit = np.nonzero(yNode > rxY)[0]
        if len(hit) == 0 or hit[0] == 0:
            raise ValueError("The receiver location seems to be out of range!")
        idn = hit[0]
        dy1 = rxY - yNode[idn - 1]
        dy2 = yNode[idn] - rxY
        Eyr[ir] = Ey0[idn - 1] * dy2 + Ey0[idn] * dy1
        Hxr[ir] = Hx0[idn - 1] * dy2 + Hx0[idn] * dy1
    return Eyr, Hxr


def _compMTResp(omega, Enum, Hden, dataType):
    """compMTRespTE (mt2DTE.jl:240-259) / compMTRespTM (mt2DTM.jl:224-242)."""
    Z = Enum / Hden
    if "Impedance" in dataType:
        return np.stack([Z.real, Z.imag], axis=1)
    rho = np.abs(Z) ** 2 / (omega * MU0)
    phs = np.arctan2(Z.imag, Z.real) * 180 / np.pi
    return np.stack([rho, phs], axis=1)


# ----------------------------------------------------------------------------
# per-frequency solves (mt2DTE.jl:19-83, mt2DTM.jl:18-83)
# ----------------------------------------------------------------------------
@dataclass
class CoeffMat:
    rAii: object
    iAii: object
    rAio: object
    iAio: object


def _compMT2D(freq, mesh, coeMat, rxLoc, dataType, mode, keep=None, bc_fixed=None):
    yLen, zLen, origin, sigma = mesh.yLen, mesh.zLen, mesh.origin, mesh.sigma
    yNode = np.concatenate([[0.0], np.cumsum(yLen)]) - origin[0]
    zNode = np.concatenate([[0.0], np.cumsum(zLen)]) - origin[1]
    ny, nz = len(yLen), len(zLen)
    omega = 2 * np.pi * freq
    Aii = (coeMat.rAii + 1j * omega * coeMat.iAii).tocsc()
    Aio = (coeMat.rAio + 1j * omega * coeMat.iAio).tocsr()
    bc = _getBoundaryMT2D(freq, yLen, zLen, sigma, mode)
    if bc_fixed is not None:              # (test hook, not in the reference: Dirichlet values held at another model's,
        bc = bc_fixed[(mode, freq)]       #  for the frozen-boundary finite differences of tests/test_gradient_pin.py)
    rhs = -(Aio @ bc)
    Ainv = spla.splu(Aii)                 # `lu(Aii)` (mt2DTE.jl:48) -> SuperLU here
    Fii = Ainv.solve(rhs)

    F2d = np.zeros((nz + 1, ny + 1), dtype=complex)
    F2d[0, :] = bc[0:ny + 1]
    F2d[1:, 0] = bc[ny + 1:ny + nz + 1]
    F2d[1:, -1] = bc[ny + nz + 1:ny + 2 * nz + 1]
    F2d[-1, 1:-1] = bc[ny + 2 * nz + 1:]
    F2d[1:-1, 1:-1] = Fii.reshape(nz - 1, ny - 1)

    zid = _find_zid(zNode, rxLoc[0, 1])
    Fr01 = F2d[zid:zid + 2, :].T.copy()             # (ny+1, 2)
    sigma1 = sigma[zid * ny:(zid + 1) * ny]
    zLen1 = zLen[zid]
    fld = F2d.reshape(-1).copy()
    if mode == "TE":
        Er, Hr = compFieldsAtRxTE(omega, rxLoc, yNode, zLen1, sigma1, Fr01)
    else:
        Er, Hr = compFieldsAtRxTM(omega, rxLoc, yNode, zLen1, sigma1, Fr01)
    resp = _compMTResp(omega, Er, Hr, dataType)
    if keep is not None:
        keep.setdefault("bc", {})[(mode, freq)] = bc
        keep.setdefault("rhs", {})[(mode, freq)] = rhs
        keep.setdefault("Aii", {})[(mode, freq)] = Aii.tocsr()
    return resp, fld, Ainv


@dataclass
class MT2DFwdData:
    exTE: np.ndarray
    hxTM: np.ndarray
    AinvTE: list
    AinvTM: list
    linearSolver: str


def MT2DFwdSolver(mesh: TensorMesh2D, mtData: MTData, linearSolver="", keep=None, bc_fixed=None):
    """MT2DFwdSolver.jl:74-216."""
    yLen, zLen, sigma = mesh.yLen, mesh.zLen, mesh.sigma
    freqs, rxLoc, dataType = mtData.freqs, mtData.rxLoc, mtData.dataType
    nFreq, nRx = len(freqs), rxLoc.shape[0]
    ny, nz = len(yLen), len(zLen)
    nNode = (ny + 1) * (nz + 1)
    mu = MU0 * np.ones(ny * nz)
    F, Grad, AveCN, AveCF = mesh.Face, mesh.Grad, mesh.AveCN, mesh.AveCF
    ii, io = getBoundaryIndex(ny, nz)
    exte = np.zeros((nNode, nFreq), dtype=complex)
    hxtm = np.zeros((nNode, nFreq), dtype=complex)
    AinvTE = [None] * nFreq
    AinvTM = [None] * nFreq
    respTE = respTM = None

    if mtData.compTE:
        MsigCN = sdiag(AveCN @ (F @ sigma))
        MmuF = sdiag(AveCF @ (F @ (1.0 / mu)))
        dGrad = (Grad.T @ MmuF @ Grad).tocsr()
        coe = CoeffMat(dGrad[ii][:, ii], MsigCN[ii][:, ii], dGrad[ii][:, io], MsigCN[ii][:, io])
        respTE = np.zeros((nFreq * nRx, 2))
        for j in range(nFreq):
            r, exte[:, j], AinvTE[j] = _compMT2D(freqs[j], mesh, coe, rxLoc, dataType, "TE", keep, bc_fixed)
            respTE[j * nRx:(j + 1) * nRx, :] = r
    if mtData.compTM:
        MmuCN = sdiag(AveCN @ (F @ mu))
        MsigF = sdiag(AveCF @ (F @ (1.0 / sigma)))
        dGrad = (Grad.T @ MsigF @ Grad).tocsr()
        coe = CoeffMat(dGrad[ii][:, ii], MmuCN[ii][:, ii], dGrad[ii][:, io], MmuCN[ii][:, io])
        respTM = np.zeros((nFreq * nRx, 2))
        for j in range(nFreq):
            r, hxtm[:, j], AinvTM[j] = _compMT2D(freqs[j], mesh, coe, rxLoc, dataType, "TM", keep, bc_fixed)
            respTM[j * nRx:(j + 1) * nRx, :] = r

    if "Impedance" in dataType:                       # :175-189
        if mtData.compTE and not mtData.compTM:
            predData = respTE[:, 0] + 1j * respTE[:, 1]
        elif mtData.compTM and not mtData.compTE:
            predData = respTM[:, 0] + 1j * respTM[:, 1]
        else:
            pTE = respTE[:, 0] + 1j * respTE[:, 1]
            pTM = respTM[:, 0] + 1j * respTM[:, 1]
            predData = np.stack([pTE, pTM], axis=1).reshape(-1)
    elif "Rho_Pha" in dataType:                       # :191-205
        if mtData.compTE and not mtData.compTM:
            predData = respTE.reshape(-1)
        elif mtData.compTM and not mtData.compTE:
            predData = respTM.reshape(-1)
        else:
            predData = np.concatenate([respTE, respTM], axis=1).reshape(-1)
    else:
        raise ValueError(dataType)
    predData = predData[mtData.dataID]
    return predData, MT2DFwdData(exte, hxtm, AinvTE, AinvTM, linearSolver)


# ----------------------------------------------------------------------------
# receiver sensitivities (sensUtils.jl:17-83,133-161; dataFuncSens.jl:21-176,197-344)
# ----------------------------------------------------------------------------
def linearInterp(point, x):
    """sensUtils.jl:133-161; returns 0-based (indL, indR, wL, wR)."""
    ind = int(np.argmin(np.abs(point - x)))
    if point - x[ind] > 0:
        indL, indR = ind, ind + 1
    else:
        indL, indR = ind - 1, ind
    n = len(x)
    indL = max(min(indL, n - 1), 0)
    indR = max(min(indR, n - 1), 0)
    if indL == indR:
        return indL, indR, 0.5, 0.5
    xLen = x[indR] - x[indL]
    wL = 1 - (point - x[indL]) / xLen
    wR = 1 - (x[indR] - point) / xLen
    return indL, indR, wL, wR


def linearInterpMat(points, x):
    """sensUtils.jl:63-83: nNode x npts sparse matrix (duplicate indices are summed as sparsevec does)."""
    rows, cols, vals = [], [], []
    for i, p in enumerate(points):
        indL, indR, wL, wR = linearInterp(p, x)
        rows += [indL, indR]; cols += [i, i]; vals += [wL, wR]
    return sp.csr_matrix((vals, (rows, cols)), shape=(len(x), len(points)))


@dataclass
class PreRxSens:                                       # MTSensitivity.jl:20-47
    zid: int
    dFn0: object
    dFn1: object
    sigma1: np.ndarray
    dsigma1: object
    yLen: np.ndarray
    zLen1: float
    linRxMap: object
    linRxMap2: object


def preSetRxFieldSens(rxLoc, yNode, zNode, sigma):
    ny, nz = len(yNode) - 1, len(zNode) - 1
    nNode, nCell = (ny + 1) * (nz + 1), ny * nz
    zLen = np.diff(zNode)
    zid = _find_zid(zNode, rxLoc[0, 1])
    Inode = spunit(nNode)
    dFn0 = Inode[zid * (ny + 1):(zid + 1) * (ny + 1), :]
    dFn1 = Inode[(zid + 1) * (ny + 1):(zid + 2) * (ny + 1), :]
    Icell = spunit(nCell)
    sigma1 = sigma[zid * ny:(zid + 1) * ny]
    dsigma1 = Icell[zid * ny:(zid + 1) * ny, :]
    yLen = np.diff(yNode)
    linRxMap = linearInterpMat(rxLoc[:, 0], yNode)
    yCen = (yNode[:-1] + yNode[1:]) / 2.0
    linRxMap2 = linearInterpMat(rxLoc[:, 0], yCen)
    return PreRxSens(zid, dFn0, dFn1, sigma1, dsigma1, yLen, zLen[zid], linRxMap, linRxMap2)


def _edge_dup(M):
    """rows [2:end-1] given -> (ny+1) rows with first/last duplicated (dataFuncSens.jl:80-88)."""
    M = sp.csr_matrix(M)
    return sp.vstack([M[0], M, M[-1]], format="csr")



def _rho_phase_rows(omega, Z, dZ, dZ_dsig):
    """The "Rho_Phs" branch of dataFuncSens.jl:130-159 (TE) / :300-330 (TM): rows of L and Q for apparent
    resistivity (first nRx rows) and phase in degrees (next nRx rows).  The reference tests the data type against
    "Rho_Phs" here while its reader and forward solver say "Rho_Pha" (readMT2DData.jl:87, MT2DFwdSolver.jl:191), so
    the branch is unreachable there; this restatement accepts either spelling -- a deliberate fix,
    not a reproduction.  `log10Rho` components (:154-160) are not restated: the reference's forward returns the
    linear apparent resistivity for them (mt2DTE.jl:253)."""
    omu = omega * MU0
    dZ = sp.csr_matrix(dZ)
    dZs = sp.csr_matrix(dZ_dsig)
    dAppRho = (2.0 / omu) * (sdiag(np.conj(Z)) @ dZ)
    dPhase = (sdiag(1.0 / np.abs(Z) ** 2) @ ((-1j) * (sdiag(np.conj(Z)) @ dZ))) * (180.0 / np.pi)
    dAppRho_dsig = (2.0 / omu) * (sdiag(Z.real) @ dZs.real + sdiag(Z.imag) @ dZs.imag)
    dPhase_dsig = (sdiag(1.0 / np.abs(Z) ** 2) @ (sdiag(Z.real) @ dZs.imag - sdiag(Z.imag) @ dZs.real)) * (180.0 / np.pi)
    return sp.vstack([dAppRho, dPhase]).tocsr(), sp.vstack([dAppRho_dsig, dPhase_dsig]).tocsr()


def _is_rho_phase(dataType):
    return "Rho_Pha" in dataType or "Rho_Phs" in dataType


def getDataFuncSensTE(omega, rx: PreRxSens, Ex01, dataType):
    """Impedance branch of dataFuncSens.jl:21-176."""
    dEx0, dEx1, sigma1, dsigma1 = rx.dFn0, rx.dFn1, rx.sigma1, rx.dsigma1
    yLen, zLen1, linRxMap = rx.yLen, rx.zLen1, rx.linRxMap
    ny = len(yLen)
    mu = MU0 * np.ones(ny)
    Bz0 = (ddx(ny) @ Ex01[:, 0]) / yLen / (1j * omega)
    Bz1 = (ddx(ny) @ Ex01[:, 1]) / yLen / (1j * omega)
    dtmp = sdiag(1.0 / yLen / (1j * omega)) @ ddx(ny)
    dBz0, dBz1 = dtmp @ dEx0, dtmp @ dEx1
    HzQ = (0.75 * Bz0 + 0.25 * Bz1) / mu
    dHzQ = sdiag(1.0 / mu) @ (0.75 * dBz0 + 0.25 * dBz1)
    HyH = -(Ex01[1:-1, 1] - Ex01[1:-1, 0]) / zLen1 / (1j * omega * MU0)
    dHyH = -(dEx1[1:-1, :] - dEx0[1:-1, :]) / zLen1 / (1j * omega * MU0)
    ExQ = 0.75 * Ex01[1:-1, 0] + 0.25 * Ex01[1:-1, 1]
    dExQ = 0.75 * dEx0[1:-1, :] + 0.25 * dEx1[1:-1, :]
    avl = avnc(ny - 1) @ yLen
    sigma1v = (avnc(ny - 1) @ (sigma1 * yLen)) / avl
    dsigma1v = sdiag(1.0 / avl) @ avnc(ny - 1) @ sdiag(yLen) @ dsigma1
    dHzQ_dy = (ddx(ny - 1) @ HzQ) / avl
    ddHzQ = sdiag(1.0 / avl) @ ddx(ny - 1) @ dHzQ
    Hy0 = np.zeros(ny + 1, dtype=complex)
    Hy0[1:-1] = HyH - (dHzQ_dy - sigma1v * ExQ) * (0.5 * zLen1)
    Hy0[0], Hy0[-1] = Hy0[1], Hy0[-2]
    dHy0 = _edge_dup(dHyH - (ddHzQ - sdiag(sigma1v) @ dExQ) * (0.5 * zLen1))
    dHy0_dsig = _edge_dup(0.5 * zLen1 * (sdiag(ExQ) @ dsigma1v))
    Exr = linRxMap.T @ Ex01[:, 0]
    Hyr = linRxMap.T @ Hy0
    dExr = linRxMap.T @ dEx0
    dHyr = linRxMap.T @ dHy0
    dHyr_dsig = linRxMap.T @ dHy0_dsig
    dZ = sdiag(1.0 / Hyr) @ dExr - sdiag(Exr / Hyr ** 2) @ dHyr
    dZ_dsig = -sdiag(Exr / Hyr ** 2) @ dHyr_dsig
    if _is_rho_phase(dataType):
        return _rho_phase_rows(omega, Exr / Hyr, dZ, dZ_dsig)
    if "Impedance" not in dataType:
        raise NotImplementedError(dataType)
    return sp.csr_matrix(dZ), sp.csr_matrix(dZ_dsig)


def getDataFuncSensTM(omega, rx: PreRxSens, Hx01, dataType):
    """Impedance branch of dataFuncSens.jl:197-344."""
    dHx0, dHx1, sigma1, dsigma1 = rx.dFn0, rx.dFn1, rx.sigma1, rx.dsigma1
    yLen, zLen1, linRxMap = rx.yLen, rx.zLen1, rx.linRxMap
    ny = len(yLen)
    Jz0 = -(ddx(ny) @ Hx01[:, 0]) / yLen
    Jz1 = -(ddx(ny) @ Hx01[:, 1]) / yLen
    dtmp = -sdiag(1.0 / yLen) @ ddx(ny)
    dJz0, dJz1 = dtmp @ dHx0, dtmp @ dHx1
    EzQ = (0.75 * Jz0 + 0.25 * Jz1) / sigma1
    dEzQ = sdiag(1.0 / sigma1) @ (0.75 * dJz0 + 0.25 * dJz1)
    dEzQ_dsig = sdiag(0.75 * Jz0 + 0.25 * Jz1) @ sdiag(-1.0 / sigma1 ** 2) @ dsigma1
    JyH = (Hx01[1:-1, 1] - Hx01[1:-1, 0]) / zLen1
    avl = avnc(ny - 1) @ yLen
    rho1v = (avnc(ny - 1) @ ((1.0 / sigma1) * yLen)) / avl
    dJyH = (dHx1[1:-1, :] - dHx0[1:-1, :]) / zLen1
    EyH = JyH * rho1v
    dEyH = sdiag(rho1v) @ dJyH
    drho1v = sdiag(1.0 / avl) @ avnc(ny - 1) @ sdiag(yLen) @ sdiag(-1.0 / sigma1 ** 2) @ dsigma1
    dEyH_dsig = sdiag(JyH) @ drho1v
    HxQ = 0.75 * Hx01[1:-1, 0] + 0.25 * Hx01[1:-1, 1]
    dHxQ = 0.75 * dHx0[1:-1, :] + 0.25 * dHx1[1:-1, :]
    dEzQ_dy = (ddx(ny - 1) @ EzQ) / avl
    dtmp = sdiag(1.0 / avl) @ ddx(ny - 1)
    ddEzQ = dtmp @ dEzQ
    ddEzQ_dsig = dtmp @ dEzQ_dsig
    Ey0 = np.zeros(ny + 1, dtype=complex)
    Ey0[1:-1] = EyH - (dEzQ_dy + 1j * omega * MU0 * HxQ) * (0.5 * zLen1)
    Ey0[0], Ey0[-1] = Ey0[1], Ey0[-2]
    dEy0 = _edge_dup(dEyH - (ddEzQ + 1j * omega * MU0 * dHxQ) * (0.5 * zLen1))
    dEy0_dsig = _edge_dup(dEyH_dsig - ddEzQ_dsig * (0.5 * zLen1))
    Hxr = linRxMap.T @ Hx01[:, 0]
    Eyr = linRxMap.T @ Ey0
    dHxr = linRxMap.T @ dHx0
    dEyr = linRxMap.T @ dEy0
    dEyr_dsig = linRxMap.T @ dEy0_dsig
    dZ = sdiag(1.0 / Hxr) @ dEyr - sdiag(Eyr / Hxr ** 2) @ dHxr
    dZ_dsig = sdiag(1.0 / Hxr) @ dEyr_dsig
    if _is_rho_phase(dataType):
        return _rho_phase_rows(omega, Eyr / Hxr, dZ, dZ_dsig)
    if "Impedance" not in dataType:
        raise NotImplementedError(dataType)
    return sp.csr_matrix(dZ), sp.csr_matrix(dZ_dsig)


# ----------------------------------------------------------------------------
# 1-D boundary-field sensitivities (MT1DSensitivity.jl:25-357)
# ----------------------------------------------------------------------------
def compImpJacMatrix(freq, sig1d, thick1d):
    """MT1DSensitivity.jl:188-243."""
    nLayer = len(sig1d)
    omega = 2 * np.pi * freq
    iom = 1j * omega * MU0
    Z = 0j
    dZ_ZP1 = np.zeros(nLayer, dtype=complex)
    dZ_sigma = np.zeros(nLayer, dtype=complex)
    zimpDeri = np.zeros(nLayer, dtype=complex)
    with np.errstate(over="ignore", invalid="ignore"):
        for j in range(nLayer - 1, -1, -1):
            k = np.sqrt(-iom * sig1d[j])
            Zt = omega * MU0 / k
            dZt = 1j * (omega * MU0) ** 2 / (2 * k ** 3)
            if j == nLayer - 1:
                Z = Zt
                dZ_sigma[j] = dZt
                continue
            RI = (Zt - Z) / (Zt + Z)
            theEXP = np.exp(-2j * k * thick1d[j])
            L = RI * theEXP
            Ztmp = Zt * (1 - L) / (1 + L)
            dL = 2 * Z / (Zt + Z) ** 2 * theEXP * dZt + (-2j * thick1d[j] * L) * (-iom / 2 / k)
            dZ_ZP1[j] = 4 * Zt * Zt * theEXP / ((1 + L) * (Zt + Z)) ** 2
            dZ_sigma[j] = dZt * (1 - L) / (1 + L) + Zt * (-2) / (1 + L) ** 2 * dL
            Z = Ztmp
        for iLayer in range(nLayer - 1, 0, -1):        # Julia nLayer:-1:2
            dZ_ZPN = 1.0 + 0j
            for j in range(iLayer):
                dZ_ZPN = dZ_ZPN * dZ_ZP1[j]
            zimpDeri[iLayer] = dZ_ZPN * dZ_sigma[iLayer]
        zimpDeri[0] = dZ_sigma[0]
    return Z, zimpDeri


def mt1DFieldSensMatrix(freq, sig1d, zNode, source="E", fTop=1.0):
    """MT1DSensitivity.jl:25-176.  Returns (field, dField) with dField of shape (nz+1, nz)."""
    sig1d = np.asarray(sig1d, dtype=float)
    if len(sig1d) != len(zNode) - 1:
        raise ValueError("layer's conductivity is not the same size with its depth.")
    omega = 2 * np.pi * freq
    omu = omega * MU0
    sigma = np.concatenate([sig1d, sig1d[-1:]])
    nLayer = len(sigma)
    zLen = np.diff(zNode)
    z1, dz1 = compImpJacMatrix(freq, sigma, zLen)

    eLayer = np.zeros((2, nLayer), dtype=complex)
    dEu = np.zeros((nLayer, nLayer), dtype=complex)
    dEd = np.zeros((nLayer, nLayer), dtype=complex)
    dHu = np.zeros((nLayer, nLayer), dtype=complex)
    dHd = np.zeros((nLayer, nLayer), dtype=complex)
    hLayer = np.zeros((2, nLayer), dtype=complex)

    with np.errstate(over="ignore", invalid="ignore"):
        ka = np.sqrt(-1j * omu * sigma)                # no displacement term (:59)
        dkaVec = (-1j * omu / 2) / ka
        dka = np.diag(dkaVec)
        k1 = ka[0]
        if source == "E":
            eLayer[0, 0] = 0.5 * fTop * (1 - omu / (z1 * k1))
            eLayer[1, 0] = 0.5 * fTop * (1 + omu / (z1 * k1))
            hLayer[0, 0] = -k1 / omu * eLayer[0, 0]
            hLayer[1, 0] = k1 / omu * eLayer[1, 0]
            dEu[0, :] = 0.5 * fTop * omu / (z1 * k1) * (1 / z1 * dz1 + 1 / k1 * dka[0, :])
            dEd[0, :] = -dEu[0, :]
            dHu[0, :] = -eLayer[0, 0] / omu * dka[0, :] - ka[0] / omu * dEu[0, :]
            dHd[0, :] = eLayer[1, 0] / omu * dka[0, :] + ka[0] / omu * dEd[0, :]
        elif source == "H":
            hLayer[0, 0] = 0.5 * fTop * (1 - z1 * k1 / omu)
            hLayer[1, 0] = 0.5 * fTop * (1 + z1 * k1 / omu)
            eLayer[0, 0] = -omu / k1 * hLayer[0, 0]
            eLayer[1, 0] = omu / k1 * hLayer[1, 0]
            dHu[0, :] = -0.5 * fTop / omu * (z1 * dka[0, :] + k1 * dz1)
            dHd[0, :] = -dHu[0, :]
            dEu[0, :] = 0.5 * fTop * (dz1 + (omu / k1 ** 2) * dka[0, :])
            dEd[0, :] = 0.5 * fTop * (dz1 - (omu / k1 ** 2) * dka[0, :])
        else:
            raise ValueError(source)

        expt = np.exp(1j * ka[:-1] * zLen)
        expr = 1.0 / expt
        dexpt_v = 1j * zLen * expt * dkaVec[:-1]
        dexpr_v = -1j * zLen * expr * dkaVec[:-1]
        dexpt = np.zeros((nLayer - 1, nLayer), dtype=complex)
        dexpr = np.zeros((nLayer - 1, nLayer), dtype=complex)
        idx = np.arange(nLayer - 1)
        dexpt[idx, idx] = dexpt_v
        dexpr[idx, idx] = dexpr_v
        kr = ka[:-1] / ka[1:]
        dkr = np.zeros((nLayer - 1, nLayer), dtype=complex)
        for j in range(nLayer - 1):
            dkr[j, :] = dka[j, :] / ka[j + 1] - ka[j] / (ka[j + 1] ** 2) * dka[j + 1, :]
        mix11 = (1 + kr) * expt
        mix12 = (1 - kr) * expr
        mix21 = (1 - kr) * expt
        mix22 = (1 + kr) * expr
        dmix11 = (1 + kr)[:, None] * dexpt + expt[:, None] * dkr
        dmix12 = (1 - kr)[:, None] * dexpr - expr[:, None] * dkr
        dmix21 = (1 - kr)[:, None] * dexpt - expt[:, None] * dkr
        dmix22 = (1 + kr)[:, None] * dexpr + expr[:, None] * dkr

        for j in range(nLayer - 1):                    # :126-157
            pInv = 0.5 * np.array([[1 + kr[j], 1 - kr[j]], [1 - kr[j], 1 + kr[j]]])
            eUD = np.array([[expt[j], 0], [0, expr[j]]])
            eLayer[:, j + 1] = (pInv @ eUD) @ eLayer[:, j]
            eu, ed = eLayer[0, j], eLayer[1, j]
            dEu[j + 1, :] = 0.5 * (dmix11[j, :] * eu + mix11[j] * dEu[j, :] +
                                   dmix12[j, :] * ed + mix12[j] * dEd[j, :])
            dEd[j + 1, :] = 0.5 * (dmix21[j, :] * eu + mix21[j] * dEu[j, :] +
                                   dmix22[j, :] * ed + mix22[j] * dEd[j, :])
            epu, epd = eLayer[0, j + 1], eLayer[1, j + 1]
            dHu[j + 1, :] = -epu / omu * dka[j + 1, :] - ka[j + 1] / omu * dEu[j + 1, :]
            dHd[j + 1, :] = epd / omu * dka[j + 1, :] + ka[j + 1] / omu * dEd[j + 1, :]
            e2 = abs(eLayer[0, j + 1] + eLayer[1, j + 1])
            e1 = abs(eLayer[0, j] + eLayer[1, j])
            if e2 - e1 > 0.0 or np.isnan(e2):
                eLayer[:, j + 1:] = 0.0
                dEu[j + 1:, j + 1:] = 0.0               # lower-right block only (App. B.7)
                dEd[j + 1:, j + 1:] = 0.0
                dHu[j + 1:, j + 1:] = 0.0
                dHd[j + 1:, j + 1:] = 0.0
                break

    dE = (dEu + dEd)[:, :-1]
    dH = (dHu + dHd)[:, :-1]
    if source == "E":
        return eLayer.sum(axis=0), dE
    hField = (eLayer[0, :] * (-ka) + eLayer[1, :] * ka) / omu
    return hField, dH


def getBCDerivParts(freq, yLen, zLen, sigma, source):
    """Structured content of getBCDerivMatrix (MT1DSensitivity.jl:253-333):
    returns bc (nb,), dF_left (nz, nz), dF_right (nz, nz), dF_mean_last (nz,)."""
    ny, nz = len(yLen), len(zLen)
    zNode = np.concatenate([[0.0], np.cumsum(zLen)])
    sig2D = sigma.reshape(nz, ny)
    nb = 2 * (ny + nz)
    bc = np.zeros(nb, dtype=complex)
    bc[0:ny + 1] = 1.0
    fL, dL = mt1DFieldSensMatrix(freq, sig2D[:, 0], zNode, source, 1.0)
    bc[ny + 1:ny + nz + 1] = fL[1:]
    fR, dR = mt1DFieldSensMatrix(freq, sig2D[:, -1], zNode, source, 1.0)
    bc[ny + nz + 1:ny + 2 * nz + 1] = fR[1:]
    fM, dM = mt1DFieldSensMatrix(freq, sig2D.mean(axis=1), zNode, source, 1.0)   # :313-314
    bc[ny + 2 * nz + 1:] = fM[-1]
    return bc, dL[1:, :], dR[1:, :], dM[-1, :]


def getBCDerivMatrix(freq, yLen, zLen, sigma, source):
    """Dense nb x nCell matrix exactly as the reference forms it (MT1DSensitivity.jl:253-333)."""
    ny, nz = len(yLen), len(zLen)
    ncell = ny * nz
    bc, dL, dR, dMlast = getBCDerivParts(freq, yLen, zLen, sigma, source)
    dBC = np.zeros((2 * (ny + nz), ncell), dtype=complex)
    dBC[ny + 1:ny + nz + 1, 0::ny] = dL
    dBC[ny + nz + 1:ny + 2 * nz + 1, ny - 1::ny] = dR
    for j in range(1, ny):                              # Julia j = 2:ny
        y1, y2 = yLen[j - 1], yLen[j]
        row = np.zeros(ncell, dtype=complex)
        row[j - 1::ny] += dMlast * (y1 / (y1 + y2))
        row[j::ny] += dMlast * (y2 / (y1 + y2))
        dBC[ny + 2 * nz + j, :] = row
    return dBC, bc


def getBCderivTE(freq, yLen, zLen, sigma):
    return getBCDerivMatrix(freq, yLen, zLen, sigma, "E")


def getBCderivTM(freq, yLen, zLen, sigma):
    return getBCDerivMatrix(freq, yLen, zLen, sigma, "H")


def _dBCT_times(freq, yLen, zLen, sigma, source, activeIdx, vecs):
    """Structured evaluation of (dBC*activeCell)^T * v for each v in vecs (no dense dBC);
    mathematically identical to the dense product, used for large test sizes."""
    ny, nz = len(yLen), len(zLen)
    bc, dL, dR, dMlast = getBCDerivParts(freq, yLen, zLen, sigma, source)
    outs = []
    for v in vecs:
        g = np.zeros((nz, ny), dtype=complex)
        g[:, 0] += dL.T @ v[ny + 1:ny + nz + 1]
        g[:, -1] += dR.T @ v[ny + nz + 1:ny + 2 * nz + 1]
        vb = v[ny + 2 * nz + 1:]                      # bottom nodes iy = 1..ny-1
        w1 = yLen[:-1] / (yLen[:-1] + yLen[1:])
        w2 = yLen[1:] / (yLen[:-1] + yLen[1:])
        colw = np.zeros(ny, dtype=complex)
        colw[:-1] += w1 * vb
        colw[1:] += w2 * vb
        g += dMlast[:, None] * colw[None, :]
        outs.append(g.reshape(-1)[activeIdx])
    return outs, bc


# ----------------------------------------------------------------------------
# J^T v (compJacTMatVec.jl:8-327)
# ----------------------------------------------------------------------------
def compJacTMatVec(exTE, hxTM, datVec, mesh, mtData, activeIdx, AinvTE, AinvTM,
                   dense_dbc=True, keep=None):
    yLen, zLen, origin, sigma = mesh.yLen, mesh.zLen, mesh.origin, mesh.sigma
    ny, nz = mesh.gridSize
    freqs, rxLoc, dataType, dataComp = mtData.freqs, mtData.rxLoc, mtData.dataType, mtData.dataComp
    rxID, freqID, dtID = mtData.rxID, mtData.freqID, mtData.dtID
    yNode = np.concatenate([[0.0], np.cumsum(yLen)]) - origin[0]
    zNode = np.concatenate([[0.0], np.cumsum(zLen)]) - origin[1]
    nFreq = len(freqs)
    nCell = ny * nz
    activeCell = activeCellMatrix(activeIdx, nCell)
    nAC = len(activeIdx)
    mu = MU0 * np.ones(nCell)
    if not mesh.setup:
        setupTensorMesh2D(mesh)
    F, Grad, AveCN, AveCF = mesh.Face, mesh.Grad, mesh.AveCN, mesh.AveCF
    ii, io = getBoundaryIndex(ny, nz)

    if mtData.compTE:
        MsigCN = sdiag(AveCN @ (F @ sigma))
        MmuF = sdiag(AveCF @ (F @ (1.0 / mu)))
        dGradTE = (Grad.T @ MmuF @ Grad).tocsr()
        rAioTE = dGradTE[ii][:, io]
        iAioTE = MsigCN[ii][:, io]
        dMsigCN = (AveCN[ii, :] @ F @ activeCell).tocsr()
    if mtData.compTM:
        MmuCN = sdiag(AveCN @ (F @ mu))
        MsigF = sdiag(AveCF @ (F @ (1.0 / sigma)))
        dGradTM = (Grad.T @ MsigF @ Grad).tocsr()
        rAioTM = dGradTM[ii][:, io]
        iAioTM = MmuCN[ii][:, io]
        Gradii = Grad[:, ii]
        Gradio = Grad[:, io]
        dMsigF = (AveCF @ F @ sdiag(-1.0 / sigma ** 2) @ activeCell).tocsr()

    rhoPhase = _is_rho_phase(dataType)                 # compJacTMatVec.jl:104-130 (see _rho_phase_rows for the spelling)
    if not rhoPhase and "Impedance" not in dataType:
        raise NotImplementedError(dataType)
    iZXY = iZYX = iRhoXY = iPhsXY = iRhoYX = iPhsYX = 0
    nRx = rxLoc.shape[0]
    for j, c in enumerate(dataComp):
        if c == "ZXY":
            iZXY = j + 1
        elif c == "ZYX":
            iZYX = j + 1
        elif "log10Rho" in c:
            raise NotImplementedError("log10Rho components: the reference's forward and sensitivity disagree on them")
        elif "RhoXY" in c:
            iRhoXY = j + 1
        elif c == "PhsXY":
            iPhsXY = j + 1
        elif "RhoYX" in c:
            iRhoYX = j + 1
        elif c == "PhsYX":
            iPhsYX = j + 1

    def pick(L, Q, i1, i2):
        """sVec and Q^T v of one system: impedance rows (i2 = 0) or rho rows + phase rows (:189-199, :260-270)"""
        idd1 = np.nonzero(subDcID == i1)[0]
        idr1 = subRxID[idd1] - 1
        sv = L[idr1, :].T @ datTmp[idd1]
        qv = Q[idr1, :].T @ datTmp[idd1]
        if i2:
            idd2 = np.nonzero(subDcID == i2)[0]
            idr2 = subRxID[idd2] - 1 + nRx
            sv = sv + L[idr2, :].T @ datTmp[idd2]
            qv = qv + Q[idr2, :].T @ datTmp[idd2]
        return sv, activeCell.T @ qv

    rxSens = preSetRxFieldSens(rxLoc, yNode, zNode, sigma)
    zid = rxSens.zid
    id0 = slice(zid * (ny + 1), (zid + 1) * (ny + 1))
    id1 = slice((zid + 1) * (ny + 1), (zid + 2) * (ny + 1))
    JTv = np.zeros(nAC, dtype=complex)
    QTv = np.zeros(nAC, dtype=complex)

    for iFreq in range(nFreq):
        freq = freqs[iFreq]
        omega = 2 * np.pi * freq
        indF = np.nonzero(freqID == iFreq + 1)[0]
        if len(indF) == 0:
            continue
        subRxID = rxID[indF]
        subDcID = dtID[indF]
        datTmp = np.conj(datVec[indF])
        calTE = any("XY" in dataComp[d - 1] for d in subDcID)
        calTM = any("YX" in dataComp[d - 1] for d in subDcID)

        if calTE:
            Ex01 = np.stack([exTE[id0, iFreq], exTE[id1, iFreq]], axis=1)
            L, Q = getDataFuncSensTE(omega, rxSens, Ex01, dataType)
            sVec, qt = pick(L, Q, iRhoXY, iPhsXY) if rhoPhase else pick(L, Q, iZXY, 0)
            QTv = QTv + qt
            AioTE = (rAioTE + 1j * omega * iAioTE).tocsr()
            eVal = AinvTE[iFreq].solve(sVec[ii])
            eVal_io = sVec[io]
            PTv = -1j * omega * (dMsigCN.T @ (exTE[ii, iFreq] * eVal))
            w = -(AioTE.T @ eVal)
            if dense_dbc:
                dBC, _ = getBCderivTE(freq, yLen, zLen, sigma)
                dBC = dBC[:, activeIdx]
                BTvii = dBC.T @ w
                BTvio = dBC.T @ eVal_io
            else:
                (BTvii, BTvio), _ = _dBCT_times(freq, yLen, zLen, sigma, "E", activeIdx, [w, eVal_io])
            JTv = JTv + PTv + BTvii + BTvio
            if keep is not None:
                keep.setdefault("terms", {})[("TE", iFreq)] = dict(
                    sVec=sVec, eVal=eVal, PTv=PTv, BTvii=BTvii, BTvio=BTvio, QTv=qt)

        if calTM:
            Hx01 = np.stack([hxTM[id0, iFreq], hxTM[id1, iFreq]], axis=1)
            L, Q = getDataFuncSensTM(omega, rxSens, Hx01, dataType)
            sVec, qt = pick(L, Q, iRhoYX, iPhsYX) if rhoPhase else pick(L, Q, iZYX, 0)
            QTv = QTv + qt
            AioTM = (rAioTM + 1j * omega * iAioTM).tocsr()
            eVal = AinvTM[iFreq].solve(sVec[ii])
            eVal_io = sVec[io]
            gl = -(Gradii @ eVal)
            PTv = dMsigF.T @ ((Gradii @ hxTM[ii, iFreq]) * gl)
            w = -(AioTM.T @ eVal)
            if dense_dbc:
                dBC, bc = getBCderivTM(freq, yLen, zLen, sigma)
                dBC = dBC[:, activeIdx]
                BTvii1 = dBC.T @ w
                BTvio = dBC.T @ eVal_io
            else:
                (BTvii1, BTvio), bc = _dBCT_times(freq, yLen, zLen, sigma, "H", activeIdx, [w, eVal_io])
            BTvii2 = dMsigF.T @ ((Gradio @ bc) * gl)
            JTv = JTv + PTv + BTvii1 + BTvii2 + BTvio
            if keep is not None:
                keep.setdefault("terms", {})[("TM", iFreq)] = dict(
                    sVec=sVec, eVal=eVal, PTv=PTv, BTvii1=BTvii1, BTvii2=BTvii2, BTvio=BTvio,
                    QTv=qt, bc_sens=bc)

    JTv = JTv + QTv
    return JTv.real


# ----------------------------------------------------------------------------
# utilities (HMCUtility.jl:69-77,168-258; HMCStruct.jl:99-125)
# ----------------------------------------------------------------------------
def modelTransform(sigModel):
    s = np.exp(sigModel)
    return s, s.copy()                  # dsigma is diag(exp(m)); returned as its diagonal


def compDataWeightMat(obsData, dataError):
    return 1.0 / np.abs(dataError)      # diagonal of dataW (HMCUtility.jl:168-190)


def getDataMisfit(dataRes):
    return float(0.5 * np.real(np.vdot(dataRes, dataRes)))


def setActiveElement(sigma, sigFix, fixIndex=None):
    """HMCUtility.jl:217-258; returns (activeIdx, bgModel)."""
    nGrid = len(sigma)
    inaInd = np.zeros(nGrid, dtype=int)
    bgModel = np.zeros(nGrid)
    for s in sigFix:
        indFix = (sigma == s)           # exact float equality, as the reference
        if not indFix.any():
            continue
        inaInd += indFix
        bgModel[indFix] += s
    if fixIndex is not None and len(fixIndex):
        inaInd[fixIndex] = 1
        bgModel[fixIndex] = sigma[fixIndex]
    return np.nonzero(inaInd == 0)[0], bgModel


def setupInverseDataModel(mesh, sigFix, sigLB, sigUB, obsData, dataErr, fixIndex=None):
    sigma = mesh.sigma
    activeIdx, bgModel = setActiveElement(sigma, sigFix, fixIndex)
    dataW = compDataWeightMat(obsData, dataErr)
    strModel = np.log(sigma[activeIdx])
    refModel = strModel.copy()
    cGrad = getCellGradient2D(mesh.yLen, mesh.zLen) @ activeCellMatrix(activeIdx, len(sigma))
    Wm = (cGrad.T @ cGrad).tocsr()
    return InvDataModel(np.asarray(obsData), dataW, strModel, refModel, activeIdx, bgModel, Wm)


# ----------------------------------------------------------------------------
# the hot path and its callers (HMCSampler.jl:206-559)
# ----------------------------------------------------------------------------
def compDataGradient(mesh, mtData, invParam, hmcprior, dense_dbc=True, keep=None):
    """HMCSampler.jl:277-330: m -> (predData, dataMisfit, dataGrad)."""
    strMod = invParam.strModel
    sig_act, dsigma = modelTransform(strMod)
    sigma = invParam.bgModel.copy()
    sigma[invParam.activeIdx] += sig_act
    mesh.sigma = sigma
    predData, fwd = MT2DFwdSolver(mesh, mtData, hmcprior.linearSolver, keep)
    dataRes = invParam.dataW * (predData - invParam.obsData)
    dataMisfit = getDataMisfit(dataRes)
    dataRes = invParam.dataW * dataRes
    g = compJacTMatVec(fwd.exTE, fwd.hxTM, dataRes, mesh, mtData, invParam.activeIdx,
                       fwd.AinvTE, fwd.AinvTM, dense_dbc, keep)
    dataGrad = dsigma * g
    if keep is not None:
        keep["exTE"], keep["hxTM"], keep["JTv"] = fwd.exTE, fwd.hxTM, g
    return predData, dataMisfit, dataGrad


def compDataMisfit(predData, invParam):
    return getDataMisfit(invParam.dataW * (predData - invParam.obsData))


def getKineticEnergy(momentum, invM):
    return 0.5 * float(np.dot(momentum, invM * momentum))


def getHamiltonian(mtData, mesh, invParam, hmcprior, momentum, invM):
    """HMCSampler.jl:358-397 (diagonal mass).  Uses mesh.sigma and invParam.strModel as left
    by the caller."""
    predData, _ = MT2DFwdSolver(mesh, mtData, hmcprior.linearSolver)
    dataMisfit = compDataMisfit(predData, invParam)
    kp = getKineticEnergy(momentum, invM)
    mprior = invParam.strModel - invParam.refModel
    mnorm = 0.5 * float(mprior @ (invParam.Wm @ mprior)) * hmcprior.regParam
    return dataMisfit, kp, dataMisfit + kp + mnorm, mnorm, predData


def checkParameterBound(model, momentum, hmcprior):
    """HMCSampler.jl:515-559 (reflection in ln sigma, momentum flip)."""
    sigmin = np.log(hmcprior.sigBounds[0])
    sigmax = np.log(hmcprior.sigBounds[1])
    for k in range(len(model)):
        if sigmin <= model[k] <= sigmax:
            continue
        niter = 0
        while True:
            niter += 1
            if model[k] < sigmin:
                model[k] = 2.0 * sigmin - model[k]
                momentum[k] *= -1.0
            if model[k] > sigmax:
                model[k] = 2.0 * sigmax - model[k]
                momentum[k] *= -1.0
            if sigmin <= model[k] <= sigmax:
                break
            if niter >= 500 and not np.isfinite(model[k]):
                raise FloatingPointError("non-finite model value; the reference would loop forever here")
    return model, momentum


def getMomentumVector(nparam, sqrtM, rng):
    """HMCSampler.jl:441-453: N(0,1) clipped to +-2.5, scaled by sqrt(M)."""
    mp = rng.standard_normal(nparam)
    mp = np.clip(mp, -2.5, 2.5)
    return sqrtM * mp


def proposeLeapfrog(currModel, currMomentum, invM, mesh, mtData, invParam, hmcprior, intstep,
                    dense_dbc=True):
    """HMCSampler.jl:206-269 with the number of steps drawn by the caller."""
    invParam.strModel = currModel.copy()
    predData, dataMisfit, dataGrad = compDataGradient(mesh, mtData, invParam, hmcprior, dense_dbc)
    hmcprior.nfevals += 1
    refModel, Wm = invParam.refModel, invParam.Wm
    dataGrad = dataGrad + (Wm @ (currModel - refModel)) * hmcprior.regParam
    dt = hmcprior.dt
    propMomentum = currMomentum - 0.5 * dt * dataGrad
    propModel = currModel.copy()
    maxStepSize = 3.0
    for k in range(1, intstep + 1):
        dm = dt * (invM * propMomentum)
        dmMax = np.max(np.abs(dm))
        if dmMax > maxStepSize:
            dm = dm / dmMax * maxStepSize
        propModel = propModel + dm
        propModel, propMomentum = checkParameterBound(propModel, propMomentum, hmcprior)
        invParam.strModel = propModel.copy()
        predData, dataMisfit, dataGrad = compDataGradient(mesh, mtData, invParam, hmcprior, dense_dbc)
        hmcprior.nfevals += 1
        dataGrad = dataGrad + (Wm @ (propModel - refModel)) * hmcprior.regParam
        delta = dt * dataGrad
        propMomentum = propMomentum - (delta if k < intstep else 0.5 * delta)
    return propModel, propMomentum


def runHMCSampler(mesh, mtData, invParam, hmcprior, rng, rhoref=None, dense_dbc=True):
    """HMCSampler.jl:72-196 with an explicit numpy Generator.  Draw order per sample:
    L ~ integers[Lmin, Lmax], accept u ~ U(0,1), fresh momentum.  `rhoref` fixes the random
    homogeneous start (the reference draws round(U(0.5 rho0, 1.5 rho0)), :100-109)."""
    nparam = len(invParam.strModel)
    ndata = len(invParam.obsData)
    invM = np.ones(nparam)
    sqrtM = np.ones(nparam)
    currModel = invParam.strModel.copy()               # file start model (:88), kept as chain state
    currMomentum = getMomentumVector(nparam, sqrtM, rng)
    sigma0 = invParam.strModel[0]                      # unique(strModel)[1]
    rho0 = 1.0 / np.exp(sigma0)
    if rhoref is None:
        rhoref = np.round(rho0 * 0.5 + (rho0 * 1.5 - rho0 * 0.5) * rng.random())
    strModel = np.log(np.ones(nparam) / rhoref)
    invParam.strModel = strModel.copy()
    invParam.refModel = strModel.copy()
    s, _ = modelTransform(invParam.strModel)           # updateStartModel (:832-847)
    sigma = invParam.bgModel.copy(); sigma[invParam.activeIdx] += s
    mesh.sigma = sigma
    startD, startK, startH, startM, predData = getHamiltonian(mtData, mesh, invParam, hmcprior,
                                                              currMomentum, invM)
    nsamples = hmcprior.totalsamples
    hmcmodel = np.zeros((nparam, nsamples))
    hmcdata = np.zeros((ndata, nsamples + 1), dtype=complex)
    hmstats = np.zeros((4, nsamples + 1))
    accept = np.zeros(nsamples, dtype=bool)
    hmstats[:, 0] = [startD, startM, startK, startH]
    hmcdata[:, 0] = predData
    for it in range(1, nsamples + 1):
        L = int(rng.integers(hmcprior.timestep[0], hmcprior.timestep[1] + 1))
        propModel, propMomentum = proposeLeapfrog(currModel, currMomentum, invM, mesh, mtData,
                                                  invParam, hmcprior, L, dense_dbc)
        finishD, finishK, finishH, finishM, predData = getHamiltonian(
            mtData, mesh, invParam, hmcprior, propMomentum, invM)
        hdif = startH - finishH
        aratio = rng.random()
        if hdif > 0 or aratio < np.exp(hdif):
            currModel, currMomentum = propModel.copy(), propMomentum.copy()
            startD, startM = finishD, finishM
            accept[it - 1] = True
            hmcdata[:, it] = predData
        else:
            hmcdata[:, it] = hmcdata[:, it - 1]
        currMomentum = getMomentumVector(nparam, sqrtM, rng)
        startK = getKineticEnergy(currMomentum, invM)
        startH = startD + startM + startK
        hmstats[:, it] = [startD, startM, startK, startH]
        hmcmodel[:, it - 1] = currModel
    return hmcmodel, dict(hmstats=hmstats, acceptstats=accept, nAccept=int(accept.sum()),
                          nReject=int((~accept).sum())), hmcdata


# ----------------------------------------------------------------------------
# explicit Jacobian (specification of J from compJacMat.jl:188-319) -- tests only
# ----------------------------------------------------------------------------
def compJacMat(mesh, mtData, activeIdx, fwd: MT2DFwdData):
    """Complex J (nData x nAC) for DataType Impedance: J = L*dF + Q*A with
    dF_ii = Aii^{-1}(P+B), dF_io = dBC (compJacMat.jl:206-248, 280-314)."""
    yLen, zLen, origin, sigma = mesh.yLen, mesh.zLen, mesh.origin, mesh.sigma
    ny, nz = mesh.gridSize
    nCell, nNode = ny * nz, (ny + 1) * (nz + 1)
    nAC = len(activeIdx)
    A = activeCellMatrix(activeIdx, nCell)
    mu = MU0 * np.ones(nCell)
    F, Grad, AveCN, AveCF = mesh.Face, mesh.Grad, mesh.AveCN, mesh.AveCF
    ii, io = getBoundaryIndex(ny, nz)
    yNode = np.concatenate([[0.0], np.cumsum(yLen)]) - origin[0]
    zNode = np.concatenate([[0.0], np.cumsum(zLen)]) - origin[1]
    rxSens = preSetRxFieldSens(mtData.rxLoc, yNode, zNode, sigma)
    zid = rxSens.zid
    id0 = slice(zid * (ny + 1), (zid + 1) * (ny + 1))
    id1 = slice((zid + 1) * (ny + 1), (zid + 2) * (ny + 1))
    dGradTE = (Grad.T @ sdiag(AveCF @ (F @ (1.0 / mu))) @ Grad).tocsr()
    MsigCN = sdiag(AveCN @ (F @ sigma))
    dMsigCN = (AveCN[ii, :] @ F @ A).tocsr()
    dGradTM = (Grad.T @ sdiag(AveCF @ (F @ (1.0 / sigma))) @ Grad).tocsr()
    MmuCN = sdiag(AveCN @ (F @ mu))
    Gradii, Gradio = Grad[:, ii], Grad[:, io]
    dMsigF = (AveCF @ F @ sdiag(-1.0 / sigma ** 2) @ A).tocsr()
    J = np.zeros((len(mtData.rxID), nAC), dtype=complex)
    for k in range(len(mtData.rxID)):
        pass
    rows = {}
    for iFreq, freq in enumerate(mtData.freqs):
        omega = 2 * np.pi * freq
        # TE
        AioTE = (dGradTE[ii][:, io] + 1j * omega * MsigCN[ii][:, io]).tocsr()
        dBC, _ = getBCderivTE(freq, yLen, zLen, sigma)
        dBC = dBC[:, activeIdx]
        PplusB = (-1j * omega * (sdiag(fwd.exTE[ii, iFreq]) @ dMsigCN)).toarray() - AioTE @ dBC
        dF = np.zeros((nNode, nAC), dtype=complex)
        dF[ii, :] = fwd.AinvTE[iFreq].solve(PplusB)
        dF[io, :] = dBC
        Ex01 = np.stack([fwd.exTE[id0, iFreq], fwd.exTE[id1, iFreq]], axis=1)
        L, Q = getDataFuncSensTE(omega, rxSens, Ex01, mtData.dataType)
        rows[("TE", iFreq)] = L @ dF + (Q @ A).toarray()
        # TM
        AioTM = (dGradTM[ii][:, io] + 1j * omega * MmuCN[ii][:, io]).tocsr()
        dBC, bc = getBCderivTM(freq, yLen, zLen, sigma)
        dBC = dBC[:, activeIdx]
        PplusB = (-(Gradii.T @ sdiag(Gradii @ fwd.hxTM[ii, iFreq]) @ dMsigF)).toarray() - AioTM @ dBC \
                 - (Gradii.T @ sdiag(Gradio @ bc) @ dMsigF).toarray()
        dF = np.zeros((nNode, nAC), dtype=complex)
        dF[ii, :] = fwd.AinvTM[iFreq].solve(PplusB)
        dF[io, :] = dBC
        Hx01 = np.stack([fwd.hxTM[id0, iFreq], fwd.hxTM[id1, iFreq]], axis=1)
        L, Q = getDataFuncSensTM(omega, rxSens, Hx01, mtData.dataType)
        rows[("TM", iFreq)] = L @ dF + (Q @ A).toarray()
    for k in range(len(mtData.rxID)):
        comp = mtData.dataComp[mtData.dtID[k] - 1]
        mode = "TE" if "XY" in comp else "TM"
        J[k, :] = rows[(mode, mtData.freqID[k] - 1)][mtData.rxID[k] - 1, :]
    return J
