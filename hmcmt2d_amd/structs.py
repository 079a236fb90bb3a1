"""Host-side data structures mirroring the reference's Julia structs field for field.

`TensorMesh2D`, `MTData`      <- HMCMT/src/HMCFileIO/HMCFileIO.jl:26-60
`HMCPrior`, `HMCParameter`,
`HMCStatus`, `InvDataModel`   <- HMCMT/src/HMCStruct/HMCStruct.jl:18-91, defaults :129-165

Index arrays (`rxID`, `freqID`, `dtID`) stay 1-based exactly as the reference reader stores
them (readMT2DData.jl:118-139); the C ABI converts.  Sparse operators of the reference
(`Face/Grad/AveCN/AveCF`, `activeCell`, `dataW`, `Wm`) are not materialised: the HIP library
builds the stencils from `yLen/zLen` directly, `activeCell` is kept as its column index list
`activeIdx` (0-based), `dataW` as its diagonal and `Wm` as a scipy CSR matrix.
"""
from __future__ import annotations

from dataclasses import dataclass, field
import numpy as np


@dataclass
class TensorMesh2D:
    yLen: np.ndarray              # cell widths in y
    zLen: np.ndarray              # cell heights in z, air layers first (top -> down)
    airLayer: np.ndarray          # air thicknesses as listed in the file (bottom -> up)
    gridSize: tuple               # (ny, nz_total)
    origin: np.ndarray            # (y0, z0); z0 already shifted by the air thickness
    sigma: np.ndarray             # cell conductivities, y fastest, air rows first
    Face: object = None           # kept for struct parity; unused by the HIP path
    Grad: object = None
    AveCN: object = None
    AveCF: object = None
    setup: bool = False


@dataclass
class MTData:
    rxLoc: np.ndarray             # (nRx, 2): y, z
    freqs: np.ndarray
    dataType: str                 # "Impedance" (the only type that works end-to-end)
    dataComp: list                # ["ZXY", "ZYX"]
    rxID: np.ndarray              # 1-based
    freqID: np.ndarray            # 1-based
    dtID: np.ndarray              # 1-based
    dataID: np.ndarray            # bool over (dt, rx, freq), dt fastest
    compTE: bool = True
    compTM: bool = True


@dataclass
class HMCPrior:
    burninsamples: int = 100
    totalsamples: int = 500
    sigBounds: list = field(default_factory=lambda: [0.01, 10.0])
    sigmastd: float = 0.05
    dt: float = 0.01
    timestep: list = field(default_factory=lambda: [10, 15])
    linearSolver: str = ""        # "" / "mumps" in the reference; "hip" selects this library
    massType: str = "diagonal"
    regParam: float = 1.0
    nfevals: int = 0


def initHMCPrior() -> HMCPrior:
    """HMCStruct.jl:129-140."""
    return HMCPrior()


@dataclass
class HMCParameter:
    nparam: int
    rhomodel: np.ndarray
    momentum: np.ndarray
    invM: np.ndarray              # diagonal of M^-1 (the reference's default "diagonal" mass)
    sqrtM: np.ndarray             # diagonal of M^1/2


def initHMCParameter(nparam: int) -> HMCParameter:
    """HMCStruct.jl:159-170."""
    return HMCParameter(nparam, np.zeros(nparam), np.zeros(nparam), np.zeros(nparam), np.zeros(nparam))


@dataclass
class HMCStatus:
    nAccept: int
    nReject: int
    acceptstats: np.ndarray       # bool[nsamples]
    hmstats: np.ndarray           # (4, nsamples+1): dataMisfit, mnorm, kinetic, hamiltonian


def initHMCStatus(nsamples: int) -> HMCStatus:
    """HMCStruct.jl:144-155."""
    return HMCStatus(0, 0, np.zeros(nsamples, dtype=bool), np.zeros((4, nsamples + 1)))


@dataclass
class InvDataModel:
    obsData: np.ndarray           # complex128[nData]
    dataW: np.ndarray             # diagonal of the data weighting matrix, 1/|err|
    strModel: np.ndarray          # ln(sigma) on active cells
    refModel: np.ndarray
    activeIdx: np.ndarray         # 0-based cell ids of the active (non-fixed) cells
    bgModel: np.ndarray           # conductivity of the fixed cells, 0 elsewhere
    Wm: object                    # scipy.sparse CSR, (G A)^T (G A)
