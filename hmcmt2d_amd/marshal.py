"""Turn the reference-shaped host structs into the flat arrays of the C ABI (include/hmcmt.h).

The same argument list is what the Julia shim (julia/HMCMTHip.jl) passes with `ccall`:
`TensorMesh2D.yLen/zLen/origin`, `MTData.freqs/rxLoc/rxID/freqID/dtID/dataID/dataComp`,
`InvDataModel.obsData/dataW/activeCell/bgModel` (HMCFileIO.jl:26-60, HMCStruct.jl:75-91).
"""
from __future__ import annotations

import ctypes as C
import numpy as np

c_double_p = C.POINTER(C.c_double)
c_int64_p = C.POINTER(C.c_int64)
c_uint8_p = C.POINTER(C.c_uint8)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _i64(a):
    return np.ascontiguousarray(a, dtype=np.int64)


COMPONENT_CODES = {"ZXY": 1, "ZYX": 2, "RhoXY": 3, "PhsXY": 4, "RhoYX": 5, "PhsYX": 6}


def comp_modes(dataComp, dataType="Impedance"):
    """Component codes of include/hmcmt.h: ZXY 1, ZYX 2 (DataType Impedance); RhoXY 3, PhsXY 4, RhoYX 5, PhsYX 6
    (DataType Rho_Pha: the names compJacTMatVec.jl:106-113 looks for).  `log10Rho*` is refused: the reference's
    forward returns the linear apparent resistivity for it while its sensitivity branch switches to log10
    (mt2DTE.jl:253 vs dataFuncSens.jl:154-160)."""
    out = []
    for c in dataComp:
        if c not in COMPONENT_CODES:
            raise ValueError(f"unsupported data component {c!r} (supported: {', '.join(COMPONENT_CODES)})")
        code = COMPONENT_CODES[c]
        if (code <= 2) != ("Impedance" in dataType):
            raise ValueError(f"data component {c!r} does not belong to DataType {dataType!r}")
        out.append(code)
    return np.asarray(out, dtype=np.int64)


class CreateArgs:
    """Keeps the numpy buffers alive and exposes them as a ctypes argument tuple."""

    def __init__(self, mtMesh, mtData, invParam):
        if "Impedance" not in mtData.dataType and "Rho_Pha" not in mtData.dataType and "Rho_Phs" not in mtData.dataType:
            raise ValueError(f"unsupported DataType {mtData.dataType!r} (Impedance or Rho_Pha)")
        self.real_data = "Impedance" not in mtData.dataType
        ny, nz = int(mtMesh.gridSize[0]), int(mtMesh.gridSize[1])
        self.ny, self.nz = ny, nz
        self.yLen = _f64(mtMesh.yLen)
        self.zLen = _f64(mtMesh.zLen)
        self.origin = _f64(mtMesh.origin)
        self.freqs = _f64(mtData.freqs)
        rx = _f64(mtData.rxLoc)
        self.rxY = _f64(rx[:, 0])
        self.rxZ = _f64(rx[:, 1])
        self.compMode = comp_modes(mtData.dataComp, mtData.dataType)
        self.freqID = _i64(mtData.freqID)
        self.rxID = _i64(mtData.rxID)
        self.dtID = _i64(mtData.dtID)
        self.dataID = np.ascontiguousarray(mtData.dataID, dtype=np.uint8)
        self.obs = np.ascontiguousarray(invParam.obsData, dtype=np.complex128)
        self.dataW = _f64(invParam.dataW)
        self.activeIdx = _i64(np.asarray(invParam.activeIdx) + 1)      # 1-based like activeCell.rowval
        self.bg = _f64(invParam.bgModel)
        self.nFreq = len(self.freqs)
        self.nRx = len(self.rxY)
        self.nComp = len(self.compMode)
        self.nData = len(self.obs)
        self.nAC = len(self.activeIdx)
        if len(self.yLen) != ny or len(self.zLen) != nz or len(self.bg) != ny * nz:
            raise ValueError("mesh arrays do not match gridSize")

    def as_tuple(self):
        p = lambda a, t: a.ctypes.data_as(t)
        return (C.c_int64(self.ny), C.c_int64(self.nz), p(self.yLen, c_double_p), p(self.zLen, c_double_p),
                p(self.origin, c_double_p), C.c_int64(self.nFreq), p(self.freqs, c_double_p),
                C.c_int64(self.nRx), p(self.rxY, c_double_p), p(self.rxZ, c_double_p),
                C.c_int64(self.nComp), p(self.compMode, c_int64_p),
                C.c_int64(self.nData), p(self.freqID, c_int64_p), p(self.rxID, c_int64_p), p(self.dtID, c_int64_p),
                p(self.dataID, c_uint8_p), p(self.obs, c_double_p), p(self.dataW, c_double_p),
                C.c_int64(self.nAC), p(self.activeIdx, c_int64_p), p(self.bg, c_double_p))


CREATE_ARGTYPES = [C.c_int64, C.c_int64, c_double_p, c_double_p, c_double_p, C.c_int64, c_double_p,
                   C.c_int64, c_double_p, c_double_p, C.c_int64, c_int64_p,
                   C.c_int64, c_int64_p, c_int64_p, c_int64_p, c_uint8_p, c_double_p, c_double_p,
                   C.c_int64, c_int64_p, c_double_p]
