"""Binary problem dump read by tests/c_abi/abi_check.c: the flat argument list of hmcmt_create (hmcmt2d_amd/marshal.py)
followed by a model vector, all native-endian."""
import numpy as np

from hmcmt2d_amd.marshal import CreateArgs


def write_dump(path, mesh, data, inv, m, device=0):
    a = CreateArgs(mesh, data, inv)
    with open(path, "wb") as f:
        np.array([a.ny, a.nz, a.nFreq, a.nRx, a.nComp, a.nData, a.nAC, device], dtype=np.int64).tofile(f)
        for arr in (a.yLen, a.zLen, a.origin, a.freqs, a.rxY, a.rxZ, a.compMode, a.freqID, a.rxID, a.dtID, a.dataID,
                    a.obs, a.dataW, a.activeIdx, a.bg, np.ascontiguousarray(m, dtype=np.float64)):
            arr.tofile(f)
    return a
