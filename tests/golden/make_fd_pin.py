"""Generates tests/golden/dprism_fd.npz: Richardson-extrapolated central differences of the ORACLE's data misfit, cell by
cell, at the model the reference's example data were generated from (tests/helpers.py::dprism_generating_problem --
where the forward map is pinned on the reference's own file, tests/test_oracle_kat.py).  Run in the build container:
`python tests/golden/make_fd_pin.py [workers]` (about 10 minutes on 7 cores).

Two differences per cell c (m = ln sigma, h = 0.01 and 0.02, R = (4 D(h) - D(2h)) / 3):
  fd_full[c]    of phi(m) with everything recomputed -- the exact derivative of the pinned forward map;
  fd_frozen[c]  of phi(m) with the Dirichlet values of all four sides held at the base model's -- the exact value of
                the P- and Q-terms of J^T v (compJacTMatVec.jl:235, 209, 306-307, 280): for cells that touch no mesh
                boundary the only other terms are the boundary-derivative terms B^T v (:237-242, :309-316), which the
                reference approximates.
Cells: 16 earth rows x 16 columns = 256 cells of the uniform core under the receiver line, none in the padding.
"""
import os
import sys
from multiprocessing import get_context

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
HERE = os.path.dirname(os.path.abspath(__file__))
H = 0.01
ROWS = [0, 1, 2, 3, 5, 7, 9, 11, 13, 15, 18, 21, 24, 28, 32, 36]        # earth rows (0 = first row below the surface)
COLS = list(range(10, 86, 5))                                            # of 96 columns, 7 padding columns per side


def pin_cells(ny):
    return np.array([kz * ny + ky for kz in ROWS for ky in COLS])


def _setup():
    from oracle import hmcmt_oracle as O
    from hmcmt2d_amd import synthetic as S, invsetup as I
    from tests.helpers import dprism_generating_problem
    mesh, data, obs, err = dprism_generating_problem()
    O.setupTensorMesh2D(mesh)
    inv = I.setupInverseDataModel(mesh, [S.SIG_AIR], 0.0, 0.0, obs, err)
    m0 = np.log(mesh.sigma[inv.activeIdx])
    return O, mesh, data, inv, m0


def _worker(cells):
    os.environ["OMP_NUM_THREADS"] = "1"
    O, mesh, data, inv, m0 = _setup()
    keep = {}
    O.MT2DFwdSolver(mesh, data, keep=keep)
    bc0 = dict(keep["bc"])

    def phi(mm, frozen):
        s = inv.bgModel.copy(); s[inv.activeIdx] += np.exp(mm); mesh.sigma = s
        return O.compDataMisfit(O.MT2DFwdSolver(mesh, data, bc_fixed=bc0 if frozen else None)[0], inv)

    out = []
    for c in cells:
        row = []
        for frozen in (False, True):
            D = []
            for hh in (H, 2 * H):
                mp, mm = m0.copy(), m0.copy()
                mp[c] += hh; mm[c] -= hh
                D.append((phi(mp, frozen) - phi(mm, frozen)) / (2 * hh))
            row.append((4 * D[0] - D[1]) / 3)
        out.append(row)
    return np.array(out)


if __name__ == "__main__":
    nw = int(sys.argv[1]) if len(sys.argv) > 1 else 7
    O, mesh, data, inv, m0 = _setup()
    cells = pin_cells(mesh.gridSize[0])
    chunks = np.array_split(cells, nw)
    with get_context("spawn").Pool(nw) as pool:
        res = np.concatenate(pool.map(_worker, chunks))
    np.savez_compressed(os.path.join(HERE, "dprism_fd.npz"), cells=cells, h=H, fd_full=res[:, 0], fd_frozen=res[:, 1])
    print("cells", len(cells), "max|fd|", np.abs(res).max())
