"""CPU prototype for preconditioner experiments (numpy/scipy; not part of the product or the tests).

Builds the TE / TM interior systems of a config with the oracle's assembly, the FDM background operator
P = A(sigma_bar(z)) (exactly what the GPU's fast-diagonalisation stage inverts, here through a sparse LU), and runs
COCG with variants of the preconditioner, printing iteration counts to the GPU's stopping rule (|z| <= tol |x|).

    python scripts/proto_precond.py [cfg] [state]     state: true | rough | rough1.0
"""
import os
import sys
import time

import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import hmcmt_oracle as O                                   # noqa: E402
from hmcmt2d_amd import synthetic as S                                 # noqa: E402

MU0 = 4e-7 * np.pi


def systems(mesh, sigma, freqs):
    """[(mode, freq, Aii csr, rhs)] with the oracle's assembly."""
    mesh.sigma = sigma
    data = S.make_data_layout(freqs, np.array([0.0, 400.0]))
    keep = {}
    O.MT2DFwdSolver(mesh, data, "", keep)
    out = []
    for md in ("TE", "TM"):
        for f in freqs:
            out.append((md, f, keep["Aii"][(md, f)].tocsr(), keep["rhs"][(md, f)]))
    return out


def lateral_mean_sigma(mesh, sigma, kind="geo"):
    ny, nz = mesh.gridSize
    s2 = sigma.reshape(nz, ny)
    m = np.exp(np.log(s2).mean(axis=1)) if kind == "geo" else s2.mean(axis=1)
    return np.repeat(m[:, None], ny, axis=1).reshape(-1)


def cocg(A, b, prec, tol=1e-11, maxit=400, x0=None):
    x = np.zeros_like(b) if x0 is None else x0.copy()
    r = b - A @ x
    z = prec(r)
    p = z.copy()
    rho = r @ z
    for it in range(1, maxit + 1):
        q = A @ p
        al = rho / (p @ q)
        x += al * p
        r -= al * q
        z = prec(r)
        if np.linalg.norm(z) <= tol * np.linalg.norm(x):
            return x, it
        rho1 = r @ z
        p = z + (rho1 / rho) * p
        rho = rho1
    return x, maxit


def make_fdmj(A, Plu, wj=0.7):
    dinv = wj / A.diagonal()

    def prec(r):
        z0 = dinv * r
        z1 = z0 + Plu.solve(r - A @ z0)
        return z1 + dinv * (r - A @ z1)
    return prec


def main():
    cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
    state = sys.argv[2] if len(sys.argv) > 2 else "true"
    mesh, data, sig_true = S.make_config(cfg)
    O.setupTensorMesh2D(mesh)
    ny, nz = mesh.gridSize
    nair = len(mesh.airLayer)
    if state == "true":
        sigma = sig_true.copy()
    else:
        std = float(state[5:]) if len(state) > 5 else 0.3
        sigma = sig_true.copy()
        n = ny * (nz - nair)
        sigma[ny * nair:] = np.exp(np.clip(np.log(0.01) + std * np.random.default_rng(1).standard_normal(n), np.log(1e-4), 0.0))
    freqs = [100.0, 4.64, 0.215, 0.01]
    sys_true = systems(mesh, sigma, freqs)
    sys_bg = systems(mesh, lateral_mean_sigma(mesh, sigma), freqs)
    for (md, f, A, b), (_, _, P, _) in zip(sys_true, sys_bg):
        t0 = time.time()
        Plu = spla.splu(P.tocsc())
        x, it = cocg(A, b, make_fdmj(A, Plu))
        print(f"{md} {f:8.3g} Hz  fdmj {it:3d}   ({time.time() - t0:.1f} s)", flush=True)


if __name__ == "__main__":
    main()
