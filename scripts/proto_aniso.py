"""CPU prototype (numpy/scipy; not part of the product): anisotropic homogenised background for the FDM stage.
The separable background P = T_y (x) M_z(q_y) + M_y (x) T_z(q_z) + i w M_y (x) M_z(s) may use DIFFERENT lateral means in its
y- and z-stiffness terms: for a coefficient that varies along y, homogenisation gives the harmonic mean for the
derivative along y and the arithmetic mean for the derivative along z.  TM: coefficient rho = 1/sigma.  TE: only the mass
term depends on the model (mean of sigma).  Iteration counts of COCG with the Jacobi / FDM / Jacobi preconditioner
(one sweep and two sweeps per side) to the GPU's stopping rule.
    python scripts/proto_aniso.py [cfg3] [true|rough|rough1.0|chain]"""
import os, sys, time
import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import hmcmt_oracle as O
from hmcmt2d_amd import synthetic as S
from scripts.proto_precond import systems, cocg, MU0

def means(mesh, sigma):
    ny, nz = mesh.gridSize
    s2 = sigma.reshape(nz, ny)
    rep = lambda m: np.repeat(m[:, None], ny, axis=1).reshape(-1)
    return dict(geo=rep(np.exp(np.log(s2).mean(1))), arith=rep(s2.mean(1)), harm=rep(1.0 / (1.0 / s2).mean(1)))

def bg_TM(mesh, rho_y_cell, rho_z_cell, omega):
    ny, nz = mesh.gridSize
    F, Grad, AveCN, AveCF = mesh.Face, mesh.Grad, mesh.AveCN, mesh.AveCF
    ii, io = O.getBoundaryIndex(ny, nz)
    ney = ny * (nz + 1)
    wy = (AveCF @ (F @ rho_y_cell))[:ney]
    wz = (AveCF @ (F @ rho_z_cell))[ney:]
    K = (Grad.T @ O.sdiag(np.concatenate([wy, wz])) @ Grad).tocsr()
    M = O.sdiag(AveCN @ (F @ (MU0 * np.ones(ny * nz))))
    return (K[ii][:, ii] + 1j * omega * M[ii][:, ii]).tocsc()

def bg_TE(mesh, sig_cell, omega):
    ny, nz = mesh.gridSize
    F, Grad, AveCN, AveCF = mesh.Face, mesh.Grad, mesh.AveCN, mesh.AveCF
    ii, io = O.getBoundaryIndex(ny, nz)
    K = (Grad.T @ O.sdiag(AveCF @ (F @ (np.ones(ny * nz) / MU0))) @ Grad).tocsr()
    M = O.sdiag(AveCN @ (F @ sig_cell))
    return (K[ii][:, ii] + 1j * omega * M[ii][:, ii]).tocsc()

def make_prec(A, Plu, sweeps, wj=0.8):
    dinv = wj / A.diagonal()
    def prec(r):
        z = dinv * r
        for _ in range(sweeps - 1): z = z + dinv * (r - A @ z)
        z = z + Plu.solve(r - A @ z)
        for _ in range(sweeps): z = z + dinv * (r - A @ z)
        return z
    return prec

def main():
    cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
    state = sys.argv[2] if len(sys.argv) > 2 else "true"
    mesh, data, sig_true = S.make_config(cfg)
    O.setupTensorMesh2D(mesh)
    ny, nz = mesh.gridSize; nair = len(mesh.airLayer)
    sigma = sig_true.copy()
    if state.startswith("rough"):
        std = float(state[5:]) if len(state) > 5 else 0.3
        n = ny * (nz - nair)
        sigma[ny * nair:] = np.exp(np.clip(np.log(0.01) + std * np.random.default_rng(1).standard_normal(n), np.log(1e-4), 0.0))
    elif state == "chain":        # smooth large-scale structure + the block: what a chain near the posterior looks like
        rng = np.random.default_rng(3)
        g = rng.standard_normal((nz - nair, ny))
        from scipy.ndimage import gaussian_filter
        g = gaussian_filter(g, 6.0); g *= 0.8 / g.std()
        sigma[ny * nair:] = np.exp(np.log(sig_true[ny * nair:]) + g.reshape(-1))
    freqs = [100.0, 4.64, 0.215, 0.01]
    st = systems(mesh, sigma, freqs)
    m = means(mesh, sigma)
    for md, f, A, b in st:
        om = 2 * np.pi * f
        res = []
        if md == "TM":
            variants = {"geo/geo": (1 / m["geo"], 1 / m["geo"]), "y:1/arith z:1/harm": (1 / m["arith"], 1 / m["harm"]),
                        "y:1/harm z:1/arith": (1 / m["harm"], 1 / m["arith"]), "arith/arith": (1 / m["arith"], 1 / m["arith"]),
                        "harm/harm": (1 / m["harm"], 1 / m["harm"])}
            for name, (ry, rz) in variants.items():
                Plu = spla.splu(bg_TM(mesh, ry, rz, om))
                its = [cocg(A, b, make_prec(A, Plu, sw))[1] for sw in (1, 2)]
                res.append(f"{name} {its[0]}/{its[1]}")
        else:
            for name in ("geo", "arith", "harm"):
                Plu = spla.splu(bg_TE(mesh, m[name], om))
                its = [cocg(A, b, make_prec(A, Plu, sw))[1] for sw in (1, 2)]
                res.append(f"{name} {its[0]}/{its[1]}")
        print(f"{md} {f:8.3g} Hz (one sweep / two sweeps):  " + "   ".join(res), flush=True)

if __name__ == "__main__":
    main()
