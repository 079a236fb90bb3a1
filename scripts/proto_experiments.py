"""The preconditioner / Krylov experiments behind DESIGN.md §9 (CPU prototype; numpy/scipy; not part of the product).

    python scripts/proto_experiments.py <experiment> [cfg3] [true | rough0.3 | rough1.0]

experiments:  damping  sweeps  lines  backgrounds  twolevel  seed  fp32  cocr

Each prints COCG (or variant) iteration counts to the GPU's stopping rule on the TE / TM systems of the config at
100, 4.64, 0.215 and 0.01 Hz; the FDM stage is an exact solve with the layered background (scripts/proto_precond.py).
"""
import sys

import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla

sys.path.insert(0, __file__.rsplit("/", 1)[0])
from proto_precond import O, S, cocg, lateral_mean_sigma, make_fdmj, systems   # noqa: E402


def setup(cfg, state):
    mesh, data, sig_true = S.make_config(cfg)
    O.setupTensorMesh2D(mesh)
    ny, nz = mesh.gridSize
    nair = len(mesh.airLayer)
    sigma = sig_true.copy()
    if state != "true":
        std = float(state[5:])
        n = ny * (nz - nair)
        sigma[ny * nair:] = np.exp(np.clip(np.log(sig_true[ny * nair:]) + std * np.random.default_rng(1).standard_normal(n),
                                           np.log(1e-4), 0.0))
    return mesh, sigma, ny, nz, nair


def sandwich(A, Plu, pre, post):
    """z = S_post-corrected( F-corrected( S_pre r ) ): the symmetric product form of the GPU preconditioner"""
    def prec(r):
        z0 = pre(r)
        z1 = z0 + Plu.solve(r - A @ z0)
        return z1 + post(r - A @ z1)
    return prec


def jacobi(A, w, nsw):
    d = A.diagonal()

    def f(r):
        z = (w / d) * r
        for _ in range(nsw - 1):
            z = z + (w / d) * (r - A @ z)
        return z
    return f


def line_jacobi(A, ny, nz, w, direction):
    n_y, n_z = ny - 1, nz - 1
    idx = np.arange(n_y * n_z).reshape(n_z, n_y)
    perm, m = (idx.T.reshape(-1), n_z) if direction == "z" else (idx.reshape(-1), n_y)
    coo = A.tocsr()[perm][:, perm].tocoo()
    keep = (coo.row // m) == (coo.col // m)
    lu = spla.splu(sp.csc_matrix((coo.data[keep], (coo.row[keep], coo.col[keep])), shape=A.shape))
    inv = np.argsort(perm)
    return lambda r: w * lu.solve(r[perm])[inv]


def rbgs(A, ny, nz):
    d = A.diagonal()
    n_y = ny - 1
    i = np.arange(A.shape[0])
    red = ((i % n_y + i // n_y) % 2 == 0)

    def pre(r):
        z = np.where(red, r / d, 0)
        return z + np.where(~red, (r - A @ z) / d, 0)

    def post(r):
        z = np.where(~red, r / d, 0)
        return z + np.where(red, (r - A @ z) / d, 0)
    return pre, post


def interp1d(nf, x):
    cidx = np.arange(1, nf, 2)
    rows, cols, vals = [], [], []
    for j, ci in enumerate(cidx):
        rows.append(ci); cols.append(j); vals.append(1.0)
        xl, xc, xm = x[ci - 1], x[ci + 1], x[ci]
        rows.append(ci - 1); cols.append(j); vals.append((xm - xl) / (xc - xl))
        if ci + 1 < nf:
            xr, xm = x[min(ci + 2, nf) + 1], x[ci + 2]
            rows.append(ci + 1); cols.append(j); vals.append((xr - xm) / (xr - xc))
    return sp.csr_matrix((vals, (rows, cols)), shape=(nf, len(cidx)))


def cocg_store(A, b, prec, tol=1e-11, maxit=400, x0=None, rnd=None):
    """COCG keeping the directions (seed projection) or rounding p, z to complex64 (rnd)"""
    c64 = (lambda v: v.astype(np.complex64).astype(np.complex128)) if rnd else (lambda v: v)
    x = np.zeros_like(b) if x0 is None else x0.copy()
    r = b - A @ x
    z = c64(prec(r)); p = z.copy(); rho = r @ z
    P, Q, PQ = [], [], []
    for it in range(1, maxit + 1):
        q = A @ p
        pq = p @ q
        P.append(p.copy()); Q.append(q.copy()); PQ.append(pq)
        al = rho / pq
        x += al * p; r -= al * q
        z = c64(prec(r))
        if np.linalg.norm(z) <= tol * np.linalg.norm(x):
            return x, it, P, Q, PQ
        rho1 = r @ z
        p = c64(z + (rho1 / rho) * p)
        rho = rho1
    return x, maxit, P, Q, PQ


def cocr(A, b, prec, tol=1e-11, maxit=400):
    x = np.zeros_like(b); r = b.copy()
    z = prec(r); p = z.copy(); w = A @ z; q = w.copy()
    zw = z @ w
    for it in range(1, maxit + 1):
        u = prec(q)
        al = zw / (q @ u)
        x += al * p; r -= al * q; z -= al * u
        if np.linalg.norm(z) <= tol * np.linalg.norm(x):
            return x, it
        w = A @ z
        zw1 = z @ w
        p = z + (zw1 / zw) * p; q = w + (zw1 / zw) * q
        zw = zw1
    return x, maxit


def main():
    exp = sys.argv[1]
    cfg = sys.argv[2] if len(sys.argv) > 2 else "cfg3"
    state = sys.argv[3] if len(sys.argv) > 3 else "true"
    mesh, sigma, ny, nz, nair = setup(cfg, state)
    freqs = [100.0, 4.64, 0.215, 0.01]
    st = systems(mesh, sigma, freqs)
    s2 = sigma.reshape(nz, ny)
    means = {"geo": lateral_mean_sigma(mesh, sigma),
             "ari": np.repeat(s2.mean(1)[:, None], ny, 1).reshape(-1),
             "har": np.repeat((1.0 / (1.0 / s2).mean(1))[:, None], ny, 1).reshape(-1),
             "median": np.repeat(np.median(s2, axis=1)[:, None], ny, 1).reshape(-1)}
    bgs = {k: systems(mesh, v, freqs) for k, v in (means.items() if exp == "backgrounds" else [("geo", means["geo"])])}
    yN = np.concatenate([[0.0], np.cumsum(mesh.yLen)]); zN = np.concatenate([[0.0], np.cumsum(mesh.zLen)])
    rng = np.random.default_rng(0)
    for i, (md, f, A, b) in enumerate(st):
        Plu = spla.splu(bgs["geo"][i][2].tocsc())
        out = []
        if exp == "damping":
            out = [f"w{w}:{cocg(A, b, make_fdmj(A, Plu, w))[1]}" for w in (0.5, 0.6, 0.7, 0.8, 0.9, 1.0)]
        elif exp == "sweeps":
            for name, (pre, post) in (("jac0.7x1", (jacobi(A, 0.7, 1),) * 2), ("jac0.7x2", (jacobi(A, 0.7, 2),) * 2),
                                      ("jac0.8x2", (jacobi(A, 0.8, 2),) * 2), ("jac0.7x3", (jacobi(A, 0.7, 3),) * 2),
                                      ("rbgs", rbgs(A, ny, nz))):
                out.append(f"{name}:{cocg(A, b, sandwich(A, Plu, pre, post))[1]}")
        elif exp == "lines":
            for name, sm in (("point0.7", jacobi(A, 0.7, 1)), ("zline0.7", line_jacobi(A, ny, nz, 0.7, "z")),
                             ("zline1.0", line_jacobi(A, ny, nz, 1.0, "z")), ("yline0.7", line_jacobi(A, ny, nz, 0.7, "y"))):
                out.append(f"{name}:{cocg(A, b, sandwich(A, Plu, sm, sm))[1]}")
            out.append(f"plainFDM:{cocg(A, b, lambda r: Plu.solve(r))[1]}")
        elif exp == "backgrounds":
            for k in bgs:
                lu = spla.splu(bgs[k][i][2].tocsc())
                out.append(f"{k}:{cocg(A, b, make_fdmj(A, lu, 0.8))[1]}")
        elif exp == "twolevel":
            P = sp.kron(interp1d(nz - 1, zN), interp1d(ny - 1, yN)).tocsr()
            Clu = spla.splu((P.T @ A @ P).tocsc())
            ops = {"S": jacobi(A, 0.7, 1), "F": lambda r: Plu.solve(r), "C": lambda r: P @ Clu.solve(P.T @ r)}
            for order in ("SFS", "SFCFS", "SCFCS", "SFCS"):
                def prec(r, order=order):
                    z = np.zeros_like(r)
                    for o in order:
                        z = z + ops[o](r - A @ z)
                    return z
                out.append(f"{order}:{cocg(A, b, prec, maxit=150)[1]}")
        elif exp == "seed":
            prec = make_fdmj(A, Plu, 0.8)
            _, itf, Ps, Qs, PQ = cocg_store(A, b, prec)
            s = np.zeros(A.shape[0], complex)
            for row in (nair - 1, nair):
                for c in np.linspace(20, ny - 21, 41).astype(int):
                    s[row * (ny - 1) + c - 1: row * (ny - 1) + c + 2] += rng.standard_normal(3) + 1j * rng.standard_normal(3)
            ita = cocg_store(A, s, prec)[1]
            x0 = np.zeros_like(s); r = s.copy()
            for p, q, pq in zip(Ps, Qs, PQ):
                c = (p @ r) / pq
                x0 += c * p; r -= c * q
            out = [f"fwd:{itf}", f"adjoint cold:{ita}", f"after seed projection:{cocg_store(A, s, prec, x0=x0)[1]}",
                   f"|r0|/|s|:{np.linalg.norm(r) / np.linalg.norm(s):.2f}"]
        elif exp == "fp32":
            prec = make_fdmj(A, Plu, 0.8)
            xe = spla.splu(A.tocsc()).solve(b)
            x0, it0 = cocg_store(A, b, prec)[:2]
            x1, it1 = cocg_store(A, b, prec, rnd=True)[:2]
            out = [f"fp64:{it0} err {np.linalg.norm(x0 - xe) / np.linalg.norm(xe):.1e}",
                   f"p,z complex64:{it1} err {np.linalg.norm(x1 - xe) / np.linalg.norm(xe):.1e}"]
        elif exp == "cocr":
            prec = make_fdmj(A, Plu, 0.8)
            out = [f"COCG:{cocg(A, b, prec)[1]}", f"COCR:{cocr(A, b, prec)[1]}"]
        else:
            raise SystemExit(__doc__)
        print(f"{md} {f:8.3g} Hz  " + "  ".join(out), flush=True)


if __name__ == "__main__":
    main()
