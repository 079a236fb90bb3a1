"""An independent check of the boundary-derivative ASSEMBLY ("what's missing" 2).

`getBCDerivMatrix` (MTSensitivity/MT1DSensitivity.jl:253-333) builds d(bc)/d(sigma) for the Dirichlet values of the four
sides out of three calls of the 1-D sensitivity `mt1DFieldSensMatrix`: the left and right columns' nz x nz blocks
(:272-285) and ONE row for the bottom side -- the layer-mean profile's last row -- spread over the two columns next to every
bottom node with the weights yLen/(yLen1 + yLen2) (:312-328).  The 1-D building blocks are held by the known-answer tests of
tests/test_oracle_kat.py; the assembly around them rested on the restatement alone.  It is the derivative of something the
reference computes itself: the boundary values of `getBoundaryMT2DTE/TM` (mt2DTE.jl:100-134, mt2DTM.jl:100-134) -- up to the
reference's documented approximations, each of which can be switched off:

  * the forward values carry the displacement term mu0 eps0 omega^2, the sensitivity does not (B.4)   -> eps0 := 0;
  * the bottom side uses ONE mean profile instead of the per-node two-column averages (B.5)        -> exact on a laterally
    uniform model, where the two coincide;
  * the last layer's derivative omits the appended half-space (B.6)                                 -> the cells of the
    bottom row are left out of the comparison.

(1) oracle, CPU: every entry of the dense dBC against Richardson quotients of the eps0-free boundary values with respect to
    single cells -- left / right blocks on a rough 2-D model, the bottom rows with their width weights on a laterally uniform
    one (all three sides there) -- and the `bc` it returns against those values.
(2) HIP, GPU, no oracle in between: on a laterally uniform model at frequencies where eps0 is negligible, the share of the
    gradient that the boundary-derivative kernels produce (k_sens_layers / k_sens_profile / k_bcsens_pre / k_bcsens_contract:
    gradient with minus gradient without hmcmt_debug_flags bit 1) against the difference of two Richardson quotients of the
    HIP path's own forward map, one with the boundary values recomputed and one with them frozen (bit 0) -- the exact value
    of what those terms approximate.
"""
import numpy as np
import pytest

from hmcmt2d_amd import synthetic as S, invsetup as I


def _richardson_vec(f, x, c, h):
    def cd(step):
        xp, xm = x.copy(), x.copy()
        xp[c] += step; xm[c] -= step
        return (f(xp) - f(xm)) / (2 * step)
    return (4 * cd(h) - cd(2 * h)) / 3


@pytest.mark.parametrize("mode,source", [("TE", "E"), ("TM", "H")])
def test_boundary_derivative_matrix_is_the_derivative_of_the_epsilon_free_boundary_values(monkeypatch, mode, source):
    from oracle import hmcmt_oracle as O
    monkeypatch.setattr(O, "EPS0", 0.0)
    rng = np.random.default_rng(21)
    ny, nz = 6, 8
    yLen = 10.0 ** rng.uniform(2.0, 3.0, ny)
    zLen = np.concatenate([[3000.0, 500.0], 10.0 ** rng.uniform(1.8, 2.8, nz - 2)])
    left = slice(ny + 1, ny + nz + 1); right = slice(ny + nz + 1, ny + 2 * nz + 1); bottom = slice(ny + 2 * nz + 1, 2 * (ny + nz))
    not_last = np.ones(ny * nz, bool); not_last[(nz - 1) * ny:] = False        # (B.6: the last layer's column is incomplete by construction)
    worst = {}
    for label in ("rough", "layered"):
        if label == "rough":
            sig2 = 10.0 ** rng.uniform(-3, -0.5, (nz, ny))
        else:
            sig2 = np.repeat(10.0 ** rng.uniform(-3, -0.5, (nz, 1)), ny, axis=1)
        sig2[:2, :] = 1e-6                                                      # air-like top layers
        sigma = sig2.reshape(-1)
        for freq in (10.0, 0.3, 0.01):
            dBC, bc = O.getBCDerivMatrix(freq, yLen, zLen, sigma, source)
            f = lambda s: O._getBoundaryMT2D(freq, yLen, zLen, s, mode)
            fwd = f(sigma)
            # the values: top 1, left / right the edge columns' 1-D solution; bottom the mean profile's (= the per-node ones on the layered model)
            sc = np.abs(fwd).max()
            assert np.abs(bc[:bottom.start] - fwd[:bottom.start]).max() < 1e-12 * sc
            if label == "layered":
                assert np.abs(bc[bottom] - fwd[bottom]).max() < 1e-12 * sc
            rows = [left, right] + ([bottom] if label == "layered" else [])
            scale = max(np.abs(dBC[r][:, not_last]).max() for r in rows)
            assert scale > 0 and np.abs(dBC[:ny + 1]).max() == 0.0                 # top side: constant
            for c in np.nonzero(not_last)[0]:
                fd = _richardson_vec(f, sigma, c, 1e-2 * sigma[c])
                for r, name in zip(rows, ("left", "right", "bottom")):
                    live = np.abs(fwd[r]) > 1e-10 * sc                           # (below the overflow cut-off the fields are zeroed, B.7)
                    if not live.any():
                        continue
                    err = np.abs(fd[r] - dBC[r, c])[live].max() / scale
                    worst[(label, name)] = max(worst.get((label, name), 0.0), err)
                    assert err < 1e-5, (label, freq, name, c, err)
            # structure: left block only on column 0, right block only on column ny-1, bottom row j on columns j-1, j with the width weights
            cols = np.arange(ny * nz) % ny
            assert np.abs(dBC[left][:, cols != 0]).max() == 0.0 and np.abs(dBC[right][:, cols != ny - 1]).max() == 0.0
            for j in range(1, ny):
                row = dBC[ny + 2 * nz + j]
                assert np.abs(row[(cols != j - 1) & (cols != j)]).max() == 0.0
                w1 = yLen[j - 1] / (yLen[j - 1] + yLen[j])
                a, b = row[cols == j - 1], row[cols == j]
                assert np.allclose(a * (1 - w1), b * w1, rtol=1e-13, atol=0)
    print("\n[boundary-derivative assembly, %s] worst |FD - dBC| / max|dBC| per block:" % mode, {k: float("%.1e" % v) for k, v in worst.items()})
    assert set(worst) == {("rough", "left"), ("rough", "right"), ("layered", "left"), ("layered", "right"), ("layered", "bottom")}


@pytest.mark.gpu
def test_hip_boundary_derivative_terms_against_differences_of_the_hip_forward():
    """(2) of the module docstring.  cfg2's mesh (50 x 25 cells + 7 air rows), a layered model with a 30-fold contrast, 0.1 /
    0.03 / 0.01 Hz (eps0 omega / sigma_air < 6e-4): B = g - g_PQ, the boundary-derivative kernels' share of the gradient, against
    D = FD(boundary recomputed) - FD(boundary frozen) on 22 cells of both edge columns, the padding and the core, every depth
    but the last row (App. B.6).  Measured: |B - D| <= 1.2e-8 max|B| (median 1.5e-9), with B itself 4.7 % of the gradient by
    maximum and up to twice the gradient's own entry in the cells compared (edge columns)."""
    from hmcmt2d_amd.lib import HipContext
    mesh = S.make_mesh(50, 25)
    ny, nzt = mesh.gridSize
    nair = len(mesh.airLayer)
    data = S.make_data_layout([0.1, 0.03, 0.01], np.arange(-3000.0, 3001.0, 600.0))
    n = len(data.rxID)
    rng = np.random.default_rng(8)
    obs = (0.004 + 0.002 * rng.standard_normal(n)) * np.where(data.dtID == 1, 1.0, -1.0) * (1 + 1j)
    mesh.sigma = np.concatenate([np.full(ny * nair, S.SIG_AIR), np.full(ny * (nzt - nair), 0.01)])
    inv = I.setupInverseDataModel(mesh, [S.SIG_AIR], 0.0, 0.0, obs, np.full(n, 2e-4))
    layers = np.log(0.01) + np.cumsum(0.5 * rng.standard_normal(nzt - nair))
    layers = np.clip(layers, np.log(0.01) - 1.7, np.log(0.01) + 1.7)
    m0 = np.repeat(layers, ny)
    assert len(m0) == len(inv.strModel)
    ctx = HipContext(mesh, data, inv, tol=1e-13, warm_start="previous")
    _, _, g = ctx.grad(m0)
    ctx.debug_flags(no_boundary_terms=True)
    _, _, pq = ctx.grad(m0)
    ctx.debug_flags()
    B = g - pq
    rows = [0, 3, 9, 16, 21, 23]                      # earth rows (of 25; the last one, 24, is left out)
    colsel = [0, ny - 1, 2, ny - 6, 20, 31]
    cells = np.array(sorted({r * ny + c for r in rows for c in colsel if (r + c) % 2 == 0 or c in (0, ny - 1)}))[:28]
    h = 0.02

    def quotients():
        out = np.zeros(len(cells))
        for j, c in enumerate(cells):
            D = []
            for hh in (h, 2 * h):
                mp, mm = m0.copy(), m0.copy(); mp[c] += hh; mm[c] -= hh
                D.append((ctx.forward(mp)[1] - ctx.forward(mm)[1]) / (2 * hh))
            out[j] = (4 * D[0] - D[1]) / 3
        return out

    fd_full = quotients()
    ctx.forward(m0)
    ctx.debug_flags(freeze_boundary=True)
    fd_frozen = quotients()
    ctx.debug_flags()
    assert ctx.stats()["status"] == 0
    ctx.close()
    D = fd_full - fd_frozen
    sc = np.abs(B).max()
    err = np.abs(B[cells] - D) / sc
    share = np.abs(B[cells]) / np.maximum(np.abs(g[cells]), 1e-3 * np.abs(g).max())
    print(f"\n[HIP boundary-derivative terms] {len(cells)} cells: |B - D| / max|B| max {err.max():.2e} median {np.median(err):.2e}; "
          f"|B| / |g| in these cells max {share.max():.2e} median {np.median(share):.2e}; max|B| / max|g| = {sc / np.abs(g).max():.2e}; "
          f"frozen-boundary FD vs P+Q terms max {(np.abs(fd_frozen - pq[cells]) / np.maximum(np.abs(g[cells]), 1e-3 * np.abs(g).max())).max():.2e}")
    assert sc > 1e-3 * np.abs(g).max()                 # (the terms are not negligible here: the comparison says something)
    assert err.max() < 1e-6
