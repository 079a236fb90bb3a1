"""The pin of the ADJOINT / GRADIENT half, as far as it can be pinned without a reference output (none exists: SURVEY
section 4).

The forward map m -> predData is pinned on the reference's own example data at the model those data were generated from
(tests/test_oracle_kat.py, first test).  At that same model the gradient of the data misfit is held, cell by cell on 256
cells of the mesh core, against Richardson-extrapolated central differences of that pinned forward map:

  (1) with the Dirichlet values of the four sides FROZEN, the difference quotient is the exact value of the P- and
      Q-terms of J^T v (compJacTMatVec.jl:235 / :306-307 and :209 / :280) -- adjoint source, adjoint solve, dA/dsigma
      contraction, the explicit sigma-dependence of the receiver functionals, the chain rule.  Asserted to 1e-6
      relative per cell (measured: 3e-9 oracle, see profiles/ for the GPU figure);
  (2) with everything recomputed it is the exact gradient.  The difference between (2) and the reference formula is
      the error of the reference's APPROXIMATE boundary-derivative terms (one mean profile for the
      bottom boundary, the last layer's half-space term dropped, boundary fields without the displacement term); it must
      stay within the size of those terms themselves, and their share of the gradient in these cells is reported:
      that share is all that rests on the restatement of `compJacTMatVec.jl:237-242,309-316` / `MT1DSensitivity.jl`
      without an independent check.

The oracle side runs on the CPU against committed difference quotients (tests/golden/make_fd_pin.py) plus two cells
recomputed live; the GPU side forms its OWN difference quotients with the HIP forward (no oracle in between) and also
compares them with the committed ones.
"""
import os
import numpy as np
import pytest

from hmcmt2d_amd import synthetic as S, invsetup as I
from hmcmt2d_amd.structs import HMCPrior
from tests.helpers import GOLDEN, dprism_generating_problem

FROZEN_TOL = 1e-6       # |fd_frozen - (P+Q terms)| / scale_c   per cell
FULL_TOL = 1e-5         # |fd_full - gradient| / scale_c, beyond the share of the boundary terms in that cell


def _problem():
    mesh, data, obs, err = dprism_generating_problem()
    inv = I.setupInverseDataModel(mesh, [S.SIG_AIR], 0.0, 0.0, obs, err)
    m0 = np.log(mesh.sigma[inv.activeIdx])
    return mesh, data, inv, m0


def _scale(g, cells):
    """per-cell denominator: the entry itself, floored at 1e-3 of the largest entry of the set (a relative error of a
    gradient entry that is itself ~0 says nothing)"""
    a = np.abs(g[cells])
    return np.maximum(a, 1e-3 * a.max())


def _report(tag, g, pq, fd_full, fd_frozen, cells):
    sc = _scale(g, cells)
    e_fro = np.abs(fd_frozen - pq[cells]) / sc
    e_full = np.abs(fd_full - g[cells]) / sc
    share = np.abs(g[cells] - pq[cells]) / sc
    print(f"\n[{tag}] {len(cells)} cells: frozen-boundary FD vs P+Q terms  max {e_fro.max():.2e} median {np.median(e_fro):.2e};  "
          f"full FD vs gradient  max {e_full.max():.2e} median {np.median(e_full):.2e};  "
          f"share of the boundary-derivative terms in these cells  max {share.max():.2e} median {np.median(share):.2e}")
    return e_fro, e_full, share


def test_oracle_gradient_against_richardson_differences_of_the_pinned_forward():
    from oracle import hmcmt_oracle as O
    fd = np.load(os.path.join(GOLDEN, "dprism_fd.npz"))
    cells = fd["cells"]
    mesh, data, inv, m0 = _problem()
    ny = mesh.gridSize[0]
    assert len(cells) >= 200 and (cells % ny).min() >= 7 and (cells % ny).max() <= ny - 8 and (cells // ny).max() < 41   # none in the padding
    O.setupTensorMesh2D(mesh)
    inv.strModel = m0.copy()
    keep = {}
    _, _, g = O.compDataGradient(mesh, data, inv, HMCPrior(), False, keep)
    pq = np.zeros(len(m0))
    for (md, f), t in keep["terms"].items():
        pq += np.real(t["PTv"] + t["QTv"]) + (np.real(t["BTvii2"]) if md == "TM" else 0.0)
    pq *= np.exp(m0)
    e_fro, e_full, share = _report("oracle", g, pq, fd["fd_full"], fd["fd_frozen"], cells)
    assert e_fro.max() < FROZEN_TOL
    assert np.all(e_full < FULL_TOL + 2.0 * share) and share.max() < 5e-3
    # two cells live (the committed quotients are what this code computes)
    from tests.golden.make_fd_pin import H
    bc0 = dict(keep["bc"])

    def phi(mm, frozen):
        s = inv.bgModel.copy(); s[inv.activeIdx] += np.exp(mm); mesh.sigma = s
        return O.compDataMisfit(O.MT2DFwdSolver(mesh, data, bc_fixed=bc0 if frozen else None)[0], inv)

    for j, frozen in ((17, True), (140, False)):
        c = cells[j]
        D = []
        for hh in (H, 2 * H):
            mp, mm = m0.copy(), m0.copy(); mp[c] += hh; mm[c] -= hh
            D.append((phi(mp, frozen) - phi(mm, frozen)) / (2 * hh))
        r = (4 * D[0] - D[1]) / 3
        ref = fd["fd_frozen" if frozen else "fd_full"][j]
        assert abs(r - ref) < 1e-7 * max(abs(ref), 1e-3 * np.abs(fd["fd_full"]).max())


@pytest.mark.gpu
def test_hip_gradient_against_richardson_differences_of_its_own_forward():
    """Oracle-free: the HIP path's gradient against difference quotients of the HIP path's own (reference-pinned,
    tests/test_gpu_parity_full.py::test_hip_forward_reproduces_...) forward map, at a tightened solver tolerance."""
    from hmcmt2d_amd.lib import HipContext
    fd = np.load(os.path.join(GOLDEN, "dprism_fd.npz"))
    cells = fd["cells"]
    mesh, data, inv, m0 = _problem()
    ctx = HipContext(mesh, data, inv, tol=1e-13, warm_start="previous")
    _, _, g = ctx.grad(m0)
    ctx.debug_flags(no_boundary_terms=True)
    _, _, pq = ctx.grad(m0)
    ctx.debug_flags()
    h = 0.02

    def quotients(frozen):
        out = np.zeros(len(cells))
        for j, c in enumerate(cells):
            D = []
            for hh in (h, 2 * h):
                mp, mm = m0.copy(), m0.copy(); mp[c] += hh; mm[c] -= hh
                D.append((ctx.forward(mp)[1] - ctx.forward(mm)[1]) / (2 * hh))
            out[j] = (4 * D[0] - D[1]) / 3
        return out

    fd_full = quotients(False)
    ctx.forward(m0)                                       # the boundary values of the base model ...
    ctx.debug_flags(freeze_boundary=True)                 # ... stay
    fd_frozen = quotients(True)
    ctx.debug_flags()
    assert ctx.stats()["status"] == 0
    ctx.close()
    e_fro, e_full, share = _report("HIP", g, pq, fd_full, fd_frozen, cells)
    assert e_fro.max() < 1e-5
    assert np.all(e_full < FULL_TOL + 2.0 * share) and share.max() < 5e-3
    # and the HIP difference quotients against the oracle's committed ones
    sc = _scale(g, cells)
    assert (np.abs(fd_full - fd["fd_full"]) / sc).max() < 1e-5 and (np.abs(fd_frozen - fd["fd_frozen"]) / sc).max() < 1e-5


@pytest.mark.gpu
def test_share_of_the_boundary_derivative_terms_on_the_headline_mesh(capsys):
    """How much of the HEADLINE gradient (cfg3: 200 x 100 cells, 16 frequencies, the rough state bench.py's chain starts
    from) rests on the restated boundary-derivative terms BTvii + BTvio (compJacTMatVec.jl:237-242, 309-316 with
    MT1DSensitivity.jl:94-243) -- the part of J^T v the difference-quotient pin above checks only to within its own size.
    Per cell: |g - g_without_boundary_terms| / max(|g|, 1e-3 max|g|); reported over the core, the padding columns and the
    deepest five rows (DESIGN section 2 quotes this line), and bounded loosely so that a change of regime is noticed."""
    import json
    from hmcmt2d_amd.lib import HipContext
    from tests.helpers import make_problem
    mesh, data, inv, m = make_problem("cfg3")
    ny = mesh.gridSize[0]
    assert len(m) % ny == 0
    rows = len(m) // ny
    ctx = HipContext(mesh, data, inv)
    _, _, g = ctx.grad(m)
    ctx.debug_flags(no_boundary_terms=True)
    _, _, pq = ctx.grad(m + 0.0)
    ctx.debug_flags()
    ctx.close()
    share = (np.abs(g - pq) / np.maximum(np.abs(g), 1e-3 * np.abs(g).max())).reshape(rows, ny)
    npy, npz = 7, 8                                        # synthetic.make_mesh defaults: padding columns per side, padding rows
    col = np.arange(ny)
    pad_c = (col < npy) | (col >= ny - npy)
    regions = {"core": share[:rows - npz][:, ~pad_c], "padding_columns": share[:, pad_c], "deepest_5_rows": share[rows - 5:, :]}
    out = {k: {"median": float(np.median(v)), "max": float(v.max())} for k, v in regions.items()}
    out["all"] = {"median": float(np.median(share)), "max": float(share.max()),
                  "l2_share": float(np.linalg.norm(g - pq) / np.linalg.norm(g))}
    with capsys.disabled():
        print("\n[cfg3, rough state] share of BTvii + BTvio per cell: " +
              "; ".join(f"{k} median {v['median']:.2e} max {v['max']:.2e}" for k, v in out.items()) +
              f"; ||g - g_PQ|| / ||g|| = {out['all']['l2_share']:.2e}")
    if os.path.isdir("gpurun_out"):
        with open("gpurun_out/boundary_share_cfg3.json", "w") as f:
            json.dump(out, f, indent=1)
    assert out["core"]["median"] < 0.05 and out["all"]["l2_share"] < 0.2
