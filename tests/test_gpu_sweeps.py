"""Two damped Jacobi sweeps on each side of the FDM stage (k_update_fused<2>, k_back_post<.,2>, k_spmv_fused<2>):
same answers as one sweep and as the oracle, fewer iterations on high-contrast models, a preconditioner that is still
linear and complex symmetric, and the per-solve choice between one and two sweeps."""
import numpy as np
import pytest

from hmcmt2d_amd.lib import HipContext
from tests.helpers import make_problem, oracle_eval, relmax, gerr_split

pytestmark = pytest.mark.gpu
LO, HI = np.log(1e-4), np.log(1.0)


def _rough(n, std, seed=2):
    return np.clip(np.log(0.01) + std * np.random.default_rng(seed).standard_normal(n), LO, HI)


def test_two_sweeps_give_the_one_sweep_answers_in_fewer_iterations(monkeypatch):
    """cfg2, ln(sigma) ~ N(ln 0.01, 1.0) clipped to the sampler's bounds (the regime of clamped burn-in trajectories): the
    forced two-sweep path agrees with the oracle and with the one-sweep path at the levels of the robustness tests and
    needs clearly fewer iterations (measured: 0.73 of the forward, 0.78 of the adjoint ones); hmcmt_stats.smoother_sweeps reports what ran."""
    mesh, data, inv, m = make_problem("cfg2")
    mm = _rough(m.size, 1.0)
    res = {}
    for sw in (1, 2):
        monkeypatch.setenv("HMCMT_SWEEPS", str(sw))
        ctx = HipContext(mesh, data, inv, verify=True)
        res[sw] = ctx.grad(mm) + (ctx.stats(),)
        ctx.close()
    (p1, f1, g1, s1), (p2, f2, g2, s2) = res[1], res[2]
    assert s1["smoother_sweeps"] == 11 and s2["smoother_sweeps"] == 22
    assert s1["status"] == 0 and s2["status"] == 0 and s2["true_res_max"] < 1e-9 and s2["fallback_solves"] == 0
    po, mo, go = oracle_eval(mesh, data, inv, mm)
    assert relmax(p2, po) < 1e-8 and abs(f2 - mo) / mo < 1e-8 and relmax(p2, p1) < 1e-8
    noise = (0.0, 0.0)                                    # the reference formula's own rounding noise on a model this rough
    for eps in (1e-14, -1e-14, 1e-13):
        _, _, gn = oracle_eval(mesh, data, inv, mm * (1 + eps))
        noise = tuple(max(a, b) for a, b in zip(noise, gerr_split(gn, go, inv, mesh)))
    shallow, deep = gerr_split(g2, go, inv, mesh)
    assert shallow < 1e-8 + 10 * noise[0] and deep < 5e-7 + 10 * noise[1], (shallow, deep, noise)
    assert s2["iters_fwd_sum"] <= 0.85 * s1["iters_fwd_sum"] and s2["iters_adj_sum"] <= 0.85 * s1["iters_adj_sum"], (s1, s2)
    assert s2["iters_fwd_max"] <= 0.8 * s1["iters_fwd_max"] and s2["iters_adj_max"] <= 0.8 * s1["iters_adj_max"], (s1, s2)


def test_two_sweep_preconditioner_is_linear_symmetric_and_close_to_its_fp64_form(monkeypatch):
    """What COCG needs of the preconditioner holds for the two-sweep one as well (x'My = y'Mx unconjugated, linearity),
    to the accuracy of its mixed-precision stages; and it is the operator the fp64 path applies (k_sweep_exp around the
    fp64 transforms) to per-cent accuracy."""
    monkeypatch.setenv("HMCMT_SWEEPS", "2")
    mesh, data, inv, m = make_problem("cfg3")
    ctx = HipContext(mesh, data, inv)
    ctx.forward(m)
    assert ctx.stats()["smoother_sweeps"] == 20
    shape = (ctx.S, ctx.NZP, ctx.NYP)
    rng = np.random.default_rng(1)

    def rand():
        v = np.zeros(shape, complex)
        v[:, 1:ctx.nz, 1:ctx.ny] = rng.standard_normal((ctx.S, ctx.nz - 1, ctx.ny - 1)) + 1j * rng.standard_normal((ctx.S, ctx.nz - 1, ctx.ny - 1))
        return v

    x, y = rand(), rand()
    Mx, My = ctx.debug_precond(x).reshape(shape), ctx.debug_precond(y).reshape(shape)
    a, b = np.sum(y * Mx, axis=(1, 2)), np.sum(x * My, axis=(1, 2))
    assert np.max(np.abs(a - b) / np.abs(a)) < 2e-3
    assert relmax(ctx.debug_precond(2.0 * x + (0.5 - 1j) * y).reshape(shape), 2.0 * Mx + (0.5 - 1j) * My) < 1e-4
    ctx.set_options(fdm_precision="fp64"); ctx.forward(m)
    M64 = ctx.debug_precond(x).reshape(shape)
    assert relmax(Mx, M64) < 2e-2
    a64 = np.sum(y * M64, axis=(1, 2)); b64 = np.sum(x * ctx.debug_precond(y).reshape(shape), axis=(1, 2))
    assert np.max(np.abs(a64 - b64) / np.abs(a64)) < 1e-9
    ctx.close()
    # ... and it is NOT the one-sweep operator
    monkeypatch.setenv("HMCMT_SWEEPS", "1")
    c1 = HipContext(mesh, data, inv)
    c1.forward(m)
    assert relmax(c1.debug_precond(x).reshape(shape), Mx) > 1e-3
    c1.close()


def test_sweeps_are_chosen_per_solve(monkeypatch):
    """Default (HMCMT_SWEEPS unset): a solve kind goes to two sweeps when its last solve needed more than HMCMT_SWEEPS_UP
    iterations (default 12 since the end of round 5; 30 -- rounds 2-4's default -- here, so that the mild model below sits
    between the thresholds), back to one below HMCMT_SWEEPS_DOWN (default 6; 12 here), and -- every 40th solve in two-sweep mode --
    runs one sweep once and keeps the cheaper of the two (iterations x 1.2 for two sweeps; x 1.15 / 1.10 in the persistent kernel).  Rough model: the first evaluation runs one sweep and is long, the next ones
    run two and are shorter; a homogeneous model (the FDM background exact: a handful of iterations) brings both kinds
    back to one sweep; a moderately rough model stays in two-sweep mode until the probe finds one sweep cheaper."""
    monkeypatch.delenv("HMCMT_SWEEPS", raising=False)
    monkeypatch.setenv("HMCMT_SWEEPS_DOWN", "12")
    monkeypatch.setenv("HMCMT_SWEEPS_UP", "30")
    mesh, data, inv, m = make_problem("cfg2")
    rough = _rough(m.size, 1.0)
    ctx = HipContext(mesh, data, inv, warm_start="cold")
    ctx.grad(rough); s0 = ctx.stats()
    assert s0["smoother_sweeps"] == 11 and s0["iters_fwd_max"] > 30 and s0["iters_adj_max"] > 30, s0
    ctx.grad(rough + 1e-3); s1 = ctx.stats()
    assert s1["smoother_sweeps"] == 22 and s1["iters_fwd_max"] <= 0.8 * s0["iters_fwd_max"], (s0, s1)
    smooth = np.full(m.size, np.log(0.01))
    ctx.grad(smooth); s2 = ctx.stats()
    assert s2["smoother_sweeps"] == 22 and s2["iters_fwd_max"] < 12
    ctx.grad(smooth + 1e-3); s3 = ctx.stats()
    assert s3["smoother_sweeps"] == 11, s3
    p_auto, f_auto, g_auto = ctx.grad(rough)                           # (one sweep again, then the rule flips it back)
    # the probe: a model of the bench's rough state (std 0.3: ~12 iterations with either smoother) in two-sweep mode
    mild = m.copy()
    ctx.grad(rough + 2e-3)
    assert ctx.stats()["smoother_sweeps"] == 22
    seen = []
    for j in range(45):
        ctx.grad(mild + 1e-4 * j)
        seen.append(ctx.stats()["smoother_sweeps"])
    assert seen[0] == 22 and seen[-1] == 11 and 22 in seen[:39], seen      # 40 solves with two sweeps, the probe, then one
    ctx.close()
    monkeypatch.setenv("HMCMT_SWEEPS", "1")
    c1 = HipContext(mesh, data, inv, warm_start="cold")
    p1, f1, g1 = c1.grad(rough)
    c1.close()
    assert relmax(p_auto, p1) < 1e-9


@pytest.mark.parametrize("persist", ["1", "0"])
@pytest.mark.parametrize("sweeps", ["1", "2"])
def test_wide_mesh_path_against_the_oracle(sweeps, persist, monkeypatch):
    """Meshes wider than 256 nodes: since round 5 the persistent kernel with two column parts per row block (272 padded columns =
    144 + 128, 8.5 K-groups of 32: the straddling one is shared by the parts' partial products); with HMCMT_PERSIST=0 -- and for
    second contexts, shared devices, meshes beyond its LDS -- the separate forward / back kernels of the launch-per-phase loop
    (k_transform_lp, k_thomas32, k_post; two sweeps: k_transform_lp<1> + k_post_w2).  A ragged 270-cell-wide problem, rough
    model, both smoothers, both paths against the oracle."""
    from tests.helpers import ragged_problem
    monkeypatch.setenv("HMCMT_SWEEPS", sweeps)
    monkeypatch.setenv("HMCMT_PERSIST", persist)
    mesh, data, inv, m = ragged_problem(270, 14, 2, 4, 4, 3)
    ny, nt = mesh.gridSize
    mm = np.clip(m + 0.8 * np.random.default_rng(3).standard_normal(m.size), LO, HI)
    ctx = HipContext(mesh, data, inv, verify=True)
    assert ctx.NYP > 256
    pred, misfit, grad = ctx.grad(mm)
    st, info = ctx.stats(), ctx.persist_info()
    assert info["column_parts"] == 2 and info["solves"] == (2 if persist == "1" else 0) and info["placement_fallbacks"] == 0, info
    assert st["status"] == 0 and st["true_res_max"] < 1e-9 and st["smoother_sweeps"] == 11 * int(sweeps), st
    po, mo, go = oracle_eval(mesh, data, inv, mm)
    noise = (0.0, 0.0)
    for eps in (1e-14, -1e-14, 1e-13):
        _, _, gn = oracle_eval(mesh, data, inv, mm * (1 + eps))
        noise = tuple(max(a, b) for a, b in zip(noise, gerr_split(gn, go, inv, mesh)))
    assert relmax(pred, po) < 1e-8 and abs(misfit - mo) / mo < 1e-8
    shallow, deep = gerr_split(grad, go, inv, mesh)
    assert shallow < 1e-8 + 10 * noise[0] and deep < 5e-6 + 10 * noise[1], (shallow, deep, noise)
    ctx.close()


def test_meshes_wider_than_the_fp64_transform_run_the_default_path():
    """include/hmcmt.h, hmcmt_create: the fp64 eigen-transform (fdm_precision = 1, and the restart of a stagnating mixed-precision
    solve) holds 28 column tiles = ny + 1 <= 448 nodes.  Round 4 refused every wider mesh at create; the reference takes any mesh
    (readEMModel2D.jl:11-154), and the default mixed-precision path needs that kernel only for its safety net -- so a wider mesh is
    refused only where fp64 is asked for.  A 500-cell-wide mesh against the oracle; NYP = 448, the widest the fp64
    transform holds, on the launch-per-phase path (outside the persistent kernel's LDS)."""
    from tests.helpers import ragged_problem
    from hmcmt2d_amd.lib import HmcmtError
    mesh, data, inv, m = ragged_problem(500, 12, 1, 4, 4, 3)
    with pytest.raises(HmcmtError) as e:
        HipContext(mesh, data, inv, fdm_precision="fp64")
    assert "EINVAL" in str(e.value) and "448" in str(e.value)
    ctx = HipContext(mesh, data, inv, verify=True)
    assert ctx.NYP == 512 and ctx.persist_info()["threads_half"] == 0
    p, f, g = ctx.grad(m)
    st = ctx.stats()
    with pytest.raises(HmcmtError):
        ctx.set_options(fdm_precision="fp64")
    ctx.close()
    assert st["status"] == 0 and st["true_res_max"] < 1e-9
    po, mo, go = oracle_eval(mesh, data, inv, m)
    assert relmax(p, po) < 1e-8 and abs(f - mo) / mo < 1e-8 and relmax(g, go) < 1e-6
    mesh, data, inv, m = ragged_problem(440, 12, 1, 4, 4, 3)          # NYP = 448
    ctx = HipContext(mesh, data, inv, verify=True)
    assert ctx.NYP == 448 and ctx.persist_info()["threads_half"] == 0
    p, f, g = ctx.grad(m)
    st = ctx.stats()
    ctx.close()
    assert st["status"] == 0 and st["true_res_max"] < 1e-9 and np.isfinite(g).all()
