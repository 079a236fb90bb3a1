"""The persistent solve kernel (hmcmt2d_amd/csrc/kernels_persist.h; one launch per solve, a system = G workgroups of one XCD):
its preconditioner against the launch-per-phase one, whole evaluations against the launch-per-phase loop and the oracle,
bitwise repeatability, the hand-back to the launch loop, and what it must leave behind for the host's fp64 restart.
The solves it replaces: MTFwdSolver/mt2DTE.jl:47-55, mt2DTM.jl:46-54, MTSensitivity/compJacTMatVec.jl:220-229, 291-300."""
import numpy as np
import pytest

from hmcmt2d_amd.lib import HipContext
from tests.helpers import make_problem, oracle_eval, relmax, ragged_problem, gerr_split

pytestmark = pytest.mark.gpu


def _ctx(monkeypatch, mesh, data, inv, persist, sweeps=None, **kw):
    monkeypatch.setenv("HMCMT_PERSIST", "1" if persist else "0")
    if sweeps is None:
        monkeypatch.delenv("HMCMT_SWEEPS", raising=False)
    else:
        monkeypatch.setenv("HMCMT_SWEEPS", str(sweeps))
    return HipContext(mesh, data, inv, **kw)


@pytest.mark.parametrize("sweeps", [1, 2])
def test_preconditioner_of_the_persistent_kernel_is_the_launch_per_phase_one(monkeypatch, sweeps):
    """z = P^-1 r through the kernel's first preconditioner application (hmcmt_debug_persist_precond) against the
    launch-per-phase kernels (hmcmt_debug_precond): the same operator up to the fp32 / bf16 rounding of its stages (the
    smoother works from a complex64 copy of r and forms its diagonal from the float couplings; measured 1e-6 .. 4e-6)."""
    mesh, data, inv, m = make_problem("cfg2")                 # 3 workgroups per system: halo rows, a short last workgroup
    ctx = _ctx(monkeypatch, mesh, data, inv, persist=False, sweeps=sweeps)
    ctx.forward(m)
    info = ctx.persist_info()
    assert info["threads_half"] == 64 and info["workgroups_per_system"] == 3
    shape = (ctx.S, ctx.NZP, ctx.NYP)
    rng = np.random.default_rng(3)
    x = np.zeros(shape, complex)
    x[:, 1:ctx.nz, 1:ctx.ny] = rng.standard_normal((ctx.S, ctx.nz - 1, ctx.ny - 1)) + 1j * rng.standard_normal((ctx.S, ctx.nz - 1, ctx.ny - 1))
    for v in (x, np.cumsum(np.cumsum(x, axis=1), axis=2) / 50.0 * (np.abs(x) > 0)):
        z0 = ctx.debug_precond(v).reshape(shape)
        z1 = ctx.debug_persist_precond(v, sweeps).reshape(shape)
        assert max(relmax(z1[s], z0[s]) for s in range(ctx.S)) < 2e-5
    ctx.close()


@pytest.mark.parametrize("name,sweeps", [("cfg2", 1), ("cfg2", 2), ("tiny", 2)])
def test_persistent_solve_equals_the_launch_per_phase_loop_and_the_oracle(monkeypatch, name, sweeps):
    """One evaluation with the true-residual check on, persistent kernel against launch-per-phase loop: the same iteration
    counts (+-1 per solve kind in the maximum), results to the solver tolerance, the oracle's values at the parity levels
    of tests/test_gpu_parity.py; the statistics say which path ran."""
    mesh, data, inv, m = make_problem(name)
    res = {}
    for persist in (False, True):
        ctx = _ctx(monkeypatch, mesh, data, inv, persist, sweeps, verify=True)
        res[persist] = ctx.grad(m) + (ctx.stats(), ctx.persist_info())
        ctx.close()
    (p0, f0, g0, s0, i0), (p1, f1, g1, s1, i1) = res[False], res[True]
    assert i0["solves"] == 0 and i1["solves"] == 2 and i1["placement_fallbacks"] == 0 and i1["enabled"] == 1
    assert s1["status"] == 0 and s1["fallback_solves"] == 0 and s1["true_res_max"] < 1e-9
    assert abs(s1["iters_fwd_max"] - s0["iters_fwd_max"]) <= 1 and abs(s1["iters_adj_max"] - s0["iters_adj_max"]) <= 1
    assert abs(s1["iters_fwd_sum"] - s0["iters_fwd_sum"]) <= 0.05 * s0["iters_fwd_sum"] + 2
    assert relmax(p1, p0) < 1e-9 and abs(f1 - f0) / abs(f0) < 1e-9 and relmax(g1, g0) < 1e-8
    po, mo, go = oracle_eval(mesh, data, inv, m)
    assert relmax(p1, po) < 1e-9 and abs(f1 - mo) / mo < 1e-9
    shallow, deep = gerr_split(g1, go, inv, mesh)
    assert shallow < 1e-8 and deep < 1e-7


def test_persistent_solve_on_a_ragged_mesh_and_masked_data(monkeypatch):
    """Sizes that are multiples of nothing (one workgroup with two own rows left over, pad columns, a polarisation with
    masked data, a fixed cell): against the oracle."""
    mesh, data, inv, m = ragged_problem(37, 22, 3, 5, 4, 3)
    ctx = _ctx(monkeypatch, mesh, data, inv, True, verify=True)
    p, f, g = ctx.grad(m)
    st, info = ctx.stats(), ctx.persist_info()
    ctx.close()
    assert info["solves"] == 2 and st["status"] == 0 and st["true_res_max"] < 1e-9
    po, mo, go = oracle_eval(mesh, data, inv, m)
    assert relmax(p, po) < 1e-9 and abs(f - mo) / mo < 1e-9 and relmax(g, go) < 1e-7


def test_persistent_solve_is_bitwise_repeatable(monkeypatch):
    """Every reduction is a fixed-order sum over per-workgroup partials and the repeated halo arithmetic is written with
    explicit FMAs: two cold evaluations of one model give the same bits -- and so do two contexts."""
    mesh, data, inv, m = make_problem("cfg2")
    outs = []
    for _ in range(2):
        ctx = _ctx(monkeypatch, mesh, data, inv, True, warm_start=False)
        a = ctx.grad(m)
        b = ctx.grad(m + 0.0)
        assert ctx.persist_info()["solves"] >= 2
        ctx.close()
        assert np.array_equal(a[0], b[0]) and a[1] == b[1] and np.array_equal(a[2], b[2])
        outs.append(a)
    assert np.array_equal(outs[0][0], outs[1][0]) and outs[0][1] == outs[1][1] and np.array_equal(outs[0][2], outs[1][2])


def test_two_contexts_use_the_launch_per_phase_loop(monkeypatch):
    """Two persistent kernels on one device could each hold CUs the other's missing workgroups need (they spin at their
    barriers): with a second context alive in the process both run the launch-per-phase loop -- same answers."""
    mesh, data, inv, m = make_problem("cfg2")
    c1 = _ctx(monkeypatch, mesh, data, inv, True)
    ref = c1.grad(m)
    assert c1.persist_info()["solves"] == 2
    c2 = _ctx(monkeypatch, mesh, data, inv, True)
    a, b = c1.grad(m + 1e-3), c2.grad(m + 1e-3)
    assert c1.persist_info()["solves"] == 2 and c2.persist_info()["solves"] == 0
    assert relmax(a[0], b[0]) < 1e-9 and relmax(a[2], b[2]) < 1e-8
    c2.close()
    c1.grad(m - 1e-3)                                      # alone again: the kernel is back
    assert c1.persist_info()["solves"] == 4
    c1.close()
    assert relmax(ref[0], a[0]) < 0.1                      # (sanity: the models are close)


def test_stagnating_persistent_solve_hands_over_to_the_fp64_restart(monkeypatch):
    """HMCMT_STALL_IT = 1 makes the stagnation watch fire at once: the kernel stops, leaves x, r, the iteration counts and
    the active flags behind, and the host's classic loop finishes the systems with the fp64 preconditioner."""
    monkeypatch.setenv("HMCMT_STALL_IT", "1")
    mesh, data, inv, m = make_problem("cfg2")
    ctx = _ctx(monkeypatch, mesh, data, inv, True, verify=True)
    p, f, g = ctx.grad(m)
    st = ctx.stats()
    ctx.close()
    assert st["fallback_solves"] >= 1 and st["status"] == 0 and st["true_res_max"] < 1e-9
    po, mo, go = oracle_eval(mesh, data, inv, m)
    assert relmax(p, po) < 1e-9 and relmax(g, go) < 1e-7


def test_iteration_cap_is_reported_by_the_persistent_kernel(monkeypatch):
    mesh, data, inv, m = make_problem("cfg2")
    ctx = _ctx(monkeypatch, mesh, data, inv, True, maxit=3)
    with pytest.raises(Exception) as e:
        ctx.grad(m)
    assert "ENOCONV" in str(e.value)
    ctx.set_options(maxit=2000)
    p, f, g = ctx.grad(m)                                  # ... and the context recovers (cold start after a failure)
    assert ctx.stats()["status"] == 0
    ctx.close()


def test_more_systems_than_slots_run_in_rounds(monkeypatch):
    """A system group (G workgroups of one XCD) takes the next system when its own is done: 44 systems on a mesh whose 100 rows
    need 8 workgroups each (4 groups per XCD, 32 at a time) -- every system solved, to the launch-per-phase loop's results."""
    from hmcmt2d_amd import synthetic as S, invsetup as I
    from tests.helpers import start_sigma
    mesh = S.make_mesh(60, 93)
    data = S.make_data_layout(S.log_freqs(22), np.arange(-2000.0, 2001.0, 500.0))
    n = len(data.rxID)
    obs = np.full(n, 0.02 + 0.02j) * np.where(data.dtID == 1, 1.0, -1.0)
    mesh.sigma = start_sigma(mesh)
    inv = I.setupInverseDataModel(mesh, [S.SIG_AIR], 0.0, 0.0, obs, np.full(n, 1e-3))
    m = S.rough_state(len(inv.strModel))
    res = {}
    for persist in (False, True):
        ctx = _ctx(monkeypatch, mesh, data, inv, persist, 2, verify=True)
        res[persist] = ctx.grad(m) + (ctx.stats(), ctx.persist_info())
        if persist:
            a = ctx.grad(m + 0.01)                                        # (a warm-started one as well)
            assert ctx.stats()["status"] == 0
        ctx.close()
    (p0, f0, g0, s0, i0), (p1, f1, g1, s1, i1) = res[False], res[True]
    assert s1["nsystems"] == 44 and i1["workgroups_per_system"] == 8 and i1["slots_per_xcd"] == 4 and i1["solves"] == 2
    assert s1["status"] == 0 and s1["true_res_max"] < 1e-9 and s1["fallback_solves"] == 0
    assert abs(s1["iters_fwd_sum"] - s0["iters_fwd_sum"]) <= 0.05 * s0["iters_fwd_sum"] + 2
    assert relmax(p1, p0) < 1e-9 and abs(f1 - f0) / abs(f0) < 1e-9 and relmax(g1, g0) < 1e-8


def test_a_second_process_on_the_device_runs_the_launch_per_phase_loop(monkeypatch, tmp_path):
    """Across processes the persistent kernel is guarded by an advisory lock per device (flock on
    $HMCMT_LOCK_DIR/hmcmt_persist_<PCI bus id>.lock, held by a process while it has a context on the device): a second process
    does not get it, runs the launch-per-phase loop -- same answers -- and this process keeps the kernel."""
    import subprocess, sys, json, os
    monkeypatch.setenv("HMCMT_LOCK_DIR", str(tmp_path))
    mesh, data, inv, m = make_problem("cfg2")
    ctx = _ctx(monkeypatch, mesh, data, inv, True)
    p, f, g = ctx.grad(m)
    assert ctx.persist_info()["usable_now"] == 1 and ctx.persist_info()["solves"] == 2 and len(os.listdir(tmp_path)) == 1
    child = ("import json, numpy as np\n"
             "from hmcmt2d_amd.lib import HipContext\n"
             "from tests.helpers import make_problem\n"
             "mesh, data, inv, m = make_problem('cfg2')\n"
             "c = HipContext(mesh, data, inv)\n"
             "p, f, g = c.grad(m)\n"
             "print(json.dumps({'info': c.persist_info(), 'misfit': f, 'g0': float(g[0])}))\n"
             "c.close()\n")
    env = dict(os.environ, HMCMT_PERSIST="1", HMCMT_LOCK_DIR=str(tmp_path))
    r = subprocess.run([sys.executable, "-c", child], capture_output=True, text=True, timeout=300, env=env,
                       cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert out["info"]["usable_now"] == 0 and out["info"]["solves"] == 0 and out["info"]["enabled"] == 1
    assert abs(out["misfit"] - f) / abs(f) < 1e-9 and abs(out["g0"] - g[0]) <= 1e-8 * np.abs(g).max()
    ctx.grad(m - 1e-3)
    assert ctx.persist_info()["solves"] == 4
    ctx.close()
    # ... and with this process's context gone the next process gets the lock
    r = subprocess.run([sys.executable, "-c", child], capture_output=True, text=True, timeout=300, env=env,
                       cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert r.returncode == 0 and json.loads(r.stdout.strip().splitlines()[-1])["info"]["solves"] == 2


def test_followers_queued_behind_the_solve_change_nothing(monkeypatch):
    """The receiver functionals / adjoint sources and the gradient's tail are queued behind a persistent solve before the host
    knows its outcome, gated on a device word (HMCMT_SPEC): warm-started evaluations with and without give the same
    bits; and when the solve does NOT end clean -- HMCMT_STALL_IT = 1 makes every solve hand over to the fp64 restart -- the
    gated kernels do nothing and the host queues them again: results to the oracle's."""
    mesh, data, inv, m = make_problem("cfg2")
    outs = {}
    for spec in ("0", "1"):
        monkeypatch.setenv("HMCMT_SPEC", spec)
        ctx = _ctx(monkeypatch, mesh, data, inv, True)
        outs[spec] = [ctx.grad(m + 0.01 * i) for i in range(5)]
        assert ctx.persist_info()["solves"] == 10
        ctx.close()
    for a, b in zip(outs["0"], outs["1"]):
        assert np.array_equal(a[0], b[0]) and a[1] == b[1] and np.array_equal(a[2], b[2])
    monkeypatch.setenv("HMCMT_SPEC", "1")
    monkeypatch.setenv("HMCMT_STALL_IT", "1")
    ctx = _ctx(monkeypatch, mesh, data, inv, True, 1)      # (one sweep per side: with two -- the default behind a cold solve of more than
                                                           #  HMCMT_SWEEPS_UP iterations -- the estimate drops tenfold every other iteration and nothing stalls)
    ctx.grad(m)                                            # (cold: no followers queued early)
    p, f, g = ctx.grad(m + 0.02)                           # warm: queued early, gate closed by the stalled solves
    st = ctx.stats()
    ctx.close()
    assert st["fallback_solves"] >= 1 and st["status"] == 0
    po, mo, go = oracle_eval(mesh, data, inv, m + 0.02)
    assert relmax(p, po) < 1e-9 and abs(f - mo) / mo < 1e-9 and relmax(g, go) < 1e-7


def test_tall_mesh_runs_the_persistent_kernel_with_16_mode_slabs(monkeypatch):
    """200 x 150 cells (+7 air rows): the 32-mode slab of the tridiagonal solves (2 x 2 x 95 rows x 32 modes x 8 B) no longer fits
    the LDS beside the coefficient planes, the kernel is instantiated with 16-mode slabs (13 slabs on 12 workgroups: one workgroup
    solves two) -- against the launch-per-phase loop, with the true-residual check."""
    from hmcmt2d_amd import synthetic as S, invsetup as I
    from tests.helpers import start_sigma
    mesh = S.make_mesh(200, 150)
    data = S.make_data_layout(S.log_freqs(4), np.arange(-8000.0, 8001.0, 2000.0))
    n = len(data.rxID)
    obs = np.full(n, 0.02 + 0.02j) * np.where(data.dtID == 1, 1.0, -1.0)
    mesh.sigma = start_sigma(mesh)
    inv = I.setupInverseDataModel(mesh, [S.SIG_AIR], 0.0, 0.0, obs, np.full(n, 1e-3))
    m = S.rough_state(len(inv.strModel))
    res = {}
    for persist in (False, True):
        ctx = _ctx(monkeypatch, mesh, data, inv, persist, 2, verify=True)
        res[persist] = ctx.grad(m) + (ctx.stats(), ctx.persist_info())
        ctx.close()
    (p0, f0, g0, s0, i0), (p1, f1, g1, s1, i1) = res[False], res[True]
    assert i1["slab_modes"] == 16 and i1["threads_half"] == 256 and i1["workgroups_per_system"] == 12 and i1["solves"] == 2
    assert s1["status"] == 0 and s1["true_res_max"] < 1e-9 and s1["fallback_solves"] == 0
    assert abs(s1["iters_fwd_max"] - s0["iters_fwd_max"]) <= 1 and abs(s1["iters_adj_max"] - s0["iters_adj_max"]) <= 1
    assert relmax(p1, p0) < 1e-9 and abs(f1 - f0) / abs(f0) < 1e-9 and relmax(g1, g0) < 1e-8


# ---- round 5: column parts (two workgroups per row block: meshes wider than one tile), robustness of the fallbacks ----

@pytest.mark.parametrize("sweeps", [1, 2])
def test_preconditioner_with_column_parts_is_the_launch_per_phase_one(monkeypatch, sweeps):
    """HMCMT_PERSIST_CS=2 forces the two-part kernel onto a mesh one tile would hold (cfg2: 64 padded columns = 32 + 32, a system
    = 3 row blocks x 2 parts): the forward transform as the sum of two partial products, the halo columns, the back transform of
    all modes per part -- the same operator as the launch-per-phase kernels' (and as the one-part kernel's) to rounding."""
    monkeypatch.setenv("HMCMT_PERSIST_CS", "2")
    mesh, data, inv, m = make_problem("cfg2")
    ctx = _ctx(monkeypatch, mesh, data, inv, persist=False, sweeps=sweeps)
    ctx.forward(m)
    info = ctx.persist_info()
    assert info["column_parts"] == 2 and info["workgroups_per_system"] == 6 and info["slab_modes"] == 16
    shape = (ctx.S, ctx.NZP, ctx.NYP)
    rng = np.random.default_rng(3)
    x = np.zeros(shape, complex)
    x[:, 1:ctx.nz, 1:ctx.ny] = rng.standard_normal((ctx.S, ctx.nz - 1, ctx.ny - 1)) + 1j * rng.standard_normal((ctx.S, ctx.nz - 1, ctx.ny - 1))
    for v in (x, np.cumsum(np.cumsum(x, axis=1), axis=2) / 50.0 * (np.abs(x) > 0)):
        z0 = ctx.debug_precond(v).reshape(shape)
        z1 = ctx.debug_persist_precond(v, sweeps).reshape(shape)
        assert max(relmax(z1[s], z0[s]) for s in range(ctx.S)) < 2e-5
    ctx.close()


@pytest.mark.parametrize("sweeps", [1, 2])
def test_solve_with_column_parts_equals_the_launch_per_phase_loop_and_the_oracle(monkeypatch, sweeps):
    monkeypatch.setenv("HMCMT_PERSIST_CS", "2")
    mesh, data, inv, m = make_problem("cfg2")
    res = {}
    for persist in (False, True):
        ctx = _ctx(monkeypatch, mesh, data, inv, persist, sweeps, verify=True)
        res[persist] = ctx.grad(m) + (ctx.stats(), ctx.persist_info())
        if persist:
            a = ctx.grad(m)
            b = ctx.grad(m + 0.0)                  # bitwise repeatable, column parts as well (fixed-order sums, bit-identical halo copies)
            assert np.array_equal(a[0], b[0]) and a[1] == b[1] and np.array_equal(a[2], b[2])
        ctx.close()
    (p0, f0, g0, s0, i0), (p1, f1, g1, s1, i1) = res[False], res[True]
    assert i1["column_parts"] == 2 and i1["solves"] == 2 and i1["placement_fallbacks"] == 0 and i1["timeouts"] == 0
    assert s1["status"] == 0 and s1["fallback_solves"] == 0 and s1["true_res_max"] < 1e-9
    assert abs(s1["iters_fwd_max"] - s0["iters_fwd_max"]) <= 1 and abs(s1["iters_adj_max"] - s0["iters_adj_max"]) <= 1
    assert relmax(p1, p0) < 1e-9 and abs(f1 - f0) / abs(f0) < 1e-9 and relmax(g1, g0) < 1e-8
    po, mo, go = oracle_eval(mesh, data, inv, m)
    assert relmax(p1, po) < 1e-9 and abs(f1 - mo) / mo < 1e-9
    shallow, deep = gerr_split(g1, go, inv, mesh)
    assert shallow < 1e-8 and deep < 1e-7


def test_wide_ragged_mesh_runs_the_column_parts_against_the_launch_loop_and_the_oracle(monkeypatch):
    """301 x 19 cells (+3 air rows), masked data, a fixed cell: wider than one tile, so the kernel runs with two column parts by
    its own choice (304 padded columns = 160 + 144; the K-group of 32 columns that straddles the split is shared by the parts'
    partial products; 2 row blocks x 2 parts per system) -- persistent kernel vs launch-per-phase loop vs the oracle."""
    mesh, data, inv, m = ragged_problem(301, 19, 2, 6, 4, 3)
    res = {}
    for persist in (False, True):
        ctx = _ctx(monkeypatch, mesh, data, inv, persist, 2, verify=True)
        res[persist] = ctx.grad(m) + (ctx.stats(), ctx.persist_info())
        ctx.close()
    (p0, f0, g0, s0, i0), (p1, f1, g1, s1, i1) = res[False], res[True]
    assert i1["column_parts"] == 2 and i1["solves"] == 2 and i1["workgroups_per_system"] == 4 and i0["solves"] == 0
    assert s1["status"] == 0 and s1["true_res_max"] < 1e-9 and s1["fallback_solves"] == 0
    assert abs(s1["iters_fwd_max"] - s0["iters_fwd_max"]) <= 1 and abs(s1["iters_adj_max"] - s0["iters_adj_max"]) <= 1
    assert relmax(p1, p0) < 1e-9 and abs(f1 - f0) / abs(f0) < 1e-9 and relmax(g1, g0) < 1e-8
    po, mo, go = oracle_eval(mesh, data, inv, m)
    assert relmax(p1, po) < 1e-9 and abs(f1 - mo) / mo < 1e-9 and relmax(g1, go) < 1e-7


def test_a_misplaced_group_leaves_the_other_groups_results_intact(monkeypatch):
    """hmcmt_debug_flags bit 2: the first system group of the next launch fails its placement check (as if its workgroups were not
    on one XCD).  Its systems stay untouched and active; every OTHER group finishes its systems (a running group gives up only on a
    timed-out wait, not on another group's misplacement -- ADVICE r4: it used to leave with x advanced and r in registers); the
    host's launch-per-phase loop takes what is left.  Results to the oracle's, true residual checked."""
    mesh, data, inv, m = make_problem("cfg2")
    ctx = _ctx(monkeypatch, mesh, data, inv, True, 2, verify=True)
    ctx.debug_flags(fail_placement=True)
    p, f, g = ctx.grad(m)
    st, info = ctx.stats(), ctx.persist_info()
    assert info["placement_fallbacks"] == 1 and info["enabled"] == 0 and st["status"] == 0 and st["true_res_max"] < 1e-9
    po, mo, go = oracle_eval(mesh, data, inv, m)
    assert relmax(p, po) < 1e-9 and abs(f - mo) / mo < 1e-9 and relmax(g, go) < 1e-7
    p2, f2, g2 = ctx.grad(m + 0.01)                        # ... and the context goes on with the launch-per-phase loop
    assert ctx.stats()["status"] == 0 and ctx.persist_info()["solves"] == 1
    ctx.close()


@pytest.mark.parametrize("strips", [2, 4])
def test_every_group_misplaced_is_the_all_fallback_regime_end_to_end(monkeypatch, capfd, strips):
    """hmcmt_debug_flags bit 3: EVERY system group of the next launch fails its placement check -- what a device in another
    partition mode, or a driver that dispatches workgroups to the XCDs in another order than b -> XCD b mod 8
    (kernels_persist.h:26-28), would do to every launch.  No system is touched by the kernel; the whole
    evaluation runs the launch-per-phase loop; the context says so ONCE on stderr, stays off for good (why_off = 1: no backoff
    re-enables it) and keeps giving the oracle's numbers."""
    monkeypatch.setenv("HMCMT_PERSIST_STRIPS", str(strips))
    mesh, data, inv, m = make_problem("cfg2")
    ctx = _ctx(monkeypatch, mesh, data, inv, True, 2, verify=True)
    assert ctx.persist_info()["strips"] == strips
    ctx.debug_flags(fail_placement="all")
    capfd.readouterr()
    p, f, g = ctx.grad(m)
    err = capfd.readouterr().err
    st, info = ctx.stats(), ctx.persist_info()
    assert "not dispatched to one XCD" in err and err.count("libhmcmt_hip") == 1
    assert info["placement_fallbacks"] == 1 and info["enabled"] == 0 and info["why_off"] == 1 and info["solves"] == 1
    assert st["status"] == 0 and st["true_res_max"] < 1e-9
    po, mo, go = oracle_eval(mesh, data, inv, m)
    assert relmax(p, po) < 1e-9 and abs(f - mo) / mo < 1e-9 and relmax(g, go) < 1e-7
    for k in range(3):                                     # ... it stays on the loop, silently, with the same answers
        p2, f2, g2 = ctx.grad(m + 0.01 * (k + 1))
    assert capfd.readouterr().err == "" and ctx.persist_info()["solves"] == 1 and ctx.persist_info()["enabled"] == 0 and ctx.stats()["status"] == 0
    po, mo, go = oracle_eval(mesh, data, inv, m + 0.03)
    assert relmax(p2, po) < 1e-9 and relmax(g2, go) < 1e-7
    ctx.close()


@pytest.mark.parametrize("which", [True, "all"])
@pytest.mark.parametrize("start", ["1", "0"])
def test_misplaced_groups_when_the_kernel_was_to_start_the_solve(monkeypatch, which, start):
    """Round 6: the initial residual and the solve's bookkeeping are formed INSIDE the persistent kernel (PsLaunch::resid / begin;
    not under options.verify, which the other placement tests use).  A group that fails its placement check -- one, or every one --
    then leaves systems that have no residual yet: it marks them "to be solved", the host forms their residual (k_resid0 on the active
    systems only: the other groups' results stay) and its launch-per-phase loop takes them.  Cold and warm-started evaluations, the
    forward and the adjoint solve (right-hand side on the receiver layer's rows), against HMCMT_PS_START=0 and the oracle."""
    monkeypatch.setenv("HMCMT_PS_START", start)
    mesh, data, inv, m = make_problem("cfg2")
    ctx = _ctx(monkeypatch, mesh, data, inv, True, 2)
    ctx.grad(m)                                            # a history for the warm start
    ctx.debug_flags(fail_placement=which)
    p, f, g = ctx.grad(m + 0.01)
    info = ctx.persist_info()
    assert info["placement_fallbacks"] == 1 and info["enabled"] == 0 and info["why_off"] == 1 and ctx.stats()["status"] == 0
    po, mo, go = oracle_eval(mesh, data, inv, m + 0.01)
    assert relmax(p, po) < 1e-9 and abs(f - mo) / mo < 1e-9 and relmax(g, go) < 1e-7
    p2, f2, g2 = ctx.grad(m + 0.02)                        # ... and on with the launch-per-phase loop
    po, mo, go = oracle_eval(mesh, data, inv, m + 0.02)
    assert relmax(p2, po) < 1e-9 and relmax(g2, go) < 1e-7 and ctx.persist_info()["solves"] == 3
    ctx.close()
    ctx = _ctx(monkeypatch, mesh, data, inv, True, 2)      # the FIRST (cold) evaluation of a context: cold adjoint start, sparse right-hand side
    ctx.debug_flags(fail_placement=which)
    p, f, g = ctx.grad(m)
    po, mo, go = oracle_eval(mesh, data, inv, m)
    assert relmax(p, po) < 1e-9 and relmax(g, go) < 1e-7 and ctx.persist_info()["placement_fallbacks"] == 1
    ctx.close()


@pytest.mark.parametrize("which", [True, "all"])
def test_jacobi_diagonals_left_out_for_the_persistent_kernel_are_made_up_on_the_way_out(monkeypatch, which):
    """Round 6: an evaluation whose solves are to run in the persistent kernel has k_coef_all write the two polarisations'
    coefficient arrays only -- that kernel forms the Jacobi diagonal itself; the 2 x nFreq per-system diagonals (Solver::dinv) are
    stores nobody reads.  When a solve leaves the kernel after all (here: misplaced groups), k_dinv makes up for them before the
    launch-per-phase loop's kernels read them (ensure_dinv): iteration counts and results BITWISE equal to HMCMT_LAZY_DINV=0,
    where every evaluation writes them (stale diagonals -- the previous model's, or none -- would only cost iterations, which
    is why the counts are compared)."""
    mesh, data, inv, m = make_problem("cfg2")
    out = {}
    for lazy in ("1", "0"):
        monkeypatch.setenv("HMCMT_LAZY_DINV", lazy)
        ctx = _ctx(monkeypatch, mesh, data, inv, True, 2)
        ctx.grad(m)
        ctx.debug_flags(fail_placement=which)
        a = ctx.grad(m + 0.5)                              # (far from the first model: its diagonals would be poor ones)
        ia = ctx.iters().copy()
        b = ctx.grad(m + 0.3)                              # ... and on with the launch-per-phase loop
        out[lazy] = (a, ia, b, ctx.iters().copy(), ctx.persist_info())
        assert ctx.stats()["status"] == 0
        ctx.close()
    for k in (0, 2):
        for x, y in zip(out["1"][k], out["0"][k]):
            assert np.array_equal(np.asarray(x), np.asarray(y))
    assert np.array_equal(out["1"][1], out["0"][1]) and np.array_equal(out["1"][3], out["0"][3])
    assert out["1"][4]["placement_fallbacks"] == 1 and out["1"][4]["enabled"] == 0
    po, mo, go = oracle_eval(mesh, data, inv, m + 0.3)
    assert relmax(out["1"][2][0], po) < 1e-9 and relmax(out["1"][2][2], go) < 1e-6      # (a rough model without options.verify's tightened solves)


@pytest.mark.parametrize("name,sweeps", [("cfg2", 1), ("cfg2", 2), ("tiny", 2), ("cfg1", 2)])
def test_four_strip_kernel_equals_the_two_half_kernel_and_the_oracle(monkeypatch, name, sweeps):
    """k_cocg_persist4 (kernels_persist4.h: four strips of six tile rows per column, 4 x threads_half threads, 128 VGPRs, four
    waves per SIMD; opt-in, HMCMT_PERSIST_STRIPS=4) against k_cocg_persist on the same problems: the same iteration counts
    (+-1), results to the solver tolerance, the oracle's values, bitwise repeatable.  (The headline mesh: tests/test_gpu_parity_full.py.)"""
    mesh, data, inv, m = make_problem(name)
    res = {}
    for strips in (2, 4):
        monkeypatch.setenv("HMCMT_PERSIST_STRIPS", str(strips))
        ctx = _ctx(monkeypatch, mesh, data, inv, True, sweeps, verify=True)
        res[strips] = ctx.grad(m) + (ctx.stats(), ctx.persist_info())
        if strips == 4:
            again = ctx.grad(m + 0.0)
            assert np.array_equal(again[0], res[4][0]) and np.array_equal(again[2], res[4][2])      # bitwise repeatable
        ctx.close()
    (p0, f0, g0, s0, i0), (p1, f1, g1, s1, i1) = res[2], res[4]
    assert i0["strips"] == 2 and i1["strips"] == 4 and i1["solves"] >= 2 and i1["placement_fallbacks"] == 0 and i1["timeouts"] == 0
    assert i1["threads_half"] == i0["threads_half"] and i1["workgroups_per_system"] == i0["workgroups_per_system"]
    assert s1["status"] == 0 and s1["fallback_solves"] == 0 and s1["true_res_max"] < 1e-9
    assert abs(s1["iters_fwd_max"] - s0["iters_fwd_max"]) <= 1 and abs(s1["iters_adj_max"] - s0["iters_adj_max"]) <= 1
    assert relmax(p1, p0) < 1e-9 and abs(f1 - f0) / abs(f0) < 1e-9 and relmax(g1, g0) < 1e-8
    po, mo, go = oracle_eval(mesh, data, inv, m)
    assert relmax(p1, po) < 1e-9 and abs(f1 - mo) / mo < 1e-9 and relmax(g1, go) < 1e-7


def test_a_timed_out_wait_falls_back_to_the_launch_per_phase_loop(monkeypatch):
    """A foreign kernel whose backlog of workgroups keeps every CU busy (hmcmt_debug_hog: 1500 workgroups with a CU's whole LDS
    each, 100 ms apiece: six rounds over the chip) competes with the persistent kernel's workgroups for the CUs that come free:
    its grid is partly resident, the placed workgroups wait for the others, the bounded waits (HMCMT_PS_SPIN shortens them to
    milliseconds here) give up, the kernel drains.  The evaluation used to end with HMCMT_EHIP -- a 10 000-sample chain died on
    it; now the context leaves the persistent kernel and the evaluation is redone with the launch-per-phase loop.  (Which of the
    bounded waits fails -- or the placement check -- depends on the dispatcher; up to three tries to see one.)"""
    import time
    monkeypatch.setenv("HMCMT_PS_SPIN", "2048")
    mesh, data, inv, m = make_problem("cfg2")
    hit = False
    for attempt in range(3):
        ctx = _ctx(monkeypatch, mesh, data, inv, True, 2)
        ctx.grad(m)
        assert ctx.persist_info()["solves"] == 2 and ctx.persist_info()["timeouts"] == 0
        ctx.debug_hog(1500, 100)
        p, f, g = ctx.grad(m + 0.01)                       # must succeed whatever the dispatcher does
        info, st = ctx.persist_info(), ctx.stats()
        assert st["status"] == 0
        hit = info["timeouts"] + info["placement_fallbacks"] >= 1
        if hit:
            assert info["enabled"] == 0 and info["usable_now"] == 0
        po, mo, go = oracle_eval(mesh, data, inv, m + 0.01)
        assert relmax(p, po) < 1e-9 and abs(f - mo) / mo < 1e-9 and relmax(g, go) < 1e-7
        ctx.grad(m + 0.02)                                 # ... and the context goes on
        assert ctx.stats()["status"] == 0
        if hit and info["timeouts"]:
            # after a timeout the kernel is tried again 256 solves later (the tenant may have left)
            time.sleep(0.8)
            for k in range(130):
                ctx.grad(m + 0.001 * k)
            back = ctx.persist_info()
            assert back["enabled"] == 1 and back["solves"] > info["solves"] and back["timeouts"] == info["timeouts"], back
        ctx.close()
        time.sleep(0.8)                                    # (the hog's backlog drains)
        if hit:
            break
    assert hit, "no wait timed out in three tries: the hog did not displace the persistent kernel"


def test_two_contexts_with_cu_shares_run_their_persistent_kernels_side_by_side(monkeypatch):
    """hmcmt_next_cu_share (HipContext(cu_share=(i, 2))): each context is confined to half the CUs of every XCD -- CU-masked
    streams -- and takes half the system slots; the two persistent kernels are co-resident (no context falls back to the
    launch-per-phase loop, no wait times out) while two host threads drive them concurrently.  A system's arithmetic does not
    depend on the slot or the round it runs in: the same bits as one context on the whole device."""
    import threading
    mesh, data, inv, m = make_problem("cfg2")
    ref_ctx = _ctx(monkeypatch, mesh, data, inv, True, 2, warm_start=False)
    ref = [ref_ctx.grad(m + 0.01 * i) for i in range(6)]
    full_slots = ref_ctx.persist_info()["slots_per_xcd"]
    ref_ctx.close()
    ctxs = [_ctx(monkeypatch, mesh, data, inv, True, 2, warm_start=False, cu_share=(i, 2)) for i in range(2)]
    outs = [[None] * 6, [None] * 6]

    def run(j):
        for rep in range(3):                               # (long enough that the two really overlap)
            for i in range(6):
                outs[j][i] = ctxs[j].grad(m + 0.01 * i)

    th = [threading.Thread(target=run, args=(j,)) for j in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    for j in range(2):
        info = ctxs[j].persist_info()
        assert info["cu_share_index"] == j and info["cu_share_count"] == 2 and info["usable_now"] == 1, info
        assert info["solves"] == 36 and info["placement_fallbacks"] == 0 and info["timeouts"] == 0 and info["enabled"] == 1, info
        assert info["slots_per_xcd"] * info["workgroups_per_system"] <= 16 and info["slots_per_xcd"] <= full_slots     # (its half of an XCD's 32 CUs)
        for i in range(6):
            assert np.array_equal(outs[j][i][0], ref[i][0]) and outs[j][i][1] == ref[i][1] and np.array_equal(outs[j][i][2], ref[i][2])
    # a third context on the whole device overlaps both shares: all three are then on the launch-per-phase loop
    c3 = _ctx(monkeypatch, mesh, data, inv, True, 2)
    assert c3.persist_info()["usable_now"] == 0 and ctxs[0].persist_info()["usable_now"] == 0
    c3.close()
    assert ctxs[0].persist_info()["usable_now"] == 1
    for c in ctxs:
        c.close()


@pytest.mark.parametrize("parts", ["1", "2"])
def test_one_sweep_solves_do_not_lose_q_to_the_next_iterations_first_write(monkeypatch, parts):
    """With one smoothing sweep per side q's fp64 rows wait for alpha in the FIRST tile's space, whose complex64 slots are other
    threads' -- and the next iteration starts by writing z1 there.  Without a barrier behind the update a wave that leaves it early
    overwrote q of a wave still reading it: found by round 5's soak on the two-part kernel forced onto cfg3's mesh (true residuals
    of 1e-8 .. 1e+3 in a fifth of the cold solves; the one-part kernel has had the same window since round 4 and never hit it).
    150 cold evaluations with the true-residual check, one sweep per side, both kernels."""
    monkeypatch.setenv("HMCMT_PERSIST_CS", parts)
    mesh, data, inv, m = make_problem("cfg3")
    ctx = _ctx(monkeypatch, mesh, data, inv, True, 1, verify=True)
    assert ctx.persist_info()["column_parts"] == int(parts)
    rng = np.random.default_rng(11)
    worst = 0.0
    for k in range(150):
        ctx.grad(m + 0.05 * rng.standard_normal(m.size))
        st = ctx.stats()
        assert st["status"] == 0 and st["smoother_sweeps"] == 11
        worst = max(worst, st["true_res_max"])
    assert ctx.persist_info()["solves"] == 300
    ctx.close()
    assert worst < 1e-8, worst


@pytest.mark.parametrize("ny,width,parts,sweeps", [(100, 112, 1, 1), (100, 112, 1, 2), (200, 208, 1, 1), (200, 208, 1, 2), (405, 416, 2, 1), (405, 416, 2, 2)])
def test_width_specialised_kernels_equal_the_generic_ones_and_the_oracle(monkeypatch, ny, width, parts, sweeps):
    """The persistent kernel has instantiations with the row width as a compile-time constant (NYK: 112 padded nodes -- the
    reference's 96-cell example meshes --, 208 -- the headline mesh --, 416 -- the stress size, two column parts): the stencil
    passes address a row's points as one register + immediates (one-part kernels: in mesh orientation, no inner / outer selects),
    the transforms' MFMA loops are explicit two-stage pipelines.  Same arithmetic as the generic kernel up to the order of two
    fp64 additions in q = A p: predicted data, misfit and gradient agree to the solver tolerance, the iteration counts to +-1 --
    and both stand against the oracle (ragged meshes of those widths: masked data, a fixed cell, a rough model; the full-size
    meshes against the oracle: tests/test_gpu_parity_full.py, which runs the specialised kernels by default).
    HMCMT_PERSIST_WIDTHK=0 forces the generic kernel; hmcmt_persist_width says which one a context launches."""
    mesh, data, inv, m = ragged_problem(ny, 23, 2, 6, 4, 3)
    res = {}
    for wk in (1, 0):
        if wk:
            monkeypatch.delenv("HMCMT_PERSIST_WIDTHK", raising=False)
        else:
            monkeypatch.setenv("HMCMT_PERSIST_WIDTHK", "0")
        ctx = _ctx(monkeypatch, mesh, data, inv, True, sweeps, verify=True)
        assert ctx.persist_width() == (width if wk else 0), ctx.persist_width()
        res[wk] = ctx.grad(m) + (ctx.stats(), ctx.persist_info(), ctx.iters())
        ctx.close()
    (p1, f1, g1, s1, i1, it1), (p0, f0, g0, s0, i0, it0) = res[1], res[0]
    for s_, i_ in ((s1, i1), (s0, i0)):
        assert s_["status"] == 0 and s_["true_res_max"] < 1e-9 and s_["fallback_solves"] == 0
        assert i_["solves"] == 2 and i_["column_parts"] == parts
    assert relmax(p1, p0) < 1e-9 and abs(f1 - f0) / abs(f0) < 1e-9 and relmax(g1, g0) < 1e-8
    assert np.abs(it1 - it0).max() <= 1, (it1, it0)
    po, mo, go = oracle_eval(mesh, data, inv, m)
    assert relmax(p1, po) < 1e-9 and abs(f1 - mo) / mo < 1e-9 and relmax(g1, go) < 1e-7


def test_balanced_queues_solve_the_same_systems_to_the_same_bits(monkeypatch):
    """Meshes whose systems take turns on the chip (more systems than 8 x slots per XCD; cfg5: 64 systems, one per XCD at a time):
    a queue's time is the SUM of its systems' iterations, and the host re-orders the queues longest-first from the previous
    solve's iteration counts (persist_balance; hmcmt_persist_order).  A 230-wide ragged mesh with 20 frequencies: 40 systems on
    32 queues (two column parts, 4 row blocks: four systems per XCD at a time), eight queues hold two systems.  The same systems
    are solved, only later or earlier: every result has the same bits as with HMCMT_PERSIST_BALANCE=0, the oracle's values at the
    parity levels, and the table is a permutation that pairs the long systems with short ones."""
    mesh, data, inv, m = ragged_problem(230, 50, 20, 6, 4, 3)
    rng = np.random.default_rng(5)
    m2 = m + 0.02 * rng.standard_normal(len(m))
    res = {}
    for bal in (1, 0):
        monkeypatch.setenv("HMCMT_PERSIST_BALANCE", str(bal))
        ctx = _ctx(monkeypatch, mesh, data, inv, True, 1)
        info = ctx.persist_info()
        assert info["column_parts"] == 2 and ctx.S == 40 and 8 * info["slots_per_xcd"] == 32, info
        order0, n0 = ctx.persist_order(0)
        assert n0 == 0 and np.array_equal(order0, np.arange(40))
        ctx.grad(m)                                   # (index order: no counts yet)
        it1 = ctx.iters()
        out = ctx.grad(m2)                            # (the queues ordered from the first evaluation's counts)
        res[bal] = out + (ctx.stats(), ctx.iters(), [ctx.persist_order(k) for k in (0, 1)], it1)
        assert ctx.persist_info()["solves"] == 4
        ctx.close()
    (p1, f1, g1, s1, i1, ord1, itA), (p0, f0, g0, s0, i0, ord0, _) = res[1], res[0]
    assert s1["status"] == 0 and s0["status"] == 0
    assert np.array_equal(p1, p0) and f1 == f0 and np.array_equal(g1, g0) and np.array_equal(i1, i0)
    for kind in (0, 1):
        (tab, n), (tab0, nn0) = ord1[kind], ord0[kind]
        assert nn0 == 0 and np.array_equal(tab0, np.arange(40))
        assert n >= 1 and sorted(tab.tolist()) == list(range(40))
        # the longest queue of the table under the counts it was made from: not longer than that of the index order
        cost = itA[kind]
        span = lambda t: max(cost[t[q::32]].sum() for q in range(32))
        assert span(tab) <= span(np.arange(40)), (span(tab), span(np.arange(40)))
    po, mo, go = oracle_eval(mesh, data, inv, m2)
    assert relmax(p1, po) < 1e-8 and abs(f1 - mo) / mo < 1e-8 and relmax(g1, go) < 2e-6
