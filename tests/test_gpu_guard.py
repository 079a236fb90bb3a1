"""The production guard of the stopping rule (hmcmt_guard): the COCG loop stops on the ESTIMATE ||z|| <= tol ||x||
(z the preconditioned residual); every HMCMT_GUARD_EVERY-th evaluation the true residual ||b - A x|| / ||b|| of both solves is
formed without options.verify, so that a chain cannot run unnoticed on an optimistic estimate.  The solves:
MTFwdSolver/mt2DTE.jl:47-55, mt2DTM.jl:46-54, MTSensitivity/compJacTMatVec.jl:220-229, 291-300 (direct solves in the reference:
residual at rounding level, which is what the guard holds this path to)."""
import numpy as np
import pytest

from hmcmt2d_amd.lib import HipContext
from tests.helpers import make_problem, relmax

pytestmark = pytest.mark.gpu


def test_guard_checks_every_nth_evaluation_and_changes_nothing(monkeypatch):
    """Warm-started evaluations along a line of models with the guard at every 2nd evaluation against no guard at all: the same
    bits (the guarded evaluation only keeps a copy of the adjoint right-hand side), the checks counted, residuals at the level
    options.verify reports."""
    mesh, data, inv, m = make_problem("cfg2")
    outs = {}
    for every in (0, 2):
        monkeypatch.setenv("HMCMT_GUARD_EVERY", str(every))
        ctx = HipContext(mesh, data, inv)
        outs[every] = [ctx.grad(m + 0.01 * i) for i in range(5)]
        g = ctx.guard()
        st = ctx.stats()
        ctx.close()
        if every == 0:
            assert g["checks"] == 0 and g["trips"] == 0 and st["true_res_max"] == 0.0
        else:
            assert g["checks"] == 4 and g["trips"] == 0            # evaluations 2 and 4, a forward and an adjoint solve each
            assert 0 < g["worst_true_res"] < 1e-9 and g["last_true_res"] <= g["worst_true_res"]
    for a, b in zip(outs[0], outs[2]):
        assert np.array_equal(a[0], b[0]) and a[1] == b[1] and np.array_equal(a[2], b[2])


def test_default_guard_period_leaves_short_runs_alone(monkeypatch):
    monkeypatch.delenv("HMCMT_GUARD_EVERY", raising=False)
    mesh, data, inv, m = make_problem("tiny")
    ctx = HipContext(mesh, data, inv)
    for i in range(3):
        ctx.grad(m + 0.01 * i)
    assert ctx.guard()["checks"] == 0
    for i in range(3, 100):
        ctx.grad(m + 0.01 * i)
    g = ctx.guard()
    ctx.close()
    assert g["checks"] == 2 and g["trips"] == 0 and 0 < g["worst_true_res"] < 1e-9     # the 100th evaluation


def test_optimistic_estimate_is_flagged(monkeypatch, capfd):
    """A stopping tolerance of 1e-2 ends every solve on an estimate that says nothing about the residual any more: the guard
    sees the true residual, counts a trip, says so on stderr, and the next evaluation starts cold (same iteration counts as a
    fresh context's first evaluation rather than a warm start's few)."""
    monkeypatch.setenv("HMCMT_GUARD_EVERY", "2")
    mesh, data, inv, m = make_problem("cfg2")
    ctx = HipContext(mesh, data, inv, tol=1e-2)
    ctx.grad(m)
    cold = ctx.stats()["iters_fwd_sum"]
    ctx.grad(m + 0.01)                                    # guarded
    g = ctx.guard()
    assert g["checks"] == 2 and g["trips"] >= 1 and g["worst_true_res"] > 1e-6
    err = capfd.readouterr().err
    assert "stopping-rule guard" in err and "true residual" in err
    ctx.grad(m + 0.02)                                    # cold again after the trip
    assert ctx.stats()["iters_fwd_sum"] >= cold - 2 * ctx.S
    ctx.set_options(tol=1e-11)
    p, f, gr = ctx.grad(m + 0.03)                         # guarded, at a production tolerance: no new trip
    g2 = ctx.guard()
    ctx.close()
    assert g2["checks"] == 4 and g2["trips"] == g["trips"] and g2["last_true_res"] < 1e-6
