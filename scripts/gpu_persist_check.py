"""Round 4 bring-up of the persistent solve kernel (kernels_persist.h): its preconditioner against the launch-per-phase
one, then whole evaluations against the launch-per-phase loop and (small configs) the oracle.
    python -m scripts.gpu_persist_check [cfg ...]"""
import os
import sys
import time
import numpy as np

from hmcmt2d_amd.lib import HipContext
from tests.helpers import make_problem, oracle_eval, relmax


def precond_check(name):
    mesh, data, inv, m = make_problem(name)
    for sw in (1, 2):
        os.environ["HMCMT_SWEEPS"] = str(sw)
        os.environ["HMCMT_PERSIST"] = "0"
        ctx = HipContext(mesh, data, inv)
        ctx.forward(m)
        shape = (ctx.S, ctx.NZP, ctx.NYP)
        rng = np.random.default_rng(3)
        x = np.zeros(shape, complex)
        x[:, 1:ctx.nz, 1:ctx.ny] = rng.standard_normal((ctx.S, ctx.nz - 1, ctx.ny - 1)) + 1j * rng.standard_normal((ctx.S, ctx.nz - 1, ctx.ny - 1))
        # smooth it a little so that it looks like a residual, keep white noise as a second case
        for label, v in (("white", x), ("smooth", np.cumsum(np.cumsum(x, axis=1), axis=2) / 50.0 * (np.abs(x) > 0))):
            z0 = ctx.debug_precond(v).reshape(shape)
            t0 = time.time()
            z1 = ctx.debug_persist_precond(v, sw).reshape(shape)
            dt = time.time() - t0
            per = [relmax(z1[s], z0[s]) for s in range(ctx.S)]
            print(f"{name} sweeps {sw} {label}: persistent vs launch-per-phase preconditioner: max rel diff {max(per):.3e} "
                  f"(median over systems {np.median(per):.3e}) |z| {np.abs(z0).max():.3e} [{dt * 1e3:.1f} ms]", flush=True)
            if max(per) > 1e-3:
                s = int(np.argmax(per))
                d = np.abs(z1[s] - z0[s])
                iz, iy = np.unravel_index(np.argmax(d), d.shape)
                print(f"   worst system {s} at row {iz} col {iy}: {z1[s, iz, iy]} vs {z0[s, iz, iy]}; rows with error > 1e-3 max: "
                      f"{np.nonzero(d.max(axis=1) > 1e-3 * np.abs(z0[s]).max())[0][:40]}", flush=True)
                print("   per-system:", " ".join(f"{p:.1e}" for p in per), flush=True)
        ctx.close()


def solve_check(name, oracle):
    mesh, data, inv, m = make_problem(name)
    res = {}
    for sw in (1, 2):
        for persist in (0, 1):
            os.environ["HMCMT_SWEEPS"] = str(sw)
            os.environ["HMCMT_PERSIST"] = str(persist)
            ctx = HipContext(mesh, data, inv, verify=True)
            t0 = time.time()
            out = ctx.grad(m)
            dt = time.time() - t0
            st = ctx.stats()
            res[(sw, persist)] = out + (st,)
            print(f"{name} sweeps {sw} persist {persist}: status {st['status']} iters {st['iters_fwd_max']}/{st['iters_adj_max']} "
                  f"(sum {st['iters_fwd_sum']}/{st['iters_adj_sum']}) true_res {st['true_res_max']:.2e} fallback {st['fallback_solves']} [{dt * 1e3:.0f} ms]", flush=True)
            # a few warm evaluations, timed
            ctx.set_options(verify=False)
            for k in range(3):
                ctx.grad(m + 0.01 * k)
            t0 = time.time()
            for k in range(5):
                ctx.grad(m + 0.01 * (k + 3))
            print(f"      warm evaluations: {(time.time() - t0) / 5 * 1e3:.2f} ms each, iters {ctx.stats()['iters_fwd_max']}/{ctx.stats()['iters_adj_max']}", flush=True)
            ctx.close()
        a, b = res[(sw, 0)], res[(sw, 1)]
        print(f"   persistent vs launch-per-phase: pred {relmax(b[0], a[0]):.2e} misfit {abs(b[1] - a[1]) / abs(a[1]):.2e} grad {relmax(b[2], a[2]):.2e}", flush=True)
    if oracle:
        po, mo, go = oracle_eval(mesh, data, inv, m)
        for key, (p, f, g, st) in res.items():
            print(f"   {key} vs oracle: pred {relmax(p, po):.2e} misfit {abs(f - mo) / mo:.2e} grad {relmax(g, go):.2e}", flush=True)


if __name__ == "__main__":
    names = sys.argv[1:] or ["tiny", "cfg2", "cfg3"]
    for n in names:
        precond_check(n)
    for n in names:
        solve_check(n, oracle=n in ("tiny", "cfg2"))
