"""Soak of the persistent solve kernel: thousands of warm-started evaluations along a random walk of models on every shape
of the kernel (64 / 128 / 256 threads per half; 1, 3, 8 workgroups per system), the stopping-rule guard on every 10th
evaluation, the true-residual check at the end.  Prints what a reader needs to see: no failed solve, no placement
fallback, no guard trip, the worst true residual, evaluations per second.
    python -m scripts.gpu_persist_soak [evaluations]"""
import os
import sys
import time
import numpy as np

os.environ["HMCMT_PERSIST"] = "1"
from hmcmt2d_amd.lib import HipContext
from tests.helpers import make_problem

N = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
os.environ["HMCMT_GUARD_EVERY"] = sys.argv[2] if len(sys.argv) > 2 else "10"
# (name, HMCMT_PERSIST_CS, HMCMT_SWEEPS, evaluations): round 5 adds the column-part shapes -- forced onto cfg1 / cfg2 / cfg3 (three thread
# counts, equal and unequal parts), cfg5's own -- and forces the smoother both ways (the library's own choice mixes them)
cases = []
for sw in ("", "1", "2"):
    cases += [("tiny", "", sw, N), ("cfg2", "", sw, N), ("cfg1", "", sw, N), ("cfg3", "", sw, N), ("cfg2", "2", sw, N), ("cfg1", "2", sw, N),
              ("cfg3", "2", sw, N), ("cfg5", "", sw, max(N // 5, 100))]
for name, cs, sw, n_eval in cases:
    for key, val in (("HMCMT_PERSIST_CS", cs), ("HMCMT_SWEEPS", sw)):
        if val:
            os.environ[key] = val
        else:
            os.environ.pop(key, None)
    mesh, data, inv, m = make_problem(name)
    ctx = HipContext(mesh, data, inv)
    rng = np.random.default_rng(11)
    mm = m.copy()
    n = n_eval
    bad = 0
    t0 = time.time()
    for k in range(n):
        mm = np.clip(mm + 0.02 * rng.standard_normal(mm.size), np.log(1e-4), 0.0)
        try:
            p, f, g = ctx.grad(mm)
            if not (np.isfinite(f) and np.isfinite(g).all()):
                bad += 1
        except Exception as e:
            bad += 1
            print(f"   {name} evaluation {k}: {e}", flush=True)
    dt = time.time() - t0
    info, gd = ctx.persist_info(), ctx.guard()
    tables = ctx.persist_order(0)[1]         # (meshes whose systems take turns: how often the queues were re-balanced)
    ctx.set_options(verify=True)
    ctx.grad(mm + 1e-3)
    st = ctx.stats()
    ctx.close()
    print(f"{name}{' (column parts forced)' if cs else ''}{' sweeps ' + sw if sw else ''} [parts {info['column_parts']}, {info['workgroups_per_system']} workgroups/system, {info['slots_per_xcd']} slots/XCD]: {n} evaluations in {dt:.1f} s ({n / dt:.0f}/s), failed {bad}, persistent solves {info['solves']} of {2 * n + 2}, placement fallbacks {info['placement_fallbacks']} timeouts {info['timeouts']}, queue tables taken {tables}, "
          f"guard checks {gd['checks']} trips {gd['trips']} worst {gd['worst_true_res']:.1e}; final verify: status {st['status']} true_res {st['true_res_max']:.1e} "
          f"iters {st['iters_fwd_max']}/{st['iters_adj_max']} fp64 restarts in the last evaluation {st['fallback_solves']}", flush=True)
