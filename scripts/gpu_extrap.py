"""Iteration counts per initial-guess mode along the bench trajectory (cfg3)."""
import sys, numpy as np
sys.path.insert(0, ".")
from hmcmt2d_amd.lib import HipContext
from tests.helpers import make_problem
mesh, data, inv, m = make_problem(sys.argv[1] if len(sys.argv) > 1 else "cfg3")
rng = np.random.default_rng(3)
for mode in ("previous", "extrapolate"):
    ctx = HipContext(mesh, data, inv, warm_start=mode)
    mm = m.copy(); p = rng.standard_normal(m.size)
    rows = []
    for j in range(12):
        mm = mm + 0.03 * p
        p = p + 0.02 * rng.standard_normal(m.size)
        ctx.grad(mm); st = ctx.stats()
        rows.append((st["iters_fwd_max"], st["iters_adj_max"], st["iters_fwd_sum"], st["iters_adj_sum"]))
    print(mode, rows)
    ctx.close()
