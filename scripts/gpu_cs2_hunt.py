"""Hunt the guard trips of the column-part kernel forced onto cfg3 (profiles/r05 soak): walk the soak's random walk with the
true-residual guard on EVERY evaluation, stop at the first trip, then evaluate that model again under other settings.
    python -m scripts.gpu_cs2_hunt [max evaluations]"""
import os, sys
import numpy as np
os.environ["HMCMT_GUARD_EVERY"] = "1"
os.environ["HMCMT_PERSIST"] = "1"
os.environ["HMCMT_PERSIST_CS"] = "2"
from hmcmt2d_amd.lib import HipContext
from tests.helpers import make_problem
N = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
mesh, data, inv, m = make_problem("cfg3")
ctx = HipContext(mesh, data, inv)
rng = np.random.default_rng(11)
mm = m.copy()
hist = []
bad = None
for k in range(N):
    mm = np.clip(mm + 0.02 * rng.standard_normal(mm.size), np.log(1e-4), 0.0)
    hist.append(mm.copy())
    ctx.grad(mm)
    st, gd = ctx.stats(), ctx.guard()
    if gd["trips"] > 0 or st["true_res_max"] > 1e-6:
        print(f"evaluation {k}: true_res {st['true_res_max']:.3e} iters {st['iters_fwd_max']}/{st['iters_adj_max']} sweeps {st['smoother_sweeps']} guard {gd}", flush=True)
        its = np.array(ctx.iters()).reshape(2, -1)
        print("   iterations fwd", its[0].tolist(), "\n   iterations adj", its[1].tolist(), flush=True)
        bad = k
        break
print("first bad evaluation:", bad, flush=True)
ctx.close()
if bad is not None:
    mbad, mprev = hist[bad], hist[bad - 1] if bad else hist[bad]
    for label, env, warm in (("CS=2 cold", {"HMCMT_PERSIST_CS": "2"}, False), ("CS=2 warm from the previous model", {"HMCMT_PERSIST_CS": "2"}, True),
                             ("CS=1 warm", {"HMCMT_PERSIST_CS": "1"}, True), ("launch loop warm", {"HMCMT_PERSIST": "0"}, True),
                             ("CS=2 warm, one sweep", {"HMCMT_PERSIST_CS": "2", "HMCMT_SWEEPS": "1"}, True),
                             ("CS=2 warm, two sweeps", {"HMCMT_PERSIST_CS": "2", "HMCMT_SWEEPS": "2"}, True)):
        for kk in ("HMCMT_PERSIST_CS", "HMCMT_PERSIST", "HMCMT_SWEEPS"):
            os.environ.pop(kk, None)
        os.environ["HMCMT_PERSIST"] = "1"
        os.environ.update(env)
        c = HipContext(mesh, data, inv)
        if warm:
            for j in range(max(0, bad - 3), bad):
                c.grad(hist[j])
        c.grad(mbad)
        st = c.stats()
        print(f"   {label}: true_res {st['true_res_max']:.3e} iters {st['iters_fwd_max']}/{st['iters_adj_max']} sweeps {st['smoother_sweeps']} status {st['status']} persistent solves {c.persist_info()['solves']}", flush=True)
        c.close()
