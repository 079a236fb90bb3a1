"""Per-system iteration counts of a warm chain-like sequence at a config, and what the static system -> XCD map of the persistent
kernel (s = xcd + 8 k) costs against a shared queue.  python -m scripts.gpu_iters_balance cfg5"""
import sys
import numpy as np
from hmcmt2d_amd.lib import HipContext
from hmcmt2d_amd import synthetic as S
from tests.helpers import make_problem
name = sys.argv[1] if len(sys.argv) > 1 else "cfg5"
mesh, data, inv, m = make_problem(name)
ctx = HipContext(mesh, data, inv)
rng = np.random.default_rng(5)
p = np.clip(rng.standard_normal(m.size), -2.5, 2.5)
slots = ctx.persist_info()["slots_per_xcd"]
for k in range(10):
    ctx.grad(m + 0.03 * k * p)
    it = np.array(ctx.iters()).reshape(2, -1)
    for kind in range(2):
        v = it[kind]
        Sn = len(v)
        per = np.zeros((8, slots))
        # static: group (xcd, slot) runs systems xcd + 8 (slot + slots r)
        for s in range(Sn):
            x = s % 8; q = s // 8; per[x, q % slots] += v[s]
        static = per.max()
        # shared queue, longest first, 8 * slots workers
        w = np.zeros(8 * slots)
        for t in sorted(v, reverse=True):
            w[np.argmin(w)] += t
        lpt = w.max()
        w = np.zeros(8 * slots)
        for t in v:
            w[np.argmin(w)] += t
        fifo = w.max()
        if k >= 2:
            print(f"eval {k} kind {kind}: sum {v.sum()} max {v.max()} mean/worker {v.sum() / (8 * slots):.1f} | makespan static {static:.0f} queue-fifo {fifo:.0f} queue-longest-first {lpt:.0f}", flush=True)
ctx.close()
