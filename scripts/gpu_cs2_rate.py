"""Failure rate of the forced column-part kernel: cold evaluations with the true-residual check, many repetitions of a few models.
    python -m scripts.gpu_cs2_rate <cfg> <n> [env KEY=VAL ...]"""
import os, sys
import numpy as np
name, n = sys.argv[1], int(sys.argv[2])
for kv in sys.argv[3:]:
    k, v = kv.split("=", 1); os.environ[k] = v
os.environ.setdefault("HMCMT_PERSIST_CS", "2")
os.environ["HMCMT_PERSIST"] = "1"
from hmcmt2d_amd.lib import HipContext
from tests.helpers import make_problem
mesh, data, inv, m = make_problem(name)
ctx = HipContext(mesh, data, inv, verify=True)
info = ctx.persist_info()
rng = np.random.default_rng(11)
bad = []
worst = 0.0
for k in range(n):
    mm = m + 0.05 * rng.standard_normal(m.size)
    try:
        ctx.grad(mm)
        st = ctx.stats()
        worst = max(worst, st["true_res_max"])
        if st["true_res_max"] > 1e-8 or st["status"] != 0:
            its = np.array(ctx.iters()).reshape(2, -1)
            bad.append((k, st["true_res_max"], int(its[0].argmax()), int(its[0].max()), int(its[1].argmax()), int(its[1].max())))
    except Exception as e:
        bad.append((k, str(e)[:60]))
print(f"{name} {sys.argv[3:]} parts {info['column_parts']} threads/2 {info['threads_half']} G {info['workgroups_per_system']} slots {info['slots_per_xcd']} modes {info['slab_modes']}: "
      f"{len(bad)} bad of {n} (worst ok/any {worst:.2e}); first: {bad[:6]}", flush=True)
ctx.close()
