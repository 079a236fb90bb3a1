"""three cold evaluations of cfg3 (the workload of scripts/gpu_persist_pmc.sh)"""
from hmcmt2d_amd.lib import HipContext
from tests.helpers import make_problem
mesh, data, inv, m = make_problem("cfg3")
ctx = HipContext(mesh, data, inv, warm_start=False)
for k in range(3):
    ctx.grad(m + 0.01 * k)
st = ctx.stats()
print("iters", st["iters_fwd_max"], st["iters_adj_max"], st["iters_fwd_sum"], st["iters_adj_sum"])
ctx.close()
