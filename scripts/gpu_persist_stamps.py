"""Phase stamps of the persistent solve kernel on one evaluation of a config (HMCMT_STAMPS=persist; printed by hmcmt_destroy).
    python -m scripts.gpu_persist_stamps [cfg3] [sweeps]"""
import os
import sys
import time
import numpy as np
os.environ["HMCMT_STAMPS"] = "persist"
os.environ["HMCMT_PERSIST"] = "1"
name = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
os.environ["HMCMT_SWEEPS"] = sys.argv[2] if len(sys.argv) > 2 else "2"
from hmcmt2d_amd.lib import HipContext
from tests.helpers import make_problem
mesh, data, inv, m = make_problem(name)
ctx = HipContext(mesh, data, inv, warm_start=False)
for k in range(3):
    t0 = time.time()
    ctx.grad(m + 0.01 * k)
    st = ctx.stats()
    print(f"eval {k}: {(time.time() - t0) * 1e3:.2f} ms iters {st['iters_fwd_max']}/{st['iters_adj_max']} sum {st['iters_fwd_sum']}/{st['iters_adj_sum']}", flush=True)
ctx.close()
