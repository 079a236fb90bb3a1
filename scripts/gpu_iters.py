"""Iteration counts of the batched solver vs model roughness (per system)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hmcmt2d_amd import synthetic as S
from hmcmt2d_amd.lib import HipContext
from scripts.common import problem
name = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
mesh, data, inv = problem(name, False)
ctx = HipContext(mesh, data, inv)
n = len(inv.strModel)
ny = mesh.gridSize[0]
for label, m in [("white 0.30", S.rough_state(n, std=0.30)), ("white 0.38", S.rough_state(n, std=0.38)),
                 ("white 0.50", S.rough_state(n, std=0.50)), ("white 1.00", S.rough_state(n, std=1.0)),
                 ("0.3 + 0.233*clipN", S.rough_state(n, seed=1) + 0.233 * np.clip(np.random.default_rng([1, 0]).standard_normal(n), -2.5, 2.5)),
                 ("homog", np.full(n, np.log(0.01)))]:
    t0 = time.time(); ctx.grad(m); dt = time.time() - t0
    it = ctx.iters()
    print(f"{label:20s} {dt*1e3:7.2f} ms  fwd TE {it[0,:16].tolist()} TM {it[0,16:].tolist()}")
    print(f"{'':20s}             adj TE {it[1,:16].tolist()} TM {it[1,16:].tolist()}")
