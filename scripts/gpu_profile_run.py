"""Workload for rocprofv3: N gradient evaluations of one config (default cfg3, 20 evals)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hmcmt2d_amd import synthetic as S
from hmcmt2d_amd.lib import HipContext
from scripts.common import problem
name = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
mesh, data, inv = problem(name, False)
m = S.rough_state(len(inv.strModel))
ctx = HipContext(mesh, data, inv)
for j in range(n):
    ctx.grad(m + 1e-3 * j)          # (distinct models: identical ones are answered from the memo)
print(ctx.stats())
