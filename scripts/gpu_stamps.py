"""Phase stamps (HMCMT_STAMPS=upd|spmv) of the last launch of a kernel in a short chain near the rough state; printed
by hmcmt_destroy.  usage: HMCMT_STAMPS=upd python scripts/gpu_stamps.py [cfg3]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hmcmt2d_amd import synthetic as S
from hmcmt2d_amd.lib import HipContext
from tests.helpers import make_problem
name = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
mesh, data, inv, m = make_problem(name)
ctx = HipContext(mesh, data, inv, tol=1e-200, maxit=6, warm_start=False)
try:
    ctx.grad(m)
except Exception as e:
    print("(expected)", str(e)[:60])
ctx.close()
