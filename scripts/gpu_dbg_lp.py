import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hmcmt2d_amd import synthetic as S
from hmcmt2d_amd.lib import HipContext, HmcmtError
from scripts.common import problem
name = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
mesh, data, inv = problem(name, False)
m = S.rough_state(len(inv.strModel))
ctx = HipContext(mesh, data, inv, maxit=40, warm_start=False, fdm_precision="fp64")
ctx.grad(m); print("fp64 iters", ctx.iters()[0].tolist())
shape = (ctx.S, ctx.NZP, ctx.NYP)
rng = np.random.default_rng(0)
P = np.zeros(shape, complex); P[:, 1:ctx.nz, 1:ctx.ny] = rng.standard_normal((ctx.S, ctx.nz-1, ctx.ny-1)) + 1j*rng.standard_normal((ctx.S, ctx.nz-1, ctx.ny-1))
z64 = ctx.debug_precond(P).reshape(shape)
t64 = ctx.debug_transform(0, P).reshape(shape)
t2 = ctx.debug_transform(2, P).reshape(shape)
print("transform mixed vs fp64 relerr (max-norm)", np.abs(t2 - t64).max() / np.abs(t64).max())
bad = np.argwhere(np.abs(t2 - t64) > 0.05 * np.abs(t64).max())
print("n bad", len(bad), bad[:10].tolist())
t1 = ctx.debug_transform(1, P).reshape(shape); t3 = ctx.debug_transform(3, P).reshape(shape)
print("transform' mixed vs fp64 relerr", np.abs(t3 - t1).max() / np.abs(t1).max())
bad = np.argwhere(np.abs(t3 - t1) > 0.05 * np.abs(t1).max()); print("n bad", len(bad), bad[:10].tolist())
ctx.set_options(fdm_precision="mixed")
try:
    ctx.grad(m)
except HmcmtError as e:
    print(e)
print("mixed iters", ctx.iters()[0].tolist())
zm = ctx.debug_precond(P).reshape(shape)
for s in [0, 8, 15, 16, 24, 31]:
    print(s, "precond mixed vs fp64", np.abs(zm[s] - z64[s]).max() / np.abs(z64[s]).max())
