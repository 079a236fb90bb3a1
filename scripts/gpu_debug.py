"""Kernel-level comparison HIP vs the host instantiation of the same bodies (bring-up aid)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hmcmt2d_amd import synthetic as S
from hmcmt2d_amd.lib import HipContext, HmcmtError
from scripts.common import problem
from tests.emul.emul_py import Emul

name = sys.argv[1] if len(sys.argv) > 1 else "tiny"
mesh, data, inv = problem(name, False)
m = S.rough_state(len(inv.strModel))
E = Emul(mesh, data, inv)
pe, me, ge = E.grad(m, True, 1, 1e-11)
print("emul iters", E.iters.max(axis=1))
ctx = HipContext(mesh, data, inv, maxit=40)
try:
    pg, mg, gg = ctx.grad(m)
    print("GPU ok; grad relerr vs emul", np.abs(gg - ge).max() / np.abs(ge).max(), "pred", np.abs(pg - pe).max() / np.abs(pe).max())
except HmcmtError as e:
    print("GPU grad failed:", e, ctx.stats(), ctx.iters())
shape = (ctx.S, ctx.NZP, ctx.NYP)
rng = np.random.default_rng(0)
A = rng.standard_normal(shape) + 1j * rng.standard_normal(shape)
Vp = E.get("Vpad").reshape(ctx.NYP, ctx.NYP)
for which, B in ((0, Vp), (1, Vp.T)):
    Cg = ctx.debug_transform(which, A).reshape(shape)
    Cr = A @ B
    print("transform", which, "relerr", np.abs(Cg - Cr).max() / np.abs(Cr).max())
    if np.abs(Cg - Cr).max() / np.abs(Cr).max() > 1e-10:
        print(" sample GPU", Cg[0, 1, :4], "\n ref", Cr[0, 1, :4])
# vectors with zero boundary / pad
P = np.zeros(shape, dtype=complex)
P[:, 1:ctx.nz, 1:ctx.ny] = A[:, 1:ctx.nz, 1:ctx.ny]
q_g = ctx.debug_spmv(P).reshape(shape); q_e = E.apply("spmv", P).reshape(shape)
print("spmv relerr", np.abs(q_g - q_e).max() / np.abs(q_e).max())
z_g = ctx.debug_precond(P).reshape(shape); z_e = E.apply("fdmj", P).reshape(shape)
print("fdm precond relerr", np.abs(z_g - z_e).max() / np.abs(z_e).max())
ex_g, hx_g = ctx.fields()
Xe = E.get("X").reshape(shape)
nF = len(data.freqs)
Xg = np.stack([ex_g[:, f].reshape(ctx.nz + 1, ctx.ny + 1) for f in range(nF)] + [hx_g[:, f].reshape(ctx.nz + 1, ctx.ny + 1) for f in range(nF)])
bmask = np.ones((ctx.nz + 1, ctx.ny + 1), bool); bmask[1:-1, 1:-1] = False
print("bc err", np.abs(Xg[:, bmask] - Xe[:, :, :ctx.ny + 1][:, bmask]).max())
