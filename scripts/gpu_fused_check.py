"""Fused forward FDM kernel vs the separate transform + tridiagonal kernels on the same input."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hmcmt2d_amd.lib import HipContext
from tests.helpers import make_problem
mesh, data, inv, m = make_problem(sys.argv[1] if len(sys.argv) > 1 else "cfg3")
ctx = HipContext(mesh, data, inv)
ctx.grad(m)
rng = np.random.default_rng(0)
shape = (ctx.S, ctx.NZP, ctx.NYP)
for trial in range(3):
    T = np.zeros(shape, complex)
    T[:, 1:ctx.nz, 1:ctx.ny] = rng.standard_normal((ctx.S, ctx.nz - 1, ctx.ny - 1)) + 1j * rng.standard_normal((ctx.S, ctx.nz - 1, ctx.ny - 1))
    f, g = ctx.debug_fdm_fwd(T)
    f = f.reshape(shape); g = g.reshape(shape)
    d = np.abs(f - g)
    scale = np.abs(g).reshape(ctx.S, -1).max(1)[:, None, None] + 1e-300
    rel = d / scale
    bad = np.argwhere(~(rel < 1e-4))
    print("trial", trial, "max rel", np.nanmax(rel), "nan", np.isnan(f).sum(), np.isnan(g).sum(), "bad entries", len(bad))
    if len(bad):
        sys_ = np.unique(bad[:, 0]); print(" bad systems", sys_.tolist())
        for s in sys_[:4]:
            b = bad[bad[:, 0] == s]
            print("  s", s, "rows", b[:, 1].min(), "..", b[:, 1].max(), "n rows", len(np.unique(b[:, 1])),
                  "cols", np.unique(b[:, 2]).tolist()[:40])
            r, c = b[0, 1], b[0, 2]
            np.set_printoptions(precision=5, linewidth=200)
            print("   first bad", (r, c), "A", f[s, max(r-1,0):r+2, c:c+4].tolist())
            print("   first bad", (r, c), "B", g[s, max(r-1,0):r+2, c:c+4].tolist())
