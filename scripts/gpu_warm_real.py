"""A real leapfrog trajectory near the true model (host loop), then the same model sequence replayed with the other
initial-guess modes / extrapolation orders: iterations per step."""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as B
from hmcmt2d_amd import synthetic as S, invsetup as I
from hmcmt2d_amd.lib import HipContext
mesh, data, inv0, sig_true = B.build_problem("cfg3")
ctx0 = HipContext(mesh, data, inv0)
m_true = np.log(sig_true[inv0.activeIdx])
pred_true, _ = ctx0.forward(m_true); ctx0.close()
obs, err = S.noisy_observations(pred_true)
inv = I.setupInverseDataModel(mesh, [S.SIG_AIR], 0.0, 0.0, obs, err)
n = len(m_true); mref = np.full(n, np.log(0.01)); Wm = inv.Wm
rng = np.random.default_rng(3)
state = sys.argv[1] if len(sys.argv) > 1 else "true"
ctx = HipContext(mesh, data, inv)
models = []
m = m_true.copy() if state == "true" else S.rough_state(n)
m_start = m.copy()
lo, hi = np.log(1e-4), 0.0
for tr in range(2):
    p = np.clip(rng.standard_normal(n), -2.5, 2.5)
    _, _, g = ctx.grad(m); g = g + Wm @ (m - mref)
    p = p - 0.5 * 0.03 * g
    if state != "true":
        m = m_start.copy()                      # (every proposal is rejected there: each trajectory restarts at the rough state)
        _, _, g = ctx.grad(m); g = g + Wm @ (m - mref)
        p = np.clip(rng.standard_normal(n), -2.5, 2.5) - 0.5 * 0.03 * g
    for k in range(1, 9):
        dm = 0.03 * p
        if np.abs(dm).max() > 3.0:
            dm = dm / np.abs(dm).max() * 3.0
        m = m + dm
        for _ in range(50):
            b = m < lo; m = np.where(b, 2 * lo - m, m); p = np.where(b, -p, p)
            a = m > hi; m = np.where(a, 2 * hi - m, m); p = np.where(a, -p, p)
            if not (a.any() or b.any()): break
        models.append(m.copy())
        _, _, g = ctx.grad(m); g = g + Wm @ (m - mref)
        p = p - (1.0 if k < 8 else 0.5) * 0.03 * g
ctx.close()
d = np.diff(np.array(models[:8]), axis=0)
print("cos of consecutive steps", [round(float(d[i] @ d[i+1] / np.linalg.norm(d[i]) / np.linalg.norm(d[i+1])), 4) for i in range(6)])
print("|step|", [round(float(np.linalg.norm(x)), 3) for x in d])
for mode, npts in (("cold", 6), ("previous", 6), ("extrapolate", 2), ("extrapolate", 3), ("extrapolate", 4), ("extrapolate", 6)):
    os.environ["HMCMT_EXTRAP_POINTS"] = str(npts)
    ctx = HipContext(mesh, data, inv, warm_start=mode)
    ctx.grad(m_start)
    out = []
    for mm in models:
        ctx.grad(mm); st = ctx.stats()
        out.append(f"{st['iters_fwd_max']}/{st['iters_adj_max']}")
    print(f"{mode:12s} np={npts}", " ".join(out), flush=True)
    ctx.close()
