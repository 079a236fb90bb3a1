"""Iteration counts on REAL leapfrog trajectories near equilibrium (cfg3, chain started at the true model):
trajectories are generated once with the host leapfrog loop, then replayed for each extrapolation order
(HMCMT_EXTRAP_POINTS must be set per process: run this script once per order)."""
import os, sys
import numpy as np
sys.path.insert(0, ".")
from hmcmt2d_amd import synthetic as S, invsetup as I
from hmcmt2d_amd.lib import HipContext
import bench
name = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
mesh, data, inv0, sig_true = bench.build_problem(name, 0)
ctx0 = HipContext(mesh, data, inv0)
m_true = np.log(sig_true[inv0.activeIdx])
pred_true, _ = ctx0.forward(m_true); ctx0.close()
obs, err = S.noisy_observations(pred_true)
inv = I.setupInverseDataModel(mesh, [S.SIG_AIR], 0.0, 0.0, obs, err)
path = "gpurun_out/traj_%s.npy" % name
ctx = HipContext(mesh, data, inv)
n = ctx.nAC
if not os.path.exists(path):
    rng = np.random.default_rng(7)
    Wm = inv.Wm; mref = m_true.copy(); lam = 1.0; dt = 0.03
    m = m_true + 0.02 * rng.standard_normal(n)
    models = []
    for traj in range(5):
        p = np.clip(rng.standard_normal(n), -2.5, 2.5)
        _, f, g = ctx.grad(m); models.append(m.copy())
        g = g + lam * (Wm @ (m - mref))
        p = p - 0.5 * dt * g
        for l in range(8):
            m = m + dt * p
            _, f, g = ctx.grad(m); models.append(m.copy())
            g = g + lam * (Wm @ (m - mref))
            p = p - (dt if l < 7 else 0.5 * dt) * g
        print("traj", traj, "misfit", f, "|g|", np.linalg.norm(g), "|p|", np.linalg.norm(p))
    np.save(path, np.array(models))
models = np.load(path)
ctx.close()
ctx = HipContext(mesh, data, inv)
tot = []
for j, mm in enumerate(models):
    ctx.grad(mm); st = ctx.stats()
    tot.append((st["iters_fwd_max"], st["iters_adj_max"], st["iters_fwd_sum"] + st["iters_adj_sum"]))
print("points", os.environ.get("HMCMT_EXTRAP_POINTS", "4"), "sum", sum(t[2] for t in tot[9:]), [t[:2] for t in tot[9:27]])
