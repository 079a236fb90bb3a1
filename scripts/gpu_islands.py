"""Untraced time of the three serial islands of an evaluation (HIP-event brackets of the library: categories assembly_bc =
sigma .. first residual, receivers = between the solves, gradient = behind the adjoint solve) on bench.py's chain."""
import os, sys, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench as B
from hmcmt2d_amd import synthetic as S, invsetup as I
from hmcmt2d_amd.lib import HipContext
state = sys.argv[1] if len(sys.argv) > 1 else "true"
mesh, data, inv0, sig_true = B.build_problem("cfg3")
ctx0 = HipContext(mesh, data, inv0)
m_true = np.log(sig_true[inv0.activeIdx])
pred_true, _ = ctx0.forward(m_true); ctx0.close()
obs, err = S.noisy_observations(pred_true)
inv = I.setupInverseDataModel(mesh, [S.SIG_AIR], 0.0, 0.0, obs, err)
ctx = HipContext(mesh, data, inv, warm_start="extrapolate")
dev = torch.device("cuda", 0)
n = ctx.nAC
start = {"true": m_true, "rough": S.rough_state(n)}[state]
c = B.Chain(ctx, torch, dev, start, np.full(n, np.log(0.01)), inv.Wm, seed=7)
for t in range(4): c.trajectory(8)
for prof in (False, True):
    if prof: ctx.profile(("assembly_bc", "receivers", "gradient"), every=1)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for t in range(6): c.trajectory(8)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"{'bracketed' if prof else 'plain'}: {dt*1e6/48:.1f} us per step")
pr = ctx.profile_read(); cn = ctx.profile_counters()
ev = max(cn["evaluations"], 1)
for k in ("assembly_bc", "receivers", "gradient"):
    print(f"  {k:12s} {pr[k][0]*1e3/ev:7.1f} us per evaluation ({pr[k][1]/ev:.1f} brackets, overhead {ctx.profile_overhead_us():.2f} us each not subtracted)")
ctx.close()
