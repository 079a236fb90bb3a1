"""Time per trajectory of bench.py's Chain (near the true model), with the accept / reject decisions."""
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
ctx = HipContext(mesh, data, inv, warm_start=os.environ.get("WS", "extrapolate"), fdm_precision=os.environ.get("FDMP", "mixed"))
dev = torch.device("cuda", 0)
n = ctx.nAC
start = {"true": m_true, "rough": S.rough_state(n), "homog": np.full(n, np.log(0.01))}[state]
c = B.Chain(ctx, torch, dev, start, np.full(n, np.log(0.01)), inv.Wm, seed=7)
for t in range(int(os.environ.get('NTRAJ', '10'))):
    a0 = c.accepted
    torch.cuda.synchronize(); t0 = time.perf_counter()
    c.trajectory(8)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    it = ctx.iters()
    print(f"traj {t}: {dt*1e3:.1f} ms ({dt*1e3/8:.2f} ms/step) {'accepted' if c.accepted > a0 else 'rejected'} last-step iters {c.iters[-1]} misfit {c.D0:.1f}", flush=True)
if len(sys.argv) > 2:
    print("second chain on the same context:", sys.argv[2])
    c2 = B.Chain(ctx, torch, dev, m_true if sys.argv[2] == "true" else S.rough_state(n), np.full(n, np.log(0.01)), inv.Wm, seed=7)
    for t in range(5):
        a0 = c2.accepted
        torch.cuda.synchronize(); t0 = time.perf_counter()
        c2.trajectory(8)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print(f"traj {t}: {dt*1e3:.1f} ms ({dt*1e3/8:.2f} ms/step) {'accepted' if c2.accepted > a0 else 'rejected'} last-step iters {c2.iters[-1]} misfit {c2.D0:.1f}", flush=True)
it = ctx.iters()
print("per-system iterations of the last evaluation (forward | adjoint), systems = TE f0..f15, TM f0..f15:")
print(" fwd", it[0].tolist())
print(" adj", it[1].tolist())
