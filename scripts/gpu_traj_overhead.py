"""Where a trajectory of bench.py's Chain spends its host time: momentum draw / kinetic energy (torch), the library call, the
wait, the read-back and the accept test."""
import os, sys, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench as B
from hmcmt2d_amd import synthetic as S, invsetup as I
from hmcmt2d_amd.lib import HipContext
mesh, data, inv0, sig_true = B.build_problem("cfg3")
ctx0 = HipContext(mesh, data, inv0)
m_true = np.log(sig_true[inv0.activeIdx])
pred_true, _ = ctx0.forward(m_true); ctx0.close()
obs, err = S.noisy_observations(pred_true)
inv = I.setupInverseDataModel(mesh, [S.SIG_AIR], 0.0, 0.0, obs, err)
ctx = HipContext(mesh, data, inv, warm_start="extrapolate")
dev = torch.device("cuda", 0)
n = ctx.nAC
c = B.Chain(ctx, torch, dev, m_true, np.full(n, np.log(0.01)), inv.Wm, seed=7)
for t in range(4): c.trajectory(8)
seg = np.zeros(5); N = 10
for t in range(N):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    c.p.normal_(generator=c.gen).clamp_(-2.5, 2.5)
    c.ham[0] = 0.5 * (c.p * c.p).sum()
    c.m_prop.copy_(c.m_cur)
    torch.cuda.current_stream().synchronize(); t1 = time.perf_counter()
    ctx.leapfrog_device(c.m_prop.data_ptr(), c.p.data_ptr(), B.DT, 8, B.LAMBDA, c.lo, c.hi, c.start_grad, c.d_pred.data_ptr(), c.scal.data_ptr(), c.scal.data_ptr() + 8)
    t2 = time.perf_counter()
    ctx.wait(); t3 = time.perf_counter()
    st = ctx.stats()
    c.ham[1] = 0.5 * (c.p * c.p).sum()
    c.ham[2:4] = c.scal
    K0, K1, D1, M1 = c.ham.tolist(); t4 = time.perf_counter()
    hdif = c.D0 + c.M0 + K0 - (D1 + M1 + K1)
    if hdif > 0 or c.host_rng.random() < np.exp(hdif):
        c.m_cur, c.m_prop = c.m_prop, c.m_cur; c.D0, c.M0 = D1, M1; c.start_grad = 1
    else:
        c.start_grad = 2
    t5 = time.perf_counter()
    seg += [t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4]
seg *= 1e6 / N
print("per trajectory of 8 steps (us): momentum + kinetic energy + copy + torch sync %.0f | leapfrog_device call (host, returns with the last step queued) %.0f | wait %.0f | "
      "kinetic energy + read-back %.0f | accept test %.0f | total %.0f" % (*seg, seg.sum()))
ctx.close()
