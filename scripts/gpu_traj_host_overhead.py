"""Host-side time of bench.py's Chain.trajectory around the library call (near the true model): how much of a trajectory
is spent outside hmcmt_leapfrog_device + hmcmt_wait."""
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
ctx = HipContext(mesh, data, inv)
dev = torch.device("cuda", 0)
c = B.Chain(ctx, torch, dev, m_true, np.full(ctx.nAC, np.log(0.01)), inv.Wm, seed=7)
c.run(32)
tl = []
orig_l, orig_w = ctx.leapfrog_device, ctx.wait
def l(*a, **k):
    tl.append(("launch0", time.perf_counter())); r = orig_l(*a, **k); tl.append(("launch1", time.perf_counter())); return r
def w(*a, **k):
    r = orig_w(*a, **k); tl.append(("wait1", time.perf_counter())); return r
ctx.leapfrog_device, ctx.wait = l, w
torch.cuda.synchronize(); t0 = time.perf_counter()
N = 20
for _ in range(N):
    tl.append(("begin", time.perf_counter())); c.trajectory(8); tl.append(("end", time.perf_counter()))
torch.cuda.synchronize(); tot = time.perf_counter() - t0
seg = {"before": 0.0, "launch": 0.0, "wait": 0.0, "after": 0.0}
for i in range(0, len(tl), 5):
    b, l0, l1, w1, e = (x[1] for x in tl[i:i + 5])
    seg["before"] += l0 - b; seg["launch"] += l1 - l0; seg["wait"] += w1 - l1; seg["after"] += e - w1
print(f"{N} trajectories, {tot / N * 1e3:.2f} ms each:", {k: f"{v / N * 1e6:.0f} us" for k, v in seg.items()})
