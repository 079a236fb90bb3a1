"""How many independent chains does ONE GPU overlap?  Aggregate leapfrog steps/s of 1..NMAX contexts, one host thread
per chain (bench.py's Chain: real trajectories near the true model; ctypes releases the GIL inside the library)."""
import os, sys, time, threading, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench as B
from hmcmt2d_amd import synthetic as S, invsetup as I
from hmcmt2d_amd.lib import HipContext
cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
NMAX = int(sys.argv[2]) if len(sys.argv) > 2 else 4
mesh, data, inv0, sig_true = B.build_problem(cfg)
ctx0 = HipContext(mesh, data, inv0)
m_true = np.log(sig_true[inv0.activeIdx])
pred_true, _ = ctx0.forward(m_true); ctx0.close()
obs, err = S.noisy_observations(pred_true)
inv = I.setupInverseDataModel(mesh, [S.SIG_AIR], 0.0, 0.0, obs, err)
dev = torch.device("cuda", 0)
n = len(m_true)
mref = np.full(n, np.log(0.01))
Ke = 64
chains = []
for i in range(NMAX):
    ctx = HipContext(mesh, data, inv)
    c = B.Chain(ctx, torch, dev, m_true, mref, inv.Wm, seed=7 + i)
    c.run(16)
    chains.append(c)
base = None
for nc in range(1, NMAX + 1):
    th = [threading.Thread(target=c.run, args=(Ke,)) for c in chains[:nc]]
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for t in th: t.start()
    for t in th: t.join()
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    r = nc * Ke / dt
    base = base or r
    print(f"{nc} chains: {r:.1f} steps/s aggregate ({r/base:.2f}x)", flush=True)
