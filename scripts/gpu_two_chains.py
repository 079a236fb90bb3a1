"""Do two independent chains on ONE GPU overlap?  Aggregate leapfrog steps/s of 1 vs 2 contexts driven from two host
threads (ctypes releases the GIL inside the library calls)."""
import os, sys, time, threading, numpy as np
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
dev = torch.device("cuda", 0)
n = len(m_true)
K = 16
def make(seed):
    ctx = HipContext(mesh, data, inv)
    rng = np.random.default_rng(seed)
    traj = np.stack([m_true + 0.03 * rng.standard_normal(n) * 1.0 for _ in range(K)])   # independent nearby models: ~cold solves
    d_m = torch.from_numpy(traj).to(dev)
    out = (torch.zeros(2 * ctx.nData, dtype=torch.float64, device=dev), torch.zeros(1, dtype=torch.float64, device=dev), torch.zeros(n, dtype=torch.float64, device=dev))
    return ctx, d_m, out
def run(ctx, d_m, out, reps):
    for _ in range(reps):
        for k in range(K):
            ctx.grad_device_async(d_m[k].data_ptr(), out[0].data_ptr(), out[1].data_ptr(), out[2].data_ptr())
        ctx.wait()
a = make(1); b = make(2)
run(*a, 1); run(*b, 1)
torch.cuda.synchronize(); t0 = time.perf_counter(); run(*a, 3); torch.cuda.synchronize(); t1 = time.perf_counter() - t0
print(f"one chain : {3*K/t1:.1f} steps/s")
th = [threading.Thread(target=run, args=(*c, 3)) for c in (a, b)]
torch.cuda.synchronize(); t0 = time.perf_counter()
for t in th: t.start()
for t in th: t.join()
torch.cuda.synchronize(); t2 = time.perf_counter() - t0
print(f"two chains: {2*3*K/t2:.1f} steps/s aggregate ({2*3*K/t2/(3*K/t1):.2f}x)")
