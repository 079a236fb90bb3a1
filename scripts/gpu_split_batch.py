"""Would splitting the 32 systems of an evaluation into two concurrent half-batches (even / odd frequencies on two
streams) pay?  Emulated with two contexts of 8 frequencies each driven from two host threads, against one context
with all 16."""
import os, sys, time, threading, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hmcmt2d_amd import synthetic as S, invsetup as I
from hmcmt2d_amd.lib import HipContext
mesh, data16, sig_true = S.make_config("cfg3")
ny, nz = mesh.gridSize; nair = len(mesh.airLayer)
mesh.sigma = np.concatenate([np.full(ny * nair, S.SIG_AIR), np.full(ny * (nz - nair), 0.01)])
dev = torch.device("cuda", 0)
K = 16
def make(freqs, seed=1):
    d = S.make_data_layout(freqs, data16.rxLoc[:, 0])
    nd = len(d.rxID)
    inv = I.setupInverseDataModel(mesh, [S.SIG_AIR], 0.0, 0.0, np.full(nd, 0.02 + 0.02j) * np.where(d.dtID == 1, 1, -1), np.full(nd, 1e-3))
    ctx = HipContext(mesh, d, inv)
    n = ctx.nAC
    m_true = np.log(sig_true[inv.activeIdx])
    rng = np.random.default_rng(seed)
    traj = np.stack([m_true + 0.03 * rng.standard_normal(n) for _ in range(K)])
    d_m = torch.from_numpy(traj).to(dev)
    out = (torch.zeros(2 * ctx.nData, dtype=torch.float64, device=dev), torch.zeros(1, dtype=torch.float64, device=dev), torch.zeros(n, dtype=torch.float64, device=dev))
    return ctx, d_m, out
def run(ctx, d_m, out, reps):
    for _ in range(reps):
        for k in range(K):
            ctx.grad_device_async(d_m[k].data_ptr(), out[0].data_ptr(), out[1].data_ptr(), out[2].data_ptr())
        ctx.wait()
full = make(data16.freqs)
even = make(data16.freqs[0::2]); odd = make(data16.freqs[1::2])
for c in (full, even, odd): run(*c, 1)
def timed(cs, reps=3):
    th = [threading.Thread(target=run, args=(*c, reps)) for c in cs]
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for t in th: t.start()
    for t in th: t.join()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / (reps * K)
tf = timed([full]); te = timed([even]); to = timed([odd]); tb = timed([even, odd])
print(f"16 freq, one batch      : {tf*1e3:.2f} ms per evaluation")
print(f" 8 freq (even) alone    : {te*1e3:.2f} ms;  8 freq (odd) alone: {to*1e3:.2f} ms")
print(f" 8 + 8 freq concurrently: {tb*1e3:.2f} ms per evaluation of all 16  ({tf/tb:.2f}x the single batch)")
