"""Per-step diagnostics of a real chain (near the true model or from the rough state): iterations per solve and
wall time of every leapfrog step, with the host loop of sampler.proposeLeapfrog on device tensors (synchronous calls)."""
import os, sys, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench as B
from hmcmt2d_amd import synthetic as S, invsetup as I
from hmcmt2d_amd.lib import HipContext
state = sys.argv[1] if len(sys.argv) > 1 else "true"
ntraj = int(sys.argv[2]) if len(sys.argv) > 2 else 3
mesh, data, inv0, sig_true = B.build_problem("cfg3")
ctx0 = HipContext(mesh, data, inv0)
m_true = np.log(sig_true[inv0.activeIdx])
pred_true, _ = ctx0.forward(m_true); ctx0.close()
obs, err = S.noisy_observations(pred_true)
inv = I.setupInverseDataModel(mesh, [S.SIG_AIR], 0.0, 0.0, obs, err)
ctx = HipContext(mesh, data, inv)
n = ctx.nAC
dev = torch.device("cuda", 0)
mref = np.full(n, np.log(0.01))
Wm = inv.Wm
m = torch.from_numpy(m_true if state == "true" else S.rough_state(n)).to(dev)
d_pred = torch.zeros(2 * ctx.nData, dtype=torch.float64, device=dev)
d_mis = torch.zeros(1, dtype=torch.float64, device=dev)
d_g = torch.zeros(n, dtype=torch.float64, device=dev)
Wt = torch.sparse_csr_tensor(torch.from_numpy(Wm.indptr.astype(np.int64)), torch.from_numpy(Wm.indices.astype(np.int64)),
                             torch.from_numpy(Wm.data), size=Wm.shape).to(dev)
mref_t = torch.from_numpy(mref).to(dev)
gen = torch.Generator(device=dev); gen.manual_seed(7)
lo, hi = np.log(1e-4), 0.0
def grad(mm):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ctx.grad_device(mm.data_ptr(), d_pred.data_ptr(), d_mis.data_ptr(), d_g.data_ptr())
    dt = time.perf_counter() - t0
    st = ctx.stats(); it = ctx.iters()
    return d_g + (Wt @ (mm - mref_t).unsqueeze(1)).squeeze(1), dt, st, it
for tr in range(ntraj):
    p = torch.zeros(n, dtype=torch.float64, device=dev).normal_(generator=gen).clamp_(-2.5, 2.5)
    g, dt0, st, it = grad(m)
    print(f"traj {tr} start: {dt0*1e3:.2f} ms iters {st['iters_fwd_max']}/{st['iters_adj_max']} sum {st['iters_fwd_sum']}/{st['iters_adj_sum']} misfit {float(d_mis):.1f} |g| {float(g.norm()):.1f}")
    p = p - 0.5 * B.DT * g
    mm = m.clone()
    for k in range(1, 9):
        dm = B.DT * p
        mx = float(dm.abs().max())
        if mx > 3.0: dm = dm / mx * 3.0
        mm = mm + dm
        for _ in range(50):
            below = mm < lo; mm = torch.where(below, 2 * lo - mm, mm); p = torch.where(below, -p, p)
            above = mm > hi; mm = torch.where(above, 2 * hi - mm, mm); p = torch.where(above, -p, p)
            if not (below.any() or above.any()): break
        g, dts, st, it = grad(mm)
        print(f"  step {k}: {dts*1e3:.2f} ms iters {st['iters_fwd_max']}/{st['iters_adj_max']} sum {st['iters_fwd_sum']}/{st['iters_adj_sum']} fb {st['fallback_solves']} |dm|max {min(mx,3.0):.3f} misfit {float(d_mis):.1f} std(m-mtrue) {float((mm.cpu()-torch.from_numpy(m_true)).std()):.3f}", flush=True)
        p = p - (1.0 if k < 8 else 0.5) * B.DT * g
    m = mm        # (always accept: diagnostics)
