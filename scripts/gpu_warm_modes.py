"""Iterations per step along trajectories around the true model for the three initial-guess modes."""
import os, sys, time, numpy as np
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
n = len(m_true)
rng = np.random.default_rng(3)
p = np.clip(rng.standard_normal(n), -2.5, 2.5)
for centre, name in ((m_true, "true"), (S.rough_state(n), "rough")):
    for scale in (1.0, 0.1):
        for mode in ("cold", "previous", "extrapolate"):
            ctx = HipContext(mesh, data, inv, warm_start=mode)
            out = []
            for k in range(9):
                ctx.grad(centre + scale * 0.03 * k * p)
                st = ctx.stats()
                out.append(f"{st['iters_fwd_max']}/{st['iters_adj_max']}")
            print(f"{name:5s} step x{scale} {mode:12s}", " ".join(out), flush=True)
            ctx.close()
