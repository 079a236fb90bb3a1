"""Per-system iteration counts at structured models: the 2-layer + block true model, with and without white noise."""
import sys
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
n = len(m_true)
ny = mesh.gridSize[0]; nze = n // ny
two_layer = m_true.reshape(nze, ny).copy(); two_layer[:] = np.median(two_layer, axis=1, keepdims=True)
rng = np.random.default_rng(3)
cases = [("homog 100 ohm-m", np.full(n, np.log(0.01))), ("2-layer", two_layer.reshape(-1)), ("2-layer + block (true)", m_true),
         ("true + 0.24 white", m_true + 0.24 * rng.standard_normal(n)), ("homog + 0.3 white (bench state)", S.rough_state(n))]
for prec in ("fdmj", "fdm"):
    ctx = HipContext(mesh, data, inv, warm_start=False, precond=prec)
    for label, m in cases:
        ctx.grad(m); it = ctx.iters()
        print(f"{prec:5s} {label:32s} fwd TE {it[0,:16].max():3d} TM {it[0,16:].max():3d} | adj TE {it[1,:16].max():3d} TM {it[1,16:].max():3d} | fwd TE {it[0,:16].tolist()}")
    ctx.close()
