"""Per-system COCG iteration counts along the headline chain (the balance of the persistent kernel's system queues):
python -m scripts.gpu_iters_by_system cfg5 [steps]"""
import sys
import numpy as np
import torch
import bench as B
from hmcmt2d_amd import synthetic as S, invsetup as I
from hmcmt2d_amd.lib import HipContext

name = sys.argv[1] if len(sys.argv) > 1 else "cfg5"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 32
mesh, data, inv0, sig_true = B.build_problem(name)
ctx0 = HipContext(mesh, data, inv0, device_id=0)
m_true = np.log(sig_true[inv0.activeIdx])
pred_true, _ = ctx0.forward(m_true)
ctx0.close()
obs, err = S.noisy_observations(pred_true)
inv = I.setupInverseDataModel(mesh, [S.SIG_AIR], 0.0, 0.0, obs, err)
ctx = HipContext(mesh, data, inv, device_id=0)
dev = torch.device("cuda", 0)
mref = np.full(ctx.nAC, np.log(0.01))
chain = B.Chain(ctx, torch, dev, S.rough_state(ctx.nAC, seed=1), mref, inv.Wm, seed=20250114)
chain.run(16)
acc = np.zeros((2, ctx.S))
n = 0
for t in range(steps // B.LTRAJ):
    chain.trajectory(B.LTRAJ)
    it = ctx.iters()
    acc += it; n += 1
    if t == 0:
        print("last step of the first trajectory: fwd", it[0].tolist()); print("adj", it[1].tolist())
# the queues as the context ordered them for the LAST solve against that solve's own counts (the table was made from the one before)
for kind in range(2):
    tab, nreb = ctx.persist_order(kind)
    NQ = 8 * ctx.persist_info()["slots_per_xcd"]
    q = np.array([it[kind][tab[j::NQ]].sum() for j in range(NQ)])
    q0 = np.array([it[kind][j::NQ].sum() for j in range(NQ)])
    print("kind", kind, "tables taken", nreb, " balanced queues", q.tolist(), "max/mean %.4f" % (q.max() / q.mean()), " index order max/mean %.4f" % (q0.max() / q0.mean()))
acc /= max(n, 1)
np.set_printoptions(linewidth=250, precision=1, suppress=True)
print("mean over", n, "trajectory ends  fwd", acc[0]); print("adj", acc[1])
for kind in range(2):
    q = np.array([acc[kind][x::8].sum() for x in range(8)])
    print(("fwd" if kind == 0 else "adj"), "queue sums (XCD x: systems x, x+8, ..):", q, " max/mean", q.max() / q.mean(), " lower bound (mean)", q.mean(), " max", q.max())
ctx.close()
