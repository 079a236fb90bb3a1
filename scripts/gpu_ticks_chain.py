"""Untraced timeline (HMCMT_TICKS) of the last evaluation of bench.py's chain, at the rough state or near the true model.
    python scripts/gpu_ticks_chain.py [true|rough] [trajectories]"""
import os, sys, numpy as np
os.environ["HMCMT_TICKS"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench as B
from hmcmt2d_amd import synthetic as S, invsetup as I
from hmcmt2d_amd.lib import HipContext
state = sys.argv[1] if len(sys.argv) > 1 else "true"
ntraj = int(sys.argv[2]) if len(sys.argv) > 2 else 6
cfgname = sys.argv[3] if len(sys.argv) > 3 else "cfg3"
mesh, data, inv0, sig_true = B.build_problem(cfgname)
ctx0 = HipContext(mesh, data, inv0)
m_true = np.log(sig_true[inv0.activeIdx])
pred_true, _ = ctx0.forward(m_true); ctx0.close()
obs, err = S.noisy_observations(pred_true)
inv = I.setupInverseDataModel(mesh, [S.SIG_AIR], 0.0, 0.0, obs, err)
ctx = HipContext(mesh, data, inv, warm_start="extrapolate")
dev = torch.device("cuda", 0)
if os.environ.get("HMCMT_BENCH_FORCE_PG"):        # (round 6: what an initialised RCCL process group does to the chain)
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29543")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev); dist.barrier()
n = ctx.nAC
start = {"true": m_true, "rough": S.rough_state(n)}[state]
c = B.Chain(ctx, torch, dev, start, np.full(n, np.log(0.01)), inv.Wm, seed=7)
for t in range(ntraj): c.trajectory(8)
torch.cuda.synchronize()
print(state, c.summary())
ctx.close()
