"""Aggregate steps/s of c chains on one GPU with CU shares (hmcmt_next_cu_share), near the true model of a config:
    python -m scripts.gpu_cu_share_probe cfg3
one chain on the whole device | one chain on half | two chains on halves | four chains on quarters"""
import sys, time, threading
import numpy as np
import torch
sys.argv = sys.argv[:2]
import bench as B
from hmcmt2d_amd import synthetic as S, invsetup as I
from hmcmt2d_amd.lib import HipContext
name = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
mesh, data, inv0, sig_true = B.build_problem(name)
dev = torch.device("cuda", 0)
ctx0 = HipContext(mesh, data, inv0)
m_true = np.log(sig_true[inv0.activeIdx])
pred_true, _ = ctx0.forward(m_true)
ctx0.close()
obs, err = S.noisy_observations(pred_true)
inv = I.setupInverseDataModel(mesh, [S.SIG_AIR], 0.0, 0.0, obs, err)
mref = np.full(len(m_true), np.log(0.01))
K = 48
for label, shares in (("1 chain, whole device", [None]), ("1 chain on half the CUs", [(0, 2)]), ("2 chains on halves", [(0, 2), (1, 2)]),
                      ("4 chains on quarters", [(i, 4) for i in range(4)])):
    ctxs = [HipContext(mesh, data, inv, cu_share=sh) for sh in shares]
    chains = [B.Chain(c, torch, dev, m_true, mref, inv.Wm, seed=7 + j) for j, c in enumerate(ctxs)]
    for c in chains:
        c.run(16)
    def run_on_own_stream(c):
        # (CU-masked streams are BLOCKING streams -- hipExtStreamCreateWithCUMask has no flags --, so torch work on the legacy
        #  default stream would wait for the other chain's trajectory and make it wait: every chain's torch ops on a stream of its own)
        with torch.cuda.stream(torch.cuda.Stream(device=dev)):
            c.run(K)
    th = [threading.Thread(target=run_on_own_stream, args=(c,)) for c in chains]
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for t in th: t.start()
    for t in th: t.join()
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    info = ctxs[0].persist_info()
    print(f"{label}: {len(chains) * K / dt:.1f} steps/s aggregate; slots/XCD {info['slots_per_xcd']} persistent solves {[c.persist_info()['solves'] for c in ctxs]} "
          f"timeouts {[c.persist_info()['timeouts'] for c in ctxs]}", flush=True)
    for c in ctxs:
        c.close()
