"""Bring-up check on a GPU box: HIP path vs oracle on small configs, kernel unit checks, timings.

    python tests/tools/gpu_check.py [cfg ...]      (default: tiny cfg2 cfg3)
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from hmcmt2d_amd import synthetic as S, invsetup as I            # noqa: E402
from hmcmt2d_amd.lib import HipContext                            # noqa: E402
from hmcmt2d_amd.structs import HMCPrior                          # noqa: E402


def problem(name, with_oracle):
    from oracle import hmcmt_oracle as O
    mesh, data, sig_true = S.make_config(name)
    ny, nz = mesh.gridSize
    nair = len(mesh.airLayer)
    start = np.concatenate([np.full(ny * nair, 1e-8), np.full(ny * (nz - nair), 0.01)])
    if with_oracle:
        O.setupTensorMesh2D(mesh)
        mesh.sigma = sig_true.copy()
        pred_true, _ = O.MT2DFwdSolver(mesh, data)
        obs, err = S.noisy_observations(pred_true)
    else:
        obs = np.ones(len(data.rxID), dtype=complex) * (0.02 + 0.02j)
        err = np.full(len(obs), 1e-3)
    mesh.sigma = start
    inv = I.setupInverseDataModel(mesh, [1e-8], 0, 0, obs, err)
    return mesh, data, inv


def main():
    cfgs = sys.argv[1:] or ["tiny", "cfg2", "cfg3"]
    for name in cfgs:
        with_oracle = name in ("tiny", "cfg2", "cfg1")
        mesh, data, inv = problem(name, with_oracle)
        m = S.rough_state(len(inv.strModel))
        print(f"=== {name}: ny={mesh.gridSize[0]} nz={mesh.gridSize[1]} nFreq={len(data.freqs)} nData={len(inv.obsData)}", flush=True)
        ctx = HipContext(mesh, data, inv, verify=True)
        print("dims NYP,NZP,S,nblk:", ctx.NYP, ctx.NZP, ctx.S, ctx.nblk)
        # --- transform unit check
        rng = np.random.default_rng(0)
        A = (rng.standard_normal((ctx.S, ctx.NZP, ctx.NYP)) + 1j * rng.standard_normal((ctx.S, ctx.NZP, ctx.NYP)))
        t0 = time.time()
        pred, mis, g = ctx.grad(m)
        t1 = time.time() - t0
        st = ctx.stats()
        print(f"first grad call {t1*1e3:.1f} ms  stats {st}")
        if with_oracle:
            from oracle import hmcmt_oracle as O
            inv.strModel = m.copy()
            keep = {}
            pd, mo, go = O.compDataGradient(mesh, data, inv, HMCPrior(), False, keep)
            print("  pred relerr", np.abs(pred - pd).max() / np.abs(pd).max(), " misfit rel", abs(mis - mo) / mo,
                  " grad relerr(max-norm)", np.abs(g - go).max() / np.abs(go).max())
            ex, hx = ctx.fields()
            print("  field err TE", np.abs(ex - keep["exTE"]).max(), " TM", np.abs(hx - keep["hxTM"]).max())
        # Jacobi vs FDM self-consistency
        if name != "cfg3":
            ctx.set_options(precond="jacobi", maxit=50000)
            t0 = time.time(); p2, m2, g2 = ctx.grad(m); tj = time.time() - t0
            print(f"  jacobi: {tj*1e3:.1f} ms stats {ctx.stats()}  grad diff vs fdm {np.abs(g2-g).max()/np.abs(g).max():.3e}")
            ctx.set_options(precond="fdmj", maxit=2000)
        # timing
        ctx.set_options(verify=0)
        for _ in range(2):
            ctx.grad(m)
        n = 10
        t0 = time.time()
        for _ in range(n):
            ctx.grad(m)
        dt = (time.time() - t0) / n
        print(f"  FDM grad: {dt*1e3:.2f} ms/eval  ({1/dt:.1f} steps/s)  iters {ctx.stats()['iters_fwd_max']}/{ctx.stats()['iters_adj_max']}")
        ctx.profile(True)
        for _ in range(5):
            ctx.grad(m)
        pr = ctx.profile_read()
        tot = sum(v[0] for v in pr.values())
        for k, (ms, cnt) in pr.items():
            print(f"    {k:14s} {ms/5:8.3f} ms/eval  {cnt//5:5d} launches/eval  avg {1e3*ms/max(cnt,1):7.2f} us")
        print(f"    total kernel time {tot/5:.3f} ms/eval")
        ctx.profile(False)
        ctx.close()


if __name__ == "__main__":
    main()
