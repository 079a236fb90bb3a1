"""Accuracy against the oracle and iteration counts as functions of the stopping tolerance (cfg2; cfg3 against its golden file).
    python -m scripts.gpu_tol_probe"""
import os
import numpy as np
from hmcmt2d_amd.lib import HipContext
from tests.helpers import make_problem, oracle_eval, relmax, GOLDEN

for name in ("cfg2", "cfg3"):
    mesh, data, inv, m = make_problem(name)
    if name == "cfg2":
        po, mo, go = oracle_eval(mesh, data, inv, m)
    else:
        g = np.load(os.path.join(GOLDEN, "cfg3.npz"))
        m, po, mo, go = g["m"], g["pred"], float(g["misfit"]), g["grad"]
    for tol in (1e-11, 1e-10, 1e-9, 1e-8):
        ctx = HipContext(mesh, data, inv, tol=tol, verify=True)
        p, f, gr = ctx.grad(m)
        st = ctx.stats()
        ctx.close()
        print(f"{name} tol {tol:.0e}: iters {st['iters_fwd_max']}/{st['iters_adj_max']} sum {st['iters_fwd_sum']}/{st['iters_adj_sum']} true_res {st['true_res_max']:.1e} "
              f"err_est {st['err_est_max']:.1e} | vs oracle: pred {relmax(p, po):.2e} misfit {abs(f - mo) / mo:.2e} grad {relmax(gr, go):.2e}", flush=True)
