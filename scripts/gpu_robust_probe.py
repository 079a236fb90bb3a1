"""Iteration counts / fallback use on rough models (cfg2), printed for choosing the robustness test cases."""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hmcmt2d_amd.lib import HipContext
from tests.helpers import make_problem, oracle_eval, relmax, gerr_split
mesh, data, inv, m = make_problem("cfg2")
lo, hi = np.log(1e-4), np.log(1.0)
rng = np.random.default_rng(2)
cases = {}
for std in (1.0, 1.5, 2.0):
    cases[f"std{std}"] = np.clip(np.log(0.01) + std * rng.standard_normal(m.size), lo, hi)
chk = np.where((np.arange(m.size) // 50 + np.arange(m.size) % 50) % 2 == 0, lo, hi)
cases["checkerboard at the bounds"] = chk
blk = np.full(m.size, lo); blk[(np.arange(m.size) % 50 > 15) & (np.arange(m.size) % 50 < 35) & (np.arange(m.size) // 50 < 12)] = hi
cases["block hi in lo"] = blk
ctx = HipContext(mesh, data, inv, verify=True)
for name, mm in cases.items():
    try:
        pred, mis, grad = ctx.grad(mm)
        st = ctx.stats()
        po, mo, go = oracle_eval(mesh, data, inv, mm)
        print(f"{name:28s} iters {st['iters_fwd_max']}/{st['iters_adj_max']} fallback {st['fallback_solves']} true_res {st['true_res_max']:.1e} "
              f"pred {relmax(pred, po):.1e} grad {gerr_split(grad, go, inv, mesh)}", flush=True)
    except Exception as e:
        print(name, "FAILED", e, flush=True)
ctx.close()
