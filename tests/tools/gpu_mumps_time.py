"""Timing of the MUMPS-symbol interface on headline-size MT systems (one TE and one TM system per listed frequency)."""
import sys, time
import numpy as np, scipy.sparse as sp
sys.path.insert(0, ".")
from hmcmt2d_amd import mumps as M, synthetic as S
from tests.helpers import make_problem, oracle_eval
name = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
mesh, data, inv, m = make_problem(name)
fs = [data.freqs[0], data.freqs[len(data.freqs) // 2], data.freqs[-1]]
d1 = S.make_data_layout(fs, data.rxLoc[:, 0])
from hmcmt2d_amd import invsetup as I
n = len(d1.rxID)
inv1 = I.setupInverseDataModel(mesh, [S.SIG_AIR], 0.0, 0.0, np.full(n, 0.02 + 0.02j), np.full(n, 1e-3))
keep = {}
oracle_eval(mesh, d1, inv1, m, keep=keep)
rng = np.random.default_rng(0)
for key in sorted(keep["Aii"]):
    A = sp.csc_matrix(keep["Aii"][key]); nn = A.shape[0]
    rhs = np.asarray(keep["rhs"][key]).reshape(nn)
    t0 = time.time(); F = M.factorMUMPS(A, 1); t1 = time.time()
    x = M.applyMUMPS(F, rhs); t2 = time.time()
    st = M.lastSolveStats(F); M.destroyMUMPS(F)
    print(f"{key[0]} {key[1]:8.3g} Hz  n={nn}  factor {1e3*(t1-t0):6.1f} ms  solve {1e3*(t2-t1):7.1f} ms  "
          f"iterations {st['iterations']:6d}  passes {st['refinement_passes']}  relres {st['relres']:.2e}")
