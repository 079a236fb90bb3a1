"""Prints the parity levels actually achieved on the GPU for the cases of tests/test_gpu_parity_full.py
(`python -m tests.tools.gpu_parity_levels` on the GPU box); the test tolerances are these x ~5."""
import os
import numpy as np

from hmcmt2d_amd import synthetic as S
from hmcmt2d_amd.fileio import readstartupFile
from hmcmt2d_amd.lib import HipContext
from tests.helpers import GOLDEN, make_problem, relmax, cfg3_subset_problem, gerr_split


def level(tag, ctx, m, pred_ref, mis_ref, grad_ref, inv, mesh):
    pred, misfit, grad = ctx.grad(m)
    st = ctx.stats()
    sh, dp = gerr_split(grad, grad_ref, inv, mesh)
    print(f"{tag:28s} pred {relmax(pred, pred_ref):.1e} misfit {abs(misfit - mis_ref) / mis_ref:.1e} grad {sh:.1e} deep {dp:.1e} "
          f"true_res {st['true_res_max']:.1e} err_est {st['err_est_max']:.1e} iters {st['iters_fwd_max']}/{st['iters_adj_max']}", flush=True)


def main():
    for tol in (1e-11, 3e-12):
        print("tol", tol)
        for name in ("tiny", "cfg2", "cfg1"):
            g = np.load(os.path.join(GOLDEN, f"{name}.npz"))
            mesh, data, inv, m = make_problem(name)
            ctx = HipContext(mesh, data, inv, verify=True, tol=tol)
            level(name, ctx, m, g["pred"], float(g["misfit"]), g["grad"], inv, mesh)
            if name == "tiny":
                ex, hx = ctx.fields()
                ny, nz = mesh.gridSize
                for got, ref, md in ((ex, g["exTE"], "TE"), (hx, g["hxTM"], "TM")):
                    for f in range(ref.shape[1]):
                        d = np.abs(got[:, f] - ref[:, f]).reshape(nz + 1, ny + 1).max(1) / np.abs(ref[:, f]).max()
                        b = np.abs(ref[:, f]).reshape(nz + 1, ny + 1)[:, 0]
                        print(md, f, "row err:", " ".join(f"{x:.0e}" for x in d))
                        print(md, f, "left bc:", " ".join(f"{x:.0e}" for x in b))
            ctx.close()
        g = np.load(os.path.join(GOLDEN, "cfg3s.npz"))
        mesh, data, inv, m, data16, inv16 = cfg3_subset_problem(g)
        ctx = HipContext(mesh, data, inv, verify=True, tol=tol)
        level("cfg3s rough", ctx, m, g["pred"], float(g["misfit"]), g["grad"], inv, mesh)
        ex, hx = ctx.fields(); ea, ha = ctx.fields(adjoint=True)
        m_true = np.log(S.make_config("cfg3")[2][inv.activeIdx])
        level("cfg3s true model", ctx, m_true, g["pred_true"], float(g["misfit_true"]), g["grad_true"], inv, mesh)
        ctx.close()
        ctx = HipContext(mesh, data16, inv16, verify=True, tol=tol)
        ctx.grad(m)
        ex16, hx16 = ctx.fields(); ea16, ha16 = ctx.fields(adjoint=True)
        f = g["fidx"]
        print("cfg3 16-freq vs subset run: fields", relmax(ex16[:, f], ex), relmax(hx16[:, f], hx), "adjoint", relmax(ea16[:, f], ea),
              relmax(ha16[:, f], ha), "bit-identical", np.array_equal(ex16[:, f], ex), np.array_equal(ea16[:, f], ea))
        ctx.close()
        for name in ("dprism3d", "coprod2"):
            g = np.load(os.path.join(GOLDEN, f"example_{name}.npz"))
            mesh, data, inv, prior = readstartupFile(os.path.join(GOLDEN, "examples", name, "startupfile"))
            ctx = HipContext(mesh, data, inv, verify=True, tol=tol)
            level(name + " start", ctx, g["m0"], g["pred0"], float(g["misfit0"]), g["grad0"], inv, mesh)
            level(name + " perturbed", ctx, g["m1"], g["pred1"], float(g["misfit1"]), g["grad1"], inv, mesh)
            ctx.close()


if __name__ == "__main__":
    main()
