"""Shapes of the persistent kernel over a grid of mesh sizes: which instantiation persist_shape picks, and cold evaluations with the
true-residual check under both smoothers, with the library's choice of column parts and with two parts forced.
    python -m scripts.gpu_shape_fuzz [evaluations per case] [verify|guard]
"verify" forms the true residuals with the verification pass (host-side start of the solves); "guard" runs the production path (the kernel
forms the initial residual itself) with the production guard on every evaluation and reads the guard's worst true residual."""
import os, sys
import numpy as np
os.environ["HMCMT_PERSIST"] = "1"
from hmcmt2d_amd.lib import HipContext
from hmcmt2d_amd import synthetic as S, invsetup as I
from tests.helpers import start_sigma
N = int(sys.argv[1]) if len(sys.argv) > 1 else 30
MODE = sys.argv[2] if len(sys.argv) > 2 else "verify"
if MODE == "guard":
    os.environ["HMCMT_GUARD_EVERY"] = "1"
bad_total = 0
for ny, nz in ((37, 22), (60, 93), (60, 40), (96, 49), (130, 30), (130, 121), (200, 100), (200, 150), (230, 60), (270, 14), (301, 19), (330, 120), (400, 200), (415, 40), (100, 250)):
    mesh = S.make_mesh(ny, nz, npad_y=min(7, (ny - 3) // 2), npad_z=min(8, nz - 2))
    data = S.make_data_layout(S.log_freqs(3), np.linspace(-1500.0, 1500.0, 5))
    n = len(data.rxID)
    obs = np.full(n, 0.02 + 0.02j) * np.where(data.dtID == 1, 1.0, -1.0)
    mesh.sigma = start_sigma(mesh)
    inv = I.setupInverseDataModel(mesh, [S.SIG_AIR], 0.0, 0.0, obs, np.full(n, 1e-3))
    m0 = S.rough_state(len(inv.strModel))
    for cs in ("", "2"):
        for sw in ("1", "2"):
            if cs:
                os.environ["HMCMT_PERSIST_CS"] = cs
            else:
                os.environ.pop("HMCMT_PERSIST_CS", None)
            os.environ["HMCMT_SWEEPS"] = sw
            try:
                ctx = HipContext(mesh, data, inv, verify=(MODE == "verify"))
            except Exception as e:
                print(f"{ny}x{nz} cs={cs or 'auto'} sw={sw}: create failed: {e}", flush=True)
                continue
            info = ctx.persist_info()
            rng = np.random.default_rng(5)
            worst, bad = 0.0, 0
            for k in range(N):
                try:
                    ctx.grad(m0 + 0.05 * rng.standard_normal(m0.size))
                    st = ctx.stats()
                    tr = ctx.guard()["last_true_res"] if MODE == "guard" else st["true_res_max"]
                    worst = max(worst, tr)
                    bad += int(not (tr <= 2e-8) or st["status"] != 0)
                except Exception as e:
                    bad += 1
            used = ctx.persist_info()["solves"]
            ctx.close()
            bad_total += bad
            print(f"{ny}x{nz} (NYP {ctx.NYP}, rows {ctx.nz}) cs={cs or 'auto'} sw={sw}: threads/2 {info['threads_half']} parts {info['column_parts']} G {info['workgroups_per_system']} slots {info['slots_per_xcd']} "
                  f"modes {info['slab_modes']} persistent solves {used}/{2 * N}; bad {bad} worst true_res {worst:.1e}", flush=True)
print("MODE", MODE, "TOTAL BAD", bad_total, flush=True)
