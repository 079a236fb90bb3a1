"""Does the persistent kernel's iteration time depend on how many systems share an XCD's L2?  cfg3's mesh with 4 / 8 / 16
frequencies (1 / 2 / 4 systems per XCD; working set per system ~1.5 MB + 0.84 MB of fp64 coefficients per polarisation).
    python -m scripts.gpu_persist_l2probe"""
import os
import subprocess
import sys

CHILD = r'''
import os, sys, numpy as np
os.environ["HMCMT_STAMPS"] = "persist"; os.environ["HMCMT_PERSIST"] = "1"; os.environ["HMCMT_SWEEPS"] = "2"
from hmcmt2d_amd import synthetic as S, invsetup as I
from hmcmt2d_amd.lib import HipContext
nf = int(sys.argv[1])
c = S.CONFIGS["cfg3"]
mesh = S.make_mesh(c["ny"], c["nz"])
y0, y1, st = c["rx"]
data = S.make_data_layout(S.log_freqs(16)[::16 // nf], np.arange(y0, y1 + 0.5 * st, st))
n = len(data.rxID)
obs = np.full(n, 0.02 + 0.02j) * np.where(data.dtID == 1, 1.0, -1.0)
from tests.helpers import start_sigma
mesh.sigma = start_sigma(mesh)
inv = I.setupInverseDataModel(mesh, [S.SIG_AIR], 0.0, 0.0, obs, np.full(n, 1e-3))
m = S.rough_state(len(inv.strModel))
ctx = HipContext(mesh, data, inv, warm_start=False)
for k in range(3):
    ctx.grad(m + 0.01 * k)
print("nfreq", nf, "systems", ctx.S, ctx.persist_info(), flush=True)
ctx.close()
'''
for nf in (4, 8, 16):
    r = subprocess.run([sys.executable, "-c", CHILD, str(nf)], capture_output=True, text=True, timeout=300)
    print(r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stdout)
    print("\n".join(l for l in r.stderr.splitlines() if "STAMPS" in l or "slabs" in l))
