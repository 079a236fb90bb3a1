"""Does a foreign kernel holding CUs (hmcmt_debug_hog) make the persistent kernel's waits time out?  python -m scripts.gpu_hog_probe"""
import os, time
os.environ["HMCMT_PS_SPIN"] = "2048"
os.environ["HMCMT_SWEEPS"] = "2"
from hmcmt2d_amd.lib import HipContext
from tests.helpers import make_problem
mesh, data, inv, m = make_problem("cfg2")
ctx = HipContext(mesh, data, inv)
ctx.grad(m)
for nb, ms in ((250, 500), (256, 500), (200, 500), (2000, 300)):
    for k in range(2):
        ctx.grad(m + 0.001 * k)
    t0 = time.time(); ctx.grad(m + 0.01); tref = time.time() - t0
    ctx.debug_hog(nb, ms)
    time.sleep(0.05)
    t0 = time.time()
    try:
        ctx.grad(m + 0.02)
        err = None
    except Exception as e:
        err = str(e)
    dt = time.time() - t0
    print(f"hog {nb} x {ms} ms: eval {dt * 1e3:.1f} ms (alone {tref * 1e3:.1f} ms) info {ctx.persist_info()} err {err}", flush=True)
    time.sleep(0.6)
    ctx.close()
    ctx = HipContext(mesh, data, inv)
    ctx.grad(m)
ctx.close()
