"""Per-launch time of the iteration kernels when only HALF of the batch is active from the start (TM-only data on cfg3: 16 of
32 systems), for A/B runs of the tile shapes (HMCMT_RT, HMCMT_UPD2, HMCMT_SPMV ...): in a real solve the TE half converges
in half the iterations of the TM half, so every second iteration runs in this state."""
import os, sys, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hmcmt2d_amd import synthetic as S, invsetup as I
from hmcmt2d_amd.lib import HipContext
which = sys.argv[1] if len(sys.argv) > 1 else "tm"
mesh, data, sig_true = S.make_config("cfg3")
if which != "all":
    keep = data.dtID == (2 if which == "tm" else 1)
    data.dataID = keep.copy()
    data.rxID, data.freqID, data.dtID = data.rxID[keep], data.freqID[keep], data.dtID[keep]
ny, nz = mesh.gridSize
nair = len(mesh.airLayer)
mesh.sigma = np.concatenate([np.full(ny * nair, S.SIG_AIR), np.full(ny * (nz - nair), 0.01)])
n = len(data.rxID)
rng = np.random.default_rng(3)
inv = I.setupInverseDataModel(mesh, [S.SIG_AIR], 0.0, 0.0, (0.02 + 0.01 * rng.standard_normal(n)) * (1 + 1j), np.full(n, 2e-3))
ctx = HipContext(mesh, data, inv, warm_start="cold")
m = S.rough_state(ctx.nAC)
for i in range(3): ctx.grad(m + 0.01 * i)
ctx.profile(True, every=1)
t0 = time.perf_counter()
for i in range(6): ctx.grad(m + 0.01 * (i + 3))
dt = time.perf_counter() - t0
pr = ctx.profile_read(); cn = ctx.profile_counters(); st = ctx.stats()
ov = ctx.profile_overhead_us()
print(f"{which}: {dt/6*1e3:.2f} ms per cold evaluation, iterations {st['iters_fwd_max']}/{st['iters_adj_max']}, two-sweep solves {cn['solves_two_sweeps']}/{cn['solves']}, "
      f"active systems per iteration launch {cn['active_iter_systems'] / max(1, pr['spmv'][1]):.1f}")
for k in ("spmv", "vector_ops", "tridiagonal", "post_smoother", "fdm_transform"):
    ms, cnt = pr[k]
    if cnt: print(f"   {k:14s} {ms*1e3/cnt - ov:6.2f} us per launch ({cnt} launches)")
ctx.close()
