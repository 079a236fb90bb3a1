"""Runs the reference's example directories (tests/golden/examples/<name>/startupfile, the reference's files unchanged)
the way HMCMT/examples/<name>/runHMCscript.jl does -- readstartupFile, runHMCSampler, getPosteriorModel, outputHMCSamples
-- on the HIP path with the device-resident trajectory, a few hundred samples long, and prints the chain's trace.
    python scripts/gpu_run_example.py [dprism3d|coprod2] [nsamples]"""
import os, sys, time, tempfile
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hmcmt2d_amd as H
from hmcmt2d_amd import sampler
name = sys.argv[1] if len(sys.argv) > 1 else "dprism3d"
ns = int(sys.argv[2]) if len(sys.argv) > 2 else 300
rhoref = float(sys.argv[3]) if len(sys.argv) > 3 else None       # None: the reference's random homogeneous start (HMCSampler.jl:100-109)
root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "examples", name)
mesh, data, inv, prior = H.readstartupFile(os.path.join(root, "startupfile"))
prior.totalsamples, prior.burninsamples = ns, ns // 3
print(f"{name}: mesh {mesh.gridSize}, {len(inv.obsData)} data, {len(data.freqs)} frequencies, {len(inv.strModel)} parameters; "
      f"dt {prior.dt}, L in {prior.timestep}, lambda {prior.regParam}, bounds sigma {prior.sigBounds}")
t0 = time.time()
model, stats, preds = sampler.runHMCSampler(mesh, data, inv, prior, rng=np.random.default_rng(0), device_leapfrog=True, rhoref=rhoref)
dt = time.time() - t0
hm = stats.hmstats
print(f"{ns} samples in {dt:.2f} s: {prior.nfevals} gradient evaluations ({prior.nfevals / dt:.0f} per second incl. host code), "
      f"accepted {stats.nAccept}, rejected {stats.nReject}")
for k in (0, 1, 2, 5, 10, 20, 50, 100, 200, ns):
    if k <= ns:
        print(f"  sample {k:4d}: data misfit {hm[0, k]:12.1f}  model norm {hm[1, k]:10.2f}  (rms misfit per datum {np.sqrt(2 * hm[0, k] / len(inv.obsData)):.2f})")
with tempfile.TemporaryDirectory() as td:
    mean, std = H.getPosteriorModel(model, mesh, inv, prior, outdir=td)
    H.outputHMCSamples(model, stats, preds, ichain=1, cputime=dt, outdir=td)
    print("wrote", sorted(os.listdir(td)))
print(f"posterior ln(sigma) after burn-in: mean in [{mean.min():.2f}, {mean.max():.2f}], std in [{std.min():.3f}, {std.max():.3f}]")
if name == "dprism3d":
    # the model the example's data were generated from (oracle/pin/recover_dprism.py): where does the
    # posterior put it?
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from tests.helpers import dprism_generating_problem
    tmesh = dprism_generating_problem()[0]
    truth = np.log(tmesh.sigma[inv.activeIdx])
    ny, nz = mesh.gridSize; nair = len(mesh.airLayer)
    T = truth.reshape(nz - nair, ny); M = np.asarray(mean).reshape(nz - nair, ny); Sd = np.asarray(std).reshape(nz - nair, ny)
    core = np.zeros_like(T, dtype=bool); core[:30, 8:88] = True          # under the receiver line, top 3 km
    l10 = lambda x: -x / np.log(10.0)                                    # ln sigma -> log10 rho
    for label, sel in (("conductive prism (10 ohm-m)", (T > np.log(0.05)) & core), ("resistive prism (1000 ohm-m)", (T < np.log(0.005)) & core),
                       ("background (100 ohm-m), top 3 km under the line", (np.abs(T - np.log(0.01)) < 1e-9) & core)):
        z = (M[sel] - T[sel]) / Sd[sel]
        print(f"  {label}: {sel.sum()} cells, true log10 rho {l10(T[sel]).mean():.2f}, posterior mean {l10(M[sel]).mean():.2f} "
              f"(cell range {l10(M[sel]).min():.2f} .. {l10(M[sel]).max():.2f}), mean posterior std {Sd[sel].mean() / np.log(10):.2f} decades, "
              f"truth within 2 std in {100 * np.mean(np.abs(z) < 2):.0f} % of the cells")
