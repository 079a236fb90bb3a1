"""End-to-end soak: parallelHMCSampler on the dprism3d example, two chains concurrently on one GPU, device-resident
trajectories, checkpoints every 500 samples; then the same call again (everything resumes from the final checkpoints and
returns at once with identical samples)."""
import os, sys, time, tempfile, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hmcmt2d_amd as H
from hmcmt2d_amd import sampler
ns = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "examples", "dprism3d")
mesh, data, inv, prior = H.readstartupFile(os.path.join(root, "startupfile"))
prior.totalsamples, prior.burninsamples = ns, 100
with tempfile.TemporaryDirectory() as td:
    ck = os.path.join(td, "soak.ckpt")
    t0 = time.time()
    models, stats, preds = sampler.parallelHMCSampler(mesh, data, inv, prior, nchains=2, seed=3, chains_per_gpu=2, device_leapfrog=True,
                                                      rhoref=100.0, checkpoint=ck, checkpoint_every=500)
    t1 = time.time() - t0
    print(f"2 chains x {ns} samples concurrently: {t1:.1f} s; accepted {[s.nAccept for s in stats]}; "
          f"final rms misfit {[round(float(np.sqrt(2 * s.hmstats[0, -1] / len(inv.obsData))), 3) for s in stats]}; files {sorted(os.listdir(td))}")
    t0 = time.time()
    m2, s2, p2 = sampler.parallelHMCSampler(mesh, data, inv, prior, nchains=2, seed=3, chains_per_gpu=2, device_leapfrog=True,
                                            rhoref=100.0, checkpoint=ck, checkpoint_every=500)
    print(f"second call (resumes behind the last sample): {time.time() - t0:.1f} s, identical samples: {all(np.array_equal(a, b) for a, b in zip(models, m2))}")
