"""Oracle chain vs GPU chain (same seed) on the tiny config: per-sample deviation, posterior mean / std."""
import copy, sys, time
import numpy as np
sys.path.insert(0, ".")
from oracle import hmcmt_oracle as O
from hmcmt2d_amd import sampler
from hmcmt2d_amd.structs import HMCPrior
from tests.helpers import make_problem, relmax
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
mesh, data, inv, m = make_problem("tiny")
prior = HMCPrior(totalsamples=n, burninsamples=2, dt=0.02, timestep=[2, 4], sigBounds=[1e-4, 1.0])
mesh_o, inv_o, prior_o = copy.deepcopy(mesh), copy.deepcopy(inv), copy.deepcopy(prior)
O.setupTensorMesh2D(mesh_o)
t0 = time.time()
mo, so, do = O.runHMCSampler(mesh_o, data, inv_o, prior_o, np.random.default_rng(5), dense_dbc=False)
t1 = time.time()
for dev in (False, True):
    inv_p, prior_p = copy.deepcopy(inv), copy.deepcopy(prior)
    t2 = time.time()
    mp, sp_, dp = sampler.runHMCSampler(copy.deepcopy(mesh), data, inv_p, prior_p, np.random.default_rng(5), device_leapfrog=dev)
    t3 = time.time()
    sampler.release_context(inv_p)
    dev_per = [relmax(mp[:, i], mo[:, i]) for i in range(n)]
    same = int(np.sum(sp_.acceptstats == so["acceptstats"]))
    print(f"device_leapfrog={dev}: accept flags equal {same}/{n}, accepted {int(sp_.nAccept)}; per-sample max rel dev: first {dev_per[0]:.1e} "
          f"median {np.median(dev_per):.1e} last {dev_per[-1]:.1e}; mean dev {relmax(mp.mean(1), mo.mean(1)):.1e} std dev {relmax(mp.std(1), mo.std(1)):.1e}; "
          f"oracle {t1 - t0:.1f} s, gpu {t3 - t2:.1f} s, nfevals {prior_p.nfevals} vs {prior_o.nfevals}")
