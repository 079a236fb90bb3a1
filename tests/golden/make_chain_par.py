"""Generates tests/golden/cfg3_chain.npz (and cfg5_chain.npz): a 40-sample HMC chain of the ORACLE at the headline size (BASELINE configs[2]: 200 x 100
cells + 7 air rows, 16 frequencies, TE+TM, 41 receivers; observations of cfg3.npz) with bench.py's sampler settings -- dt = 0.03,
L in [6, 10], lambda = 1, bounds rho in [1, 1e4] ohm-m (examples/dprism3d/startupfile:3-8) --, homogeneous 100 ohm-m reference
model, the chain state started at the synthetic's true model (bench.py's `near_true_state` chain), numpy Generator seed 2025:
make_chain.py's chain, at a size where one evaluation of the oracle is 16 core-seconds (sparse dBC).

The chain is oracle/hmcmt_oracle.py's runHMCSampler, unchanged.  Its two entry points into the hot path -- compDataGradient
(HMCSampler.jl:277-330) and the forward solve of getHamiltonian (:358-397) -- are replaced, for this run, by versions that
hand the oracle's own frequency loop (MT2DFwdSolver.jl:140-146, compJacTMatVec.jl:202-325: the reference's only parallel
axis) to one single-threaded worker process per pair of frequencies: every worker runs the oracle's compDataGradient /
MT2DFwdSolver on ITS frequencies' data (the data are sorted by frequency: contiguous blocks), the parent concatenates the
predicted data and adds the misfits and the gradients in frequency order.  Nothing of the arithmetic changes but the order
of that last sum (the serial oracle adds the frequencies' J^T v inside compJacTMatVec).

    python tests/golden/make_chain_par.py [cfg3|cfg5] [samples for a trial run: nothing is written | traj: one trajectory from the rough state]
(cfg3: 11 minutes on 8 cores; cfg5 -- the stress size, 64 systems of 82 194 unknowns, make_chain.py's CHAINS["cfg5"] -- about an hour)
"""
import copy
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
HERE = os.path.dirname(os.path.abspath(__file__))
NAME = (sys.argv[1] if len(sys.argv) > 1 else "cfg3") if __name__ == "__main__" else os.environ.get("HMCMT_CHAIN_NAME", "cfg3")     # (the spawned workers read the parent's choice from the environment)
PER = 2                                    # frequencies per worker task

_cache = {}


def _sub_problem(w):
    """(mesh, data, inv) of the frequencies [PER w, PER (w+1)) of cfg3 -- built once per worker process and task index"""
    if w not in _cache:
        from oracle import hmcmt_oracle as O
        from hmcmt2d_amd import synthetic as S
        from tests.helpers import make_problem
        mesh, data, inv, _ = make_problem(NAME)
        nR = len(data.rxLoc)
        f0, f1 = PER * w, min(PER * (w + 1), len(data.freqs))
        sl = slice(2 * nR * f0, 2 * nR * f1)                   # (freq, rx, comp) order: a frequency's data are one block
        d = S.make_data_layout(data.freqs[f0:f1], data.rxLoc[:, 0])
        assert np.array_equal(d.rxID, data.rxID[sl]) and np.array_equal(d.dtID, data.dtID[sl]) and np.array_equal(d.freqID, data.freqID[sl] - f0)
        inv.obsData = inv.obsData[sl].copy(); inv.dataW = inv.dataW[sl].copy()
        O.setupTensorMesh2D(mesh)
        _cache[w] = (mesh, d, inv, sl)
    return _cache[w]


def _task(job):
    w, vec, want_grad = job
    from threadpoolctl import threadpool_limits
    from oracle import hmcmt_oracle as O
    from hmcmt2d_amd.structs import HMCPrior
    mesh, d, inv, sl = _sub_problem(w)
    with threadpool_limits(limits=1):
        if want_grad:                                          # vec = the model (ln sigma on the active cells)
            inv.strModel = vec.copy()
            pred, mis, g = O.compDataGradient(mesh, d, inv, HMCPrior(), False)
            return pred, mis, g
        mesh.sigma = vec.copy()                                # vec = the conductivity the caller left in mesh.sigma
        pred, _ = O.MT2DFwdSolver(mesh, d, "")
        return pred, None, None


if __name__ == "__main__":
    import multiprocessing as mp
    from oracle import hmcmt_oracle as O
    from tests.helpers import make_problem
    from tests.golden.make_chain import chain_prior_of, start_model_of, SEED, RHOREF
    nmax = int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2] != "traj" else None
    mesh, data, inv, _ = make_problem(NAME)
    O.setupTensorMesh2D(mesh)
    prior = chain_prior_of(NAME)
    inv.strModel = start_model_of(NAME, mesh, inv)
    if nmax:
        prior.totalsamples = nmax; prior.burninsamples = min(prior.burninsamples, nmax // 2)
    nW = (len(data.freqs) + PER - 1) // PER
    os.environ["HMCMT_CHAIN_NAME"] = NAME
    pool = mp.get_context("spawn").Pool(min(nW, len(os.sched_getaffinity(0))))
    nev = [0, 0]

    def par_grad(mesh_, mtData, invParam, hmcprior, dense_dbc=True, keep=None):
        sig_act, dsigma = O.modelTransform(invParam.strModel)
        sigma = invParam.bgModel.copy(); sigma[invParam.activeIdx] += sig_act
        mesh_.sigma = sigma                                    # (as compDataGradient leaves it: getHamiltonian reads it)
        res = pool.map(_task, [(w, invParam.strModel, True) for w in range(nW)], chunksize=1)
        nev[0] += 1
        g = res[0][2].copy()
        for r in res[1:]:
            g = g + r[2]
        return np.concatenate([r[0] for r in res]), float(sum(r[1] for r in res)), g

    def par_fwd(mesh_, mtData, linearSolver="", keep=None, bc_fixed=None):
        res = pool.map(_task, [(w, mesh_.sigma, False) for w in range(nW)], chunksize=1)
        nev[1] += 1
        return np.concatenate([r[0] for r in res]), None

    O.compDataGradient, O.MT2DFwdSolver = par_grad, par_fwd
    t0 = time.time()
    if len(sys.argv) > 2 and sys.argv[2] == "traj":
        # ONE trajectory of bench.py's headline chain -- the chain started at SURVEY 8(d)'s rough state, the regime `value` is timed in: clamped
        # steps, bound reflections, models that get rougher with every step --: proposeLeapfrog (HMCSampler.jl:206-269) from the rough state
        # with the first momentum of the generator, L = 8, and the Hamiltonian terms at the proposal -> tests/golden/<name>_rough_traj.npz
        from hmcmt2d_amd import synthetic as S
        n = len(inv.strModel)
        m0 = S.rough_state(n)
        inv.refModel = np.full(n, np.log(1.0 / RHOREF))
        invM = np.ones(n)
        p0 = O.getMomentumVector(n, np.ones(n), np.random.default_rng(SEED))
        L = 8
        inv2 = copy.deepcopy(inv)
        m1, p1 = O.proposeLeapfrog(m0.copy(), p0.copy(), invM, mesh, data, inv2, prior, L, False)
        D, K, H, M, pred = O.getHamiltonian(data, mesh, inv2, prior, p1, invM)
        pool.close()
        print("trajectory done in %.0f s (%d gradient + %d forward evaluations): misfit at the proposal %.1f, kinetic %.1f, model norm %.1f; max |m1 - m0| %.2f" % (
            time.time() - t0, nev[0], nev[1], D, K, M, np.abs(m1 - m0).max()), flush=True)
        # (m0 and p0 are S.rough_state(n) and the generator's first momentum: kept in the small file only, the test re-draws them)
        start = dict(m0=m0, p0=p0) if n <= 20000 else {}
        np.savez_compressed(os.path.join(HERE, f"{NAME}_rough_traj.npz"), m1=m1, p1=p1, D=D, K=K, H=H, M=M, pred=pred, dt=prior.dt, L=L,
                            bounds=np.array(prior.sigBounds), mref=inv.refModel[:1], seed=SEED, **start)
        sys.exit(0)
    hm, st, hd = O.runHMCSampler(mesh, data, copy.deepcopy(inv), prior, np.random.default_rng(SEED), rhoref=RHOREF, dense_dbc=False)
    pool.close()
    print("chain done in %.0f s (%d gradient + %d forward evaluations): accepted %d of %d, nfevals %d, misfit %.1f -> %.1f" % (
        time.time() - t0, nev[0], nev[1], st["nAccept"], prior.totalsamples, prior.nfevals, st["hmstats"][0, 0], st["hmstats"][0, -1]), flush=True)
    print("decisions", st["acceptstats"].astype(int))
    if not nmax:
        np.savez_compressed(os.path.join(HERE, f"{NAME}_chain.npz"), hmstats=st["hmstats"], acceptstats=st["acceptstats"],
                            samples32=hm.astype(np.float32), mean=hm.mean(1), std=hm.std(1), nfevals=prior.nfevals,
                            first=hm[:, :5], last=hm[:, -1], data_last=hd[:, -1])
