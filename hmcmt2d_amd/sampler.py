"""Host-side mirror of the reference sampler API on top of the HIP library.

Same function names, argument order and return values as HMCMT/src/HMCSampler/HMCSampler.jl:
`compDataGradient` (:277-330), `getHamiltonian` (:358-397), `proposeLeapfrog` (:206-269),
`getKineticEnergy` / `getKineticGradient` (:407-431), `getMomentumVector` (:441-453),
`setMassMatrix` (:463-474), `checkParameterBound` (:515-559), `runHMCSampler` (:72-196) and
`parallelHMCSampler` (parallelHMC.jl:10-49).  The only compute they do themselves is O(nparam)
vector arithmetic; the forward / adjoint solves go through `HipContext` (include/hmcmt.h).

Differences that are deliberate and documented:
  * random numbers come from an explicit `numpy.random.Generator` (the reference uses Julia's
    unseeded global RNG, so sample-level parity with it is impossible by construction);
  * `getHamiltonian` reuses the forward response of the last gradient evaluation when it was
    computed at the same model (the reference repeats an identical forward solve,
    HMCSampler.jl:251 vs :364); `reuse_forward=False` restores the reference's cost structure;
  * a non-finite model raises instead of spinning forever in the bound reflection (:546-548).
"""
from __future__ import annotations

import os
import time
import weakref
import numpy as np

from .lib import HipContext
from .structs import HMCParameter, HMCStatus, initHMCParameter, initHMCStatus

_contexts: dict = {}          # (id(invParam), device) -> HipContext; entries die with their InvDataModel (weakref.finalize)


def default_device() -> int:
    """The GPU of this process: an explicit HMCMT_DEVICE first (the same precedence as julia/HMCMTHip.jl's
    defaultDevice), else LOCAL_RANK under torch.distributed.run / torchrun (one process per GPU), else 0.  The
    reference's counterpart is the worker id of `pmap` (parallelHMC.jl:23-40)."""
    for var in ("HMCMT_DEVICE", "LOCAL_RANK"):
        v = os.environ.get(var)
        if v not in (None, ""):
            return int(v)
    return 0


def _drop_context(key):
    ctx = _contexts.pop(key, None)
    if ctx is not None:
        ctx.close()


def get_context(mtMesh, mtData, invParam, device_id=None, **opts) -> HipContext:
    """One HIP context per (InvDataModel, device), created lazily on `device_id` (default: this process's GPU,
    `default_device()`).  The cache entry is tied to the life of `invParam` (a recycled id() can never return a
    context built for another problem) and is rebuilt when different solver options are requested."""
    dev = default_device() if device_id is None else int(device_id)
    key = (id(invParam), dev)
    ctx = _contexts.get(key)
    if ctx is not None and (ctx._owner() is not invParam or (opts and opts != ctx._opts)):
        _drop_context(key)
        ctx = None
    if ctx is None:
        ctx = HipContext(mtMesh, mtData, invParam, device_id=dev, **opts)
        ctx._cache = None
        ctx._opts = dict(opts)
        ctx._owner = weakref.ref(invParam)
        ctx.device_id = dev
        _contexts[key] = ctx
        weakref.finalize(invParam, _drop_context, key)
    return ctx


def cu_shares_for(chains_per_gpu, nlocal):
    """CU shares (hmcmt_next_cu_share) for `nlocal` chains of this rank under chains_per_gpu: the largest of 4, 2, 1 that exceeds
    neither -- every share in use, none idle (parallelHMC.jl:23-45 runs np chains over fewer workers the same way)."""
    return max(sh for sh in (4, 2, 1) if sh <= max(1, min(int(chains_per_gpu), int(nlocal))))


def release_context(invParam, device_id=None):
    for key in [k for k in _contexts if k[0] == id(invParam) and (device_id is None or k[1] == int(device_id))]:
        _drop_context(key)


def _check_solver(hmcprior):
    if hmcprior.linearSolver.lower() not in ("", "hip"):
        raise ValueError(f"linearsolver {hmcprior.linearSolver!r}: this package only has the HIP path "
                         "(use `linearsolver: hip`)")


# --------------------------------------------------------------------------------------------
def compDataGradient(mtMesh, mtData, invParam, hmcprior, ctx: HipContext | None = None):
    """m = invParam.strModel -> (predData, dataMisfit, dataGrad)."""
    _check_solver(hmcprior)
    ctx = ctx or get_context(mtMesh, mtData, invParam)
    m = np.asarray(invParam.strModel, dtype=np.float64)
    pred, misfit, grad = ctx.grad(m)
    sigma = invParam.bgModel.copy()
    sigma[invParam.activeIdx] += np.exp(m)
    mtMesh.sigma = sigma                                   # HMCSampler.jl:294
    ctx._cache = (m.copy(), pred, misfit)
    return pred, misfit, grad


def compDataMisfit(predData, invParam):
    res = invParam.dataW * (predData - invParam.obsData)
    return 0.5 * float(np.real(np.vdot(res, res)))


def getKineticEnergy(momentum, hmcParam: HMCParameter):
    return 0.5 * float(np.dot(momentum, hmcParam.invM * momentum))


def getKineticGradient(momentum, hmcParam: HMCParameter):
    return hmcParam.invM * momentum


def getMomentumVector(nparam, hmcParam: HMCParameter, rng):
    mp = np.clip(rng.standard_normal(nparam), -2.5, 2.5)   # clipped N(0,1), HMCSampler.jl:444-447
    return hmcParam.sqrtM * mp


def setMassMatrix(nparam, scaling=1.0):
    mass = scaling * np.ones(nparam)
    return 1.0 / mass, np.sqrt(mass)


def checkParameterBound(model, momentum, hmcprior):
    """Reflect at the ln(sigma) bounds and flip the momentum (vectorised; same fixed point as the
    reference's per-element while loop)."""
    lo, hi = np.log(hmcprior.sigBounds[0]), np.log(hmcprior.sigBounds[1])
    if not (np.all(np.isfinite(model)) and hi > lo):
        raise FloatingPointError("non-finite model value in checkParameterBound")
    for _ in range(500):
        below, above = model < lo, model > hi
        if not (below.any() or above.any()):
            return model, momentum
        model = np.where(below, 2.0 * lo - model, model)
        momentum = np.where(below, -momentum, momentum)
        above = model > hi
        model = np.where(above, 2.0 * hi - model, model)
        momentum = np.where(above, -momentum, momentum)
    raise FloatingPointError("bound reflection did not terminate")


def getHamiltonian(mtData, mtMesh, invParam, hmcprior, hmcParam: HMCParameter,
                   ctx: HipContext | None = None, reuse_forward=True):
    """(dataMisfit, kinetic, hamiltonian, mnorm, predData) at invParam.strModel."""
    _check_solver(hmcprior)
    ctx = ctx or get_context(mtMesh, mtData, invParam)
    m = np.asarray(invParam.strModel, dtype=np.float64)
    cache = getattr(ctx, "_cache", None)
    if reuse_forward and cache is not None and np.array_equal(cache[0], m):
        pred, misfit = cache[1], cache[2]
    else:
        pred, misfit = ctx.forward(m)
        ctx._cache = (m.copy(), pred, misfit)
    kp = getKineticEnergy(hmcParam.momentum, hmcParam)
    mprior = m - invParam.refModel
    mnorm = 0.5 * float(mprior @ (invParam.Wm @ mprior)) * hmcprior.regParam
    return misfit, kp, misfit + kp + mnorm, mnorm, pred


def proposeLeapfrog(hmcParamCurrent: HMCParameter, mtMesh, mtData, invParam, hmcprior, rng=None,
                    intstep=None, ctx: HipContext | None = None):
    """Leapfrog trajectory; L ~ U{timestep[0]..timestep[1]} unless `intstep` is given."""
    ctx = ctx or get_context(mtMesh, mtData, invParam)
    currModel = hmcParamCurrent.rhomodel
    currMomentum = hmcParamCurrent.momentum
    invParam.strModel = currModel.copy()
    _, _, dataGrad = compDataGradient(mtMesh, mtData, invParam, hmcprior, ctx)
    hmcprior.nfevals += 1
    refModel, Wm = invParam.refModel, invParam.Wm
    dataGrad = dataGrad + (Wm @ (currModel - refModel)) * hmcprior.regParam
    dt = hmcprior.dt
    propMomentum = currMomentum - 0.5 * dt * dataGrad
    propModel = currModel.copy()
    if intstep is None:
        intstep = int(rng.integers(hmcprior.timestep[0], hmcprior.timestep[1] + 1))
    maxStepSize = 3.0
    for k in range(1, intstep + 1):
        dm = dt * getKineticGradient(propMomentum, hmcParamCurrent)
        dmMax = np.max(np.abs(dm))
        if dmMax > maxStepSize:
            dm = dm / dmMax * maxStepSize
        propModel = propModel + dm
        propModel, propMomentum = checkParameterBound(propModel, propMomentum, hmcprior)
        invParam.strModel = propModel.copy()
        _, _, dataGrad = compDataGradient(mtMesh, mtData, invParam, hmcprior, ctx)
        hmcprior.nfevals += 1
        dataGrad = dataGrad + (Wm @ (propModel - refModel)) * hmcprior.regParam
        delta = dt * dataGrad
        propMomentum = propMomentum - (delta if k < intstep else 0.5 * delta)
    return propModel, propMomentum


def proposeLeapfrogDevice(hmcParamCurrent: HMCParameter, mtMesh, mtData, invParam, hmcprior, rng=None,
                          intstep=None, ctx: HipContext | None = None):
    """Same contract as proposeLeapfrog, with the whole trajectory kept on the GPU
    (`hmcmt_leapfrog`): only (m, p) go in and (m', p', predData, misfit) come out.  The prior
    (refModel, Wm, mass) must have been registered with `ctx.set_prior`."""
    ctx = ctx or get_context(mtMesh, mtData, invParam)
    if intstep is None:
        intstep = int(rng.integers(hmcprior.timestep[0], hmcprior.timestep[1] + 1))
    lo, hi = np.log(hmcprior.sigBounds[0]), np.log(hmcprior.sigBounds[1])
    m1, p1, pred, misfit, mnorm, nf = ctx.leapfrog(hmcParamCurrent.rhomodel, hmcParamCurrent.momentum, hmcprior.dt,
                                                   intstep, hmcprior.regParam, lo, hi)
    hmcprior.nfevals += nf
    invParam.strModel = m1.copy()
    sigma = invParam.bgModel.copy()
    sigma[invParam.activeIdx] += np.exp(m1)
    mtMesh.sigma = sigma
    ctx._cache = (m1.copy(), pred, misfit)                 # getHamiltonian reuses the proposal's forward
    return m1, p1


def _run_fingerprint(invParam, hmcprior, shape):
    """What a checkpoint belongs to: sizes, sampler settings, data, weights and reference model (sha256)."""
    import hashlib
    h = hashlib.sha256()
    h.update(np.asarray(shape, dtype=np.int64).tobytes())
    h.update(np.asarray([hmcprior.dt, hmcprior.regParam, *hmcprior.timestep, *hmcprior.sigBounds], dtype=np.float64).tobytes())
    for a in (invParam.obsData, invParam.dataW, invParam.refModel):
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()


def _save_checkpoint(path, it, done, hmcmodel, hmcdata, stats, cur, start, rng, invParam, hmcprior):
    """Flushes samples `done + 1 .. it` and the loop state.  Two files: `path + ".samples"`, append-only records of one
    sample each (model column, predicted-data column) -- a flush writes only the new samples, so a run costs O(N) I/O,
    not O(N^2 / k) -- and `path`, the small state (counters, statistics, current model / momentum, RNG state, the run's
    fingerprint), replaced atomically AFTER the samples are on disk (flush + fsync before os.replace; a crash between
    the two leaves surplus records that a resume ignores)."""
    import json
    import os
    with open(path + ".samples", "r+b" if os.path.exists(path + ".samples") else "w+b") as f:
        rec = (hmcmodel.shape[0] + 2 * hmcdata.shape[0]) * 8
        f.seek(done * rec)
        f.truncate()
        for j in range(done + 1, it + 1):
            f.write(np.ascontiguousarray(hmcmodel[:, j - 1]).tobytes())
            f.write(np.ascontiguousarray(hmcdata[:, j]).tobytes())
        f.flush()
        os.fsync(f.fileno())
    tmp = path + ".tmp"
    with open(tmp, "wb") as f:
        np.savez(f, it=it, data0=hmcdata[:, 0], hmstats=stats.hmstats[:, :it + 1],
                 acceptstats=stats.acceptstats[:it], counts=np.array([stats.nAccept, stats.nReject, hmcprior.nfevals]),
                 rhomodel=cur.rhomodel, momentum=cur.momentum, start=np.array(start),
                 fingerprint=np.array(_run_fingerprint(invParam, hmcprior, hmcmodel.shape)),
                 rng=np.array(json.dumps(rng.bit_generator.state)))
        f.flush()
        os.fsync(f.fileno())
    os.replace(tmp, path)


def _load_checkpoint(path, hmcmodel, hmcdata, stats, cur, rng, invParam, hmcprior):
    import json
    with np.load(path) as g:
        if str(g["fingerprint"]) != _run_fingerprint(invParam, hmcprior, hmcmodel.shape):
            raise ValueError(f"{path}: checkpoint of a different run (sizes, dt / timestep / regParam / sigBounds, data, "
                             "weights or reference model differ)")
        it = int(g["it"])
        nparam, ndata = hmcmodel.shape[0], hmcdata.shape[0]
        raw = np.fromfile(path + ".samples", dtype=np.float64, count=it * (nparam + 2 * ndata))
        if raw.size != it * (nparam + 2 * ndata):
            raise ValueError(f"{path}.samples holds fewer than the {it} samples the state file records")
        raw = raw.reshape(it, nparam + 2 * ndata)
        hmcmodel[:, :it] = raw[:, :nparam].T
        hmcdata[:, 1:it + 1] = raw[:, nparam:].copy().view(np.complex128).T
        hmcdata[:, 0] = g["data0"]
        stats.hmstats[:, :it + 1] = g["hmstats"]
        stats.acceptstats[:it] = g["acceptstats"]
        stats.nAccept, stats.nReject, hmcprior.nfevals = (int(c) for c in g["counts"])
        cur.rhomodel, cur.momentum = g["rhomodel"].copy(), g["momentum"].copy()
        rng.bit_generator.state = json.loads(str(g["rng"]))
        return it, tuple(float(x) for x in g["start"])


def runHMCSampler(mtMesh, mtData, invParam, hmcprior, rng=None, rhoref=None, ctx: HipContext | None = None,
                  verbose=False, reuse_forward=True, device_leapfrog=False, device_id=None,
                  checkpoint=None, checkpoint_every=0):
    """Returns (hmcmodel[nparam, nsamples], hmcstats, hmcdata[ndata, nsamples+1]).  The chain runs on GPU `device_id`
    (default: `default_device()`, i.e. LOCAL_RANK) unless a context is passed in.

    `checkpoint` (a file path) with `checkpoint_every` = k > 0: the chain's state -- samples so far, statistics, current
    model and momentum, the RNG state -- is flushed every k samples; if the file exists when the sampler starts, the
    chain RESUMES behind its last flushed sample and draws the same random numbers an uninterrupted run would have
    (the reference keeps every sample in memory until the chain ends, HMCSampler.jl:785-828: a crash loses the run;
    SURVEY section 5 lists this as the one piece of fault tolerance worth adding).  The resumed chain is the
    uninterrupted chain to solver tolerance, not bit for bit, on a real GPU context: the iterative solves of the first
    resumed trajectory start cold instead of from the interrupted run's field history (bit-identical with a
    deterministic direct-solve context, tests/test_host.py).  A checkpoint of a different run -- other sizes, dt,
    timestep, regParam, sigBounds, data, weights or reference model -- is refused."""
    _check_solver(hmcprior)
    rng = rng or np.random.default_rng()
    ctx = ctx or get_context(mtMesh, mtData, invParam, device_id=device_id)
    nparam, ndata = len(invParam.strModel), len(invParam.obsData)
    if hmcprior.massType != "diagonal":
        raise NotImplementedError("only the reference's default diagonal mass matrix is supported "
                                  "(the dense variant needs nparam^2 memory, SURVEY App. B.14)")
    cur = initHMCParameter(nparam)
    cur.invM, cur.sqrtM = setMassMatrix(nparam, 1.0)
    cur.rhomodel = invParam.strModel.copy()               # file start model stays the chain state (:88)
    cur.momentum = getMomentumVector(nparam, cur, rng)
    prop = HMCParameter(nparam, cur.rhomodel.copy(), cur.momentum.copy(), cur.invM, cur.sqrtM)
    # random homogeneous start / reference model (:100-109)
    rho0 = 1.0 / np.exp(invParam.strModel[0])
    if rhoref is None:
        rhoref = np.round(rho0 * 0.5 + (rho0 * 1.5 - rho0 * 0.5) * rng.random())
    if verbose:
        print(f"Homogeneous starting model with a resistivity of {rhoref} Ωm is used.")
    strModel = np.log(np.ones(nparam) / rhoref)
    invParam.strModel = strModel.copy()
    invParam.refModel = strModel.copy()
    startD, startK, startH, startM, predData = getHamiltonian(mtData, mtMesh, invParam, hmcprior, cur, ctx,
                                                              reuse_forward)
    if device_leapfrog:
        ctx.set_prior(invParam.refModel, invParam.Wm, cur.invM)
    nsamples = hmcprior.totalsamples
    hmcmodel = np.zeros((nparam, nsamples))
    hmcdata = np.zeros((ndata, nsamples + 1), dtype=np.complex128)
    stats: HMCStatus = initHMCStatus(nsamples)
    stats.hmstats[:, 0] = [startD, startM, startK, startH]
    hmcdata[:, 0] = predData
    first = 1
    flushed = 0
    if checkpoint:
        import os
        if os.path.exists(checkpoint):
            done, (startD, startM, startK, startH) = _load_checkpoint(checkpoint, hmcmodel, hmcdata, stats, cur, rng,
                                                                      invParam, hmcprior)
            first = done + 1
            flushed = done
            if verbose:
                print(f"resuming behind sample {done} of {nsamples} ({checkpoint})")
    for it in range(first, nsamples + 1):
        propose = proposeLeapfrogDevice if device_leapfrog else proposeLeapfrog
        propModel, propMomentum = propose(cur, mtMesh, mtData, invParam, hmcprior, rng, None, ctx)
        prop.rhomodel, prop.momentum = propModel.copy(), propMomentum.copy()
        finishD, finishK, finishH, finishM, predData = getHamiltonian(mtData, mtMesh, invParam, hmcprior, prop,
                                                                      ctx, reuse_forward)
        hdif = startH - finishH
        aratio = rng.random()
        if hdif > 0 or aratio < np.exp(hdif):
            cur.rhomodel, cur.momentum = propModel.copy(), propMomentum.copy()
            startD, startM = finishD, finishM
            stats.nAccept += 1
            stats.acceptstats[it - 1] = True
            hmcdata[:, it] = predData
        else:
            stats.nReject += 1
            hmcdata[:, it] = hmcdata[:, it - 1]
        if verbose:
            print(f"iterNo={it:6d} dtMisfit={finishD:8.3e} mNorm={finishM:8.3e} KEnergy={finishK:8.3e} "
                  f"HEnergy={finishH:8.3e} {'accepted' if stats.acceptstats[it - 1] else 'rejected'} "
                  f"p={min(1.0, float(np.exp(min(hdif, 0.0)))):.3f}")
        cur.momentum = getMomentumVector(nparam, cur, rng)
        startK = getKineticEnergy(cur.momentum, cur)
        startH = startD + startM + startK
        stats.hmstats[:, it] = [startD, startM, startK, startH]
        hmcmodel[:, it - 1] = cur.rhomodel
        if checkpoint and checkpoint_every > 0 and (it % checkpoint_every == 0 or it == nsamples):
            _save_checkpoint(checkpoint, it, flushed, hmcmodel, hmcdata, stats, cur, (startD, startM, startK, startH), rng,
                             invParam, hmcprior)
            flushed = it
    return hmcmodel, stats, hmcdata


# --------------------------------------------------------------------------------------------
def parallelHMCSampler(mtMesh, mtData, invParam, hmcprior, pids=None, seed=0, outdir=None, nchains=None,
                       run_chain=None, device_id=None, context_factory=None, chains_per_gpu=1, gather="torch", **sampler_kw):
    """Independent chains, one process per GPU (parallelHMC.jl:10-49).

    With `torch.distributed` initialised (backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in
    CPU tests) rank r runs chains r, r+W, ... on ITS OWN GPU -- `device_id`, default `default_device()` =
    LOCAL_RANK; with the nccl backend the process's current device is set to it before the collective --
    with RNG stream (seed, chain), and the sample blocks are ALL-GATHERED so every rank holds every chain;
    rank 0 writes the per-chain files the reference writes.  Without a process group the chains run one
    after another.  `pids` is kept for signature parity: its length is the number of chains (default:
    world size).  `context_factory(mesh, data, inv, device_id)` replaces `get_context` (the CPU tests run
    the real sampler on an oracle-backed stand-in); `run_chain(chain_index, rng)` replaces the sampler.
    `chains_per_gpu` = 2 or 4 runs that many of the rank's chains CONCURRENTLY on its GPU, one context and one host
    thread each, every context confined to its share of the CUs of every XCD (HipContext(cu_share=...),
    hmcmt_next_cu_share: CU-masked streams): the persistent solve kernels of the chains are co-resident, each with its share of
    the system slots.  Measured at the headline size near the true model (bench.py `two_chains_per_gpu`,
    scripts/gpu_cu_share_probe.py): two chains on halves 1.13-1.15x the aggregate steps/s of one chain on the whole device, four
    on quarters 1.17x.  (The shares' streams are BLOCKING HIP streams: keep other GPU work of the process off the legacy
    default stream while chains sample, or it serialises them -- DESIGN 7.)  A mesh whose systems need
    more workgroups than a share's CUs per XCD hold (the stress size: 30 of 16) runs the launch-per-phase loop in each share.  The chains
    and their results are the same as run one after another (independent contexts, per-chain RNG streams; the same solver, so
    the same bits).  Rounds 2-3 ran the concurrent contexts on the launch-per-phase loop (1.34x then); round 4 had no mode that
    composed with the persistent kernel (0.95x).
    `gather="library"`: the sample blocks are all-gathered by the library's own RCCL entry point
    (hmcmt_allgather_samples, include/hmcmt.h -- what a non-Python host would call) instead of torch's; the process
    group then only carries the 128-byte RCCL id from rank 0 to the others.  Works without a process group too (one rank).
    Further keyword arguments go to runHMCSampler (e.g. device_leapfrog=True; `checkpoint=path` becomes
    `path.chain<k>` per chain).
    Returns (hmcmodel[list], hmcstats[list], hmcdata[list]) indexed by chain.
    """
    import copy
    try:
        import torch
        import torch.distributed as dist
        have_pg = dist.is_available() and dist.is_initialized()
    except Exception:                                       # pragma: no cover
        have_pg = False
    world, rank = (dist.get_world_size(), dist.get_rank()) if have_pg else (1, 0)
    dev_id = default_device() if device_id is None else int(device_id)
    use_cuda = have_pg and dist.get_backend() == "nccl"
    if use_cuda:
        torch.cuda.set_device(dev_id)                       # the collective below runs on this rank's own GPU
    if nchains is None:
        nchains = len(pids) if pids is not None else world
    mine = list(range(rank, nchains, world))
    results = {}

    # shares of every XCD's CUs = chains that actually run at a time: the largest of (4, 2, 1) that neither exceeds chains_per_gpu nor
    # the chains this rank has (chains_per_gpu = 4 with two local chains: halves, not quarters -- a quarter to a half of the GPU would
    # idle for the whole run), and that many worker threads (three chains on two shares: two at a time, then the third)
    shares = cu_shares_for(chains_per_gpu, len(mine))
    if int(chains_per_gpu) > 1 and int(chains_per_gpu) not in (2, 4):
        import warnings
        warnings.warn("chains_per_gpu must be 1, 2 or 4 (shares of every XCD's CUs): other values run concurrent contexts on the "
                      "launch-per-phase loop, slower than one chain after another")
    import threading
    free_shares, share_lock = list(range(shares)), threading.Lock()

    def one_chain(c):
        rng = np.random.default_rng([seed, c])
        t0 = time.time()
        if run_chain is not None:
            model, stats, data = run_chain(c, rng)
        else:
            inv_c, prior_c, mesh_c = copy.deepcopy(invParam), copy.deepcopy(hmcprior), copy.deepcopy(mtMesh)
            with share_lock:
                my_share = free_shares.pop(0) if shares > 1 else None
            try:
                # (the context is created INSIDE the try: a create that raises hands its share back, and the next chain fails with
                #  the real error instead of an IndexError on an empty list)
                ctx_kw = {} if my_share is None else {"cu_share": (my_share, shares)}
                ctx_c = context_factory(mesh_c, mtData, inv_c, dev_id) if context_factory is not None else \
                    get_context(mesh_c, mtData, inv_c, device_id=dev_id, **ctx_kw)
                kw = dict(sampler_kw)
                if kw.get("checkpoint"):                        # one checkpoint file per chain
                    kw["checkpoint"] = f"{kw['checkpoint']}.chain{c + 1}"
                model, stats, data = runHMCSampler(mesh_c, mtData, inv_c, prior_c, rng, ctx=ctx_c, **kw)
            finally:
                release_context(inv_c)
                if my_share is not None:
                    with share_lock:
                        free_shares.append(my_share)
        results[c] = (model, stats, data, time.time() - t0)

    if chains_per_gpu > 1 and len(mine) > 1:
        from concurrent.futures import ThreadPoolExecutor
        workers = shares if int(chains_per_gpu) in (2, 4) else int(chains_per_gpu)
        with ThreadPoolExecutor(max_workers=workers) as pool:      # (the library calls release the GIL)
            list(pool.map(one_chain, mine))
    else:
        for c in mine:
            one_chain(c)

    nparam, nsamples = next(iter(results.values()))[0].shape if results else (len(invParam.strModel), hmcprior.totalsamples)
    ndata = len(invParam.obsData)
    per = (nchains + world - 1) // world                   # chain slots per rank (padded)
    blk = nparam * nsamples + 4 * (nsamples + 1) + nsamples + 2 * ndata * (nsamples + 1) + 2

    def pack(slot):
        c = rank + slot * world
        buf = np.zeros(blk)
        if c in results:
            model, stats, data, secs = results[c]
            o = 0
            buf[o:o + model.size] = model.reshape(-1); o += model.size
            buf[o:o + stats.hmstats.size] = stats.hmstats.reshape(-1); o += stats.hmstats.size
            buf[o:o + nsamples] = stats.acceptstats.astype(float); o += nsamples
            buf[o:o + 2 * data.size] = data.reshape(-1).view(np.float64); o += 2 * data.size
            buf[o] = secs; buf[o + 1] = 1.0
        return buf

    local = np.concatenate([pack(s) for s in range(per)]) if per else np.zeros(0)
    if gather == "library":
        from .lib import SampleComm
        uid = [SampleComm.unique_id() if rank == 0 else None]
        if have_pg and world > 1:
            dist.broadcast_object_list(uid, src=0)
        comm = SampleComm(dev_id, world, rank, uid[0])
        allbuf = comm.allgather(local).reshape(world, per, blk)
        comm.close()
    elif have_pg and world > 1:
        dev = torch.device("cuda", dev_id) if use_cuda else torch.device("cpu")
        send = torch.from_numpy(local).to(dev)
        recv = torch.empty(world * local.size, dtype=torch.float64, device=dev)
        dist.all_gather_into_tensor(recv, send)             # RCCL ring over xGMI on the GPU box
        allbuf = recv.cpu().numpy().reshape(world, per, blk)
    else:
        allbuf = local.reshape(1, per, blk)

    hmcmodel, hmcstats, hmcdata, secs = [None] * nchains, [None] * nchains, [None] * nchains, [0.0] * nchains
    for r in range(world):
        for s in range(per):
            c = r + s * world
            if c >= nchains:
                continue
            buf = allbuf[r, s]
            o = 0
            model = buf[o:o + nparam * nsamples].reshape(nparam, nsamples).copy(); o += nparam * nsamples
            hm = buf[o:o + 4 * (nsamples + 1)].reshape(4, nsamples + 1).copy(); o += 4 * (nsamples + 1)
            acc = buf[o:o + nsamples] > 0.5; o += nsamples
            data = buf[o:o + 2 * ndata * (nsamples + 1)].copy().view(np.complex128).reshape(ndata, nsamples + 1)
            o += 2 * ndata * (nsamples + 1)
            secs[c] = float(buf[o])
            hmcmodel[c], hmcdata[c] = model, data
            hmcstats[c] = HMCStatus(int(acc.sum()), int((~acc).sum()), acc, hm)
    if outdir is not None and rank == 0:
        from .fileio import outputHMCSamples
        for c in range(nchains):
            outputHMCSamples(hmcmodel[c], hmcstats[c], hmcdata[c], ichain=c + 1, cputime=secs[c], outdir=outdir)
    return hmcmodel, hmcstats, hmcdata
