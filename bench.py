#!/usr/bin/env python3
"""Headline benchmark: leapfrog steps/s (= forward + gradient evaluations/s) on the 200x100-cell,
16-frequency synthetic of BASELINE.json, one independent HMC chain per GPU.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config cfg3]

A step is one leapfrog step of a REAL trajectory of the sampler (proposeLeapfrog, HMCSampler.jl:206-269): position
update with the step clamp and the bound reflection, one pass of the hot path -- compDataGradient (:277-330) -- at
the new model, prior gradient lambda*Wm*(m - mref), momentum update; trajectories of L = 8 steps (timestep 6..10 in
examples/dprism3d/startupfile), dt = 0.03, momentum redrawn per trajectory from the clipped N(0,1) of
getMomentumVector (:441-453), accept / reject on the Hamiltonian (:149-171).  Model, momentum, gradient, predicted
data stay in HBM (hmcmt_leapfrog_device); per trajectory three scalars come back for the accept test.  The gradient
at a trajectory's start model is the one of the previous trajectory's end (accepted) or start (rejected) model, so a
trajectory costs L evaluations; the timed region holds exactly K of them.

  value               chain started at the rough state of SURVEY 8(d): m = ln 0.01 + 0.3 N(0,1) (seed 1) -- the
                      burn-in regime (misfit 8e5, gradients 1e5: the step clamp and the bounds are active)
  near_true_state     the same chain started at the synthetic's true model: the regime a converged chain samples in
  straight_line       round 1's idealised figure: evaluations along m0 + k*dt*p (flatters the initial-guess extrapolation)
  cold_start          evaluations at the rough state's neighbourhood with zero initial guesses (options.warm_start = 0)

For N > 1 the driver launches this file with torch.distributed.run (one rank per GPU, RCCL); ranks are independent
chains (weak scaling), the only collective in the timed region is the barrier; after it the ranks all-gather a block
of samples (what parallelHMCSampler does at the end of a run) and report its bandwidth.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0             # MI355X HBM3E peak (MI355X_MICROARCH.md: 8 TB/s spec, ~6.3 TB/s achievable)
LTRAJ, DT, LAMBDA = 8, 0.03, 1.0
RHO_BOUNDS = (1.0, 1e4)           # examples/dprism3d/startupfile:5


def build_problem(name):
    """Synthetic workload of SURVEY §8(d): observed data = GPU forward of the true model + 3 % noise."""
    from hmcmt2d_amd import synthetic as S, invsetup as I
    mesh, data, sig_true = S.make_config(name)
    ny, nz = mesh.gridSize
    nair = len(mesh.airLayer)
    mesh.sigma = np.concatenate([np.full(ny * nair, S.SIG_AIR), np.full(ny * (nz - nair), 0.01)])
    n = len(data.rxID)
    inv0 = I.setupInverseDataModel(mesh, [S.SIG_AIR], 0.0, 0.0, np.zeros(n, complex), np.ones(n))
    return mesh, data, inv0, sig_true


def _cpu_worker(job):
    """One worker process = one core: the oracle's compDataGradient restricted to the given frequencies."""
    name, fs = job
    from threadpoolctl import threadpool_limits
    from oracle import hmcmt_oracle as O
    from hmcmt2d_amd import synthetic as S, invsetup as I
    from hmcmt2d_amd.structs import HMCPrior
    mesh, data, _ = S.make_config(name)
    d1 = S.make_data_layout(list(fs), data.rxLoc[:, 0])
    ny, nz = mesh.gridSize
    nair = len(mesh.airLayer)
    mesh.sigma = np.concatenate([np.full(ny * nair, S.SIG_AIR), np.full(ny * (nz - nair), 0.01)])
    n = len(d1.rxID)
    inv = I.setupInverseDataModel(mesh, [S.SIG_AIR], 0.0, 0.0, np.full(n, 0.02 + 0.02j), np.full(n, 1e-3))
    inv.strModel = S.rough_state(len(inv.strModel))
    with threadpool_limits(limits=1):
        O.setupTensorMesh2D(mesh)
        t0 = time.time()
        O.compDataGradient(mesh, d1, inv, HMCPrior(), True)      # dense dBC, as the reference forms it
        return time.time() - t0


def cpu_baseline(name):
    """Oracle (numpy/scipy restatement, SuperLU direct solves, dense dBC) on a bounded sample, using every
    host core: one single-threaded worker process per core, each evaluating compDataGradient on its own
    frequencies (TE + TM) of the config -- the frequency loop is the reference's only parallelisable axis
    (MT2DFwdSolver.jl:140-146, compJacTMatVec.jl:202-325).  value = frequencies done / wall time / nFreq.
    Runs BEFORE this process touches the GPU (the workers are spawned, not forked)."""
    import multiprocessing as mp
    from hmcmt2d_amd import synthetic as S
    _, data, _ = S.make_config(name)
    nF = len(data.freqs)
    cores = max(1, min(len(os.sched_getaffinity(0)), nF))
    per = 2                                                      # frequencies per worker: ~10-20 core-seconds in all
    jobs = [(name, tuple(float(data.freqs[((j * nF) // cores + i * (nF // per)) % nF]) for i in range(per)))
            for j in range(cores)]
    ctx = mp.get_context("spawn")
    with ctx.Pool(cores) as pool:
        t0 = time.time()
        times = pool.map(_cpu_worker, jobs, chunksize=1)
        wall = time.time() - t0
    tmax = max(times)
    return {"value": per * len(jobs) / (tmax * nF), "unit": "steps/s", "cores": cores, "kind": "port",
            "sample": f"oracle compDataGradient (numpy/scipy, SuperLU direct solves, dense dBC as the reference) at "
                      f"{per} of {nF} frequencies (TE+TM) of {name} per worker, one single-threaded worker process per core on "
                      f"{cores} cores concurrently: slowest worker {tmax:.2f} s (sum {sum(times):.1f} core-s, wall "
                      f"incl. start-up {wall:.1f} s), scaled to {nF} frequencies"}


def rocprof_avg_us(config, cat):
    """rocprofv3 --stats average duration (us) of the kernels of a category in the committed profile of this config
    (profiles/pmc_traffic.json `rocprof_avg_us`, written by scripts/make_profile_summary.py); None if absent."""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
            return json.load(f).get(config, {}).get("rocprof_avg_us", {}).get(cat)
    except (OSError, ValueError):
        return None


def pmc_traffic(config, cat):
    """HBM-side bytes per launch from rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE collected in separate runs of
    this same command, FETCH_SIZE doubled per MI355X_MICROARCH.md 'HBM'), summarised by scripts/pmc_summary.py into
    profiles/pmc_traffic.json keyed by config; None when that file has no entry for this config."""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
            return json.load(f).get(config, {}).get("per_launch_bytes", {}).get(cat)
    except (OSError, ValueError):
        return None


from hmcmt2d_amd.lib import HmcmtError  # noqa: E402  (pure ctypes: does not touch the GPU at import)


class Chain:
    """The sampler of HMCSampler.jl:72-196 around hmcmt_leapfrog_device: everything O(nparam) stays on the GPU
    (torch tensors), the accept test reads three scalars per trajectory."""

    def __init__(self, ctx, torch, dev, m_start, mref, Wm, seed):
        self.ctx, self.torch, self.dev = ctx, torch, dev
        n = ctx.nAC
        self.gen = torch.Generator(device=dev); self.gen.manual_seed(seed)
        self.host_rng = np.random.default_rng([seed, 99])
        self.m_cur = torch.from_numpy(np.ascontiguousarray(m_start)).to(dev)
        self.m_prop = self.m_cur.clone()
        self.p = torch.zeros(n, dtype=torch.float64, device=dev)
        self.d_pred = torch.zeros(2 * ctx.nData, dtype=torch.float64, device=dev)
        self.scal = torch.zeros(2, dtype=torch.float64, device=dev)          # [misfit, mnorm] of the proposal
        self.ham = torch.zeros(4, dtype=torch.float64, device=dev)           # [K0, K1, D1, M1] read back together
        self.lo, self.hi = float(np.log(1.0 / RHO_BOUNDS[1])), float(np.log(1.0 / RHO_BOUNDS[0]))
        ctx.set_prior(mref, Wm, np.ones(n))
        # Hamiltonian terms at the start model (getHamiltonian, :358-397)
        d = m_start - mref
        self.M0 = 0.5 * LAMBDA * float(d @ (Wm @ d))
        ctx.forward_device(self.m_cur.data_ptr(), self.d_pred.data_ptr(), self.scal.data_ptr())
        self.D0 = float(self.scal[0].item())
        self.start_grad = 0
        self.accepted = self.rejected = self.failed = 0
        self.last_error = None
        self.iters = []
        self.ms_per_step = []

    def trajectory(self, L):
        torch, ctx = self.torch, self.ctx
        t_begin = time.perf_counter()
        self.p.normal_(generator=self.gen).clamp_(-2.5, 2.5)
        self.ham[0] = 0.5 * (self.p * self.p).sum()                          # kinetic energy at the start (stays on the device)
        self.m_prop.copy_(self.m_cur)
        torch.cuda.current_stream().synchronize()                            # torch's stream before the library's
        try:
            ctx.leapfrog_device(self.m_prop.data_ptr(), self.p.data_ptr(), DT, L, LAMBDA, self.lo, self.hi, self.start_grad,
                                self.d_pred.data_ptr(), self.scal.data_ptr(), self.scal.data_ptr() + 8)
            ctx.wait()
        except HmcmtError as e:                       # a proposal the solver gave up on is a rejected proposal
            self.failed += 1
            self.rejected += 1
            self.start_grad = 0
            self.last_error = str(e)
            return None
        st = ctx.stats()
        self.iters.append((st["iters_fwd_max"], st["iters_adj_max"], st["fallback_solves"]))
        self.ms_per_step.append(1e3 * (time.perf_counter() - t_begin) / L)
        self.ham[1] = 0.5 * (self.p * self.p).sum()
        self.ham[2:4] = self.scal
        K0, K1, D1, M1 = self.ham.tolist()                                    # the trajectory's one read-back
        hdif = self.D0 + self.M0 + K0 - (D1 + M1 + K1)
        if hdif > 0 or self.host_rng.random() < np.exp(hdif):
            self.m_cur, self.m_prop = self.m_prop, self.m_cur
            self.D0, self.M0 = D1, M1
            self.start_grad = 1
            self.accepted += 1
        else:
            self.start_grad = 2
            self.rejected += 1
        return D1

    def run(self, nsteps):
        """exactly nsteps leapfrog steps (the last trajectory is shorter if need be)"""
        left = nsteps
        while left > 0:
            L = min(LTRAJ, left)
            self.trajectory(L)
            left -= L

    def summary(self):
        it = np.array(self.iters) if self.iters else np.zeros((1, 3))
        ms = self.ms_per_step[-len(self.iters):] if self.iters else []
        return {"trajectories": len(self.iters), "accepted": self.accepted, "rejected": self.rejected,
                # samples/s = trajectories/s (a sample costs L new evaluations here; the reference pays
                # L + 1 gradients and one more forward solve for the same sample, HMCSampler.jl:136,141)
                "samples_per_s": (1e3 / (LTRAJ * float(np.mean(ms)))) if ms else None,
                "iters_fwd_max_last_step_mean": float(it[:, 0].mean()), "iters_adj_max_last_step_mean": float(it[:, 1].mean()),
                "fp64_restarts": int(it[:, 2].sum()), "misfit_last": self.D0, "failed_trajectories": self.failed,
                "ms_per_step_by_trajectory": [round(x, 3) for x in self.ms_per_step[-len(self.iters):]],
                "last_error": self.last_error}


def timed(torch, dist, fn):
    import gc
    gc.collect()
    gc.disable()                      # (no collector pauses inside a timed region)
    try:
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        return time.perf_counter() - t0
    finally:
        gc.enable()


def two_chains_leg(torch, HipContext, mesh, data, inv, local, dev, m_true, mref, Ke):
    """(v) two chains on this GPU at once (what parallelHMCSampler(chains_per_gpu=2) does): two contexts, each confined to half
    the CUs of every XCD (hmcmt_next_cu_share: CU-masked streams, half the system slots of the persistent kernel each), one host
    thread per chain; aggregate steps/s of both chains near the true model.  Runs with the bench's own context closed: a context
    on the whole device overlaps both shares, and all three would use the launch-per-phase loop."""
    import threading
    ctxa = HipContext(mesh, data, inv, device_id=local, cu_share=(0, 2))
    ctxb = HipContext(mesh, data, inv, device_id=local, cu_share=(1, 2))
    ca = Chain(ctxa, torch, dev, m_true, mref, inv.Wm, seed=7)
    cb_ = Chain(ctxb, torch, dev, m_true, mref, inv.Wm, seed=8)
    # Each thread creates its stream, runs two warm trajectories, meets the other at a barrier and THEN times a FIXED number of
    # steps (Ke = 48 whatever --steps is; VERDICT r5: timed from thread start with the stream creation and the first trajectory's
    # transients inside, 20 steps per chain, the leg read 0.62x of one chain where a 48-step run reads 1.15x).  The timed region of
    # a chain is its own: per-chain ms/step; the aggregate is both chains' steps over the span from the common start to the last end.
    gate = threading.Barrier(2)
    span = {}

    def run_on_own_stream(c, key):
        # CU-masked streams are BLOCKING streams (hipExtStreamCreateWithCUMask takes no flags): torch work on the legacy default
        # stream would wait for the other chain's trajectory in flight, and make it wait -- each chain's torch ops (momentum draw,
        # kinetic energies) go to a stream of its own, as a multi-chain host would arrange it
        st = torch.cuda.Stream(device=dev)
        with torch.cuda.stream(st):
            c.run(2 * LTRAJ)
            st.synchronize()
            gate.wait()
            t0 = time.perf_counter()
            c.run(Ke)
            st.synchronize()
            span[key] = (t0, time.perf_counter())
    th = [threading.Thread(target=run_on_own_stream, args=(c, i)) for i, c in enumerate((ca, cb_))]
    torch.cuda.synchronize()
    import gc
    gc.collect()
    gc.disable()                      # (no collector pauses inside the timed region, as in timed(): a collection holds the GIL -- BOTH chains' host
    try:                              #  threads -- for 35-70 ms: one run in seven read 530-650 instead of 840-880 steps/s, one trajectory of both chains at 6-11 ms per step)
        for t in th:
            t.start()
        for t in th:
            t.join()
    finally:
        gc.enable()
    torch.cuda.synchronize()
    t5 = max(e for _, e in span.values()) - min(b for b, _ in span.values())
    res = {"steps_per_s_aggregate": 2 * Ke / t5, "steps_per_chain": Ke,
           "ms_per_step_by_chain": [1e3 * (e - b) / Ke for b, e in (span[0], span[1])],
           "timed": "inside each chain's thread, behind its stream's creation, two warm trajectories and a barrier with the other chain",
           "persistent_solves": [ctxa.persist_info()["solves"], ctxb.persist_info()["solves"]],
           "timeouts_and_placement_fallbacks": [[c.persist_info()["timeouts"], c.persist_info()["placement_fallbacks"]] for c in (ctxa, ctxb)],
           "ms_per_step_by_trajectory": [[round(x, 2) for x in c.ms_per_step[-(Ke // LTRAJ):]] for c in (ca, cb_)],
           "iters_last_step_by_trajectory": [[list(t[:2]) for t in c.iters[-(Ke // LTRAJ):]] for c in (ca, cb_)],
           "slots_per_xcd": [ctxa.persist_info()["slots_per_xcd"], ctxb.persist_info()["slots_per_xcd"]],
           "state": "two independent chains near the true model on ONE GPU: two contexts, two host threads, "
                    "each context on half the CUs of every XCD (CU-masked streams), their persistent solve "
                    "kernels side by side with half the system slots each, each chain's torch ops on a stream of its own; "
                    "compare with near_true_state (ONE chain on the whole device)"}
    ctxa.close()
    ctxb.close()
    return res


def spawn_ranks(n):
    """N child processes, one rank per GPU (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* as torch.distributed.run sets them), started
    BEFORE this process has made any GPU call (it never makes one: no exec of a process that has initialised the GPU).  Rank 0's
    stdout -- the one JSON line -- is this process's stdout; the exit code is the first non-zero one of the ranks."""
    import socket
    import subprocess
    # (a free port by bind-and-close: another process can take it before rank 0 binds it -- MASTER_PORT in the environment overrides
    #  it, and a rendezvous that fails on "address in use" shows up as rank 0's exit within the poll loop below, not as a hang)
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=os.environ.get("MASTER_PORT", str(port)), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    # all ranks polled in one loop: a rank r > 0 that dies while rank 0 sits in init_process_group or a barrier is seen at once
    # (waiting for the ranks in order, rank 0's wait blocked until the store / NCCL timeout -- minutes -- before the failure showed)
    rc = 0
    left = dict(enumerate(procs))
    while left:
        for r, p in list(left.items()):
            c = p.poll()
            if c is None:
                continue
            del left[r]
            if c != 0 and rc == 0:
                rc = c if c > 0 else 1
                print(f"bench.py: rank {r} exited with {c}", file=sys.stderr)
                for q in left.values():              # (the others would wait at the barrier for ever)
                    q.terminate()
        if left:
            time.sleep(0.05)
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=96)
    ap.add_argument("--warmup", type=int, default=16)
    ap.add_argument("--config", default="cfg3")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the untimed-region legs (near_true_state, straight_line, cold_start)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N` (parallelHMC.jl:10-49 is one call, too): this process touches no GPU, it starts N rank
        # processes with the environment torch.distributed.run would give them and hands rank 0's JSON line through
        raise SystemExit(spawn_ranks(args.gpus))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # stdout carries exactly ONE line, the JSON: whatever libraries print there (RCCL's version banner under
    # NCCL_DEBUG=VERSION goes to the C stdout) is sent to stderr for the life of the process
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    cpu = None
    if world == 1 and rank == 0 and not args.no_cpu_baseline:
        cpu = cpu_baseline(args.config)          # before this process touches the GPU: the workers are spawned
    import torch
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}, or without it")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    torch.cuda.set_device(local)
    dist = None
    if world > 1 or os.environ.get("HMCMT_BENCH_FORCE_PG"):      # (FORCE_PG: exercise the RCCL calls with a single rank)
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))

    from hmcmt2d_amd import synthetic as S, invsetup as I
    from hmcmt2d_amd.lib import HipContext
    name = args.config
    mesh, data, inv0, sig_true = build_problem(name)
    # observed data from the GPU forward of the true model (+ seeded noise), then the real context
    ctx0 = HipContext(mesh, data, inv0, device_id=local)
    m_true = np.log(sig_true[inv0.activeIdx])
    pred_true, _ = ctx0.forward(m_true)
    ctx0.close()
    obs, err = S.noisy_observations(pred_true)
    inv = I.setupInverseDataModel(mesh, [S.SIG_AIR], 0.0, 0.0, obs, err)
    ctx = HipContext(mesh, data, inv, device_id=local)
    nAC, nData = ctx.nAC, ctx.nData
    dev = torch.device("cuda", local)
    mref = np.full(nAC, np.log(0.01))            # homogeneous 100 Ohm-m reference / start model (HMCSampler.jl:100-109)
    K, W = args.steps, args.warmup

    # ---- headline: real trajectories from the rough state -------------------------------------------------------
    chain = Chain(ctx, torch, dev, S.rough_state(nAC, seed=1 + rank), mref, inv.Wm, seed=20250114 + rank)
    # warm-up: the W steps asked for, but never less than two WHOLE trajectories (the first one starts from a context without history --
    # cold solves, the sweep count not yet chosen --, and a 5-step warm-up leaves the timed region starting in mid-trajectory)
    Wrun = max(W, 2 * LTRAJ)
    chain.run(Wrun)
    chain.iters.clear(); acc0, rej0 = chain.accepted, chain.rejected
    prof_every, prof_overhead_us = 0, 0.0
    if not os.environ.get("HMCMT_BENCH_NOPROF"):
        # HIP events between the launches of the iteration kernels, in every n-th evaluation of the timed region, n chosen so
        # that four evaluations (~200 iterations, ~800 launches) are sampled whatever K is: the driver's 20-step run samples
        # every 5th, the default 96-step run every 25th.  An event is a marker packet in the queue: 3 us per sampled launch,
        # +24 % on a sampled evaluation -- 20 steps with 7 sampled evaluations ran at 301 steps/s, without sampling at 337
        # (round 2 sampled every 6th = 3 evaluations of a 20-step run, with two events per launch: the same cost)
        prof_every = max(1, K // 4) | 1            # (odd: a trajectory has 8 steps -- an even interval would sample the same steps of every trajectory)
        ctx.profile(["fdm_transform", "tridiagonal", "spmv", "vector_ops", "post_smoother"], every=prof_every)
    if prof_every:
        prof_overhead_us = ctx.profile_overhead_us()
    elapsed = timed(torch, dist, lambda: chain.run(K))
    prof = ctx.profile_read()
    cnt = ctx.profile_counters()
    ctx.profile(False)
    st = ctx.stats()
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    head = chain.summary()
    head["accepted"] -= acc0; head["rejected"] -= rej0

    # one evaluation of the timed chain's last model again with the true-residual check on (cold start, verify):
    # the numbers the timed region produced are solutions of the systems it claims to have solved
    ctx.set_options(verify=1)
    d_g = torch.zeros(nAC, dtype=torch.float64, device=dev)
    d_mis = torch.zeros(1, dtype=torch.float64, device=dev)
    ctx.grad_device(chain.m_cur.data_ptr(), chain.d_pred.data_ptr(), d_mis.data_ptr(), d_g.data_ptr())
    stv = ctx.stats()
    check = {"true_res_max_at_last_model": stv["true_res_max"], "misfit_at_last_model": float(d_mis.item()),
             "chain_misfit": head["misfit_last"], "solver_status": stv["status"],
             "grad_l2_at_last_model": float(torch.linalg.vector_norm(d_g).item())}
    ctx.set_options(verify=0)

    # ---- RCCL all-gather of a sample block (what parallelHMCSampler does with the chains' samples, parallelHMC.jl:23-45)
    gather = None
    if dist is not None:
        ks = 64                                                   # samples per rank in the block
        send = torch.randn(ks * nAC, dtype=torch.float64, device=dev)
        recv = torch.empty(world * ks * nAC, dtype=torch.float64, device=dev)
        dist.all_gather_into_tensor(recv, send)                   # warm-up (communicator set-up)
        tg = timed(torch, dist, lambda: dist.all_gather_into_tensor(recv, send))
        ok = bool(torch.equal(recv[rank * ks * nAC:(rank + 1) * ks * nAC], send))
        gather = {"samples_per_rank": ks, "bytes_per_rank": ks * nAC * 8, "seconds": tg,
                  "algbw_GBps": world * ks * nAC * 8 / tg / 1e9, "own_block_intact": ok,
                  "note": "all_gather_into_tensor of k x nparam float64 sample blocks over RCCL, outside the timed region"}

    extras = {}
    if world == 1 and not args.no_extras:            # (single-GPU runs only: keeps the ranks of an N-GPU run symmetric)
        Ke = min(K, 48)
        # (ii) the same sampler started at the true model
        c2 = Chain(ctx, torch, dev, m_true, mref, inv.Wm, seed=7)
        c2.run(max(W, 2 * LTRAJ))
        for _ in range(4):                # (... and until the chain has taken BOTH branches of the accept test once, six trajectories at most: the first
            if c2.accepted and c2.rejected:   #  accepted proposal is the first use of the device leapfrog's "start from the end gradient" path -- on a fresh
                break                     #  box its first launch cost 26 ms inside a 29 ms timed region: near_true_state read 360 for 690)
            c2.run(LTRAJ)
        c2.iters.clear(); a0, r0 = c2.accepted, c2.rejected     # (two trajectories: the first one starts from
        #  the headline chain's fields and iteration counts, a transient of the context's history, not of this state)
        t2 = timed(torch, None, lambda: c2.run(Ke))
        s2 = c2.summary(); s2["accepted"] -= a0; s2["rejected"] -= r0
        extras["near_true_state"] = dict(s2, steps_per_s=Ke / t2, steps=Ke,
                                         state="chain started at the true model (100 over 10 ohm-m + 10 ohm-m block)")
        # (iii) round 1's idealised straight-line trajectories
        rng = np.random.default_rng([20250114, 7, rank])
        m0 = S.rough_state(nAC, seed=1 + rank)
        traj = np.empty((Ke + W, nAC))
        for k in range(Ke + W):
            if k % LTRAJ == 0:
                p = np.clip(rng.standard_normal(nAC), -2.5, 2.5)
            traj[k] = m0 + DT * (k % LTRAJ) * p
        d_m = torch.from_numpy(traj).to(dev)

        def line(k0, k1):
            for k in range(k0, k1):
                ctx.grad_device_async(d_m[k].data_ptr(), chain.d_pred.data_ptr(), d_mis.data_ptr(), d_g.data_ptr())
            ctx.wait()
        line(0, W)
        t3 = timed(torch, None, lambda: line(W, W + Ke))
        s3 = ctx.stats()
        extras["straight_line"] = {"steps_per_s": Ke / t3, "steps": Ke, "iters_fwd_max": s3["iters_fwd_max"],
                                   "iters_adj_max": s3["iters_adj_max"],
                                   "state": "m0 + k*dt*p around the rough state, momentum redrawn every 8 steps (round 1's headline)"}
        # (iv) cold start: zero initial guesses
        ctx.set_options(warm_start=0)
        line(0, 2)
        t4 = timed(torch, None, lambda: line(W, W + min(Ke, 16)))
        s4 = ctx.stats()
        extras["cold_start"] = {"steps_per_s": min(Ke, 16) / t4, "steps": min(Ke, 16), "iters_fwd_max": s4["iters_fwd_max"],
                                "iters_adj_max": s4["iters_adj_max"], "state": "the straight-line models, options.warm_start = 0"}
        ctx.set_options(warm_start=2)
        # (iv-b) the iteration kernels with EVERY system active in every launch: cold solves that cannot converge
        # (tolerance 1e-200) cut off after 12 iterations -- the per-launch cost at full batch, beside the timed region's
        # figures at the real chains' 60-65 % activity
        ctx.set_options(warm_start=0, tol=1e-200, maxit=12)
        ctx.profile(["fdm_transform", "tridiagonal", "spmv", "vector_ops", "post_smoother"], every=1)
        for k in range(4):
            try:
                ctx.grad_device(d_m[k].data_ptr(), chain.d_pred.data_ptr(), d_mis.data_ptr(), d_g.data_ptr())
            except HmcmtError:
                pass                                     # (ENOCONV by construction)
        full_prof, full_cnt = ctx.profile_read(), ctx.profile_counters()
        ctx.profile(False)
        ctx.set_options(warm_start=2, tol=1e-11, maxit=2000)
    if rank == 0:
        # Algorithmic bytes per launch: U = S*(nz-1)*(ny-1) interior unknowns, complex128 = 16 B,
        # complex64 / split-bf16 = 8 B; the real stencil coefficients are shared by all frequencies of a mode (not
        # counted).  One preconditioned COCG iteration = these four launches on the fused path (five when the back
        # transform and the post-smoother run as separate kernels: wide meshes, HMCMT_FUSED_BACK=0):
        #   k_fdm_fwd      forward eigen-transform + tridiagonal solves: read t (8) + inverse pivots (8), write y (8) = 24 U
        #                  (+ 40 U since round 3: x += alpha p -- x in and out, p in -- by its waves that wait for the sweeps; the
        #                  preconditioner application in front of the first iteration has no x update)
        #   k_back_post    back transform + both Jacobi halves + dots: read y (8), dinv (16), r (16), write z (8)   = 48 U
        #                  (separate: k_transform_lp<2> 56 U with z1 written, k_post: read r, z1, dinv (48), write z (8) = 56 U)
        #   k_spmv_fused   p = z + beta p, q = A p, p'q: read z, p (8 + 8: complex64), write p (8), q (16)          = 40 U
        #   k_update_fused x, r updates + Jacobi pre-smoothing: read p (8), q, r, x, dinv (64), write x, r (32), t (8) = 112 U
        #                  (72 U with the x update in k_fdm_fwd)
        # A launch works on the systems still active; U_launch = U * (active systems / S), the active count from the
        # device counter of hmcmt_profile_counters over the SAME sampled launches the HIP events time (every launch of
        # every n-th evaluation of the timed region, the empty ones behind a convergence poll included).
        nyi, nzi = ctx.ny - 1, ctx.nz - 1
        Usys = nzi * nyi
        U = ctx.S * Usys
        ws_mb = 15 * ctx.S * ctx.NZP * ctx.NYP * 16 / 1e6

        # SURVEY 8(d)'s canonical figure beside it: what a CSR implementation (complex128 values + int32 column indices,
        # complex128 vectors, one pass per vector operation, nothing fused) would move for the operations a kernel performs:
        # B_spmv = nnz*(16+4) + (N+1)*4 + 2*16*N per system and stencil product, 16*N per vector pass.
        nnz_sys = 5 * Usys - 2 * (nyi + nzi)
        B_spmv = nnz_sys * 20 + (Usys + 1) * 4 + 32 * Usys
        # category -> (stencil products, vector passes) with one / two smoothing sweeps per side
        CANON = {"spmv": ((1, 5), (2, 11)),            # p = z + beta p (3), q = A p, p'q (2)   [two sweeps: + second post-sweep: A z4, 6 passes]
                 "vector_ops": ((1, 12), (2, 18)),     # x += a p (3), r -= a q (3), z1 = D r (3), t = r - A z1 (2), |x|^2 (1)   [+ A z1 ..., 6]
                 "tridiagonal": ((0, 3), (0, 3)),      # read t, pivots, write y (the eigen-transform itself has no CSR analogue)
                 "fdm_transform": ((1, 13), (1, 13)),  # read y (1), z0 = D r (3), z = V y + z0 (3), A z, 4 passes, r'z and |z|^2 (2)
                 "post_smoother": ((1, 6), (1, 6))}

        def persist_design_bytes(f2):
            """What k_cocg_persist's OWN design moves through global memory, per OWNED unknown: (per iteration, per application of
            the preconditioner in front of the first iteration).  r lives in registers, the tiles and the float stencil
            coefficients in LDS; what crosses the CU boundary is
              x                     read 16 + write 16 (fp64 complex, by its owner, once)
              r' (complex64)        published 8, the neighbours' halo rows read back: 2 x 5 halo rows per 14 own rows
              z2, p (complex64)     published 8 each; read back on the WHOLE tile (24 rows per 14 own rows, W_t columns per W_o own)
              t (two sweeps)        written 8 + read 8 (the rho identity behind the FDM stage)
              yhat                  the forward transform's rows: written 8 per column part, read by the slab owner
              inverse pivots        8 (complex64)
              solved slabs          written 8 (split bf16 planes), read back on the tile's 24 rows (every mode: no column factor)
              fp64 coefficients     dM, cY, cZ of the own rows for q = A p: 24 B per node and POLARISATION, shared by the systems of
                                    that polarisation an XCD runs at a time (half its slots)
            The bf16 eigenvector fragments (shared by every workgroup of the chip, 87-350 KB) are served by the L2 and not counted."""
            pi = ctx.persist_info()
            cs = max(int(pi.get("column_parts", 1)), 1)
            own_w = ctx.NYP / cs
            tile_w = own_w + (8 if cs > 1 else 0) * min(cs - 1, 2)       # halo columns of a column part (kernels_persist.h: PS_HC)
            rows, cols = 24.0 / 14.0, tile_w / own_w
            halo_r = (10.0 / 14.0) * cols + (cols - 1.0)                 # halo rows + (column split) the halo columns of the own rows
            common = 8 + 8 * halo_r + 8 + 8 * rows * cols + 8 * cs + 8 * cs + 8 + 8 + 8 * rows + 16 * f2
            it = 32 + common + 8 + 8 * rows * cols + 24.0 / max(pi["slots_per_xcd"] / 2.0, 1.0)
            pre = common
            return it, pre

        def build_roofline_persistent(prof, cnt, population, every):
            """The whole solve is ONE launch of k_cocg_persist (kernels_persist.h).  `achieved` / `frac`: the bytes the kernel's OWN
            design moves (persist_design_bytes: ~168 B per unknown and iteration with two sweeps per side at cfg3; the PMC passes
            measure 173) x the system-iterations the launch performed (device counter) / the HIP-event duration of the launch.
            `frac_four_kernel_bytes` keeps rounds 1-4's figure (every vector of the launch-per-phase iteration streamed once: 224 /
            264 B) for round-to-round comparison; `frac_traffic` is the PMC-measured rate."""
            f2 = cnt.get("solves_two_sweeps", 0) / max(cnt["solves"], 1)
            it_sys, pre_sys = cnt["active_iter_systems"], cnt["start_systems"]
            ms_c, n_c = prof["spmv"]                         # (the launch is timed under this category)
            avg_us = 1e3 * ms_c / max(n_c, 1)
            it_bpu, pre_bpu = persist_design_bytes(f2)
            it4_bpu, pre4_bpu = 224.0 + 40.0 * f2, 96.0 + 8.0 * f2
            tot_bytes = Usys * (it_bpu * it_sys + pre_bpu * pre_sys)
            nbytes = tot_bytes / max(n_c, 1)
            ach = nbytes / (avg_us * 1e-6) / 1e9 if n_c else 0.0
            ach4 = Usys * (it4_bpu * it_sys + pre4_bpu * pre_sys) / max(n_c, 1) / (avg_us * 1e-6) / 1e9 if n_c else 0.0
            (sp1, vp1), (sp2, vp2) = (3, 33), (5, 45)        # canonical CSR: stencil products and vector passes of an iteration (sum of CANON)
            can_it = (sp1 * B_spmv + vp1 * 16 * Usys) * (1 - f2) + (sp2 * B_spmv + vp2 * 16 * Usys) * f2
            can = can_it * it_sys / max(n_c, 1) / (avg_us * 1e-6) / 1e9 if n_c else 0.0
            serial = max(cnt.get("serial_iterations", 0), 1)
            it_us = 1e3 * ms_c / serial                      # kernel time / iterations of its slowest system (+ 1 application): the iteration's latency
            sys_per_it = (it_sys + pre_sys) / serial
            kpad = 32 * ((ctx.NYP + 31) // 32)
            flops = 2 * 2.0 * (2 * ctx.NZP) * ctx.NYP * kpad * (1.0 + 24.0 / 14.0) * (it_sys + pre_sys) / max(n_c, 1)
            entry = {"kernel": "k_cocg_persist (the whole COCG solve of all systems in one launch: a system = %d workgroups of one XCD, r in "
                               "registers, tiles and stencil coefficients in LDS, four per-system synchronisations per iteration)" % ctx.persist_info()["workgroups_per_system"],
                     "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                     "traffic": pmc_traffic(name, "persist"), "avg_launch_us": avg_us,
                     "frac_traffic": (pmc_traffic(name, "persist") / (avg_us * 1e-6) / 1e9 / HBM_PEAK_GBS) if (n_c and pmc_traffic(name, "persist")) else None,
                     "frac_canonical_csr": can / HBM_PEAK_GBS, "evaluations_sampled": cnt["evaluations"], "sampled_every": every,
                     "event_bracket_overhead_us_subtracted": prof_overhead_us, "rocprofv3_avg_launch_us": rocprof_avg_us(name, "persist"),
                     "launches_timed": n_c, "bytes_per_launch": nbytes, "ms_timed": ms_c,
                     "system_iterations_per_launch": it_sys / max(n_c, 1), "preconditioner_applications_per_launch": pre_sys / max(n_c, 1),
                     "bytes_per_unknown_and_iteration": it_bpu, "frac_four_kernel_bytes": ach4 / HBM_PEAK_GBS,
                     "four_kernel_bytes_per_unknown_and_iteration": it4_bpu, "us_per_iteration": it_us, "active_systems_per_iteration": sys_per_it,
                     "active_systems_per_launch": sys_per_it, "launches_per_iteration": 0,
                     "mfma": {"flops_per_launch": flops, "achieved_tflops": flops / (avg_us * 1e-6) / 1e12 if n_c else 0.0, "peak_tflops_bf16_dense": 2500.0},
                     "population": population}
            iteration = {"kernels": 1, "two_sweep_fraction": f2, "bytes": Usys * it_bpu * sys_per_it, "us": it_us,
                         "achieved": Usys * it_bpu * sys_per_it / (it_us * 1e-6) / 1e9 if n_c else 0.0, "unit": "GB/s",
                         "frac": Usys * it_bpu * sys_per_it / (it_us * 1e-6) / 1e9 / HBM_PEAK_GBS if n_c else 0.0,
                         "note": "one preconditioned COCG iteration inside the persistent kernel: kernel time / iterations of its slowest system; "
                                 "bytes = the kernel's own design bytes x the systems active in an average iteration"}
            step_bytes = tot_bytes / max(cnt["evaluations"], 1)
            return [entry], iteration, step_bytes

        def build_roofline(prof, cnt, population, every):
            if (cnt.get("persistent_solves", 0) > 0 and cnt["persistent_solves"] >= cnt["solves"]) or \
               (cnt.get("solves", 0) == 0 and ctx.persist_info()["solves"] > 0):       # (nothing sampled, HMCMT_BENCH_NOPROF: name the path that ran)
                return build_roofline_persistent(prof, cnt, population, every)
            fwd_fused = ctx.NYP <= 256                      # the library's own rule (launch_fdm_fwd): wide meshes run the separate kernels
            back_fused = fwd_fused                          # ... and k_back_post goes with it (launch_back_post)
            # round 3: on the fused path x += alpha p and |x|^2 (x in and out, p in: 40 B per unknown) ride along in k_fdm_fwd,
            # done by the waves that wait for the tridiagonal sweeps (Solver::xInFwd; HMCMT_XFWD=0 puts them back)
            x_in_fwd = fwd_fused and (ctx.nz - 1) * ctx.NYP >= 12000       # (the library's rule, hmcmt_create)
            if os.environ.get("HMCMT_XFWD"):
                x_in_fwd = fwd_fused and os.environ["HMCMT_XFWD"][:1] != "0"
            xb = 40.0 if x_in_fwd else 0.0
            # two damped Jacobi sweeps per side of the FDM stage (chosen per solve by the library, hmcmt_stats.smoother_sweeps):
            # a fifth launch, k_post2 (category post_smoother on the fused path), and 8 B/unknown more in two others.  f2 = the
            # fraction of the sampled preconditioner applications that ran it
            n7 = prof["post_smoother"][1]
            f2 = cnt.get("solves_two_sweeps", 0) / max(cnt["solves"], 1)
            # the second post-sweep inside k_spmv_fused<2> (default) or as k_post2 (HMCMT_POST2=1; fused path only)
            merged = f2 > 0 and (n7 == 0 or not back_fused)
            it_sys = cnt["active_iter_systems"]             # sum over sampled iterations of active systems
            pre_sys = cnt["start_systems"]                  # + one preconditioner application per solve before the first iteration
            fams = {("k_fdm_fwd (split-bf16 MFMA eigen-transform + LDS-resident complex64 tridiagonal sweeps, one launch)" if fwd_fused else
                     "k_thomas32 (batched complex64 tridiagonal solve of the FDM stage)"): ("tridiagonal", 24.0 + xb * it_sys / max(it_sys + pre_sys, 1), 1, it_sys + pre_sys),
                    ("k_back_post (split-bf16 MFMA back transform + both Jacobi halves of the post-smoother + dot products)" if back_fused else
                     "k_transform_lp<2> (split-bf16 MFMA back transform fused with the first Jacobi half)" if fwd_fused else
                     "k_transform_lp<0>,<2> (split-bf16 MFMA forward and back transforms: 16 U and 56 U)"):
                        ("fdm_transform", ((48.0 + 8.0 * f2) if back_fused else 56.0) if fwd_fused else (16.0 + 56.0 * (1 - f2) + 24.0 * f2) / 2, 1 if fwd_fused else 2, (it_sys + pre_sys) * (1 if fwd_fused else 2)),
                    "k_spmv_fused (p-update + 5-point stencil product + dot)": ("spmv", 40.0 + (24.0 * f2 if merged else 0.0), 1, it_sys),
                    ("k_update_fused (r update + Jacobi pre-smoothing; x update in k_fdm_fwd)" if x_in_fwd else
                     "k_update_fused (x, r updates + Jacobi pre-smoothing)"): ("vector_ops", 112.0 + 8.0 * f2 - xb, 1, it_sys)}
            # two sweeps, bytes per unknown (the Jacobi diagonal is complex64 there, -8 per read): k_update_fused<2> also writes
            # the pre-smoothed iterate z2 and the smoothed residual t as complex64 (+16 - 8), k_back_post<.,2> reads both on top
            # of dinv, r (+16 - 8), k_spmv_fused<2> reads z4, r, dinv instead of z (+24) for the second post-sweep it does itself
            if not back_fused:
                # (two sweeps on this path: the back transform writes F t alone, 24 U, and k_post_w2 reads F t, z2, t, r, dinv (56) and
                # writes z4 (8) = 64 U)
                fams["k_post / k_post_w2 (post-smoothing sweep: 5-point stencil + dot products)"] = ("post_smoother", 56.0 * (1 - f2) + 64.0 * f2, 1, it_sys + pre_sys)
            elif n7:
                # HMCMT_POST2=1: read z4 (8), r (16), dinv (16), write z (8)
                fams["k_post2 (second post-sweep of the two-sweep smoother: 5-point stencil + dot products)"] = \
                    ("post_smoother", 48.0, f2, (it_sys + pre_sys) * f2)
            roofs = []
            it_bytes = it_us = step_bytes = 0.0
            nev = max(cnt["evaluations"], 1)
            for kname, (cat, bpu, per_it, sys_launches) in fams.items():
                ms_c, n_c = prof[cat]
                avg_us = 1e3 * ms_c / max(n_c, 1)
                act = sys_launches / max(n_c, 1)                       # active systems per launch, averaged over the timed launches
                nbytes = bpu * Usys * act
                ach = nbytes / (avg_us * 1e-6) / 1e9 if n_c else 0.0
                it_bytes += per_it * nbytes
                it_us += per_it * avg_us
                step_bytes += bpu * Usys * sys_launches / nev
                (sp1, vp1), (sp2, vp2) = CANON[cat]
                can_sys = (sp1 * B_spmv + vp1 * 16 * Usys) * (1 - f2) + (sp2 * B_spmv + vp2 * 16 * Usys) * f2
                can = can_sys * act / (avg_us * 1e-6) / 1e9 if n_c else 0.0
                entry = {"kernel": kname, "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": ach / HBM_PEAK_GBS, "traffic": pmc_traffic(name, cat), "avg_launch_us": avg_us,
                         "frac_canonical_csr": can / HBM_PEAK_GBS, "canonical_csr_bytes_per_launch": can_sys * act,
                         "evaluations_sampled": cnt["evaluations"], "sampled_every": every,
                         "event_bracket_overhead_us_subtracted": prof_overhead_us,
                         "rocprofv3_avg_launch_us": rocprof_avg_us(name, cat),
                         "launches_timed": n_c, "launches_per_iteration": per_it, "bytes_per_launch": nbytes,
                         "bytes_per_launch_all_systems_active": bpu * U, "active_systems_per_launch": act, "ms_timed": ms_c,
                         "population": population}
                if cat in ("tridiagonal", "fdm_transform") and n_c:
                    # the two MFMA kernels, for reference: two bf16 products (input hi and lo parts x the bf16 eigenvectors) of
                    # a [2*S*NZP real rows] x [NYP] x [K = NYP padded to 32] real matrix product
                    kpad = 32 * ((ctx.NYP + 31) // 32)
                    flops = 2 * 2.0 * (2 * act * ctx.NZP) * ctx.NYP * kpad
                    entry["mfma"] = {"flops_per_launch": flops, "achieved_tflops": flops / (avg_us * 1e-6) / 1e12,
                                     "peak_tflops_bf16_dense": 2500.0}
                roofs.append(entry)
            iteration = {"kernels": len(fams), "two_sweep_fraction": f2, "bytes": it_bytes, "us": it_us,
                         "achieved": it_bytes / (it_us * 1e-6) / 1e9 if it_us else 0.0, "unit": "GB/s",
                         "frac": it_bytes / (it_us * 1e-6) / 1e9 / HBM_PEAK_GBS if it_us else 0.0,
                         "note": "one preconditioned COCG iteration of the systems still active = %d launches%s; " % (
                                     len(fams), " (two smoothing sweeps per side in %.0f %% of the sampled solves)" % (100 * f2) if f2 > 0 else "") +
                                 ("the working set of a solve (~15 vectors = %.0f MB) fits the 256 MB Infinity Cache, so launches are "
                                  "latency- not HBM-bound" if ws_mb <= 256 else
                                  "the working set of a solve (~15 vectors = %.0f MB) is beyond the 256 MB Infinity Cache: HBM-bound") % ws_mb}
            roofs.sort(key=lambda r: -r["ms_timed"])             # strict arg-max of the measured total time, no tie-break
            return roofs, iteration, step_bytes

        roofs, iteration, step_bytes = build_roofline(
            prof, cnt, f"every launch of this kernel in every {prof_every}. evaluation of the timed region (HIP events on the "
                       "library's stream), launches that found all systems converged included", prof_every)
        # Everything under `roofline` is measured in THIS run (HIP events, the device's iteration counters).  What comes from the
        # committed rocprofv3 passes of this command (profiles/pmc_traffic.json: PMC bytes per launch, --stats average) describes
        # another run's launch population: it goes under `roofline_committed_profile`, labelled; `roofline.traffic` is null here.
        committed = {"source": "profiles/pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes and --kernel-trace --stats of this command, "
                               "collected by scripts/gpu_profile_all.sh in the round that file was committed; another run's launch population)",
                     "kernels": []}
        for r in roofs:
            committed["kernels"].append({"kernel": r["kernel"][:60], "traffic_bytes_per_launch": r.pop("traffic", None),
                                         "frac_traffic_against_this_runs_launch_time": r.pop("frac_traffic", None),
                                         "rocprofv3_avg_launch_us": r.pop("rocprofv3_avg_launch_us", None)})
            r["traffic"] = None
            r["traffic_source"] = "not collected in this run (PMC counters need rocprofv3): see roofline_committed_profile"
        ms_step = 1e3 * elapsed / K
        traj_ms = head.get("ms_per_step_by_trajectory") or []
        step = {"bytes_per_step": step_bytes, "ms_per_step": ms_step, "achieved": step_bytes / (ms_step * 1e-3) / 1e9,
                "unit": "GB/s", "frac": step_bytes / (ms_step * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "note": "algorithmic bytes of the iteration kernels per leapfrog step (sampled evaluations) / wall time per step / 8 TB/s"}
        out = {
            "metric": "leapfrog steps/sec (= fwd+grad evals/sec), 200x100 mesh x 16 freq",
            "value": world * K / elapsed, "unit": "steps/s", "n_gpus": world, "steps": K, "warmup": W,
            "ms_per_step": ms_step, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"{name}: {mesh.gridSize[0]}x{mesh.gridSize[1] - len(mesh.airLayer)}-cell mesh "
                                   f"(+{len(mesh.airLayer)} air rows), {len(data.freqs)} freq, TE+TM, "
                                   f"{data.rxLoc.shape[0]} receivers, 1 independent chain per GPU; real leapfrog trajectories "
                                   f"(L = {LTRAJ}, dt = {DT}, prior lambda = {LAMBDA}, bounds rho in [1, 1e4] ohm-m, accept/reject) "
                                   f"started at the rough state m = ln 0.01 + 0.3 N(0,1)",
                       "systems_per_step": ctx.S, "unknowns_per_system": nyi * nzi, "nparam": nAC,
                       "solver": "batched fp64 COCG, Jacobi/fast-diagonalisation/Jacobi preconditioner (FDM stage in bf16/fp32; one or two Jacobi sweeps per side, chosen per solve), tol 1e-11 (error estimate), warm start; " + ("one persistent launch per solve (kernels_persist.h)" if ctx.persist_info()["solves"] else "four launches per iteration"),
                       "iters_fwd_max": st["iters_fwd_max"], "iters_adj_max": st["iters_adj_max"],
                       "smoother_sweeps_last_evaluation": st["smoother_sweeps"],
                       "parallelism": f"chains x{world}" if world > 1 else "1 chain"},
            "chain": head,
            "median_ms_per_step_by_trajectory": float(np.median(traj_ms)) if traj_ms else None,
            "warmup_steps_run": Wrun,
            "roofline": roofs[0], "roofline_other": roofs[1:], "roofline_iteration": iteration, "roofline_step": step,
            "roofline_committed_profile": committed,
            "check": check,
        }
        if world == 1 and not args.no_extras:
            fr, fi, _ = build_roofline(full_prof, full_cnt, "every launch of 4 cold evaluations whose solves are cut off after 12 "
                                                           "iterations with all systems still active (tolerance 1e-200)", 1)
            extras["roofline_all_systems_active"] = {"kernels": fr, "iteration": fi}
        out.update(extras)
        if gather is not None:
            out["allgather"] = gather
        if cpu is not None:
            out["cpu_baseline"] = cpu
        if world == 1 and not args.no_extras:
            ctx.close()                          # (a context on the whole device overlaps the two chains' shares)
            out["two_chains_per_gpu"] = two_chains_leg(torch, HipContext, mesh, data, inv, local, dev, m_true, mref, 48)
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    ctx.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
