#!/usr/bin/env python3
"""Headline benchmark: leapfrog steps/s (= forward + gradient evaluations/s) on the 200x100-cell,
16-frequency synthetic of BASELINE.json, one independent chain per GPU.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config cfg3]

A step is one pass of the hot path -- compDataGradient (HMCSampler.jl:277-330) -- at the next model
of a synthetic leapfrog trajectory m_k = m_0 + k*dt*p; all K models are resident in HBM before the
timed region and predData / misfit / gradient stay in HBM.  For N > 1 the driver launches this file
with torch.distributed.run (one rank per GPU, RCCL); ranks are independent chains (weak scaling) and
the only collective in the timed region is the barrier.  Rank 0 prints ONE JSON line.
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


def build_problem(name, seed):
    """Synthetic workload of SURVEY §8(d): observed data = GPU forward of the true model + 3 % noise."""
    from hmcmt2d_amd import synthetic as S, invsetup as I
    from hmcmt2d_amd.lib import HipContext
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
    frequency (TE + TM) of the config -- the frequency loop is the reference's only parallelisable axis
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
    # wall includes each worker's imports and set-up; the rate uses the slowest worker's own timer
    tmax = max(times)
    return {"value": per * len(jobs) / (tmax * nF), "unit": "steps/s", "cores": cores, "kind": "port",
            "sample": f"oracle compDataGradient (numpy/scipy, SuperLU direct solves, dense dBC as the reference) at "
                      f"{per} of {nF} frequencies (TE+TM) of {name} per worker, one single-threaded worker process per core on "
                      f"{cores} cores concurrently: slowest worker {tmax:.2f} s (sum {sum(times):.1f} core-s, wall "
                      f"incl. start-up {wall:.1f} s), scaled to {nF} frequencies"}


def pmc_traffic(cat):
    """HBM-side bytes per launch from rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE collected in separate runs of
    this same command, FETCH_SIZE doubled per MI355X_MICROARCH.md 'HBM'), summarised by scripts/pmc_summary.py into
    profiles/pmc_traffic.json; None when that file has no entry."""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
            return json.load(f)["per_launch_bytes"].get(cat)
    except (OSError, KeyError, ValueError):
        return None


def sampler_leg(step, forward, K, W, Ltraj, nsamples=6):
    """samples/s with the reference's cost structure -- per sample one trajectory of (1 + L)
    gradient evaluations plus one forward-only solve at the proposal (HMCSampler.jl:136,141) -- on the synthetic
    trajectories of the timed region (L = Ltraj - 1 = 7 position steps, timestep 6..10 in
    examples/dprism3d/startupfile).  The proposal's model equals the last gradient's, which the library
    recognises (the forward-only solve then costs about one iteration)."""
    import torch
    first = ((W + Ltraj - 1) // Ltraj) * Ltraj            # trajectories start at multiples of Ltraj
    n = min(nsamples, (W + K - first) // Ltraj)
    if n < 1:
        return None
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        k0 = first + i * Ltraj
        for j in range(Ltraj):
            step(k0 + j)                         # gradient at the start model and after each of the L position steps
        forward(k0 + Ltraj - 1)                  # getHamiltonian's forward-only solve at the proposal
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return {"samples_per_s": n / dt, "samples": n, "evals_per_sample": f"{Ltraj} gradients + 1 forward-only"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default="cfg3")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-sampler", action="store_true", help="skip the untimed samples/s leg")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    cpu = None
    if world == 1 and rank == 0 and not args.no_cpu_baseline:
        cpu = cpu_baseline(args.config)          # before this process touches the GPU: the workers are spawned
    import torch
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    torch.cuda.set_device(local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))

    from hmcmt2d_amd import synthetic as S, invsetup as I
    from hmcmt2d_amd.lib import HipContext
    name = args.config
    mesh, data, inv0, sig_true = build_problem(name, rank)
    # observed data from the GPU forward of the true model (+ seeded noise), then the real context
    ctx0 = HipContext(mesh, data, inv0, device_id=local)
    m_true = np.log(sig_true[inv0.activeIdx])
    pred_true, _ = ctx0.forward(m_true)
    ctx0.close()
    obs, err = S.noisy_observations(pred_true)
    inv = I.setupInverseDataModel(mesh, [S.SIG_AIR], 0.0, 0.0, obs, err)
    ctx = HipContext(mesh, data, inv, device_id=local)
    nAC, nData = ctx.nAC, ctx.nData

    K, W = args.steps, args.warmup
    # synthetic leapfrog trajectories: position steps dm = dt*p with dt = 0.03 (examples/dprism3d/startupfile:5),
    # momentum redrawn every L = 8 steps (timestep 6..10) from the clipped N(0,1) of getMomentumVector
    rng = np.random.default_rng([20250114, 7, rank])
    m0 = S.rough_state(nAC, seed=1 + rank)
    Ltraj, dt = 8, 0.03
    traj = np.empty((K + W, nAC))
    for k in range(K + W):
        if k % Ltraj == 0:
            p = np.clip(rng.standard_normal(nAC), -2.5, 2.5)
        traj[k] = m0 + dt * (k % Ltraj) * p
    dev = torch.device("cuda", local)
    d_m = torch.from_numpy(traj).to(dev)
    d_pred = torch.zeros(2 * nData, dtype=torch.float64, device=dev)
    d_mis = torch.zeros(1, dtype=torch.float64, device=dev)
    d_grad = torch.zeros(nAC, dtype=torch.float64, device=dev)

    # hmcmt_grad_device_async: as a device-resident leapfrog would call it (the next model comes from this gradient on
    # the device); the step's two convergence polls still block, its gradient tail overlaps the next step's launches
    def step(k):
        ctx.grad_device_async(d_m[k].data_ptr(), d_pred.data_ptr(), d_mis.data_ptr(), d_grad.data_ptr())

    for k in range(W):
        step(k)
    ctx.wait()
    if not os.environ.get("HMCMT_BENCH_NOPROF"):
        # HIP events around the two heaviest kernel families, in every 6th step of the timed region
        # (bracketing every launch of every step costs ~20 % of the throughput)
        ctx.profile(["fdm_transform", "tridiagonal", "spmv", "vector_ops", "post_smoother"], every=6)
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(W, W + K):
        step(k)
    ctx.wait()                                           # (status of the last step; every earlier one was checked by its successor)
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    prof = ctx.profile_read()
    ctx.profile(False)
    st = ctx.stats()
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    misfit = float(d_mis.item())
    gnorm = float(torch.linalg.vector_norm(d_grad).item())
    def forward(k):
        ctx.forward_device(d_m[k].data_ptr(), d_pred.data_ptr(), d_mis.data_ptr())

    samples = None
    structured = None
    if world == 1 and not args.no_sampler:            # (single-GPU runs only: keeps the ranks of an N-GPU run symmetric)
        samples = sampler_leg(step, forward, K, W, Ltraj)
        # Secondary, untimed-region figure: the same trajectories laid around the TRUE model (2 layers + a 10x
        # conductive block) instead of the homogeneous reference model of SURVEY 8(d).  Lateral structure is what
        # the laterally averaged FDM background cannot see, so this state needs 2-3x the iterations; it is the
        # regime a converged chain samples in.
        nst = min(K + W, 24)
        d_ms = torch.from_numpy(traj[:nst] - m0 + m_true).to(dev)
        for k in range(min(8, nst)):
            ctx.grad_device_async(d_ms[k].data_ptr(), d_pred.data_ptr(), d_mis.data_ptr(), d_grad.data_ptr())
        ctx.wait()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for k in range(8, nst):
            ctx.grad_device_async(d_ms[k].data_ptr(), d_pred.data_ptr(), d_mis.data_ptr(), d_grad.data_ptr())
        ctx.wait()
        torch.cuda.synchronize()
        sst = ctx.stats()
        if nst > 8:
            structured = {"steps_per_s": (nst - 8) / (time.perf_counter() - t1), "steps": nst - 8,
                          "iters_fwd_max": sst["iters_fwd_max"], "iters_adj_max": sst["iters_adj_max"],
                          "state": "true model (100 over 10 ohm-m + 10 ohm-m block) + the same dt*j*p trajectories"}

    if rank == 0:
        # Algorithmic bytes per launch: U = S*(nz-1)*(ny-1) interior unknowns, complex128 = 16 B,
        # complex64 / split-bf16 = 8 B; the real stencil coefficients are shared by all frequencies of a mode (not
        # counted).  One preconditioned COCG iteration = these four launches on the fused path (five when the back
        # transform and the post-smoother run as separate kernels: wide meshes, HMCMT_FUSED_BACK=0):
        #   k_fdm_fwd      forward eigen-transform + tridiagonal solves: read t (8) + inverse pivots (8), write y (8) = 24 U
        #   k_back_post    back transform + both Jacobi halves + dots: read y (8), dinv (16), r (16), write t (16)   = 56 U
        #                  (separate: k_transform_lp<2> 56 U with z written, k_post: read r, z, dinv (48), write t (16) = 64 U)
        #   k_spmv_fused   p = z + beta p, q = A p, p'q: read z, p (32), write p, q (32)                             = 64 U
        #   k_update_fused x, r updates + Jacobi pre-smoothing: read p, q, r, x, dinv (80), write x, r (32), t (8)   = 120 U
        nyi, nzi = ctx.ny - 1, ctx.nz - 1
        U = ctx.S * nzi * nyi
        back_fused = prof["post_smoother"][1] == 0
        fwd_fused = ctx.NYP <= 256                      # the library's own rule (launch_fdm_fwd): wide meshes run the separate kernels
        fams = {("k_fdm_fwd (split-bf16 MFMA eigen-transform + LDS-resident complex64 tridiagonal sweeps, one launch)" if fwd_fused else
                 "k_thomas32 (batched complex64 tridiagonal solve of the FDM stage)"): ("tridiagonal", 24.0 * U, 1),
                ("k_back_post (split-bf16 MFMA back transform + both Jacobi halves of the post-smoother + dot products)" if back_fused else
                 "k_transform_lp<2> (split-bf16 MFMA back transform fused with the first Jacobi half)" if fwd_fused else
                 "k_transform_lp<0>,<2> (split-bf16 MFMA forward and back transforms: 16 U and 56 U)"):
                    ("fdm_transform", 56.0 * U if fwd_fused else 36.0 * U, 1 if fwd_fused else 2),
                "k_spmv_fused (p-update + 5-point stencil product + dot)": ("spmv", 64.0 * U, 1),
                "k_update_fused (x, r updates + Jacobi pre-smoothing)": ("vector_ops", 120.0 * U, 1)}
        if not back_fused:
            fams["k_post (second Jacobi half: 5-point stencil + dot products)"] = ("post_smoother", 64.0 * U, 1)
        roofs = []
        it_bytes = it_us = 0.0
        for kname, (cat, nbytes, per_it) in fams.items():
            ms_c, n_c = prof[cat]
            avg_us = 1e3 * ms_c / max(n_c, 1)
            ach = nbytes / (avg_us * 1e-6) / 1e9 if n_c else 0.0
            it_bytes += per_it * nbytes
            it_us += per_it * avg_us
            entry = {"kernel": kname, "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": ach / HBM_PEAK_GBS, "traffic": pmc_traffic(cat), "avg_launch_us": avg_us,
                     "launches_timed": n_c, "launches_per_iteration": per_it, "bytes_per_launch": nbytes,
                     "ms_timed": ms_c}
            if cat in ("tridiagonal", "fdm_transform") and n_c:
                # the two MFMA kernels, for reference: three bf16 products (hi*hi, hi*lo, lo*hi) of a
                # [2*S*NZP real rows] x [NYP] x [K = NYP padded to 32] real matrix product
                kpad = 32 * ((ctx.NYP + 31) // 32)
                flops = 3 * 2.0 * (2 * ctx.S * ctx.NZP) * ctx.NYP * kpad
                entry["mfma"] = {"flops_per_launch": flops, "achieved_tflops": flops / (avg_us * 1e-6) / 1e12,
                                 "peak_tflops_bf16_dense": 2500.0}
            roofs.append(entry)
        iteration = {"kernels": len(fams), "bytes": it_bytes, "us": it_us,
                     "achieved": it_bytes / (it_us * 1e-6) / 1e9 if it_us else 0.0, "unit": "GB/s",
                     "frac": it_bytes / (it_us * 1e-6) / 1e9 / HBM_PEAK_GBS if it_us else 0.0,
                     "note": "one preconditioned COCG iteration of all systems = %d launches;" % len(fams) + " the working set "
                             "(~15 vectors) fits the 256 MB Infinity Cache at cfg3, so launches are latency- not HBM-bound"}
        roofs.sort(key=lambda r: -r["ms_timed"])
        # The four kernels of an iteration take 10-17 us each and two of them (k_update_fused, k_fdm_fwd, the latter with
        # one launch more per solve) are within a few per cent of each other in total time, so "the dominant kernel"
        # would flip from run to run: among kernels within 5 % of the largest total, name the one that moves the most
        # bytes (the HBM-bound one); the others follow in roofline_other, the whole iteration in roofline_iteration.
        top = [r for r in roofs if r["ms_timed"] >= 0.95 * roofs[0]["ms_timed"]]
        if len(top) > 1:
            lead = max(top, key=lambda r: r["bytes_per_launch"])
            roofs.remove(lead)
            roofs.insert(0, lead)
            lead["note"] = "total time within 5 % of: " + ", ".join(r["kernel"].split(" ")[0] for r in top if r is not lead)
        out = {
            "metric": "leapfrog steps/sec (= fwd+grad evals/sec), 200x100 mesh x 16 freq",
            "value": world * K / elapsed, "unit": "steps/s", "n_gpus": world, "steps": K, "warmup": W,
            "ms_per_step": 1e3 * elapsed / K, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"{name}: {mesh.gridSize[0]}x{mesh.gridSize[1] - len(mesh.airLayer)}-cell mesh "
                                   f"(+{len(mesh.airLayer)} air rows), {len(data.freqs)} freq, TE+TM, "
                                   f"{data.rxLoc.shape[0]} receivers, 1 independent chain per GPU",
                       "systems_per_step": ctx.S, "unknowns_per_system": nyi * nzi, "nparam": nAC,
                       "solver": "batched fp64 COCG, Jacobi/fast-diagonalisation/Jacobi preconditioner (FDM stage in split-bf16/fp32), tol 1e-11 (error estimate), warm start",
                       "iters_fwd_max": st["iters_fwd_max"], "iters_adj_max": st["iters_adj_max"],
                       "parallelism": f"chains x{world}" if world > 1 else "1 chain"},
            "roofline": roofs[0], "roofline_other": roofs[1:], "roofline_iteration": iteration,
            "samples": samples, "structured_state": structured,
            "check": {"misfit_last": misfit, "grad_l2_last": gnorm, "solver_status": st["status"]},
        }
        if cpu is not None:
            out["cpu_baseline"] = cpu
        print(json.dumps(out))
    ctx.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
