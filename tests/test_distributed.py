"""N > 1 path: chains sharded over ranks, sample blocks all-gathered (gloo, world_size 2, CPU)."""
import os
import socket
import sys

import numpy as np
import pytest

from tests.conftest import ROOT


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, nchains, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    from hmcmt2d_amd import sampler
    from hmcmt2d_amd.structs import HMCPrior, HMCStatus
    from tests.helpers import make_problem
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mesh, data, inv, m = make_problem("tiny")
    prior = HMCPrior(totalsamples=4)
    nparam, ndata = len(inv.strModel), len(inv.obsData)

    def run_chain(c, rng):
        model = rng.standard_normal((nparam, 4)) + c
        acc = np.array([True, False, True, c % 2 == 0])
        st = HMCStatus(int(acc.sum()), int((~acc).sum()), acc, rng.standard_normal((4, 5)))
        dat = rng.standard_normal((ndata, 5)) + 1j * rng.standard_normal((ndata, 5))
        return model, st, dat

    hm, hs, hd = sampler.parallelHMCSampler(mesh, data, inv, prior, nchains=nchains, seed=11, run_chain=run_chain)
    digest = [float(np.sum(x)) for x in hm] + [float(np.sum(np.abs(x))) for x in hd] + [s.nAccept for s in hs]
    # what rank-independent generation gives for every chain
    exp = []
    for c in range(nchains):
        mdl, st, dat = run_chain(c, np.random.default_rng([11, c]))
        exp.append((float(np.sum(mdl)), float(np.sum(np.abs(dat))), st.nAccept))
    q.put((rank, digest, exp))
    dist.destroy_process_group()


def _worker_real(rank, world, port, nchains, q):
    """The real runHMCSampler on every rank (host leapfrog loop, accept/reject, packing), the compute interface
    served by the oracle-backed stand-in of tests/helpers.py -- so the all-gather carries real chain output."""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    import copy
    import torch.distributed as dist
    from hmcmt2d_amd import sampler
    from hmcmt2d_amd.structs import HMCPrior
    from tests.helpers import make_problem, OracleContext
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mesh, data, inv, m = make_problem("tiny")
    prior = HMCPrior(totalsamples=2, burninsamples=0, dt=0.02, timestep=[1, 2], sigBounds=[1e-4, 1.0])
    devices = []

    def factory(mesh_c, data_c, inv_c, dev):
        devices.append(dev)
        return OracleContext(mesh_c, data_c, inv_c)

    hm, hs, hd = sampler.parallelHMCSampler(mesh, data, inv, prior, nchains=nchains, seed=11, context_factory=factory)
    exp = []
    for c in range(nchains):        # the same chains, one after another in this process
        mdl, st, dat = sampler.runHMCSampler(copy.deepcopy(mesh), data, copy.deepcopy(inv), copy.deepcopy(prior),
                                             np.random.default_rng([11, c]), ctx=OracleContext(mesh, data, inv))
        exp.append((mdl, st.hmstats, st.acceptstats, dat))
    ok = all(np.array_equal(hm[c], exp[c][0]) and np.array_equal(hs[c].hmstats, exp[c][1])
             and np.array_equal(hs[c].acceptstats, exp[c][2]) and np.array_equal(hd[c], exp[c][3]) for c in range(nchains))
    q.put((rank, ok, devices, [float(np.abs(x).sum()) for x in hm]))
    dist.destroy_process_group()


def test_real_sampler_chains_allgathered_with_per_rank_device():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    nchains = 3
    procs = [ctx.Process(target=_worker_real, args=(r, 2, port, nchains, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = sorted(q.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, ok0, dev0, s0), (r1, ok1, dev1, s1) = out
    assert ok0 and ok1, "gathered chains must equal the same chains run one after another"
    assert s0 == s1 and all(x > 0 for x in s0)
    assert dev0 == [0, 0] and dev1 == [1]            # rank r builds its contexts on device LOCAL_RANK = r (chains 0, 2 | 1)


def test_default_device_follows_local_rank(monkeypatch):
    """LOCAL_RANK (one process per GPU) unless HMCMT_DEVICE says otherwise -- the precedence of julia/HMCMTHip.jl's
    defaultDevice."""
    from hmcmt2d_amd import sampler
    monkeypatch.delenv("LOCAL_RANK", raising=False); monkeypatch.delenv("HMCMT_DEVICE", raising=False)
    assert sampler.default_device() == 0
    monkeypatch.setenv("LOCAL_RANK", "5")
    assert sampler.default_device() == 5
    monkeypatch.setenv("HMCMT_DEVICE", "3")
    assert sampler.default_device() == 3


@pytest.mark.parametrize("nchains", [2, 3])
def test_chains_shard_and_allgather(nchains):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, nchains, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    out.sort()
    (r0, d0, e0), (r1, d1, e1) = out
    assert d0 == d1, "every rank must hold every chain after the all-gather"
    sums = d0[:nchains]; dsum = d0[nchains:2 * nchains]; acc = d0[2 * nchains:]
    for c in range(nchains):
        assert abs(sums[c] - e0[c][0]) < 1e-9 and abs(dsum[c] - e0[c][1]) < 1e-9 and acc[c] == e0[c][2]


def test_single_process_runs_chains_sequentially():
    from hmcmt2d_amd import sampler
    from hmcmt2d_amd.structs import HMCPrior, HMCStatus
    from tests.helpers import make_problem
    mesh, data, inv, m = make_problem("tiny")
    nparam, ndata = len(inv.strModel), len(inv.obsData)
    seen = []

    def run_chain(c, rng):
        seen.append(c)
        return (np.full((nparam, 2), float(c)), HMCStatus(2, 0, np.array([True, True]), np.zeros((4, 3))),
                np.zeros((ndata, 3), complex))

    hm, hs, hd = sampler.parallelHMCSampler(mesh, data, inv, HMCPrior(totalsamples=2), pids=[0, 1, 2], run_chain=run_chain)
    assert seen == [0, 1, 2] and [float(x[0, 0]) for x in hm] == [0.0, 1.0, 2.0] and hs[1].nAccept == 2
