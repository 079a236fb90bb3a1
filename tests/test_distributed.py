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
