"""the steps of tests/test_gpu_parity.py::test_library_allgather... one by one (run under `timeout`)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from hmcmt2d_amd.lib import SampleComm, HmcmtError
def step(name, f):
    t = time.time(); r = f(); print(name, round(time.time() - t, 3), flush=True); return r
uid = step("unique id", SampleComm.unique_id)
comm = step("comm create", lambda: SampleComm(0, 1, 0, uid))
blk = np.random.default_rng(0).standard_normal(200_000)
out = step("allgather host", lambda: comm.allgather(blk)); print(np.array_equal(out[0], blk))
d_s = torch.from_numpy(blk).to("cuda:0"); d_r = torch.zeros_like(d_s); torch.cuda.synchronize()
step("allgather device", lambda: comm.allgather_device(d_s.data_ptr(), d_r.data_ptr(), blk.size)); print(torch.equal(d_r, d_s))
step("close", comm.close)
try:
    SampleComm(0, 2, 5, uid)
except HmcmtError as e:
    print("refused:", e, flush=True)
uid2 = step("unique id 2", SampleComm.unique_id)
comm2 = step("comm create 2", lambda: SampleComm(0, 1, 0, uid2))
step("allgather 2", lambda: comm2.allgather(blk))
step("close 2", comm2.close)
