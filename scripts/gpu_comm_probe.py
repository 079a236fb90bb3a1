"""hmcmt_comm_* on one rank, step by step (run under `timeout`)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from hmcmt2d_amd.lib import SampleComm
t = time.time(); uid = SampleComm.unique_id(); print("unique id", time.time() - t, flush=True)
t = time.time(); comm = SampleComm(0, 1, 0, uid); print("comm create", time.time() - t, flush=True)
blk = np.arange(1000.0)
t = time.time(); out = comm.allgather(blk); print("allgather", time.time() - t, np.array_equal(out[0], blk), flush=True)
comm.close(); print("closed", flush=True)
