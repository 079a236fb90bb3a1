"""As tests/conftest.py does: /opt/rocm's HIP runtime first, then torch, then the library's RCCL gather (run under `timeout`)."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
hip = ctypes.CDLL("libamdhip64.so"); n = ctypes.c_int(0); hip.hipGetDeviceCount(ctypes.byref(n)); print("devices", n.value, flush=True)
import numpy as np
t = time.time(); import torch; print("torch import", round(time.time() - t, 1), flush=True)
from hmcmt2d_amd.lib import SampleComm
def step(name, f):
    t = time.time(); r = f(); print(name, round(time.time() - t, 3), flush=True); return r
uid = step("unique id", SampleComm.unique_id)
comm = step("comm create", lambda: SampleComm(0, 1, 0, uid))
blk = np.arange(1000.0)
out = step("allgather host", lambda: comm.allgather(blk)); print(np.array_equal(out[0], blk))
step("close", comm.close)
os.system("grep -E 'rccl|amdhip' /proc/%d/maps | awk '{print $6}' | sort -u" % os.getpid())
