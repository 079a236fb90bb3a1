"""The event-bracket calibration kernel (k_spin, 12 us on the device's wall clock) under rocprofv3: what duration does the
profiler report for it?  The difference to 12 us is the dispatch ramp rocprofv3 counts as kernel time."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hmcmt2d_amd.lib import HipContext
from tests.helpers import make_problem
mesh, data, inv, m = make_problem("tiny")
ctx = HipContext(mesh, data, inv)
for _ in range(4):
    ctx.profile(True)
    print("calibrated bracket overhead (us):", ctx.profile_overhead_us())
ctx.profile(False)
ctx.close()
