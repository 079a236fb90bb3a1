import os
import sys

# The oracle's sparse direct solves (SuperLU) call BLAS on small supernodes: a multi-threaded BLAS gains nothing there and,
# when anything else runs on the machine, its spinning worker threads cost 50x (measured: 4 s -> 300 s for one test).
# One thread per process, set before numpy / scipy load their BLAS.
for _v in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):
    os.environ.setdefault(_v, "1")

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")


def _have_gpu():
    try:
        import ctypes
        hip = ctypes.CDLL("libamdhip64.so")
        n = ctypes.c_int(0)
        return hip.hipGetDeviceCount(ctypes.byref(n)) == 0 and n.value > 0
    except Exception:
        return False


HAVE_GPU = _have_gpu()


def pytest_collection_modifyitems(config, items):
    if HAVE_GPU:
        return
    skip = pytest.mark.skip(reason="no HIP device in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)
