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


@pytest.fixture(autouse=True)
def _no_context_left_alive(request):
    """A HipContext left alive by one test silently moves every later context of the process to the launch-per-phase loop
    (one persistent kernel per device: hmcmt_persist_info 'usable_now') -- a parity test would then test the other solver
    without saying so.  A test that leaks a context fails here, and the leak is cleaned up for the tests behind it."""
    if "gpu" not in request.keywords:
        yield
        return
    import gc
    from hmcmt2d_amd.lib import HipContext
    before = HipContext.live              # (contexts of module-scoped fixtures: set up before this one, theirs to close)
    known = {id(o) for o in gc.get_objects() if isinstance(o, HipContext) and getattr(o, "h", None)} if before else set()
    yield
    if HipContext.live > before:
        gc.collect()                      # (contexts dropped without close(): __del__ closes them)
    left = HipContext.live - before
    if left > 0:
        for o in gc.get_objects():
            if isinstance(o, HipContext) and getattr(o, "h", None) and id(o) not in known:
                o.close()
        pytest.fail(f"{left} HipContext(s) left open by this test")
