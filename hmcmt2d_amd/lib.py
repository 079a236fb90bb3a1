"""ctypes binding of libhmcmt_hip.so (include/hmcmt.h).

There is no fallback: if the shared library is missing, cannot be loaded, or no HIP device is
present, the calls raise.  `build_library()` compiles the in-tree source for gfx950 with hipcc.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
import numpy as np

from .marshal import CreateArgs, CREATE_ARGTYPES, c_double_p, c_int64_p

HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.environ.get("HMCMT_LIB_PATH") or os.path.join(HERE, "libhmcmt_hip.so")   # (override: A/B runs of two builds)
CSRC = os.path.join(HERE, "csrc")
SOURCES = [os.path.join(CSRC, "hmcmt_hip.hip"), os.path.join(CSRC, "mumps_shim.hip"), os.path.join(CSRC, "comm.hip")]
HEADERS = [os.path.join(CSRC, h) for h in ("hmcmt_math.h", "hmcmt_items.h", "hmcmt_host.h", "kernels_cocg.h", "kernels_fdm.h",
                                            "kernels_fused.h", "kernels_persist.h", "kernels_persist4.h", "kernels_path.h")] + \
          [os.path.join(HERE, "..", "include", h) for h in ("hmcmt.h", "hmcmt_debug.h", "hmcmt_mumps.h")]

HMCMT_NCAT = 8
CATEGORIES = ["fdm_transform", "tridiagonal", "spmv", "vector_ops", "assembly_bc", "receivers", "gradient",
              "post_smoother"]
PRECOND = {"jacobi": 0, "fdm": 1, "fdmj": 2}
# initial guess of both solves: cold start, the previous evaluation's fields, or those extrapolated along the model path
def _warm_start_code(v):
    if isinstance(v, bool):
        return 2 if v else 0
    return {"cold": 0, "previous": 1, "extrapolate": 2, 0: 0, 1: 1, 2: 2}[v]
ERRORS = {-1: "EINVAL", -2: "ENODEV", -3: "EHIP", -10: "ENOCONV", -11: "EBREAKDOWN", -13: "ENOMEM"}


class HmcmtError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libhmcmt_hip: {ERRORS.get(code, code)}: {msg}")
        self.code = code


class Options(C.Structure):
    _fields_ = [("precond", C.c_int32), ("maxit", C.c_int32), ("tol", C.c_double),
                ("check_every", C.c_int32), ("verify", C.c_int32), ("warm_start", C.c_int32),
                ("fdm_precision", C.c_int32)]


class Stats(C.Structure):
    _fields_ = [("iters_fwd_max", C.c_int32), ("iters_adj_max", C.c_int32),
                ("iters_fwd_sum", C.c_int32), ("iters_adj_sum", C.c_int32),
                ("err_est_max", C.c_double), ("true_res_max", C.c_double),
                ("status", C.c_int32), ("nsystems", C.c_int32), ("fallback_solves", C.c_int32), ("smoother_sweeps", C.c_int32)]


def build_library(force=False, verbose=False):
    """hipcc --offload-arch=gfx950 -shared: cross-compiles without a GPU."""
    newest = max(os.path.getmtime(f) for f in SOURCES + HEADERS)
    if not force and os.path.exists(SO_PATH) and os.path.getmtime(SO_PATH) >= newest:
        return SO_PATH
    cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC",
           "-Wno-unused-value", "-Wno-pass-failed", "-o", SO_PATH] + SOURCES + ["-ldl"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return SO_PATH


_lib = None


def load_library():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(SO_PATH):
        raise HmcmtError(-2, f"{SO_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                             "(there is no CPU fallback)")
    lib = C.CDLL(SO_PATH)
    vp = C.c_void_p
    lib.hmcmt_default_options.argtypes = [C.POINTER(Options)]
    lib.hmcmt_default_options.restype = None
    lib.hmcmt_create.argtypes = [C.POINTER(vp), C.c_int32] + CREATE_ARGTYPES + [C.POINTER(Options)]
    lib.hmcmt_destroy.argtypes = [vp]
    lib.hmcmt_last_error.argtypes = [vp]
    lib.hmcmt_last_error.restype = C.c_char_p
    lib.hmcmt_set_options.argtypes = [vp, C.POINTER(Options)]
    lib.hmcmt_get_stats.argtypes = [vp, C.POINTER(Stats)]
    lib.hmcmt_get_iters.argtypes = [vp, C.POINTER(C.c_int32)]
    lib.hmcmt_grad.argtypes = [vp, c_double_p, c_double_p, c_double_p, c_double_p]
    lib.hmcmt_forward.argtypes = [vp, c_double_p, c_double_p, c_double_p]
    lib.hmcmt_grad_device.argtypes = [vp, vp, vp, vp, vp]
    lib.hmcmt_forward_device.argtypes = [vp, vp, vp, vp]
    lib.hmcmt_grad_device_async.argtypes = [vp, vp, vp, vp, vp]
    lib.hmcmt_wait.argtypes = [vp]
    lib.hmcmt_set_prior.argtypes = [vp, c_double_p, c_int64_p, c_int64_p, c_double_p, c_double_p]
    lib.hmcmt_leapfrog.argtypes = [vp, c_double_p, c_double_p, C.c_double, C.c_int32, C.c_double, C.c_double,
                                   C.c_double, c_double_p, c_double_p, c_double_p, c_double_p, c_double_p,
                                   C.POINTER(C.c_int32)]
    lib.hmcmt_leapfrog_device.argtypes = [vp, vp, vp, C.c_double, C.c_int32, C.c_double, C.c_double, C.c_double, C.c_int32,
                                          vp, vp, vp, C.POINTER(C.c_int32)]
    lib.hmcmt_get_fields.argtypes = [vp, C.c_int32, c_double_p, c_double_p]
    lib.hmcmt_profile.argtypes = [vp, C.c_int32]
    lib.hmcmt_profile_every.argtypes = [vp, C.c_int32]
    lib.hmcmt_profile_read.argtypes = [vp, c_double_p, c_int64_p]
    lib.hmcmt_profile_counters.argtypes = [vp, c_int64_p]
    lib.hmcmt_profile_overhead.argtypes = [vp, C.POINTER(C.c_double)]
    lib.hmcmt_dims.argtypes = [vp, C.POINTER(C.c_int32)]
    lib.hmcmt_debug_transform.argtypes = [vp, C.c_int32, c_double_p, c_double_p]
    lib.hmcmt_comm_id.argtypes = [C.c_char_p]
    lib.hmcmt_comm_create.argtypes = [C.POINTER(vp), C.c_int32, C.c_int32, C.c_int32, C.c_char_p]
    lib.hmcmt_allgather_samples.argtypes = [vp, vp, vp, C.c_int64, C.c_int32]
    lib.hmcmt_comm_destroy.argtypes = [vp]
    lib.hmcmt_comm_last_error.argtypes = [vp]
    lib.hmcmt_comm_last_error.restype = C.c_char_p
    for name in ("hmcmt_comm_id", "hmcmt_comm_create", "hmcmt_allgather_samples", "hmcmt_comm_destroy"):
        getattr(lib, name).restype = C.c_int
    lib.hmcmt_debug_flags.argtypes = [vp, C.c_int32]
    lib.hmcmt_debug_spmv.argtypes = [vp, c_double_p, c_double_p]
    lib.hmcmt_debug_precond.argtypes = [vp, c_double_p, c_double_p]
    lib.hmcmt_debug_persist_precond.argtypes = [vp, C.c_int32, c_double_p, c_double_p]
    lib.hmcmt_persist_info.argtypes = [vp, c_int64_p, C.c_int32]
    lib.hmcmt_debug_hog.argtypes = [vp, C.c_int32, C.c_int32]
    lib.hmcmt_next_cu_share.argtypes = [C.c_int32, C.c_int32]
    lib.hmcmt_persist_width.argtypes = [vp, C.POINTER(C.c_int32)]
    lib.hmcmt_persist_order.argtypes = [vp, C.c_int32, C.POINTER(C.c_int32), c_int64_p]
    lib.hmcmt_persist_pack.argtypes = [c_double_p, C.c_int32, C.c_int32, C.POINTER(C.c_int32), c_double_p]
    lib.hmcmt_persist_envelope.argtypes = [C.c_int64, C.c_int64, C.c_int32, C.c_int64, c_int64_p]
    lib.hmcmt_guard.argtypes = [vp, c_double_p]
    lib.hmcmt_debug_fdm_fwd.argtypes = [vp, c_double_p, c_double_p]
    lib.hmcmt_debug_back_post.argtypes = [vp, c_double_p, c_double_p, c_double_p, c_double_p]
    for name in ("hmcmt_create", "hmcmt_destroy", "hmcmt_set_options", "hmcmt_get_stats", "hmcmt_get_iters",
                 "hmcmt_grad", "hmcmt_forward", "hmcmt_grad_device", "hmcmt_forward_device", "hmcmt_grad_device_async",
                 "hmcmt_wait", "hmcmt_set_prior", "hmcmt_leapfrog", "hmcmt_leapfrog_device", "hmcmt_get_fields", "hmcmt_profile", "hmcmt_profile_every", "hmcmt_profile_read", "hmcmt_profile_counters", "hmcmt_profile_overhead",
                 "hmcmt_dims", "hmcmt_debug_transform", "hmcmt_debug_flags", "hmcmt_debug_spmv", "hmcmt_debug_precond",
                 "hmcmt_debug_fdm_fwd", "hmcmt_debug_back_post", "hmcmt_debug_persist_precond", "hmcmt_persist_info", "hmcmt_guard", "hmcmt_debug_hog", "hmcmt_next_cu_share", "hmcmt_persist_envelope", "hmcmt_persist_width", "hmcmt_persist_order", "hmcmt_persist_pack"):
        getattr(lib, name).restype = C.c_int
    _lib = lib
    return lib


# include/hmcmt.h: the drop-in boundary
PRODUCT_SYMBOLS = ["hmcmt_default_options", "hmcmt_create", "hmcmt_destroy", "hmcmt_last_error",
                   "hmcmt_set_options", "hmcmt_get_stats", "hmcmt_get_iters", "hmcmt_grad", "hmcmt_forward",
                   "hmcmt_grad_device", "hmcmt_forward_device", "hmcmt_grad_device_async", "hmcmt_wait",
                   "hmcmt_set_prior", "hmcmt_leapfrog", "hmcmt_leapfrog_device", "hmcmt_get_fields", "hmcmt_guard", "hmcmt_next_cu_share",
                   "hmcmt_comm_id", "hmcmt_comm_create", "hmcmt_allgather_samples", "hmcmt_comm_destroy", "hmcmt_comm_last_error"]
# include/hmcmt_debug.h: instrumentation, introspection of the persistent kernel, test hooks
DEBUG_SYMBOLS = ["hmcmt_profile", "hmcmt_profile_every", "hmcmt_profile_read", "hmcmt_profile_counters", "hmcmt_profile_overhead", "hmcmt_dims",
                 "hmcmt_debug_transform", "hmcmt_debug_flags", "hmcmt_debug_spmv", "hmcmt_debug_precond", "hmcmt_debug_fdm_fwd",
                 "hmcmt_debug_back_post", "hmcmt_debug_persist_precond", "hmcmt_persist_info", "hmcmt_debug_hog", "hmcmt_persist_envelope",
                 "hmcmt_persist_width", "hmcmt_persist_order", "hmcmt_persist_pack"]
EXPORTED_SYMBOLS = PRODUCT_SYMBOLS + DEBUG_SYMBOLS


def _dp(a):
    return a.ctypes.data_as(c_double_p)


class SampleComm:
    """hmcmt_comm: the RCCL communicator of this process (one process per GPU) for the all-gather of sample blocks."""
    ID_BYTES = 128

    @staticmethod
    def unique_id():
        lib = load_library()
        buf = C.create_string_buffer(SampleComm.ID_BYTES)
        rc = lib.hmcmt_comm_id(buf)
        if rc != 0:
            raise HmcmtError(rc, (lib.hmcmt_comm_last_error(None) or b"").decode())
        return buf.raw

    def __init__(self, device_id, nranks, rank, uid):
        self.lib = load_library()
        if len(uid) != self.ID_BYTES:
            raise ValueError("the RCCL unique id has 128 bytes")
        h = C.c_void_p()
        rc = self.lib.hmcmt_comm_create(C.byref(h), int(device_id), int(nranks), int(rank), uid)
        if rc != 0:
            raise HmcmtError(rc, (self.lib.hmcmt_comm_last_error(None) or b"").decode())
        self.h, self.nranks, self.rank = h, int(nranks), int(rank)

    def allgather(self, block):
        """host float64 block of this rank -> [nranks, len(block)] with every rank's block"""
        send = np.ascontiguousarray(block, dtype=np.float64).reshape(-1)
        recv = np.empty(self.nranks * send.size)
        rc = self.lib.hmcmt_allgather_samples(self.h, send.ctypes.data, recv.ctypes.data, send.size, 0)
        if rc != 0:
            raise HmcmtError(rc, (self.lib.hmcmt_comm_last_error(self.h) or b"").decode())
        return recv.reshape(self.nranks, send.size)

    def allgather_device(self, d_send, d_recv, count):
        rc = self.lib.hmcmt_allgather_samples(self.h, d_send, d_recv, int(count), 1)
        if rc != 0:
            raise HmcmtError(rc, (self.lib.hmcmt_comm_last_error(self.h) or b"").decode())

    def close(self):
        if self.h:
            self.lib.hmcmt_comm_destroy(self.h)
            self.h = None


def persist_envelope(ny, nz, cus_per_xcd=32, nsystems=32):
    """Would a mesh of ny x nz cells (nz incl. the air layers) run the one-launch-per-solve kernel, and in which shape
    (hmcmt_persist_envelope: pure arithmetic, no GPU needed)?  column_parts == 0: outside its envelope."""
    lib = load_library()
    out = (C.c_int64 * 6)()
    rc = lib.hmcmt_persist_envelope(int(ny), int(nz), int(cus_per_xcd), int(nsystems), out)
    if rc != 0:
        raise HmcmtError(rc, "hmcmt_persist_envelope: ny >= 2, nz >= 3, cus_per_xcd >= 1, nsystems >= 1")
    return dict(zip(("column_parts", "threads_half", "workgroups_per_system", "slab_modes", "lds_bytes", "slots_per_xcd"), (int(x) for x in out)))


def persist_pack(cost, queues):
    """(order, makespan, makespan of the index order): the packing behind HipContext.persist_order on the caller's costs
    (hmcmt_persist_pack: pure arithmetic, no GPU needed) -- systems onto `queues` queues that take turns, position
    queue + queues * round -> system."""
    lib = load_library()
    cost = np.ascontiguousarray(cost, dtype=np.float64)
    order = (C.c_int32 * len(cost))()
    m1, m0 = C.c_double(0.0), C.c_double(0.0)
    rc = lib.hmcmt_persist_pack(_dp(cost), len(cost), int(queues), order, C.byref(m1))
    if rc == 0:
        rc = lib.hmcmt_persist_pack(_dp(cost), len(cost), int(queues), None, C.byref(m0))
    if rc != 0:
        raise HmcmtError(rc, "hmcmt_persist_pack: costs >= 0, at least one system and one queue")
    return np.array(list(order)), float(m1.value), float(m0.value)


class HipContext:
    """One GPU context: the drop-in for the reference's per-call solver state."""
    live = 0          # contexts created and not yet closed in this process (a second one on a device switches BOTH to the
                      # launch-per-phase loop: tests/conftest.py fails a test that leaves one behind)

    def __init__(self, mtMesh, mtData, invParam, device_id=0, precond="fdmj", tol=None, maxit=None,
                 verify=False, check_every=None, warm_start=True, fdm_precision="mixed", cu_share=None):
        """cu_share = (index, count): the context is confined to that share (count 1, 2 or 4) of the CUs of every XCD
        (hmcmt_next_cu_share): `count` contexts -- chains -- on one device then run their persistent solve kernels side by side."""
        self.lib = load_library()
        self.args = CreateArgs(mtMesh, mtData, invParam)
        opts = Options()
        self.lib.hmcmt_default_options(C.byref(opts))
        opts.precond = PRECOND[precond]
        if precond == "jacobi" and maxit is None:
            maxit = 20000
        if tol is not None:
            opts.tol = tol
        if maxit is not None:
            opts.maxit = maxit
        if check_every is not None:
            opts.check_every = check_every
        opts.verify = int(verify)
        opts.warm_start = _warm_start_code(warm_start)
        opts.fdm_precision = {"mixed": 0, "fp64": 1}[fdm_precision]
        self.opts = opts
        h = C.c_void_p()
        if cu_share is not None:
            rc = self.lib.hmcmt_next_cu_share(int(cu_share[0]), int(cu_share[1]))       # (this thread's next create)
            if rc != 0:
                raise HmcmtError(rc, f"cu_share {cu_share}: (index, count) with count 1, 2 or 4")
        rc = self.lib.hmcmt_create(C.byref(h), device_id, *self.args.as_tuple(), C.byref(opts))
        if rc != 0:
            raise HmcmtError(rc, (self.lib.hmcmt_last_error(None) or b"").decode())
        self.h = h
        HipContext.live += 1
        d = (C.c_int32 * 7)()
        self.lib.hmcmt_dims(self.h, d)
        self.NYP, self.NZP, self.S, self.ny, self.nz, self.zid, self.nblk = list(d)
        self.nAC, self.nData, self.nFreq = self.args.nAC, self.args.nData, self.args.nFreq

    # -- helpers ------------------------------------------------------------------------------
    def _check(self, rc):
        if rc != 0:
            raise HmcmtError(rc, (self.lib.hmcmt_last_error(self.h) or b"").decode())

    def set_options(self, **kw):
        before = {k: getattr(self.opts, k) for k in kw}
        try:
            self._set_options(**kw)
        except HmcmtError:
            for k, v in before.items():              # (the library kept its options: so does the mirror)
                setattr(self.opts, k, v)
            raise

    def _set_options(self, **kw):
        for k, v in kw.items():
            if k == "precond":
                v = PRECOND[v]
            if k == "fdm_precision":
                v = {"mixed": 0, "fp64": 1}[v]
            if k == "warm_start":
                v = _warm_start_code(v)
            setattr(self.opts, k, v)
        self._check(self.lib.hmcmt_set_options(self.h, C.byref(self.opts)))

    def stats(self):
        s = Stats()
        self._check(self.lib.hmcmt_get_stats(self.h, C.byref(s)))
        return {f: getattr(s, f) for f, _ in Stats._fields_}

    def iters(self):
        it = (C.c_int32 * (2 * self.S))()
        self._check(self.lib.hmcmt_get_iters(self.h, it))
        return np.array(list(it)).reshape(2, self.S)

    # -- hot path -----------------------------------------------------------------------------
    def grad(self, m):
        """compDataGradient: m -> (predData, dataMisfit, dataGrad)."""
        m = np.ascontiguousarray(m, dtype=np.float64)
        if m.shape != (self.nAC,):
            raise ValueError("model vector has the wrong length")
        pred = np.empty(self.nData, dtype=np.complex128)
        grad = np.empty(self.nAC)
        mis = C.c_double()
        self._check(self.lib.hmcmt_grad(self.h, _dp(m), _dp(pred), C.byref(mis), _dp(grad)))
        return (pred.real.copy() if self.args.real_data else pred), mis.value, grad

    def forward(self, m):
        """MT2DFwdSolver + compDataMisfit: m -> (predData, dataMisfit)."""
        m = np.ascontiguousarray(m, dtype=np.float64)
        if m.shape != (self.nAC,):
            raise ValueError("model vector has the wrong length")
        pred = np.empty(self.nData, dtype=np.complex128)
        mis = C.c_double()
        self._check(self.lib.hmcmt_forward(self.h, _dp(m), _dp(pred), C.byref(mis)))
        return (pred.real.copy() if self.args.real_data else pred), mis.value

    def grad_device(self, d_m, d_pred, d_misfit, d_grad):
        """Raw device pointers (ints), e.g. torch tensors' data_ptr()."""
        self._check(self.lib.hmcmt_grad_device(self.h, d_m, d_pred, d_misfit, d_grad))

    def grad_device_async(self, d_m, d_pred, d_misfit, d_grad):
        """Enqueue only (hmcmt_grad_device_async); finish with wait()."""
        self._check(self.lib.hmcmt_grad_device_async(self.h, d_m, d_pred, d_misfit, d_grad))

    def wait(self):
        self._check(self.lib.hmcmt_wait(self.h))

    def forward_device(self, d_m, d_pred, d_misfit):
        self._check(self.lib.hmcmt_forward_device(self.h, d_m, d_pred, d_misfit))

    def fields(self, adjoint=False):
        nn = (self.ny + 1) * (self.nz + 1)
        e = np.empty(nn * self.nFreq, dtype=np.complex128)
        h = np.empty(nn * self.nFreq, dtype=np.complex128)
        self._check(self.lib.hmcmt_get_fields(self.h, int(adjoint), _dp(e), _dp(h)))
        return e.reshape(self.nFreq, nn).T.copy(), h.reshape(self.nFreq, nn).T.copy()

    # -- prior / leapfrog ---------------------------------------------------------------------
    def set_prior(self, mref, Wm, invM):
        Wm = Wm.tocsr()
        self._prior = (np.ascontiguousarray(mref, dtype=np.float64),
                       np.ascontiguousarray(Wm.indptr, dtype=np.int64),
                       np.ascontiguousarray(Wm.indices, dtype=np.int64),
                       np.ascontiguousarray(Wm.data, dtype=np.float64),
                       np.ascontiguousarray(invM, dtype=np.float64))
        a = self._prior
        self._check(self.lib.hmcmt_set_prior(self.h, _dp(a[0]), a[1].ctypes.data_as(c_int64_p),
                                             a[2].ctypes.data_as(c_int64_p), _dp(a[3]), _dp(a[4])))

    def leapfrog(self, m0, p0, dt, L, regParam, lnSigMin, lnSigMax):
        m0 = np.ascontiguousarray(m0, dtype=np.float64)
        p0 = np.ascontiguousarray(p0, dtype=np.float64)
        m1 = np.empty(self.nAC); p1 = np.empty(self.nAC)
        pred = np.empty(self.nData, dtype=np.complex128)
        mis = C.c_double(); mnorm = C.c_double(); nf = C.c_int32()
        self._check(self.lib.hmcmt_leapfrog(self.h, _dp(m0), _dp(p0), dt, L, regParam, lnSigMin, lnSigMax,
                                            _dp(m1), _dp(p1), _dp(pred), C.byref(mis), C.byref(mnorm),
                                            C.byref(nf)))
        return m1, p1, (pred.real.copy() if self.args.real_data else pred), mis.value, mnorm.value, nf.value

    def leapfrog_device(self, d_m, d_p, dt, L, regParam, lnSigMin, lnSigMax, start_grad=0, d_pred=None, d_misfit=None,
                        d_mnorm=None):
        """hmcmt_leapfrog_device: raw device pointers (ints); d_m, d_p are updated in place; finish with wait().
        start_grad: 0 evaluate / 1 start = end model of the previous trajectory / 2 start = its start model."""
        nf = C.c_int32()
        self._check(self.lib.hmcmt_leapfrog_device(self.h, d_m, d_p, dt, L, regParam, lnSigMin, lnSigMax, start_grad,
                                                   d_pred, d_misfit, d_mnorm, C.byref(nf)))
        return nf.value

    # -- instrumentation ----------------------------------------------------------------------
    def profile(self, enable=True, every=1):
        """enable: True (all categories), False, or an iterable of category names to time;
        every: bracket the kernels of every n-th evaluation only."""
        self._check(self.lib.hmcmt_profile_every(self.h, int(every)))
        if enable is True:
            mask = (1 << HMCMT_NCAT) - 1
        elif not enable:
            mask = 0
        else:
            mask = sum(1 << CATEGORIES.index(c) for c in enable)
        self._check(self.lib.hmcmt_profile(self.h, mask))

    def profile_read(self):
        ms = np.zeros(HMCMT_NCAT)
        n = np.zeros(HMCMT_NCAT, dtype=np.int64)
        self._check(self.lib.hmcmt_profile_read(self.h, _dp(ms), n.ctypes.data_as(c_int64_p)))
        return {c: (float(ms[i]), int(n[i])) for i, c in enumerate(CATEGORIES)}

    def profile_overhead_us(self):
        us = C.c_double()
        self._check(self.lib.hmcmt_profile_overhead(self.h, C.byref(us)))
        return us.value

    def profile_counters(self):
        """{active_iter_systems, start_systems, evaluations, solves, solves_two_sweeps, serial_iterations, persistent_solves}
        of the sampled evaluations (hmcmt_profile_counters): serial_iterations = sum over the sampled solves of the slowest
        system's iterations + 1 (the preconditioner application in front of the first one)."""
        o = np.zeros(7, dtype=np.int64)
        self._check(self.lib.hmcmt_profile_counters(self.h, o.ctypes.data_as(c_int64_p)))
        return dict(zip(("active_iter_systems", "start_systems", "evaluations", "solves", "solves_two_sweeps", "serial_iterations",
                         "persistent_solves"), (int(x) for x in o)))

    def _vec(self, a):
        a = np.ascontiguousarray(a, dtype=np.complex128)
        if a.size != self.S * self.NZP * self.NYP:
            raise ValueError("vector must be in the padded nodal layout [S][NZP][NYP]")
        return a

    def debug_fdm_fwd(self, T):
        T = self._vec(T)
        out = np.empty((2,) + T.shape, dtype=np.complex128)
        self._check(self.lib.hmcmt_debug_fdm_fwd(self.h, _dp(T), _dp(out)))
        return out[0], out[1]

    def debug_back_post(self, Yv, R):
        Yv, R = self._vec(Yv), self._vec(R)
        out = np.empty((2,) + Yv.shape, dtype=np.complex128)
        sums = np.zeros(6)
        self._check(self.lib.hmcmt_debug_back_post(self.h, _dp(Yv), _dp(R), _dp(out), _dp(sums)))
        return out[0], out[1], sums

    def debug_transform(self, which, A):
        A = self._vec(A)
        Cc = np.empty_like(A)
        self._check(self.lib.hmcmt_debug_transform(self.h, which, _dp(A), _dp(Cc)))
        return Cc

    def debug_flags(self, freeze_boundary=False, no_boundary_terms=False, fail_placement=False):
        """Test hook (include/hmcmt_debug.h): hold the Dirichlet values at the previous evaluation's / leave dBC^T w out of the gradient /
        (one-shot) let the first system group -- fail_placement="all": EVERY group -- of the next persistent launch fail its placement check."""
        self._cache = None
        place = 8 if fail_placement == "all" else (4 if fail_placement else 0)
        self._check(self.lib.hmcmt_debug_flags(self.h, (1 if freeze_boundary else 0) | (2 if no_boundary_terms else 0) | place))

    def debug_spmv(self, p):
        p = self._vec(p)
        q = np.empty_like(p)
        self._check(self.lib.hmcmt_debug_spmv(self.h, _dp(p), _dp(q)))
        return q

    def debug_precond(self, r):
        r = self._vec(r)
        z = np.empty_like(r)
        self._check(self.lib.hmcmt_debug_precond(self.h, _dp(r), _dp(z)))
        return z

    def guard(self):
        """Production guard of the stopping rule (hmcmt_guard): true residuals formed every HMCMT_GUARD_EVERY-th evaluation."""
        out = (C.c_double * 4)()
        self._check(self.lib.hmcmt_guard(self.h, out))
        return {"checks": int(out[0]), "worst_true_res": float(out[1]), "last_true_res": float(out[2]), "trips": int(out[3])}

    def persist_info(self):
        """The persistent solve kernel and this context (kernels_persist.h): shape, whether it is enabled, how many solves it ran."""
        out = (C.c_int64 * 14)()
        self._check(self.lib.hmcmt_persist_info(self.h, out, 14))
        return dict(zip(("threads_half", "workgroups_per_system", "slots_per_xcd", "enabled", "solves", "placement_fallbacks", "usable_now", "slab_modes",
                         "column_parts", "timeouts", "cu_share_index", "cu_share_count", "strips", "why_off"), (int(x) for x in out)))

    def persist_width(self):
        """Row width (padded nodes) of the width-specialised persistent kernel this context launches; 0: the generic kernel."""
        w = C.c_int32(0)
        self._check(self.lib.hmcmt_persist_width(self.h, C.byref(w)))
        return int(w.value)

    def persist_order(self, kind=0):
        """(order, rebalanced): the order in which the persistent kernel's queues take the systems of a forward (0) / adjoint (1)
        solve -- position queue + queues * round -> system -- and how often the context has re-balanced it (hmcmt_persist_order)."""
        order = (C.c_int32 * self.S)()
        n = C.c_int64(0)
        self._check(self.lib.hmcmt_persist_order(self.h, int(kind), order, C.byref(n)))
        return np.array(list(order)), int(n.value)

    def debug_hog(self, nblocks, ms):
        """Test hook: nblocks workgroups holding a CU's LDS each for ms milliseconds on a stream of their own (returns once they are resident)."""
        self._check(self.lib.hmcmt_debug_hog(self.h, int(nblocks), int(ms)))

    def debug_persist_precond(self, r, sweeps=1):
        """z = P^-1 r by the persistent solve kernel's preconditioner (kernels_persist.h)."""
        r = self._vec(r)
        z = np.empty_like(r)
        self._check(self.lib.hmcmt_debug_persist_precond(self.h, int(sweeps), _dp(r), _dp(z)))
        return z

    def close(self):
        if getattr(self, "h", None):
            self.lib.hmcmt_destroy(self.h)
            self.h = None
            HipContext.live -= 1

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
