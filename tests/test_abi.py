"""The C-ABI shared library: builds for gfx950, loads, exports every symbol include/hmcmt.h declares,
and refuses to run without a HIP device (no CPU fallback).  No compute calls here."""
import ctypes
import os
import re

import numpy as np
import pytest

from hmcmt2d_amd import lib as L
from tests.conftest import HAVE_GPU, ROOT


@pytest.fixture(scope="module")
def so():
    return ctypes.CDLL(L.build_library())


def declared_functions(header="hmcmt.h", pattern=r"\b(hmcmt_[a-z_]+)\s*\("):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(pattern, text)))


def test_header_symbols_are_exported(so):
    """include/hmcmt.h = the drop-in boundary (create / destroy / grad / forward / leapfrog / fields / stats /
    options / guard / CU shares / comm), include/hmcmt_debug.h = instrumentation, persistent-kernel introspection and test hooks
    (round 6: split).  Every declared symbol is exported, and the Python mirror lists exactly these."""
    names = declared_functions()
    assert len(names) >= 18
    for n in names:
        assert hasattr(so, n), f"{n} declared in include/hmcmt.h but not exported"
    assert set(names) == set(L.PRODUCT_SYMBOLS)
    assert not [n for n in names if n.startswith(("hmcmt_debug_", "hmcmt_profile", "hmcmt_persist_"))]
    dbg = declared_functions("hmcmt_debug.h")
    for n in dbg:
        assert hasattr(so, n), f"{n} declared in include/hmcmt_debug.h but not exported"
    assert set(dbg) == set(L.DEBUG_SYMBOLS)


def test_mumps_interface_symbols_are_exported(so):
    """include/hmcmt_mumps.h: the eight Fortran-convention symbols MUMPS/src/MUMPSfuncs.jl binds (:32,49,105,115,
    128,139,155,170) plus the statistics call."""
    names = declared_functions("hmcmt_mumps.h", r"\b([a-z_]+mumps[a-z_]*)\s*\(")
    assert set(names) == {"factor_mumps_cmplx_", "factor_mumps_", "solve_mumps_cmplx_", "solve_mumps_",
                          "solve_mumps_cmplx_sparse_rhs_", "solve_mumps_sparse_rhs_", "destroy_mumps_cmplx_",
                          "destroy_mumps_", "hmcmt_mumps_last_solve"}
    for n in names:
        assert hasattr(so, n), f"{n} declared in include/hmcmt_mumps.h but not exported"


def test_default_options(so):
    o = L.Options()
    L.load_library().hmcmt_default_options(ctypes.byref(o))
    assert o.precond == 2 and o.maxit >= 100 and 0 < o.tol < 1e-8 and o.check_every >= 1


@pytest.mark.skipif(HAVE_GPU, reason="checks the no-device error path")
def test_create_fails_loudly_without_a_device():
    from tests.helpers import make_problem
    mesh, data, inv, m = make_problem("tiny")
    with pytest.raises(L.HmcmtError) as e:
        L.HipContext(mesh, data, inv)
    assert e.value.code == -2 and "no host compute path" in str(e.value)


def test_create_rejects_bad_arguments_before_touching_the_device():
    from tests.helpers import make_problem
    import numpy as np
    mesh, data, inv, m = make_problem("tiny")
    data.rxLoc = data.rxLoc.copy(); data.rxLoc[:, 1] = 37.0      # no grid node at that depth
    with pytest.raises(L.HmcmtError) as e:
        L.HipContext(mesh, data, inv)
    assert e.value.code == -1 and "receiver depth" in str(e.value)
    mesh, data, inv, m = make_problem("tiny")
    data.dataType = "Rho_Pha"
    with pytest.raises(ValueError):
        L.HipContext(mesh, data, inv)


@pytest.mark.skipif(HAVE_GPU, reason="needs a GPU-less box")
def test_mumps_interface_without_a_device_fails_loudly():
    """No CPU fallback behind the MUMPS symbols either: factor reports stat < 0, which the wrapper raises
    (MUMPSfuncs.jl:59-73)."""
    import numpy as np
    import scipy.sparse as sp
    from hmcmt2d_amd import mumps as M
    A = sp.csc_matrix(np.array([[2.0, -1.0], [-1.0, 2.0]]))
    with pytest.raises(RuntimeError, match="MUMPS: error"):
        M.factorMUMPS(A, 1)


def _build_abi_check(tmp_path):
    import subprocess
    exe = str(tmp_path / "abi_check")
    subprocess.check_call(["gcc", "-std=c11", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "c_abi", "abi_check.c"), "-o", exe, "-ldl"])
    return exe


def test_headers_compile_as_c11_and_match_the_ctypes_mirror(tmp_path):
    """include/hmcmt.h and include/hmcmt_mumps.h through a real C compiler (gcc -std=c11 -Wall -Wextra -Werror):
    sizeof / offsetof of hmcmt_options and hmcmt_stats must equal the ctypes structs of hmcmt2d_amd/lib.py (and of
    julia/HMCMTHip.jl, which mirrors the same field order), the constants the Python side hard-codes must agree."""
    import subprocess
    out = subprocess.check_output([_build_abi_check(tmp_path)], text=True)
    sizes, offs, consts = {}, {}, {}
    for line in out.splitlines():
        t = line.split()
        if t[0] == "sizeof":
            sizes[t[1]] = int(t[2])
        elif t[0] == "offsetof":
            offs[t[1]] = int(t[2])
        elif t[0] == "const":
            consts[t[1]] = int(t[2])
    assert "mumps prototypes" in out
    for cname, cls in (("hmcmt_options", L.Options), ("hmcmt_stats", L.Stats)):
        assert sizes[cname] == ctypes.sizeof(cls)
        fields = [f for f, _ in cls._fields_]
        assert sorted(k.split(".")[1] for k in offs if k.startswith(cname + ".")) == sorted(f for f in fields if f != "reserved_")
        for f in fields:
            if f != "reserved_":
                assert offs[f"{cname}.{f}"] == getattr(cls, f).offset, f
    assert consts["HMCMT_NCAT"] == L.HMCMT_NCAT == len(L.CATEGORIES)
    assert consts["HMCMT_ENOCONV"] == -10 and consts["HMCMT_EBREAKDOWN"] == -11 and consts["HMCMT_ENODEV"] == -2
    assert L.ERRORS[consts["HMCMT_ENOCONV"]] == "ENOCONV" and L.PRECOND["fdmj"] == consts["HMCMT_PRECOND_FDM_JACOBI"]


def test_julia_binding_mirrors_the_struct_layout():
    """julia/HMCMTHip.jl cannot be executed here (no julia binary): at least its struct field lists must be the
    header's, in order, and it must bind every hot-path entry point."""
    src = open(os.path.join(ROOT, "julia", "HMCMTHip.jl")).read()
    def fields(name):
        body = re.search(r"struct %s\b(.*?)\nend" % name, src, flags=re.S).group(1)
        return re.findall(r"^\s*([a-z_]+)::", body, flags=re.M)
    assert fields("HmcmtOptions") == [f for f, _ in L.Options._fields_]
    assert fields("HmcmtStats") == [f for f, _ in L.Stats._fields_]
    assert "using LinearAlgebra" in src
    for sym in ("hmcmt_create", "hmcmt_grad", "hmcmt_forward", "hmcmt_destroy", "hmcmt_set_prior", "hmcmt_leapfrog",
                "hmcmt_leapfrog_device", "hmcmt_wait", "hmcmt_get_stats", "hmcmt_last_error", "hmcmt_comm_id", "hmcmt_comm_create",
                "hmcmt_allgather_samples", "hmcmt_comm_destroy"):
        assert f":{sym}" in src, sym
    # the component map is the Python binding's (include/hmcmt.h: 1 ZXY .. 6 PhsYX), both data types are accepted, and
    # the ccall signatures carry as many argument types as the header's prototypes have parameters
    from hmcmt2d_amd.marshal import COMPONENT_CODES
    jl = dict((k, int(v)) for k, v in re.findall(r'"(\w+)"\s*=>\s*(\d)', re.search(r"const COMPONENT_CODES = Dict\((.*?)\)", src, flags=re.S).group(1)))
    assert jl == COMPONENT_CODES
    assert "only DataType Impedance" not in src and 'occursin("Rho_Pha", dataType)' in src
    assert "ctx.havePrior ||" not in src and "isdiag(hmcParam.invM)" in src
    hdr = open(os.path.join(ROOT, "include", "hmcmt.h")).read()
    for sym in ("hmcmt_grad", "hmcmt_forward", "hmcmt_set_prior", "hmcmt_leapfrog", "hmcmt_leapfrog_device", "hmcmt_create"):
        proto = re.search(r"int %s\((.*?)\);" % sym, hdr, flags=re.S).group(1)
        nargs_c = len(proto.split(","))
        call = re.search(r"ccall\(\(:%s, libhmcmt\), Cint,\s*\((.*?)\),\s*\n\s*ctx" % sym, src, flags=re.S).group(1)
        nargs_jl = len([t for t in re.sub(r"\{[^}]*\}", "", call).split(",") if t.strip()])
        assert nargs_c == nargs_jl, (sym, nargs_c, nargs_jl)


def test_rccl_loader_reports_a_missing_library_instead_of_crashing():
    """hmcmt_comm_id / hmcmt_comm_create on a box whose librccl cannot be loaded must return HMCMT_ENODEV
    (include/hmcmt.h), not crash: a fresh process points the loader at a path that does not exist."""
    import subprocess
    import sys
    code = ("import ctypes, sys\n"
            "so = ctypes.CDLL(sys.argv[1])\n"
            "buf = ctypes.create_string_buffer(128)\n"
            "rc = so.hmcmt_comm_id(buf)\n"
            "h = ctypes.c_void_p()\n"
            "rc2 = so.hmcmt_comm_create(ctypes.byref(h), 0, 1, 0, buf)\n"
            "print(rc, rc2)\n")
    env = dict(os.environ, HMCMT_RCCL_PATH="/nonexistent/librccl.so.1")
    out = subprocess.run([sys.executable, "-c", code, L.SO_PATH], env=env, capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr[-400:]
    assert out.stdout.split() == ["-2", "-2"]


@pytest.mark.gpu
def test_compiled_c_consumer_runs_the_hot_path(tmp_path):
    """The same C program dlopen()s libhmcmt_hip.so and runs hmcmt_create / hmcmt_grad / hmcmt_get_stats /
    hmcmt_destroy on the tiny config from a binary dump: no Python, no ctypes between the header and the library."""
    import subprocess
    import numpy as np
    from tests.c_abi.dump import write_dump
    from tests.helpers import make_problem, oracle_eval, relmax
    exe = _build_abi_check(tmp_path)
    mesh, data, inv, m = make_problem("tiny")
    dump = str(tmp_path / "tiny.bin")
    a = write_dump(dump, mesh, data, inv, m)
    out = subprocess.check_output([exe, "run", L.SO_PATH, dump], text=True)
    kv = {ln.split()[0]: ln.split()[1:] for ln in out.splitlines()}
    assert kv["status"] == ["0"] and kv["destroy"] == ["0"] and int(kv["nsystems"][0]) == 2 * a.nFreq
    raw = np.fromfile(dump + ".out")
    pred = raw[:2 * a.nData].view(np.complex128); grad = raw[2 * a.nData:]
    po, mo, go = oracle_eval(mesh, data, inv, m)
    assert relmax(pred, po) < 1e-9 and relmax(grad, go) < 1e-7 and abs(float(kv["misfit"][0]) - mo) / mo < 1e-9
    assert float(kv["true_res"][0]) < 1e-9 and int(kv["iters"][0]) > 0


def test_envelope_of_the_persistent_kernel_without_a_device():
    """hmcmt_persist_envelope: which meshes run the one-launch-per-solve kernel and in which shape -- pure arithmetic in the
    library (the same function hmcmt_create uses), so it is checked here, in the GPU-less container, against the shapes the GPU
    runs reported (profiles/r05_shape_fuzz.log, tests/test_gpu_persist.py): BASELINE's configurations, the width limits of one
    and two column parts, tall narrow meshes that take more threads than their width needs, CU shares."""
    from hmcmt2d_amd.lib import persist_envelope as env
    e = env(200, 107, nsystems=32)                       # cfg3: one part, 8 workgroups of 512 threads per system, 4 systems per XCD
    assert (e["column_parts"], e["threads_half"], e["workgroups_per_system"], e["slab_modes"], e["slots_per_xcd"]) == (1, 256, 8, 32, 4)
    assert 150 * 1024 < e["lds_bytes"] <= 160 * 1024
    e = env(400, 207, nsystems=64)                       # cfg5: two column parts, 15 x 2 workgroups, one system per XCD at a time
    assert (e["column_parts"], e["threads_half"], e["workgroups_per_system"], e["slab_modes"], e["slots_per_xcd"]) == (2, 256, 30, 16, 1)
    assert 158 * 1024 < e["lds_bytes"] <= 160 * 1024
    assert (env(50, 32, nsystems=16)["column_parts"], env(50, 32, nsystems=16)["threads_half"]) == (1, 64)        # cfg2
    assert env(96, 56, nsystems=8)["workgroups_per_system"] == 4                                                    # cfg1
    assert env(207, 107)["column_parts"] == 1 and env(208, 107)["column_parts"] == 2                                # the width of one tile
    assert env(415, 47)["column_parts"] == 2 and env(416, 47)["column_parts"] == 0 and env(440, 15)["column_parts"] == 0
    e = env(60, 100)                                      # narrow and tall: 2 x 128 threads for a 64-column tile, 16-mode slabs
    assert (e["column_parts"], e["threads_half"], e["slab_modes"]) == (1, 128, 16)
    e = env(100, 257)                                     # 19 row blocks: 2 x 256 threads, 16-mode slabs
    assert (e["column_parts"], e["threads_half"], e["workgroups_per_system"], e["slab_modes"]) == (1, 256, 19, 16)
    assert env(100, 600)["column_parts"] == 0             # 43 row blocks do not fit an XCD
    assert env(200, 107, cus_per_xcd=16)["slots_per_xcd"] == 2 and env(400, 207, cus_per_xcd=16, nsystems=64)["column_parts"] == 0   # CU shares


def test_queue_packing_of_the_persistent_kernel_without_a_device():
    """hmcmt_persist_pack: the packing behind the balanced queues of the persistent kernel (meshes whose systems take turns on the
    chip; persist_balance makes its tables with the same function) -- pure arithmetic, checked here: the table is a permutation,
    every queue keeps the positions the index order gives it, the largest queue sum is not above the index order's and within
    the longest-first bound (mean load + the largest cost); cfg5's measured counts (64 systems, 8 queues: index order 157 of a
    mean of 151.75) come out within 1 % of the mean."""
    from hmcmt2d_amd.lib import persist_pack
    rng = np.random.default_rng(3)
    for S, NQ in ((64, 8), (40, 32), (32, 16), (7, 3), (5, 8), (1, 1)):
        for trial in range(20):
            cost = np.round(rng.uniform(5, 40, S)) if trial else np.full(S, 7.0)
            order, span, span0 = persist_pack(cost, NQ)
            assert sorted(order.tolist()) == list(range(S))
            sums = [cost[order[q::NQ]].sum() for q in range(min(NQ, S))]
            assert abs(max(sums) - span) < 1e-9 and abs(max(cost[q::NQ].sum() for q in range(min(NQ, S))) - span0) < 1e-9
            assert span <= span0 + 1e-9
            if S % NQ == 0:                               # (equal queues: the classical bound of longest-first list scheduling)
                assert span <= cost.sum() / NQ + cost.max() + 1e-9
    fwd = [21, 22.2, 20.5, 18.8, 17.8, 17, 17, 15, 14, 14, 12, 12, 11.5, 11, 10.5, 10.2, 10.8, 10.8, 10.8, 11.2, 12, 12, 12, 12, 12, 12, 12, 11, 11, 11,
           11, 10, 23, 23, 23.5, 23.8, 23.8, 23.8, 23, 23, 23, 23, 22.8, 22.5, 22.5, 22, 22, 22, 23, 24.2, 24.5, 25.2, 26.5, 26.5, 28, 27.8, 27.8,
           27.8, 28.2, 27, 27, 25.8, 25.8, 26.5]
    order, span, span0 = persist_pack(fwd, 8)
    assert abs(span0 - 157.0) < 0.3 and span < 1.01 * sum(fwd) / 8
    from hmcmt2d_amd.lib import HmcmtError
    with pytest.raises(HmcmtError):
        persist_pack([1.0, -2.0], 2)



def _c_prototypes():
    """name -> (return type, [(type, name)]) of every function the two headers declare."""
    protos = {}
    for header in ("hmcmt.h", "hmcmt_debug.h"):
        text = open(os.path.join(ROOT, "include", header)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        for m in re.finditer(r"^\s*(int|void|const char\*)\s+(hmcmt_[a-z_]+)\s*\(([^;]*?)\)\s*;", text, flags=re.M | re.S):
            params = []
            for p in [q.strip() for q in m.group(3).replace("\n", " ").split(",") if q.strip() and q.strip() != "void"]:
                mm = re.match(r"(.*?)([A-Za-z_][A-Za-z_0-9]*)\s*(\[[^\]]*\])?$", p)
                ty = (mm.group(1) + ("*" if mm.group(3) else "")).replace(" ", "")
                params.append((ty, mm.group(2)))
            protos[m.group(2)] = (m.group(1), params)
    return protos


# which Julia ccall types may carry a C parameter type (julia/HMCMTHip.jl: ComplexF64 arrays are interleaved doubles, Ref{T} is T*)
_JL_OK = {
    "hmcmt_ctx**": {"Ref{Ptr{Cvoid}}"}, "hmcmt_comm**": {"Ref{Ptr{Cvoid}}"},
    "hmcmt_ctx*": {"Ptr{Cvoid}"}, "consthmcmt_ctx*": {"Ptr{Cvoid}"}, "hmcmt_comm*": {"Ptr{Cvoid}"}, "consthmcmt_comm*": {"Ptr{Cvoid}"},
    "int32_t": {"Int32", "Cint"}, "int64_t": {"Int64"}, "double": {"Float64", "Cdouble"},
    "constdouble*": {"Ptr{Float64}", "Ptr{ComplexF64}", "Ptr{Cvoid}"}, "double*": {"Ptr{Float64}", "Ptr{ComplexF64}", "Ref{Float64}", "Ptr{Cvoid}"},
    "constint64_t*": {"Ptr{Int64}"}, "int64_t*": {"Ptr{Int64}", "Ref{Int64}"}, "int32_t*": {"Ptr{Int32}", "Ref{Int32}"},
    "constuint8_t*": {"Ptr{UInt8}"}, "void*": {"Ptr{UInt8}", "Ptr{Cvoid}"}, "constvoid*": {"Ptr{UInt8}", "Ptr{Cvoid}"},
    "consthmcmt_options*": {"Ptr{HmcmtOptions}", "Ref{HmcmtOptions}"}, "hmcmt_options*": {"Ref{HmcmtOptions}"}, "hmcmt_stats*": {"Ref{HmcmtStats}"},
}


def test_julia_ccalls_match_the_c_prototypes_argument_by_argument():
    """julia/HMCMTHip.jl has never been executed (no Julia in the image): every `ccall` in it is held to the C
    prototype of include/hmcmt.h / hmcmt_debug.h -- return type, NUMBER, ORDER and TYPE of the arguments -- and the values handed to
    the 25-argument hmcmt_create are checked name by name against the header's parameter names."""
    src = open(os.path.join(ROOT, "julia", "HMCMTHip.jl")).read()
    protos = _c_prototypes()
    seen = set()
    for m in re.finditer(r"ccall\(\(:(\w+), libhmcmt\),\s*(\w+),\s*\(", src):
        name, ret = m.group(1), m.group(2)
        i = j = m.end(); depth = 1
        while depth:
            depth += src[j] == "("; depth -= src[j] == ")"; j += 1
        types = [t.strip() for t in re.split(r",(?![^{]*\})", src[i:j - 1]) if t.strip()]
        assert name in protos, f"{name}: ccall of a symbol no header declares"
        cret, params = protos[name]
        assert (ret, cret) in (("Cint", "int"), ("Cstring", "const char*"), ("Cvoid", "void")), (name, ret, cret)
        assert len(types) == len(params), f"{name}: {len(types)} ccall argument types, {len(params)} C parameters"
        for k, (jt, (ct, pname)) in enumerate(zip(types, params)):
            assert jt in _JL_OK[ct], f"{name}: argument {k + 1} ({pname}) is {ct} in C, {jt} in the ccall"
        seen.add(name)
        if name == "hmcmt_create":
            # the values: one per type, in the header's order (lengths for the counts)
            k = j
            while src[k] != ",":
                k += 1
            depth = 0; k += 1; start = k
            while not (src[k] == ")" and depth == 0):
                depth += src[k] in "(["; depth -= src[k] in ")]"; k += 1
            vals = [v.strip() for v in re.split(r",(?![^(]*\))", src[start:k]) if v.strip()]
            assert len(vals) == len(params) == 25
            expect = {"ny": "ny", "nz": "nz", "yLen": "yLen", "zLen": "zLen", "origin": "origin", "nFreq": "length(mtData.freqs)", "freqs": "freqs",
                      "nRx": "length(rxY)", "rxY": "rxY", "rxZ": "rxZ", "nComp": "length(compMode)", "compMode": "compMode",
                      "nData": "length(obs)", "freqID": "freqID", "rxID": "rxID", "dtID": "dtID", "dataID": "dataID", "obs": "obs",
                      "dataW": "dataW", "nAC": "length(activeIdx)", "activeIdx": "activeIdx", "bgModel": "bgModel", "device_id": "device"}
            for (ct, pname), v in zip(params, vals):
                if pname in expect:
                    assert expect[pname] in v, f"hmcmt_create: parameter {pname} receives `{v}`"
    for sym in ("hmcmt_create", "hmcmt_grad", "hmcmt_forward", "hmcmt_leapfrog", "hmcmt_leapfrog_device", "hmcmt_set_prior", "hmcmt_get_stats"):
        assert sym in seen
