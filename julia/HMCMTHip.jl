#-------------------------------------------------------------------------------
# HMCMTHip -- the reference-side binding of libhmcmt_hip.so (include/hmcmt.h).
#
# Drop this module next to the reference's HMCMT package.  It overrides exactly the hot path:
#   compDataGradient(mtMesh, mtData, invParam, hmcprior)   (HMCSampler/HMCSampler.jl:277-330)
#   the forward solve inside getHamiltonian                (HMCSampler/HMCSampler.jl:358-397)
#   proposeLeapfrog(hmcParamCurrent, mtMesh, ...)          (HMCSampler/HMCSampler.jl:206-269), optional:
#                                                          the whole trajectory on the GPU (hmcmt_leapfrog)
# and leaves every other reference function (readstartupFile, runHMCSampler, parallelHMCSampler, writers)
# untouched.  Select it with `linearsolver: hip` in the startup file.
#
# One process = one GPU: under `addprocs(n)` + parallelHMCSampler (parallelHMC.jl:23-40) worker p uses device
# (p - 2) mod ndevices (worker ids start at 2), the master process device 0; HMCMT_DEVICE overrides.
#
# NOTE: this file could not be executed in the build container (no julia binary; tests/test_abi.py checks its
# struct field lists and bound symbols against include/hmcmt.h mechanically).  It is a thin `ccall` layer: every
# call passes the reference's own arrays straight through -- Julia's Vector{Float64} / Vector{ComplexF64} /
# Vector{Int64} have exactly the memory layout the C ABI expects (interleaved re/im doubles, 1-based int64
# indices).  The same ABI is exercised from Python/ctypes by tests/test_gpu_parity*.py and from compiled C by
# tests/c_abi/abi_check.c.
#-------------------------------------------------------------------------------
module HMCMTHip

using LinearAlgebra            # diag
using SparseArrays
using Distributed              # myid
using HMCMT.HMCFileIO, HMCMT.HMCStruct, HMCMT.HMCUtility

export HipContext, hipContext, compDataGradient, hipForward, setPrior!, proposeLeapfrog, proposeLeapfrogDevice!,
       hipWait, hipStats, hipGuard, hipPersistInfo, hipPersistWidth, hipPersistEnvelope, hipPersistOrder, hipPersistPack, hipNextCuShare, destroy!, commId, SampleComm, allgatherSamples

const libhmcmt = get(ENV, "HMCMT_HIP_LIB", joinpath(@__DIR__, "..", "hmcmt2d_amd", "libhmcmt_hip.so"))

# field order and types of include/hmcmt.h (checked by tests/test_abi.py against the ctypes mirror)
struct HmcmtOptions
    precond::Int32
    maxit::Int32
    tol::Float64
    check_every::Int32
    verify::Int32
    warm_start::Int32
    fdm_precision::Int32
end

struct HmcmtStats
    iters_fwd_max::Int32
    iters_adj_max::Int32
    iters_fwd_sum::Int32
    iters_adj_sum::Int32
    err_est_max::Float64
    true_res_max::Float64
    status::Int32
    nsystems::Int32
    fallback_solves::Int32
    smoother_sweeps::Int32
end

mutable struct HipContext
    ptr::Ptr{Cvoid}
    nAC::Int
    nData::Int
    device::Int
    realData::Bool          # DataType Rho_Pha: the reference's obsData / predData are real vectors
end

# component codes of include/hmcmt.h (the same table as hmcmt2d_amd/marshal.py COMPONENT_CODES; tests/test_abi.py compares them)
const COMPONENT_CODES = Dict("ZXY" => 1, "ZYX" => 2, "RhoXY" => 3, "PhsXY" => 4, "RhoYX" => 5, "PhsYX" => 6)

function compModes(dataComp, dataType::AbstractString)
    isimp = occursin("Impedance", dataType)
    (isimp || occursin("Rho_Pha", dataType) || occursin("Rho_Phs", dataType)) ||
        error("unsupported DataType $dataType (Impedance or Rho_Pha)")
    out = Int64[]
    for c in dataComp
        haskey(COMPONENT_CODES, c) || error("unsupported data component $c (log10Rho*: the reference's forward and sensitivity disagree on it)")
        code = COMPONENT_CODES[c]
        (code <= 2) == isimp || error("data component $c does not belong to DataType $dataType")
        push!(out, code)
    end
    return out
end

function checkerr(ctx::Ptr{Cvoid}, rc::Cint)
    rc == 0 && return
    msg = unsafe_string(ccall((:hmcmt_last_error, libhmcmt), Cstring, (Ptr{Cvoid},), ctx))
    error("libhmcmt_hip error $rc: $msg")        # same behaviour as checkMUMPSerror (MUMPSfuncs.jl:59-73)
end

"""
    defaultDevice()

GPU of this Julia process: `HMCMT_DEVICE` if set, else worker id - 2 modulo `HMCMT_NDEVICES` (default 8) on a
worker started by `addprocs`, else 0.
"""
function defaultDevice()
    haskey(ENV, "HMCMT_DEVICE") && return parse(Int, ENV["HMCMT_DEVICE"])
    ndev = parse(Int, get(ENV, "HMCMT_NDEVICES", "8"))
    return myid() >= 2 ? mod(myid() - 2, ndev) : 0
end

"""
    hipContext(mtMesh, mtData, invParam; device=defaultDevice())

Builds the GPU context once per run (replaces the operator set-up the reference redoes on every
call: setupTensorMesh2D!, getBoundaryIndex, preSetRxFieldSens).
"""
function hipContext(mtMesh::TensorMesh2D, mtData::MTData, invParam::InvDataModel; device::Integer=defaultDevice())
    ny, nz = mtMesh.gridSize
    compMode = compModes(mtData.dataComp, mtData.dataType)
    realData = !occursin("Impedance", mtData.dataType)
    rxY = Vector{Float64}(mtData.rxLoc[:, 1]); rxZ = Vector{Float64}(mtData.rxLoc[:, 2])
    dataID = Vector{UInt8}(mtData.dataID)
    obs = Vector{ComplexF64}(invParam.obsData)
    dataW = Vector{Float64}(diag(invParam.dataW))
    activeIdx = Vector{Int64}(invParam.activeCell.rowval)        # 1-based cell id of each column
    ctxref = Ref{Ptr{Cvoid}}(C_NULL)
    rc = ccall((:hmcmt_create, libhmcmt), Cint,
               (Ref{Ptr{Cvoid}}, Int32,
                Int64, Int64, Ptr{Float64}, Ptr{Float64}, Ptr{Float64},
                Int64, Ptr{Float64},
                Int64, Ptr{Float64}, Ptr{Float64},
                Int64, Ptr{Int64},
                Int64, Ptr{Int64}, Ptr{Int64}, Ptr{Int64},
                Ptr{UInt8}, Ptr{ComplexF64}, Ptr{Float64},
                Int64, Ptr{Int64}, Ptr{Float64}, Ptr{HmcmtOptions}),
               ctxref, Int32(device),
               ny, nz, mtMesh.yLen, mtMesh.zLen, mtMesh.origin,
               length(mtData.freqs), mtData.freqs,
               length(rxY), rxY, rxZ,
               length(compMode), compMode,
               length(obs), mtData.freqID, mtData.rxID, mtData.dtID,
               dataID, obs, dataW,
               length(activeIdx), activeIdx, invParam.bgModel, C_NULL)
    checkerr(C_NULL, rc)
    ctx = HipContext(ctxref[], length(activeIdx), length(obs), Int(device), realData)
    finalizer(destroy!, ctx)
    return ctx
end

function destroy!(ctx::HipContext)
    if ctx.ptr != C_NULL
        ccall((:hmcmt_destroy, libhmcmt), Cint, (Ptr{Cvoid},), ctx.ptr)
        ctx.ptr = C_NULL
    end
end

# one context per InvDataModel of this process (WeakKeyDict: the context goes with its problem)
const contexts = WeakKeyDict{Any,HipContext}()
getctx(mtMesh, mtData, invParam) = get!(() -> hipContext(mtMesh, mtData, invParam), contexts, invParam)

"""
    compDataGradient(mtMesh, mtData, invParam, hmcprior) -> (predData, dataMisfit, dataGrad)

Same signature and return values as HMCSampler.compDataGradient (HMCSampler.jl:277-330).
"""
function compDataGradient(mtMesh::TensorMesh2D, mtData::MTData, invParam::InvDataModel, hmcprior::HMCPrior)
    ctx = getctx(mtMesh, mtData, invParam)
    m = invParam.strModel
    pred = Vector{ComplexF64}(undef, ctx.nData)
    grad = Vector{Float64}(undef, ctx.nAC)
    misfit = Ref{Float64}(0.0)
    rc = ccall((:hmcmt_grad, libhmcmt), Cint,
               (Ptr{Cvoid}, Ptr{Float64}, Ptr{ComplexF64}, Ref{Float64}, Ptr{Float64}),
               ctx.ptr, m, pred, misfit, grad)
    checkerr(ctx.ptr, rc)
    # keep mtMesh.sigma in the state the reference leaves it in (HMCSampler.jl:293-294)
    mtMesh.sigma = invParam.activeCell * exp.(m) + invParam.bgModel
    return (ctx.realData ? real.(pred) : pred), misfit[], grad
end

"""
    hipForward(mtMesh, mtData, invParam) -> (predData, dataMisfit)

Replaces `MT2DFwdSolver` + `compDataMisfit` in getHamiltonian (HMCSampler.jl:364,384).
"""
function hipForward(mtMesh::TensorMesh2D, mtData::MTData, invParam::InvDataModel)
    ctx = getctx(mtMesh, mtData, invParam)
    pred = Vector{ComplexF64}(undef, ctx.nData)
    misfit = Ref{Float64}(0.0)
    rc = ccall((:hmcmt_forward, libhmcmt), Cint,
               (Ptr{Cvoid}, Ptr{Float64}, Ptr{ComplexF64}, Ref{Float64}),
               ctx.ptr, invParam.strModel, pred, misfit)
    checkerr(ctx.ptr, rc)
    return (ctx.realData ? real.(pred) : pred), misfit[]
end

"""
    setPrior!(ctx, invParam, hmcParam)

Registers the prior (refModel, Wm) and the diagonal of M^-1 for device-resident trajectories (hmcmt_set_prior).
Wm is symmetric, so its CSC arrays are its CSR arrays; they go over 0-based.  Only the reference's default, diagonal
mass matrix is supported (setMassMatrix's dense option, HMCSampler.jl:478-489, is refused, not silently truncated).
"""
function setPrior!(ctx::HipContext, invParam::InvDataModel, hmcParam::HMCParameter)
    (hmcParam.invM isa Diagonal || isdiag(hmcParam.invM)) ||
        error("HMCMTHip: only diagonal mass matrices are supported on the device (masstype: diagonal)")
    Wm = invParam.Wm
    rowptr = Vector{Int64}(Wm.colptr .- 1)
    colind = Vector{Int64}(Wm.rowval .- 1)
    val = Vector{Float64}(Wm.nzval)
    invM = Vector{Float64}(diag(hmcParam.invM))
    rc = ccall((:hmcmt_set_prior, libhmcmt), Cint,
               (Ptr{Cvoid}, Ptr{Float64}, Ptr{Int64}, Ptr{Int64}, Ptr{Float64}, Ptr{Float64}),
               ctx.ptr, invParam.refModel, rowptr, colind, val, invM)
    checkerr(ctx.ptr, rc)
    return ctx
end

"""
    proposeLeapfrog(hmcParamCurrent, mtMesh, mtData, invParam, hmcprior) -> (propModel, propMomentum)

Same signature and return values as HMCSampler.proposeLeapfrog (HMCSampler.jl:206-269); the L + 1 gradient
evaluations, the momentum / position updates, the step clamp and the bound reflection run on the GPU
(hmcmt_leapfrog).  The forward response at the proposal comes back with the trajectory: getHamiltonian's
`hipForward` at the same model is then answered from the library's memo without a solve.
"""
function proposeLeapfrog(hmcParamCurrent::HMCParameter, mtMesh::TensorMesh2D, mtData::MTData,
                         invParam::InvDataModel, hmcprior::HMCPrior)
    ctx = getctx(mtMesh, mtData, invParam)
    # every trajectory: runHMCSampler re-randomises invParam.refModel on each run (HMCSampler.jl:100-109) and the mass
    # matrix may change between runs; the upload is O(nnz(Wm)), nothing beside a trajectory
    setPrior!(ctx, invParam, hmcParamCurrent)
    intstep = rand(hmcprior.timestep[1]:hmcprior.timestep[2])          # unirandInteger (HMCUtility.jl)
    n = ctx.nAC
    m1 = Vector{Float64}(undef, n); p1 = Vector{Float64}(undef, n)
    pred = Vector{ComplexF64}(undef, ctx.nData)
    misfit = Ref{Float64}(0.0); mnorm = Ref{Float64}(0.0); nf = Ref{Int32}(0)
    rc = ccall((:hmcmt_leapfrog, libhmcmt), Cint,
               (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Float64, Int32, Float64, Float64, Float64,
                Ptr{Float64}, Ptr{Float64}, Ptr{ComplexF64}, Ref{Float64}, Ref{Float64}, Ref{Int32}),
               ctx.ptr, hmcParamCurrent.rhomodel, hmcParamCurrent.momentum, hmcprior.dt, Int32(intstep),
               hmcprior.regParam, log(hmcprior.sigBounds[1]), log(hmcprior.sigBounds[2]),
               m1, p1, pred, misfit, mnorm, nf)
    checkerr(ctx.ptr, rc)
    hmcprior.nfevals += nf[]
    invParam.strModel = copy(m1)
    mtMesh.sigma = invParam.activeCell * exp.(m1) + invParam.bgModel
    return m1, p1
end

"""
    proposeLeapfrogDevice!(ctx, d_m, d_p, dt, L, regParam, lnSigMin, lnSigMax; startGrad=0,
                           d_pred=C_NULL, d_misfit=C_NULL, d_mnorm=C_NULL) -> nfevals

hmcmt_leapfrog_device for callers that keep the chain state in GPU memory (e.g. AMDGPU.jl `ROCArray`s: pass
`pointer(a)`): d_m, d_p are updated in place, nothing crosses PCIe; complete with `hipWait(ctx)`.  `startGrad` as in
include/hmcmt.h (0 evaluate, 1 previous proposal accepted, 2 rejected).  Call `setPrior!` first.
"""
function proposeLeapfrogDevice!(ctx::HipContext, d_m::Ptr, d_p::Ptr, dt::Real, L::Integer, regParam::Real,
                                lnSigMin::Real, lnSigMax::Real; startGrad::Integer=0,
                                d_pred::Ptr=C_NULL, d_misfit::Ptr=C_NULL, d_mnorm::Ptr=C_NULL)
    nf = Ref{Int32}(0)
    rc = ccall((:hmcmt_leapfrog_device, libhmcmt), Cint,
               (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Float64, Int32, Float64, Float64, Float64, Int32,
                Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ref{Int32}),
               ctx.ptr, d_m, d_p, Float64(dt), Int32(L), Float64(regParam), Float64(lnSigMin), Float64(lnSigMax),
               Int32(startGrad), d_pred, d_misfit, d_mnorm, nf)
    checkerr(ctx.ptr, rc)
    return Int(nf[])
end

hipWait(ctx::HipContext) = checkerr(ctx.ptr, ccall((:hmcmt_wait, libhmcmt), Cint, (Ptr{Cvoid},), ctx.ptr))

"""
    commId() -> Vector{UInt8}(128);  SampleComm(device, nranks, rank, id);  allgatherSamples(comm, block) -> Matrix

The RCCL all-gather of the chains' sample blocks (hmcmt_comm_id / hmcmt_comm_create / hmcmt_allgather_samples): what
replaces `remotecall_fetch` of every worker's samples in parallelHMCSampler (parallelHMC.jl:23-45) when each worker
process owns a GPU.  The master obtains the id once and ships it to the workers (e.g. `remotecall_fetch(() -> commId(), 2)`
on rank 0, then as an argument of the workers' calls); every worker then holds every chain's block.
"""
commId() = (id = Vector{UInt8}(undef, 128);
            checkerr(C_NULL, ccall((:hmcmt_comm_id, libhmcmt), Cint, (Ptr{UInt8},), id)); id)

mutable struct SampleComm
    ptr::Ptr{Cvoid}
    nranks::Int
    function SampleComm(device::Integer, nranks::Integer, rank::Integer, id::Vector{UInt8})
        ref = Ref{Ptr{Cvoid}}(C_NULL)
        rc = ccall((:hmcmt_comm_create, libhmcmt), Cint, (Ref{Ptr{Cvoid}}, Int32, Int32, Int32, Ptr{UInt8}),
                   ref, Int32(device), Int32(nranks), Int32(rank), id)
        rc == 0 || error("libhmcmt_hip error $rc: " *
                         unsafe_string(ccall((:hmcmt_comm_last_error, libhmcmt), Cstring, (Ptr{Cvoid},), C_NULL)))
        c = new(ref[], Int(nranks))
        finalizer(x -> (x.ptr != C_NULL && ccall((:hmcmt_comm_destroy, libhmcmt), Cint, (Ptr{Cvoid},), x.ptr); x.ptr = C_NULL), c)
        return c
    end
end

function allgatherSamples(comm::SampleComm, block::Vector{Float64})
    recv = Matrix{Float64}(undef, length(block), comm.nranks)          # column r+1 = rank r's block
    rc = ccall((:hmcmt_allgather_samples, libhmcmt), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Int64, Int32),
               comm.ptr, block, recv, length(block), Int32(0))
    rc == 0 || error("libhmcmt_hip error $rc: " *
                     unsafe_string(ccall((:hmcmt_comm_last_error, libhmcmt), Cstring, (Ptr{Cvoid},), comm.ptr)))
    return recv
end

"""
    hipStats(ctx) -> HmcmtStats   (iteration counts, error estimate, fallback counter of the last call)
"""
function hipStats(ctx::HipContext)
    st = Ref{HmcmtStats}()
    rc = ccall((:hmcmt_get_stats, libhmcmt), Cint, (Ptr{Cvoid}, Ref{HmcmtStats}), ctx.ptr, st)
    checkerr(ctx.ptr, rc)
    return st[]
end

"""
    hipGuard(ctx) -> (checks, worst, last, trips)

The production guard of the iterative solves (`hmcmt_guard`, DESIGN 4.3): true residuals of both solves, formed every
`HMCMT_GUARD_EVERY`-th evaluation; `trips` counts the checks above `HMCMT_GUARD_LIMIT`.  A sampler loop reads it once per
output interval next to `hmcstats` and stops the chain on a trip.
"""
function hipGuard(ctx::HipContext)
    out = zeros(Float64, 4)
    rc = ccall((:hmcmt_guard, libhmcmt), Cint, (Ptr{Cvoid}, Ptr{Float64}), ctx.ptr, out)
    checkerr(ctx.ptr, rc)
    return (checks = Int(out[1]), worst = out[2], last = out[3], trips = Int(out[4]))
end

"""
    hipPersistInfo(ctx) -> NamedTuple

Shape and use of the one-launch-per-solve kernel (`hmcmt_persist_info`, include/hmcmt.h; DESIGN 5.0): `threads_half == 0` --
the mesh is outside its envelope; `usable_now == 0` -- this context is not alone on its share of the device (all solves then run
the launch-per-phase loop: same results, slower); `timeouts` / `placement_fallbacks` > 0 -- the device is shared with something
this library cannot see.
"""
function hipPersistInfo(ctx::HipContext)
    out = zeros(Int64, 14)
    rc = ccall((:hmcmt_persist_info, libhmcmt), Cint, (Ptr{Cvoid}, Ptr{Int64}, Int32), ctx.ptr, out, Int32(length(out)))
    checkerr(ctx.ptr, rc)
    return (threads_half = out[1], workgroups_per_system = out[2], slots_per_xcd = out[3], enabled = out[4], solves = out[5],
            placement_fallbacks = out[6], usable_now = out[7], slab_modes = out[8], column_parts = out[9], timeouts = out[10],
            cu_share_index = out[11], cu_share_count = out[12], strips = out[13], why_off = out[14])
end

"""
    hipPersistWidth(ctx) -> Int

Row width (padded nodes: 112 / 208 / 416) of the width-specialised persistent kernel the context launches (`hmcmt_persist_width`);
0: the generic kernel.
"""
function hipPersistWidth(ctx::HipContext)
    w = Ref{Int32}(0)
    rc = ccall((:hmcmt_persist_width, libhmcmt), Cint, (Ptr{Cvoid}, Ptr{Int32}), ctx.ptr, w)
    rc == 0 || error("hmcmt_persist_width failed")
    return Int(w[])
end

"""
    hipPersistOrder(ctx, kind = 0) -> (order, rebalanced)

The order (0-based system indices; position queue + queues * round) in which the persistent kernel's queues take the systems of a
forward (`kind = 0`) or adjoint (`1`) solve, and how often the context has re-balanced it from the iteration counts of the
previous solve (`hmcmt_persist_order`; meshes whose systems take turns on the chip).
"""
function hipPersistOrder(ctx::HipContext, kind::Integer = 0)
    dims = zeros(Int32, 7)
    ccall((:hmcmt_dims, libhmcmt), Cint, (Ptr{Cvoid}, Ptr{Int32}), ctx.ptr, dims) == 0 || error("hmcmt_dims failed")
    order = zeros(Int32, dims[3])          # (systems = 2 x frequencies)
    n = Ref{Int64}(0)
    rc = ccall((:hmcmt_persist_order, libhmcmt), Cint, (Ptr{Cvoid}, Int32, Ptr{Int32}, Ptr{Int64}), ctx.ptr, Int32(kind), order, n)
    rc == 0 || error("hmcmt_persist_order failed")
    return order, Int(n[])
end

"""
    hipPersistPack(cost, queues) -> (order, makespan)

The packing behind `hipPersistOrder` on the caller's costs (`hmcmt_persist_pack`; no device needed): systems onto `queues`
queues that take turns, longest first into the least loaded queue that has room; `order` holds 0-based system indices by
position queue + queues * round, `makespan` the largest queue sum.
"""
function hipPersistPack(cost::AbstractVector{<:Real}, queues::Integer)
    c = Vector{Float64}(cost)
    order = zeros(Int32, length(c))
    m = Ref{Float64}(0.0)
    rc = ccall((:hmcmt_persist_pack, libhmcmt), Cint, (Ptr{Float64}, Int32, Int32, Ptr{Int32}, Ptr{Float64}), c, Int32(length(c)), Int32(queues), order, m)
    rc == 0 || error("hmcmt_persist_pack failed")
    return order, m[]
end

"""
    hipPersistEnvelope(ny, nz; cus_per_xcd = 32, nsystems = 32) -> NamedTuple

Would a mesh of ny x nz cells (nz including the air layers) run the one-launch-per-solve kernel, and in which shape
(`hmcmt_persist_envelope`; pure arithmetic, no device needed)?  `column_parts == 0`: outside its envelope.
"""
function hipPersistEnvelope(ny::Integer, nz::Integer; cus_per_xcd::Integer = 32, nsystems::Integer = 32)
    out = zeros(Int64, 6)
    rc = ccall((:hmcmt_persist_envelope, libhmcmt), Cint, (Int64, Int64, Int32, Int64, Ptr{Int64}), ny, nz, Int32(cus_per_xcd), nsystems, out)
    rc == 0 || error("hmcmt_persist_envelope: invalid sizes")
    return (column_parts = out[1], threads_half = out[2], workgroups_per_system = out[3], slab_modes = out[4], lds_bytes = out[5], slots_per_xcd = out[6])
end

"""
    hipNextCuShare(index, count)

The calling task's NEXT `HipContext(...)` is confined to share `index` (0-based) of `count` = 1, 2 or 4 equal shares of the CUs of
every XCD (`hmcmt_next_cu_share`): what `parallelHMCSampler` calls before it builds the contexts of `count` chains that are to
run concurrently on one GPU (parallelHMC.jl:23-45 with more chains than devices).  Julia tasks may migrate between threads:
call it and the constructor without a yield in between (or pin the task).
"""
function hipNextCuShare(index::Integer, count::Integer)
    rc = ccall((:hmcmt_next_cu_share, libhmcmt), Cint, (Int32, Int32), Int32(index), Int32(count))
    rc == 0 || error("hmcmt_next_cu_share($index, $count): (index, count) with count 1, 2 or 4")
    return nothing
end

end # module
