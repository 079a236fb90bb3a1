#-------------------------------------------------------------------------------
# HMCMTHip -- the reference-side binding of libhmcmt_hip.so (include/hmcmt.h).
#
# Drop this module next to the reference's HMCMT package.  It overrides exactly the hot path:
#   compDataGradient(mtMesh, mtData, invParam, hmcprior)   (HMCSampler/HMCSampler.jl:277-330)
#   the forward solve inside getHamiltonian                (HMCSampler/HMCSampler.jl:358-397)
# and leaves every other reference function (readstartupFile, runHMCSampler, proposeLeapfrog,
# parallelHMCSampler, writers) untouched.  Select it with `linearsolver: hip` in the startup file.
#
# NOTE: this file could not be executed in the build container (no julia binary).  It is a thin
# `ccall` layer: every call passes the reference's own arrays straight through -- Julia's
# Vector{Float64}/Vector{ComplexF64}/Vector{Int64} have exactly the memory layout the C ABI
# expects (interleaved re/im doubles, 1-based int64 indices).  The same ABI is exercised from
# Python/ctypes by tests/test_gpu_parity.py.
#-------------------------------------------------------------------------------
module HMCMTHip

using SparseArrays
using HMCMT.HMCFileIO, HMCMT.HMCStruct, HMCMT.HMCUtility

export HipContext, hipContext, compDataGradient, hipForward, destroy!

const libhmcmt = get(ENV, "HMCMT_HIP_LIB", joinpath(@__DIR__, "..", "hmcmt2d_amd", "libhmcmt_hip.so"))

struct HmcmtOptions
    precond::Int32
    maxit::Int32
    tol::Float64
    check_every::Int32
    verify::Int32
    warm_start::Int32
    fdm_precision::Int32
end

mutable struct HipContext
    ptr::Ptr{Cvoid}
    nAC::Int
    nData::Int
end

function checkerr(ctx::Ptr{Cvoid}, rc::Cint)
    rc == 0 && return
    msg = unsafe_string(ccall((:hmcmt_last_error, libhmcmt), Cstring, (Ptr{Cvoid},), ctx))
    error("libhmcmt_hip error $rc: $msg")        # same behaviour as checkMUMPSerror (MUMPSfuncs.jl:59-73)
end

"""
    hipContext(mtMesh, mtData, invParam; device=0)

Builds the GPU context once per run (replaces the operator set-up the reference redoes on every
call: setupTensorMesh2D!, getBoundaryIndex, preSetRxFieldSens).
"""
function hipContext(mtMesh::TensorMesh2D, mtData::MTData, invParam::InvDataModel; device::Integer=0)
    occursin("Impedance", mtData.dataType) || error("only DataType Impedance is supported")
    ny, nz = mtMesh.gridSize
    compMode = Int64[occursin("XY", c) ? 1 : (occursin("YX", c) ? 2 : error("unsupported component $c"))
                     for c in mtData.dataComp]
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
    ctx = HipContext(ctxref[], length(activeIdx), length(obs))
    finalizer(destroy!, ctx)
    return ctx
end

function destroy!(ctx::HipContext)
    if ctx.ptr != C_NULL
        ccall((:hmcmt_destroy, libhmcmt), Cint, (Ptr{Cvoid},), ctx.ptr)
        ctx.ptr = C_NULL
    end
end

const contexts = IdDict{Any,HipContext}()
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
    return pred, misfit[], grad
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
    return pred, misfit[]
end

end # module
