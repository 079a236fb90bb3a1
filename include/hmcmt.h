/* hmcmt.h -- C ABI of libhmcmt_hip.so: the MI355X-native hot path of CUG-EMI/HMCMT2D.
 *
 * One context = one GPU = one HMC chain's worth of state.  A context is NOT thread-safe;
 * distinct contexts are independent.  All pointers are caller-owned; the library copies what it
 * needs during hmcmt_create and never retains host pointers.  Complex arrays are interleaved
 * (re, im) doubles: the memory layout of Julia's ComplexF64, numpy complex128 and C's
 * `double _Complex`.  Index arrays are 1-based int64, exactly as the reference stores them.
 *
 * Every function returns 0 on success or a negative HMCMT_E* code; it never throws or aborts.
 * hmcmt_last_error() gives the message of the last failure.  There is NO host compute path:
 * without a HIP device hmcmt_create fails with HMCMT_ENODEV.
 *
 * What each entry point replaces in the reference (/root/reference, all under HMCMT/src/):
 *
 *   hmcmt_create / hmcmt_destroy
 *       the per-run set-up the reference redoes inside every call: setupTensorMesh2D!
 *       (MTFwdSolver/MT2DOperators.jl:16-27), getBoundaryIndex (MT2DFwdSolver.jl:227-248),
 *       preSetRxFieldSens (MTSensitivity/sensUtils.jl:17-52), compDataWeightMat / activeCell
 *       plumbing (HMCStruct/HMCStruct.jl:99-125).
 *   hmcmt_grad   == compDataGradient(mtMesh, mtData, invParam, hmcprior)
 *       (HMCSampler/HMCSampler.jl:277-330): m = ln(sigma) on active cells ->
 *       (predData, dataMisfit, dataGrad); internally MT2DFwdSolver (MT2DFwdSolver.jl:74-216),
 *       compMT2DTE/TM (mt2DTE.jl:19-83, mt2DTM.jl:18-83), compJacTMatVec
 *       (MTSensitivity/compJacTMatVec.jl:8-327) and the MUMPS/UMFPACK factor+solve they call
 *       (MUMPS/src/MUMPSfuncs.jl:32,128; mt2DTE.jl:47-55).
 *   hmcmt_forward == MT2DFwdSolver + compDataMisfit as used by getHamiltonian
 *       (HMCSampler.jl:358-397, :498-507).
 *   hmcmt_leapfrog == proposeLeapfrog (HMCSampler.jl:206-269) with the trajectory kept on the
 *       device, plus the Hamiltonian terms getHamiltonian needs at the proposal.
 *   hmcmt_get_fields: exTE / hxTM of MT2DFwdData (MT2DFwdSolver.jl:44-53), reference node order.
 */
#ifndef HMCMT_H
#define HMCMT_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HMCMT_OK          0
#define HMCMT_EINVAL     -1   /* bad argument (message says which) */
#define HMCMT_ENODEV     -2   /* no usable HIP device */
#define HMCMT_EHIP       -3   /* HIP runtime error */
#define HMCMT_ENOCONV   -10   /* an iterative solve hit maxit (cf. MUMPS -10 "singular", MUMPSfuncs.jl:59-73) */
#define HMCMT_EBREAKDOWN -11  /* Krylov breakdown / non-finite values (a NaN model would hang the reference, HMCSampler.jl:546-548) */
#define HMCMT_ENOMEM    -13   /* device allocation failed (cf. MUMPS -13) */

#define HMCMT_PRECOND_JACOBI 0
#define HMCMT_PRECOND_FDM    1   /* fast diagonalisation with a laterally averaged background */
#define HMCMT_PRECOND_FDM_JACOBI 2 /* damped point-Jacobi / FDM / point-Jacobi, symmetric product form (default) */

typedef struct hmcmt_ctx hmcmt_ctx;

typedef struct hmcmt_options {
    int32_t precond;      /* HMCMT_PRECOND_* ; default FDM_JACOBI */
    int32_t maxit;        /* iteration cap per solve; default 2000 (FDM) */
    double  tol;          /* stop when ||P^-1 r|| <= tol*||x|| (error estimate); default 1e-11 */
    int32_t check_every;  /* host convergence poll interval in iterations of the classic (non-default) solver loops; default 2.
                             The default path looks at a mapped counter once per iteration without waiting */
    int32_t verify;       /* 1: also compute true relative residuals ||b-Ax||/||b|| after each solve */
    int32_t warm_start;   /* initial guess of both solves: 0 zero, 1 the previous evaluation's fields, 2 (default) those
                             fields extrapolated along the model path from up to the last six evaluations (Lagrange
                             extrapolation over the nearly collinear part of the history; leapfrog trajectories are nearly
                             straight lines at nearly constant speed).  The adjoint solve starts from zero unless at least
                             three collinear points exist */
    int32_t fdm_precision;/* 0 (default): bf16 transforms (fp32 accumulate) + complex64 tridiagonal inside the
                             preconditioner; 1: fp64 throughout.  x, r, p and all inner products are fp64 either way */
} hmcmt_options;

typedef struct hmcmt_stats {
    int32_t iters_fwd_max, iters_adj_max;   /* max over systems of the last call */
    int32_t iters_fwd_sum, iters_adj_sum;   /* sum over systems */
    double  err_est_max;                    /* max over systems of ||P^-1 r||/||x|| at exit */
    double  true_res_max;                   /* max ||b-Ax||/||b|| (with options.verify, and in the guarded evaluations: hmcmt_guard) */
    int32_t status;                         /* 0 or HMCMT_ENOCONV / HMCMT_EBREAKDOWN */
    int32_t nsystems;                       /* 2*nFreq */
    int32_t fallback_solves;                /* solves of the last call (0..2) whose stragglers were restarted with the fp64
                                               preconditioner because the device's stagnation watch fired (a system that did
                                               not improve its error estimate 10-fold within 30 iterations) */
    int32_t smoother_sweeps;                /* damped Jacobi sweeps on each side of the FDM stage in the last call: 10 * (forward
                                               solve) + (adjoint solve), e.g. 11 or 22; chosen per solve unless HMCMT_SWEEPS is
                                               set (was `reserved_`: same offset, same size) */
} hmcmt_stats;

void hmcmt_default_options(hmcmt_options* opts);

/* Builds a context on HIP device `device_id`.
 *   ny, nz            cells in y / z (nz INCLUDES the air layers, TensorMesh2D.gridSize)
 *   yLen[ny], zLen[nz], origin[2]      TensorMesh2D fields (HMCFileIO.jl:45-60)
 *   freqs[nFreq]; rxY[nRx], rxZ[nRx]   MTData.freqs, columns of MTData.rxLoc
 *   compMode[nComp]   component code per entry of MTData.dataComp: 1 ZXY, 2 ZYX (DataType Impedance, complex data);
 *                     3 RhoXY, 4 PhsXY, 5 RhoYX, 6 PhsYX (DataType Rho_Pha: apparent resistivity |Z|^2/(w mu0) in Ohm-m and
 *                     phase in degrees, mt2DTE.jl:253-255 -- real data: obs / pred keep the complex layout with zero
 *                     imaginary parts).  The two families cannot be mixed
 *   freqID/rxID/dtID[nData]  1-based, MTData fields; dataID[nComp*nRx*nFreq] mask, dt fastest
 *   obs[nData] complex, dataW[nData] = diag of InvDataModel.dataW
 *   activeIdx[nAC]    1-based cell id of each active cell (= activeCell.rowval), bgModel[ny*nz]
 *   opts              NULL for defaults
 * Limits: the one-launch-per-solve kernel runs meshes up to 415 cells wide (two column parts per row block beyond 207; nz up to
 * 225 rows at that width, more on narrower meshes: DESIGN 5.0, "Envelope"), all others four to six launches per iteration
 * (DESIGN 5.1).  options.fdm_precision = 1 needs ny + 1 <= 448 nodes (HMCMT_EINVAL otherwise: the fp64 eigen-transform of the
 * preconditioner holds 28 column tiles); wider meshes run the default mixed-precision path WITHOUT its fp64 safety net (a
 * stagnating solve then fails its evaluation with HMCMT_ENOCONV instead of being restarted);
 * hmcmt_persist_info tells which. */
int hmcmt_create(hmcmt_ctx** ctx, int32_t device_id,
                 int64_t ny, int64_t nz, const double* yLen, const double* zLen, const double* origin,
                 int64_t nFreq, const double* freqs,
                 int64_t nRx, const double* rxY, const double* rxZ,
                 int64_t nComp, const int64_t* compMode,
                 int64_t nData, const int64_t* freqID, const int64_t* rxID, const int64_t* dtID,
                 const uint8_t* dataID, const double* obs, const double* dataW,
                 int64_t nAC, const int64_t* activeIdx, const double* bgModel,
                 const hmcmt_options* opts);
int hmcmt_destroy(hmcmt_ctx* ctx);
const char* hmcmt_last_error(const hmcmt_ctx* ctx);   /* ctx may be NULL: create-time error */

int hmcmt_set_options(hmcmt_ctx* ctx, const hmcmt_options* opts);
int hmcmt_get_stats(const hmcmt_ctx* ctx, hmcmt_stats* out);
/* per-system iteration counts of the last call: iters[2*S] (forward, then adjoint) */
int hmcmt_get_iters(const hmcmt_ctx* ctx, int32_t* iters);

/* Host-buffer entry points (synchronous). pred: complex[nData]; grad: [nAC].
 * A model identical (bit for bit) to one of the last two evaluated through these entry points is answered from their
 * stored results without touching the GPU (a sampler re-evaluates the proposal's model and, after a rejection, the
 * previous start model); hmcmt_get_stats then reports zero iterations.  Dropped by hmcmt_set_options and when
 * options.verify is set. */
int hmcmt_grad(hmcmt_ctx* ctx, const double* m, double* pred, double* misfit, double* grad);
int hmcmt_forward(hmcmt_ctx* ctx, const double* m, double* pred, double* misfit);

/* Device-buffer entry points: all pointers are DEVICE pointers on the context's GPU; work is
 * enqueued on the context's stream and complete when the call returns. */
int hmcmt_grad_device(hmcmt_ctx* ctx, const double* d_m, double* d_pred, double* d_misfit, double* d_grad);
int hmcmt_forward_device(hmcmt_ctx* ctx, const double* d_m, double* d_pred, double* d_misfit);

/* Asynchronous variant for a device-resident caller (a leapfrog loop whose next model is computed on the device from
 * this gradient, bench.py): returns when the evaluation is ENQUEUED; the outputs are ordered on the context's stream.
 * The call still blocks at its two convergence polls, but not on the gradient assembly behind the adjoint solve, so
 * the host issues the next evaluation's boundary-value stage while the device finishes this one.  A solve that gives
 * up (iteration cap, breakdown) is seen at its poll and returned by the call itself -- nothing is built on it: no
 * adjoint solve on a failed forward solve, no gradient from a failed adjoint solve; the statistics of a successful
 * asynchronous evaluation are collected by the next evaluation on the context or by hmcmt_wait; between hmcmt_grad_device_async and hmcmt_wait only further
 * hmcmt_grad_device_async calls are allowed on the context. */
int hmcmt_grad_device_async(hmcmt_ctx* ctx, const double* d_m, double* d_pred, double* d_misfit, double* d_grad);
/* waits for everything enqueued on the context; returns the status of the last asynchronous evaluation */
int hmcmt_wait(hmcmt_ctx* ctx);

/* One leapfrog trajectory on the device (proposeLeapfrog, HMCSampler.jl:206-269; diagonal mass).
 *   m0, p0 [nAC]      current model / momentum (host)
 *   invM [nAC]        diagonal of M^-1
 *   mref [nAC]        prior reference model; Wm in CSR (rowptr[nAC+1], colind, val; 0-based int64)
 *   dt, L, regParam, lnSigMin, lnSigMax     as in HMCPrior
 * Outputs (host): m1, p1 [nAC]; pred complex[nData] and misfit at the proposal (what the next
 * getHamiltonian call would recompute, HMCSampler.jl:364); mnorm = 0.5*lambda*(m-mref)'Wm(m-mref);
 * nfevals = number of gradient evaluations performed (L+1). */
int hmcmt_set_prior(hmcmt_ctx* ctx, const double* mref, const int64_t* wm_rowptr,
                    const int64_t* wm_colind, const double* wm_val, const double* invM);
int hmcmt_leapfrog(hmcmt_ctx* ctx, const double* m0, const double* p0, double dt, int32_t L,
                   double regParam, double lnSigMin, double lnSigMax,
                   double* m1, double* p1, double* pred, double* misfit, double* mnorm,
                   int32_t* nfevals);

/* The same trajectory on DEVICE vectors: d_m, d_p [nAC] are updated in place (start model / momentum -> proposal), nothing
 * crosses PCIe.  start_grad says where the data gradient at the start model comes from:
 *   0  evaluate it (first trajectory of a chain, or a start model the context has not seen);
 *   1  the start model is the END model of the previous trajectory on this context (the proposal was accepted,
 *      HMCSampler.jl:155-163): its gradient is still on the device -- L new evaluations instead of L + 1;
 *   2  the start model is the START model of the previous trajectory (the proposal was rejected, :164-168).
 * d_pred complex[nData], d_misfit, d_mnorm (device, each may be NULL) receive the proposal's predicted data, data
 * misfit and 0.5*lambda*(m-mref)'Wm(m-mref).  Returns when the whole trajectory is enqueued and the solver status of
 * every evaluation -- the last one included: the call waits for its two solves' records, not for the gradient assembly,
 * the final momentum update and the Hamiltonian terms queued behind them -- has been checked; hmcmt_wait completes it
 * (and reports a non-finite model met on the way).  nfevals counts as the reference does (L + 1). */
int hmcmt_leapfrog_device(hmcmt_ctx* ctx, double* d_m, double* d_p, double dt, int32_t L, double regParam,
                          double lnSigMin, double lnSigMax, int32_t start_grad, double* d_pred, double* d_misfit,
                          double* d_mnorm, int32_t* nfevals);

/* Solution fields of the last evaluation THAT RAN (a call answered from the stored results runs nothing) in the
 * reference's layout: complex[(ny+1)*(nz+1)*nFreq],
 * node index (iz*(ny+1)+iy) fastest, then frequency (MT2DFwdSolver.jl:111-112).  adjoint=1 returns
 * the adjoint fields instead (interior = eVal of compJacTMatVec.jl:221, boundary 0). */
int hmcmt_get_fields(hmcmt_ctx* ctx, int32_t adjoint, double* exTE, double* hxTM);

/* All-gather of the chains' sample blocks over RCCL (xGMI inside a node): one process per GPU, one communicator per
 * process.  Replaces parallelHMCSampler's collection of the workers' results (HMCSampler/parallelHMC.jl:23-45:
 * remotecall_fetch of hmcmodel / hmcstats / hmcdata per worker) for hosts that hold their chains in this library:
 *   hmcmt_comm_id        rank 0 obtains the 128-byte RCCL id and hands it to the other ranks by its own means (the Julia
 *                        host: a remotecall; Python: torch.distributed's store; a file) -- ncclGetUniqueId
 *   hmcmt_comm_create    every rank, with the same id: ncclCommInitRank on `device_id` (collective)
 *   hmcmt_allgather_samples   `count` doubles per rank: recv[r*count .. (r+1)*count) = rank r's send -- ncclAllGather on
 *                        the communicator's stream, complete on return.  on_device = 1: send / recv are device pointers
 *                        on the communicator's GPU; 0: host buffers, staged through device memory
 * librccl.so is loaded on first use; without it these calls fail with HMCMT_ENODEV and nothing else is affected. */
#define HMCMT_COMM_ID_BYTES 128
typedef struct hmcmt_comm hmcmt_comm;
int hmcmt_comm_id(void* id /*[HMCMT_COMM_ID_BYTES]*/);
int hmcmt_comm_create(hmcmt_comm** comm, int32_t device_id, int32_t nranks, int32_t rank, const void* id);
int hmcmt_allgather_samples(hmcmt_comm* comm, const double* send, double* recv, int64_t count, int32_t on_device);
int hmcmt_comm_destroy(hmcmt_comm* comm);
const char* hmcmt_comm_last_error(const hmcmt_comm* comm);   /* comm may be NULL: hmcmt_comm_id / hmcmt_comm_create errors */

/* Production guard of the stopping rule, and chains that share a device. */
int hmcmt_guard(const hmcmt_ctx* ctx, double* out4);   /* {checks, worst true residual seen, last, trips (checks above HMCMT_GUARD_LIMIT, default 1e-6)}: the production guard of the stopping rule (every HMCMT_GUARD_EVERY-th evaluation, default 100) */
int hmcmt_next_cu_share(int32_t index, int32_t count);   /* the calling thread's NEXT hmcmt_create builds a context confined to share `index` of `count`
                                                             (1, 2, 4) equal shares of the CUs of every XCD (CU-masked streams): the persistent solve kernels of
                                                             `count` such contexts -- independent chains on one device, parallelHMC.jl:23-45 -- run side by side,
                                                             each with its share of the system slots (cfg3: two chains 1.15x one chain's steps/s).  Such a context's
                                                             streams are blocking HIP streams: they synchronise with the legacy default stream.
                                                             Consumed by that create; default: the whole device */

/* Instrumentation, introspection of the persistent solve kernel and the test hooks are declared in hmcmt_debug.h (same library):
 * hmcmt_profile*, hmcmt_dims, hmcmt_persist_*, hmcmt_debug_*.  Nothing in INTEGRATION.md section 1 needs them. */

#ifdef __cplusplus
}
#endif
#endif /* HMCMT_H */
