/* hmcmt_debug.h -- instrumentation, introspection and test hooks of libhmcmt_hip.so (round 6: split off include/hmcmt.h, whose
 * entry points are the drop-in boundary of INTEGRATION.md section 1).  Nothing here is needed to run the hot path; the
 * measurement harness (bench.py), the profiling scripts and tests/ use it.  Same library, same calling conventions (return 0 or
 * a negative HMCMT_E* code).  Stability: fields of the out-arrays are only ever appended.
 * Replaces nothing in the reference: the reference's instrumentation is `@time` / `@elapsed` around proposeLeapfrog and the
 * chain (HMCSampler.jl:136, examples/dprism3d/runHMCscript.jl:26). */
#ifndef HMCMT_DEBUG_H
#define HMCMT_DEBUG_H

#include "hmcmt.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Kernel-time accounting with HIP events on the context's stream.
 * categories: 0 fdm-transform (MFMA), 1 tridiagonal, 2 stencil SpMV, 3 vector ops,
 *             4 assembly+boundary, 5 receivers+sources, 6 gradient accumulation */
#define HMCMT_NCAT 8
int hmcmt_profile(hmcmt_ctx* ctx, int32_t category_mask);     /* bit c enables category c; 0 = off; resets the counters */
int hmcmt_profile_every(hmcmt_ctx* ctx, int32_t n);           /* time only every n-th evaluation (event brackets cost ~20 % when always on) */
int hmcmt_profile_read(hmcmt_ctx* ctx, double* ms /*[HMCMT_NCAT]*/, int64_t* launches /*[HMCMT_NCAT]*/);
/* the per-launch bracket overhead (microseconds) hmcmt_profile calibrated -- a kernel that spins for a known time on the
 * device's wall clock, bracketed back to back -- and subtracts from every sampled launch */
int hmcmt_profile_overhead(const hmcmt_ctx* ctx, double* us);

/* What the sampled launches worked on (roofline numerators): out[0] = sum over the sampled iterations of the number of
 * systems still active (device counter, incremented by k_spmv_fused), out[1] = sum over the sampled solves of the systems
 * active at their start (the preconditioner is applied once before the first iteration), out[2] = evaluations sampled,
 * out[3] = solves sampled, out[4] = those of them that ran two smoothing sweeps per side (hmcmt_stats.smoother_sweeps).
 * out[5] = sum over the sampled solves of (iterations of the slowest system + 1): the serial length of the solves; out[6] = sampled
 * solves run by the persistent solve kernel (one launch per solve).  Reset by hmcmt_profile. */
int hmcmt_profile_counters(hmcmt_ctx* ctx, int64_t* out /*[7]*/);

/* sizes the roofline accounting needs: out = {NYP, NZP, S, ny, nz, zid, nblk} */
int hmcmt_dims(const hmcmt_ctx* ctx, int32_t* out);

/* Test hooks (exercise single kernels through the ABI).
 * hmcmt_debug_transform: C = A*V (which=0) or A*V' (which=1) with the context's FDM matrices (fp64 kernel);
 *   which=2/3: the mixed-precision kernels (bf16 operands, fp32 accumulation), same products;
 *   A, C: host complex[S*NZP*NYP] in the padded nodal layout.
 * hmcmt_debug_spmv: q = A_s p for all systems at the model of the last evaluation. */
int hmcmt_debug_transform(hmcmt_ctx* ctx, int32_t which, const double* A, double* C);
/* hmcmt_debug_flags (for the gradient pin, tests/test_gradient_pin.py): bit 0 -- the Dirichlet values of all four sides
 *   (mt2DTE.jl:100-134, mt2DTM.jl:100-134) are NOT recomputed from the model but stay those of the previous evaluation;
 *   bit 1 -- the boundary-derivative terms dBC^T w (compJacTMatVec.jl:237-242, :309-313, :316) are left out of the
 *   gradient.  A frozen-boundary finite difference of the misfit must then equal the gradient with bit 1 set.
 *   bit 2 (one-shot, not stored) -- the first system group of the NEXT persistent launch fails its placement check, as if its
 *   workgroups were not on one XCD (tests/test_gpu_persist.py: the other groups finish, the launch-per-phase loop takes the rest).
 *   bit 3 (one-shot) -- EVERY group of the next persistent launch fails it: the all-fallback regime of a device in another partition
 *   mode or a driver with another dispatch order (the whole evaluation then runs the launch-per-phase loop, one line on stderr).
 *   0 restores the product behaviour; stored results of earlier calls are dropped. */
int hmcmt_debug_flags(hmcmt_ctx* ctx, int32_t flags);
int hmcmt_debug_spmv(hmcmt_ctx* ctx, const double* p, double* q);
int hmcmt_debug_precond(hmcmt_ctx* ctx, const double* r, double* z);
int hmcmt_persist_envelope(int64_t ny, int64_t nz, int32_t cus_per_xcd, int64_t nsystems, int64_t* out6);   /* no device needed: would a mesh of ny x nz cells (nz incl. air
                                                             layers) run the one-launch-per-solve kernel on a device with cus_per_xcd CUs per XCD (MI355X: 32; a half / quarter CU
                                                             share: 16 / 8), and how: {column parts (0 = outside its envelope: the launch-per-phase loop), threads / 2, workgroups per
                                                             system, modes per slab, LDS bytes per workgroup, systems per XCD at a time} */
#define HMCMT_PERSIST_INFO_FIELDS 14
int hmcmt_persist_info(const hmcmt_ctx* ctx, int64_t* out, int32_t nout);  /* writes min(nout, HMCMT_PERSIST_INFO_FIELDS) values -- fields are only ever APPENDED, so a caller built against an
                                                                   older header passes its own count and gets the fields it knows --:
                                                                   {threads per strip (0: not applicable), workgroups per system, slots per XCD, enabled, solves, placement fallbacks,
                                                                   usable now (this context alone on its device in the process AND the process holds the device's advisory lock),
                                                                   modes per slab of its tridiagonal solves (32; 16 on tall meshes and with column parts),
                                                                   column parts per row block (1; 2 on meshes wider than one tile: the stress size),
                                                                   timed-out waits (each one: the evaluation redone with the launch-per-phase loop),
                                                                   CU share index, CU share count (hmcmt_next_cu_share),
                                                                   strips of tile rows per column (2: k_cocg_persist, 4: k_cocg_persist4; threads per workgroup = strips x threads per strip),
                                                                   why the kernel is off (0: it is not, or HMCMT_PERSIST=0; 1: a placement fallback, for good; 2: a timed-out wait, tried again later)} */
int hmcmt_persist_order(const hmcmt_ctx* ctx, int32_t kind, int32_t* order, int64_t* rebalanced);   /* meshes whose systems take turns on the chip (more systems than 8 x
                                                             slots per XCD: cfg5): the order in which the persistent kernel's queues take the systems of a solve of
                                                             `kind` (0 forward, 1 adjoint) -- order[nsystems], position queue + queues * round -> system --, balanced from
                                                             the previous solve's iteration counts (HMCMT_PERSIST_BALANCE=0: never; the same systems, the same results);
                                                             *rebalanced = tables taken so far.  Either pointer may be NULL */
int hmcmt_persist_pack(const double* cost, int32_t nsystems, int32_t queues, int32_t* order, double* makespan);   /* no device needed: the packing behind
                                                             hmcmt_persist_order on the caller's costs -- nsystems systems onto `queues` queues that take turns (position
                                                             queue + queues * round), longest first into the least loaded queue that has room; order[nsystems] = system at
                                                             each position, *makespan = the largest queue sum (order NULL: that of the index order) */
int hmcmt_persist_width(const hmcmt_ctx* ctx, int32_t* width);   /* the compile-time row width (padded nodes: 112 / 208 / 416) of the width-specialised persistent
                                                                     kernel this context launches; 0: the generic kernel (HMCMT_PERSIST_WIDTHK=0 forces it) */
int hmcmt_debug_hog(hmcmt_ctx* ctx, int32_t nblocks, int32_t ms);   /* test hook: nblocks workgroups that each hold a CU's LDS for ms milliseconds on a stream of their own (a foreign tenant on the device); returns once they are resident, without waiting for them to end */
int hmcmt_debug_persist_precond(hmcmt_ctx* ctx, int32_t sweeps, const double* r, double* z);   /* the persistent solve kernel's preconditioner (tests) */
int hmcmt_debug_fdm_fwd(hmcmt_ctx* ctx, const double* t, double* out);   /* [2][S*vstride] complex: fused kernel | separate kernels */
int hmcmt_debug_back_post(hmcmt_ctx* ctx, const double* y, const double* r, double* out, double* sums);   /* out: [2][S*vstride] complex (fused | separate), sums[6] */

#ifdef __cplusplus
}
#endif
#endif /* HMCMT_DEBUG_H */
