/*
 * hmcmt_mumps.h -- the reference's EXISTING native boundary: the eight Fortran-convention symbols its MUMPS
 * wrapper binds with ccall (MUMPS/src/MUMPSfuncs.jl), exported by libhmcmt_hip.so so that the UNMODIFIED
 * reference package can load this library in place of its `MUMPS/lib/MUMPS` binary (MUMPS/src/MUMPS.jl:14)
 * and run with `linearsolver: mumps`.
 *
 * This is the compatibility path, not the fast path: one blocking call per factor / solve /
 * destroy (128 per gradient at 16 frequencies), no batching across frequencies.  The fast path is hmcmt.h.
 *
 * Behind the symbols sits an iterative solver on the GPU, not a factorisation: `factor` uploads the matrix
 * (CSC of a symmetric matrix = its CSR) and its inverse diagonal; `solve` runs Jacobi-preconditioned
 * conjugate-orthogonal CG (= CG for the real SPD case) in fp64 per right-hand side, with iterative refinement
 * on the true residual, until ||b - A x|| <= 1e-14 ||b|| or stagnation.  Only symmetric matrices are accepted
 * (sym = 1 or 2, both triangles stored, as the reference passes them: mt2DTE.jl:51-52, MUMPS/test/testDivGrad.jl);
 * sym = 0 reports stat = -1.
 *
 * Conventions (exactly the reference's ccall signatures): every argument by pointer, Int64 integers, 1-based
 * rowval / colptr (Julia SparseMatrixCSC), complex = interleaved (re, im) doubles, arrays caller-owned and
 * copied during `factor` (never retained), x caller-allocated (column-major n x nrhs), handle = opaque Int64.
 * stat[0] < 0 after factor is an error (MUMPSfuncs.jl:59-73): -10 zero on the diagonal ("numerically singular"),
 * -13 allocation failure, -1 anything else (no HIP device, unsupported sym).  A solve that fails to reach 1e-10
 * prints a warning to stderr and returns -10; the reference ignores the solve return value.
 */
#ifndef HMCMT_MUMPS_H
#define HMCMT_MUMPS_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* MUMPSfuncs.jl:32-35 (factorMUMPS, ComplexF64) / :49-52 (Float64) */
int64_t factor_mumps_cmplx_(const int64_t* n, const int64_t* sym, const int64_t* ooc, const double* nzval /* [2*nnz] */,
                            const int64_t* rowval, const int64_t* colptr, int64_t* stat);
int64_t factor_mumps_(const int64_t* n, const int64_t* sym, const int64_t* ooc, const double* nzval, const int64_t* rowval,
                      const int64_t* colptr, int64_t* stat);
/* MUMPSfuncs.jl:128-130 (applyMUMPS!, dense complex rhs) / :105-107 (real) */
int64_t solve_mumps_cmplx_(const int64_t* handle, const int64_t* nrhs, const double* rhs, double* x, const int64_t* transpose);
int64_t solve_mumps_(const int64_t* handle, const int64_t* nrhs, const double* rhs, double* x, const int64_t* transpose);
/* MUMPSfuncs.jl:139-143 (sparse complex rhs, CSC) / :115-118 (real) */
void solve_mumps_cmplx_sparse_rhs_(const int64_t* handle, const int64_t* nzrhs, const int64_t* nrhs, const double* nzval,
                                   const int64_t* rowval, const int64_t* colptr, double* x, const int64_t* transpose);
void solve_mumps_sparse_rhs_(const int64_t* handle, const int64_t* nzrhs, const int64_t* nrhs, const double* nzval,
                             const int64_t* rowval, const int64_t* colptr, double* x, const int64_t* transpose);
/* MUMPSfuncs.jl:170-171 / :155-156 */
int64_t destroy_mumps_cmplx_(const int64_t* handle);
int64_t destroy_mumps_(const int64_t* handle);

/* Not part of the reference's interface: statistics of the last solve on a handle (tests):
 * out[0] = iterations of the last right-hand side, out[1] = refinement passes, out[2] = max over rhs of
 * ||b - A x|| / ||b||. */
int64_t hmcmt_mumps_last_solve(const int64_t* handle, double* out /* [3] */);

#ifdef __cplusplus
}
#endif
#endif
