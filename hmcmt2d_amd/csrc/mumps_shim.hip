// The reference's existing native boundary (include/hmcmt_mumps.h): the eight MUMPS-wrapper symbols of
// MUMPS/src/MUMPSfuncs.jl, implemented as a GPU iterative solver for symmetric matrices.
//
//   factor : validate + upload the matrix once (CSC of a symmetric matrix is its CSR), inverse diagonal
//   solve  : per right-hand side, Jacobi-preconditioned conjugate-orthogonal CG in fp64 (unconjugated inner
//            products: complex symmetric matrices, MT2DFwdSolver.jl:254-255; plain CG for real SPD ones) with
//            iterative refinement on the true residual b - A x
//
// Two launches per iteration: k_sp_dir (p = z + beta p recomputed at every gathered column, q = A p, p'q) and
// k_sp_upd (x += alpha p, r -= alpha q, z = D^-1 r, r'z, |r|^2).  Every block reduces the previous kernel's
// per-block partial sums itself, in the same order, so all blocks agree on alpha / beta / convergence without a
// scalar kernel or a grid barrier; the host looks at one flag every 32 iterations.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../include/hmcmt_mumps.h"
#include "hmcmt_math.h"

namespace {
using hmcmt::cplx;

constexpr int SB = 256;      // threads per block
constexpr int MAXB = 512;    // blocks per launch = partial sums per reduction

__host__ __device__ inline double abs2v(double a) { return a * a; }
__host__ __device__ inline double abs2v(cplx a) { return a.re * a.re + a.im * a.im; }
__host__ __device__ inline void zerov(double& a) { a = 0.0; }
__host__ __device__ inline void zerov(cplx& a) { a = cplx{0.0, 0.0}; }
__device__ inline double shfl_down_v(double a, int o) { return __shfl_down(a, o); }
__device__ inline cplx shfl_down_v(cplx a, int o) { return cplx{__shfl_down(a.re, o), __shfl_down(a.im, o)}; }

template <class T>
struct Sys {
    int n, nblk;
    const int* rp;      // [n+1] 0-based row pointers (= the caller's colptr - 1)
    const int* ci;      // [nnz] 0-based column indices (= rowval - 1)
    const T* va;        // [nnz]
    const T* dinv;      // [n]
    const T* b;
    T *x, *r, *z, *q;
    T *partPQ, *partRZ; // [MAXB]
    double* partRR;     // [MAXB]
    T* rho;             // [2] r'z by iteration parity
    int* flag;          // 0 running, 1 converged, 2 breakdown
    double bb, tol2;
};

// sum of the nblk per-block partials, identical in every block (same order of operations)
template <class T>
__device__ T block_total(const T* part, int nblk, T* sh) {
    T v;
    zerov(v);
    for (int i = threadIdx.x; i < nblk; i += SB) v = v + part[i];
    for (int o = 32; o > 0; o >>= 1) v = v + shfl_down_v(v, o);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    T t = sh[0];
    for (int w = 1; w < SB / 64; ++w) t = t + sh[w];
    return t;
}
template <class T>
__device__ void block_store(T v, T* dst, T* sh) {
    for (int o = 32; o > 0; o >>= 1) v = v + shfl_down_v(v, o);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        T t = sh[0];
        for (int w = 1; w < SB / 64; ++w) t = t + sh[w];
        *dst = t;
    }
}

// r = b - A x, z = D^-1 r, partial r'z and |r|^2 (start of a pass of the refinement loop); LPR lanes per row
template <class T, int LPR>
__global__ __launch_bounds__(SB) void k_sp_resid(Sys<T> s) {
    __shared__ T sh[SB / 64];
    __shared__ double shd[SB / 64];
    const int lr = threadIdx.x % LPR;
    T rz;
    zerov(rz);
    double rr = 0.0;
    for (long row0 = (long)blockIdx.x * (SB / LPR); row0 < s.n; row0 += (long)gridDim.x * (SB / LPR)) {
        const long row = row0 + threadIdx.x / LPR;
        T acc;
        zerov(acc);
        if (row < s.n)
            for (int k = s.rp[row] + lr; k < s.rp[row + 1]; k += LPR) acc = acc + s.va[k] * s.x[s.ci[k]];
        for (int o = LPR / 2; o > 0; o >>= 1) acc = acc + shfl_down_v(acc, o);
        if (row < s.n && lr == 0) {
            const T rv = s.b[row] - acc, zv = s.dinv[row] * rv;
            s.r[row] = rv; s.z[row] = zv;
            rz = rz + rv * zv; rr += abs2v(rv);
        }
    }
    block_store(rz, s.partRZ + blockIdx.x, sh);
    block_store(rr, s.partRR + blockIdx.x, shd);
}

// convergence test, beta, p = z + beta p_old (recomputed at every gathered column), q = A p, partial p'q
template <class T, int LPR>
__global__ __launch_bounds__(SB) void k_sp_dir(Sys<T> s, int it, const T* __restrict__ pold, T* __restrict__ pnew) {
    __shared__ T sh[SB / 64];
    __shared__ double shd[SB / 64];
    if (*s.flag) return;
    const double rr = block_total(s.partRR, s.nblk, shd);
    if (rr <= s.tol2 * s.bb) { if (blockIdx.x == 0 && threadIdx.x == 0) *s.flag = 1; return; }
    const T rz = block_total(s.partRZ, s.nblk, sh);
    T beta;
    zerov(beta);
    if (it > 0) beta = rz / s.rho[(it - 1) & 1];
    if (blockIdx.x == 0 && threadIdx.x == 0) s.rho[it & 1] = rz;
    const int lr = threadIdx.x % LPR;
    T pq;
    zerov(pq);
    for (long row0 = (long)blockIdx.x * (SB / LPR); row0 < s.n; row0 += (long)gridDim.x * (SB / LPR)) {
        const long row = row0 + threadIdx.x / LPR;
        T acc;
        zerov(acc);
        if (row < s.n)
            for (int k = s.rp[row] + lr; k < s.rp[row + 1]; k += LPR) {
                const int c = s.ci[k];
                acc = acc + s.va[k] * (it > 0 ? s.z[c] + beta * pold[c] : s.z[c]);
            }
        for (int o = LPR / 2; o > 0; o >>= 1) acc = acc + shfl_down_v(acc, o);
        if (row < s.n && lr == 0) {
            const T pn = it > 0 ? s.z[row] + beta * pold[row] : s.z[row];
            pnew[row] = pn; s.q[row] = acc;
            pq = pq + pn * acc;
        }
    }
    block_store(pq, s.partPQ + blockIdx.x, sh);
}

// alpha, x += alpha p, r -= alpha q, z = D^-1 r, partial r'z and |r|^2
template <class T>
__global__ __launch_bounds__(SB) void k_sp_upd(Sys<T> s, int it, const T* __restrict__ p) {
    __shared__ T sh[SB / 64];
    __shared__ double shd[SB / 64];
    if (*s.flag) return;
    const T pq = block_total(s.partPQ, s.nblk, sh);
    if (!(abs2v(pq) > 0.0)) { if (blockIdx.x == 0 && threadIdx.x == 0) *s.flag = 2; return; }
    const T alpha = s.rho[it & 1] / pq;
    T rz;
    zerov(rz);
    double rr = 0.0;
    for (long i = (long)blockIdx.x * SB + threadIdx.x; i < s.n; i += (long)gridDim.x * SB) {
        s.x[i] = s.x[i] + alpha * p[i];
        const T rv = s.r[i] - alpha * s.q[i], zv = s.dinv[i] * rv;
        s.r[i] = rv; s.z[i] = zv;
        rz = rz + rv * zv; rr += abs2v(rv);
    }
    block_store(rz, s.partRZ + blockIdx.x, sh);
    block_store(rr, s.partRR + blockIdx.x, shd);
}

struct HandleBase {
    bool cmplx = false;
    virtual ~HandleBase() {}
};

template <class T>
struct Handle : HandleBase {
    int n = 0, lpr = 4;
    long nnz = 0;
    int device = 0;
    hipStream_t stream = nullptr;
    std::vector<void*> allocs;
    Sys<T> s{};
    T *p0 = nullptr, *p1 = nullptr, *d_b = nullptr;
    int* h_flag = nullptr;       // pinned
    double* h_part = nullptr;    // pinned [MAXB]
    double lastIters = 0, lastPasses = 0, lastRes = 0;

    ~Handle() override {
        for (void* p : allocs) hipFree(p);
        if (h_flag) hipHostFree(h_flag);
        if (h_part) hipHostFree(h_part);
        if (stream) hipStreamDestroy(stream);
    }
    template <class U>
    bool alloc(U** p, size_t count) {
        void* q = nullptr;
        if (hipMalloc(&q, std::max<size_t>(count, 1) * sizeof(U)) != hipSuccess) return false;
        allocs.push_back(q);
        *p = (U*)q;
        return true;
    }
};

template <class T> T make_val(const double* nz, long k);
template <> double make_val<double>(const double* nz, long k) { return nz[k]; }
template <> cplx make_val<cplx>(const double* nz, long k) { return cplx{nz[2 * k], nz[2 * k + 1]}; }

template <class T>
int64_t factor_impl(const int64_t* np, const int64_t* symp, const double* nzval, const int64_t* rowval,
                    const int64_t* colptr, int64_t* stat) {
    auto fail = [&](int64_t code, const char* why) -> int64_t {
        if (stat) stat[0] = code;
        fprintf(stderr, "libhmcmt_hip (MUMPS interface): factor failed: %s\n", why);
        return 0;
    };
    if (!np || !symp || !nzval || !rowval || !colptr || !stat) return fail(-1, "null argument");
    const int64_t n = *np;
    if (n < 1 || n > 0x7fffffff) return fail(-1, "bad dimension");
    if (*symp != 1 && *symp != 2) return fail(-1, "only symmetric matrices (sym = 1 or 2) are supported");
    if (colptr[0] != 1) return fail(-1, "colptr must be 1-based");
    const int64_t nnz = colptr[n] - 1;
    if (nnz < n || nnz > 0x7fffffff) return fail(-1, "bad number of non-zeros");
    std::vector<int> rp(n + 1), ci(nnz);
    std::vector<T> va(nnz), dinv(n);
    for (int64_t j = 0; j <= n; ++j) {
        if (j && colptr[j] < colptr[j - 1]) return fail(-1, "colptr not monotone");
        rp[j] = (int)(colptr[j] - 1);
    }
    for (int64_t j = 0; j < n; ++j) {
        bool haveDiag = false;
        for (int64_t k = rp[j]; k < rp[j + 1]; ++k) {
            const int64_t i = rowval[k] - 1;
            if (i < 0 || i >= n) return fail(-1, "row index out of range");
            ci[k] = (int)i;
            va[k] = make_val<T>(nzval, k);
            if (i == j) {
                if (!(abs2v(va[k]) > 0.0)) return fail(-10, "zero on the diagonal");
                dinv[j] = T{1.0} / va[k];
                haveDiag = true;
            }
        }
        if (!haveDiag) return fail(-10, "structurally zero diagonal");
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return fail(-1, "no HIP device (there is no CPU path)");
    auto* h = new Handle<T>();
    h->cmplx = sizeof(T) == sizeof(cplx);
    h->n = (int)n; h->nnz = nnz;
    h->lpr = (double)nnz / (double)n <= 12.0 ? 4 : 16;
    const char* dev = getenv("HMCMT_DEVICE");
    h->device = dev ? atoi(dev) : 0;
    bool ok = hipSetDevice(h->device) == hipSuccess && hipStreamCreate(&h->stream) == hipSuccess;
    int *d_rp = nullptr, *d_ci = nullptr;
    T *d_va = nullptr, *d_dinv = nullptr;
    Sys<T>& s = h->s;
    ok = ok && h->alloc(&d_rp, n + 1) && h->alloc(&d_ci, nnz) && h->alloc(&d_va, nnz) && h->alloc(&d_dinv, n) &&
         h->alloc(&s.x, n) && h->alloc(&s.r, n) && h->alloc(&s.z, n) && h->alloc(&s.q, n) && h->alloc(&h->p0, n) &&
         h->alloc(&h->p1, n) && h->alloc(&h->d_b, n) && h->alloc(&s.partPQ, MAXB) && h->alloc(&s.partRZ, MAXB) &&
         h->alloc(&s.partRR, MAXB) && h->alloc(&s.rho, 2) && h->alloc(&s.flag, 1) &&
         hipHostMalloc((void**)&h->h_flag, sizeof(int)) == hipSuccess &&
         hipHostMalloc((void**)&h->h_part, sizeof(double) * MAXB) == hipSuccess;
    if (!ok) { delete h; return fail(-13, "device allocation failed"); }
    ok = hipMemcpy(d_rp, rp.data(), sizeof(int) * (n + 1), hipMemcpyHostToDevice) == hipSuccess &&
         hipMemcpy(d_ci, ci.data(), sizeof(int) * nnz, hipMemcpyHostToDevice) == hipSuccess &&
         hipMemcpy(d_va, va.data(), sizeof(T) * nnz, hipMemcpyHostToDevice) == hipSuccess &&
         hipMemcpy(d_dinv, dinv.data(), sizeof(T) * n, hipMemcpyHostToDevice) == hipSuccess;
    if (!ok) { delete h; return fail(-1, "upload failed"); }
    s.n = (int)n; s.rp = d_rp; s.ci = d_ci; s.va = d_va; s.dinv = d_dinv; s.b = h->d_b;
    const long rowsPerBlock = SB / h->lpr;
    s.nblk = (int)std::min<long>(MAXB, (n + rowsPerBlock - 1) / rowsPerBlock);
    stat[0] = 0;
    return (int64_t)(intptr_t) static_cast<HandleBase*>(h);
}

template <class T, int LPR>
void launch_resid(Handle<T>* h) { hipLaunchKernelGGL((k_sp_resid<T, LPR>), dim3(h->s.nblk), dim3(SB), 0, h->stream, h->s); }
template <class T, int LPR>
void launch_dir(Handle<T>* h, int it, const T* po, T* pn) {
    hipLaunchKernelGGL((k_sp_dir<T, LPR>), dim3(h->s.nblk), dim3(SB), 0, h->stream, h->s, it, po, pn);
}

// one right-hand side (host pointers); returns the final relative residual
template <class T>
double solve_one(Handle<T>* h, const T* rhs, T* x) {
    Sys<T>& s = h->s;
    const int n = s.n;
    double bb = 0.0;
    for (int i = 0; i < n; ++i) bb += abs2v(rhs[i]);
    if (!(bb > 0.0)) {
        for (int i = 0; i < n; ++i) zerov(x[i]);
        return 0.0;
    }
    hipMemcpyAsync(h->d_b, rhs, sizeof(T) * n, hipMemcpyHostToDevice, h->stream);
    hipMemsetAsync(s.x, 0, sizeof(T) * n, h->stream);
    s.bb = bb;
    const double target = 1e-14;
    s.tol2 = target * target;
    const int maxit = std::max(2000, 20 * n), chk = 32;
    double rel = 1.0, prev = 1e300;
    int itTotal = 0, pass = 0;
    for (; pass < 8; ++pass) {
        if (h->lpr == 4) launch_resid<T, 4>(h); else launch_resid<T, 16>(h);
        hipMemcpyAsync(h->h_part, s.partRR, sizeof(double) * s.nblk, hipMemcpyDeviceToHost, h->stream);
        hipStreamSynchronize(h->stream);
        double rr = 0.0;
        for (int i = 0; i < s.nblk; ++i) rr += h->h_part[i];
        rel = std::sqrt(rr / bb);
        if (!(rel > target) || (pass > 0 && rel > 0.5 * prev) || itTotal >= maxit) break;
        prev = rel;
        hipMemsetAsync(s.flag, 0, sizeof(int), h->stream);
        *h->h_flag = 0;
        T* pb[2] = {h->p0, h->p1};
        for (int it = 0; itTotal < maxit; ++it, ++itTotal) {
            if (h->lpr == 4) launch_dir<T, 4>(h, it, pb[(it + 1) & 1], pb[it & 1]);
            else launch_dir<T, 16>(h, it, pb[(it + 1) & 1], pb[it & 1]);
            hipLaunchKernelGGL(k_sp_upd<T>, dim3(s.nblk), dim3(SB), 0, h->stream, s, it, pb[it & 1]);
            if ((it + 1) % chk == 0) {
                hipMemcpyAsync(h->h_flag, s.flag, sizeof(int), hipMemcpyDeviceToHost, h->stream);
                hipStreamSynchronize(h->stream);
                if (*h->h_flag) break;
            }
        }
    }
    hipMemcpyAsync(x, s.x, sizeof(T) * n, hipMemcpyDeviceToHost, h->stream);
    hipStreamSynchronize(h->stream);
    h->lastIters = itTotal; h->lastPasses = pass;
    return rel;
}

template <class T>
int64_t solve_impl(const int64_t* handle, const int64_t* nrhsp, const double* rhs, double* x) {
    if (!handle || !nrhsp || !rhs || !x || !*handle) return -1;
    auto* hb = reinterpret_cast<HandleBase*>((intptr_t)*handle);
    if (hb->cmplx != (sizeof(T) == sizeof(cplx))) {
        fprintf(stderr, "libhmcmt_hip (MUMPS interface): real/complex solve called on a handle of the other kind\n");
        return -1;
    }
    auto* h = static_cast<Handle<T>*>(hb);
    if (hipSetDevice(h->device) != hipSuccess) return -1;
    const int64_t nrhs = *nrhsp;
    double worst = 0.0;
    for (int64_t j = 0; j < nrhs; ++j) {
        const double rel = solve_one<T>(h, reinterpret_cast<const T*>(rhs) + j * h->n, reinterpret_cast<T*>(x) + j * h->n);
        worst = std::max(worst, rel);
    }
    h->lastRes = worst;
    if (!(worst <= 1e-10)) {
        fprintf(stderr, "libhmcmt_hip (MUMPS interface): solve stopped at relative residual %.3e\n", worst);
        return -10;
    }
    return 0;
}

template <class T>
void solve_sparse_impl(const int64_t* handle, const int64_t* nrhsp, const double* nzval, const int64_t* rowval,
                       const int64_t* colptr, double* x) {
    if (!handle || !nrhsp || !nzval || !rowval || !colptr || !x || !*handle) return;
    auto* hb = reinterpret_cast<HandleBase*>((intptr_t)*handle);
    if (hb->cmplx != (sizeof(T) == sizeof(cplx))) return;
    auto* h = static_cast<Handle<T>*>(hb);
    const int64_t nrhs = *nrhsp;
    std::vector<T> dense((size_t)h->n * nrhs);
    for (auto& v : dense) zerov(v);
    for (int64_t j = 0; j < nrhs; ++j)
        for (int64_t k = colptr[j] - 1; k < colptr[j + 1] - 1; ++k) {
            const int64_t i = rowval[k] - 1;
            if (i >= 0 && i < h->n) dense[(size_t)j * h->n + i] = make_val<T>(nzval, k);
        }
    solve_impl<T>(handle, nrhsp, reinterpret_cast<const double*>(dense.data()), x);
}

int64_t destroy_impl(const int64_t* handle) {
    if (!handle || !*handle) return -1;
    delete reinterpret_cast<HandleBase*>((intptr_t)*handle);
    return 0;
}

}  // namespace

extern "C" {

int64_t factor_mumps_cmplx_(const int64_t* n, const int64_t* sym, const int64_t* ooc, const double* nzval,
                            const int64_t* rowval, const int64_t* colptr, int64_t* stat) {
    (void)ooc;
    return factor_impl<cplx>(n, sym, nzval, rowval, colptr, stat);
}
int64_t factor_mumps_(const int64_t* n, const int64_t* sym, const int64_t* ooc, const double* nzval, const int64_t* rowval,
                      const int64_t* colptr, int64_t* stat) {
    (void)ooc;
    return factor_impl<double>(n, sym, nzval, rowval, colptr, stat);
}
int64_t solve_mumps_cmplx_(const int64_t* handle, const int64_t* nrhs, const double* rhs, double* x, const int64_t* transpose) {
    (void)transpose;   // symmetric matrices only: A' = A
    return solve_impl<cplx>(handle, nrhs, rhs, x);
}
int64_t solve_mumps_(const int64_t* handle, const int64_t* nrhs, const double* rhs, double* x, const int64_t* transpose) {
    (void)transpose;
    return solve_impl<double>(handle, nrhs, rhs, x);
}
void solve_mumps_cmplx_sparse_rhs_(const int64_t* handle, const int64_t* nzrhs, const int64_t* nrhs, const double* nzval,
                                   const int64_t* rowval, const int64_t* colptr, double* x, const int64_t* transpose) {
    (void)nzrhs; (void)transpose;
    solve_sparse_impl<cplx>(handle, nrhs, nzval, rowval, colptr, x);
}
void solve_mumps_sparse_rhs_(const int64_t* handle, const int64_t* nzrhs, const int64_t* nrhs, const double* nzval,
                             const int64_t* rowval, const int64_t* colptr, double* x, const int64_t* transpose) {
    (void)nzrhs; (void)transpose;
    solve_sparse_impl<double>(handle, nrhs, nzval, rowval, colptr, x);
}
int64_t destroy_mumps_cmplx_(const int64_t* handle) { return destroy_impl(handle); }
int64_t destroy_mumps_(const int64_t* handle) { return destroy_impl(handle); }

int64_t hmcmt_mumps_last_solve(const int64_t* handle, double* out) {
    if (!handle || !*handle || !out) return -1;
    auto* hb = reinterpret_cast<HandleBase*>((intptr_t)*handle);
    if (hb->cmplx) {
        auto* h = static_cast<Handle<cplx>*>(hb);
        out[0] = h->lastIters; out[1] = h->lastPasses; out[2] = h->lastRes;
    } else {
        auto* h = static_cast<Handle<double>*>(hb);
        out[0] = h->lastIters; out[1] = h->lastPasses; out[2] = h->lastRes;
    }
    return 0;
}

}  // extern "C"
