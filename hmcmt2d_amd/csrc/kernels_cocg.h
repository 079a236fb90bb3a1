// kernels_cocg.h -- part of libhmcmt_hip.so; included by hmcmt_hip.hip INSIDE its anonymous namespace (one translation unit).
// Solver state shared by the COCG kernels, deterministic wave / block reductions, and the classic (one kernel per
// vector operation) COCG kernels: the Jacobi and plain-FDM preconditioners, the fp64 restart of the default path.
#pragma once

// ----------------------------------------------------------------------------------------------
// solver state shared by the COCG kernels
// ----------------------------------------------------------------------------------------------
struct Solver {
    int S, NB, NYP, NZP, ny, nz, nFreq;
    long vstride, chunk;
    const double* omega;
    const double *cY, *cZ, *dK, *dM;      // [2][vstride]
    const float4* cf32;                   // [2][vstride][2] the same six stencil coefficients of a node as floats: {dK, dM, cY[e], cY[e-1]}, {cZ[e], cZ[e-NYP], 0, 0}
                                          // (two 16-byte loads instead of six 8-byte ones, for the preconditioner's own stencils; interior nodes, else 0)
    const double* ofz;                    // [2][NZP]
    const cplx* invp;                     // [S][vstride]
    cplx *x, *r, *p, *q, *z, *y, *t;      // [S][vstride]
    cplx *dinv;                           // [S][vstride] omegaJ / diag(A) on interior nodes, 0 elsewhere
    float2 *dinv32;                       // ... the same as complex64, for the two-sweep smoother's kernels (it is read four times per iteration there)
    // mixed-precision FDM stage (options.fdm_precision == 0): bf16 transform operands, fp32 tridiagonal
    float2* t32;                          // [S][vstride] complex64 transform input (or its pre-split bf16 form, see store_t32)
    int splitT;                           // 1: t32 / y32 hold bf16 hi/lo planes instead of complex64
    int twist;                            // = View.twist: the inverse pivots are those of the twisted factorisation
    float2* y32;                          // [S][vstride] complex64
    const float2* invp32;                 // [S][vstride]
    cplx *p2, *r2;                        // second buffers of p and r for the fused kernels
    float2 *zs32, *z4_32;                 // two smoothing sweeps per side (sweeps == 2): the pre-smoothed iterate z2 and the iterate
                                          // after the first post-sweep z4, complex64 (k_update_fused<2> -> k_back_post<.,2> -> k_spmv_fused<2>)
    float2 *t2_32;                        // ... the smoothed residual t the FDM stage was given, as complex64 (for the rho identity below)
    cplx *partR;                          // [S][MAXNB] two sweeps: sum over a tile of (r + t) .* z2 -- with k_back_post<.,2>'s sum of t .* (V y) it
                                          // makes rho = r'z WITHOUT the second post-sweep: r'z5 = t'z3 + r'z2 (z3 = z2 + F t), an identity of the
                                          // symmetric construction (G'^2 r = t for the smoother's iteration matrix G) -- so that sweep can run
                                          // inside k_spmv_fused<2>, after the reduction that needs rho
    float w2;                             // two sweeps: damping of the INNER sweeps (second pre-sweep, first post-sweep) relative to the outer ones
    int merged2;                          // two sweeps: 1 = second post-sweep inside k_spmv_fused<2> (always, in solves), 0 = k_post2 launch (the test hook that needs z in memory)
    int sweeps;                           // damped Jacobi sweeps on each side of the FDM stage in the solve at hand (1 or 2)
    float2 *z32, *p32a, *p32b;            // fused path: preconditioned residual and the two search-direction buffers as complex64
                                          // (x, r, q and every inner product stay fp64; see k_spmv_fused)
    int RT, NTR;                          // rows per tile / row tiles per system of the fused kernels (NTR <= MAXNB)
    int xInFwd;                           // 1: x += alpha p and |x|^2 are done by k_fdm_fwd's idle waves (its pX argument), not by k_update_fused
    int RT2;                              // rows per tile of k_update_fused<2> (its two halo rows per side cost less on taller tiles)
    int RTS;                              // ... of k_spmv_fused<2> (HMCMT_RTS)
    cplx *partPQ;                         // [S][MAXNB]  p'q of the fused path
    cplx *rho2;                           // [2][S] rho by iteration parity (fused path)
    cplx *partA;                          // [S][MAXNB]  p'q   | r'z
    double *partB;                        // [S][MAXNB]  |x|^2 | |z|^2
    cplx *rho, *alphaBeta;                // [S]
    int *active, *iters, *status, *nactive;
    int* nactHost;                        // pinned host copy of *nactive (device address): the convergence polls only synchronise
    double *errEst;                       // [S] (zz/xx)
    double tol2;
    double* errRef;                       // [S] best error estimate so far / 10-fold improvements (stagnation watch of the mixed-precision solve)
    int* errRefIt;                        // [S] iteration at which errRef was set
    int stallIt;                          // iterations allowed per 10-fold drop of the error estimate (STALL_IT; HMCMT_STALL_IT)
    int* progHost;                        // pinned host word: the iteration whose k_spmv_fused has STARTED (host throttle, see solve())
    int* stallHost;                       // pinned host flag: a system has not improved its error estimate 10-fold in STALL_IT iterations
    int* failHost;                        // pinned host word: status (HMCMT_ENOCONV / HMCMT_EBREAKDOWN) of a system that has just given up --
                                          // the host must not build on this solve (adjoint on a failed forward, next leapfrog step)
    long long* ticks;                     // HMCMT_TICKS: View::ticks
    const cplx* xTickF;                   //   (the forward fields: k_extrap's forward call is the one on the timeline)
    long long* stamps;                    // HMCMT_STAMPS=<kernel>: per-workgroup s_memtime stamps of that kernel's phases ([workgroup][8]; printed at hmcmt_destroy)
    int stampKernel;                      // 1 k_update_fused<2>, 2 k_spmv_fused<2>
    unsigned long long* cntActive;        // non-null in an evaluation sampled by hmcmt_profile: += systems still active per iteration
};

// Sum over the 64 lanes of a wave, the total returned in EVERY lane.  Data-parallel-primitive moves inside the rows
// of 16 lanes (quad_perm xor 1, xor 2, row_ror 4, row_ror 8: ~4 cycles each) and one v_readlane per row instead of a
// butterfly of 6 ds_bpermute round trips per 32-bit half (~1 us for the three sums at the end of k_back_post).
// The order of the additions is fixed, and the final value is formed from lanes 0/16/32/48 only, so it is the same
// bit pattern in every lane and in every workgroup that reduces the same numbers.
template <int CTRL>
__device__ __forceinline__ double dpp_mov_f64(double v) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xF, 0xF, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xF, 0xF, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double readlane_f64(double v, int lane) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), lane), __builtin_amdgcn_readlane(__double2loint(v), lane));
}
// HMCMT_TICKS: the constant-rate wall clock at the start of the first workgroup / the end of the last one of kernel `id`
// (a store per workgroup when enabled, a null test otherwise); printed at hmcmt_destroy as the untraced timeline of the
// last evaluation -- rocprofv3's kernel trace makes the host's launches the bottleneck around the solves
enum { TK_SIGMA = 0, TK_BC, TK_EXTW, TK_EXT, TK_COEF, TK_PIVOT, TK_RESID_F, TK_SPMV, TK_UPDATE, TK_FWD, TK_BACK, TK_SOLVE_END, TK_RXALL,
       TK_SRC, TK_RESID_A, TK_WB, TK_BCSENS, TK_GRADCELL, TK_GRADFINAL, TK_LF_MOM, TK_LF_MAX, TK_LF_STEP, TK_SENS, TK_PERSIST_F, TK_PERSIST_A, TK_N };
__device__ __forceinline__ void tick_begin(long long* t, int id) {          // first workgroup of the first launch since the reset
    if (t && threadIdx.x == 0 && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && t[id] == -1) t[id] = wall_clock64();
}
__device__ __forceinline__ void tick_end(long long* t, int id) {            // every workgroup: plain store into one of 64 slots (the host takes the maximum)
    if (t && threadIdx.x == 0) {
        const unsigned b = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
        t[32 + 64 * id + (b & 63)] = wall_clock64();
    }
}
__device__ __forceinline__ double wave_sum(double v) {
    v += dpp_mov_f64<0xB1>(v);          // quad_perm [1,0,3,2]
    v += dpp_mov_f64<0x4E>(v);          // quad_perm [2,3,0,1]
    v += dpp_mov_f64<0x124>(v);         // row_ror:4
    v += dpp_mov_f64<0x128>(v);         // row_ror:8  -> the sum of the lane's row of 16
    return (readlane_f64(v, 0) + readlane_f64(v, 16)) + (readlane_f64(v, 32) + readlane_f64(v, 48));
}

// block-wide deterministic sum of up to 2 doubles; result valid in thread 0
__device__ __forceinline__ void block_sum2(double& a, double& b, double* sh /* [2*16]: workgroups of up to 16 waves */) {
    a = wave_sum(a);
    b = wave_sum(b);
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    if (l == 0) { sh[w] = a; sh[16 + w] = b; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const int nw = (blockDim.x + 63) >> 6;
        double sa = 0, sb = 0;
        for (int i = 0; i < nw; ++i) { sa += sh[i]; sb += sh[16 + i]; }
        a = sa; b = sb;
    }
}

// the same for workgroups of up to 8 waves (sh: [2*8])
__device__ __forceinline__ void block_sum2_8(double& a, double& b, double* sh) {
    a = wave_sum(a);
    b = wave_sum(b);
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    if (l == 0) { sh[w] = a; sh[8 + w] = b; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const int nw = (blockDim.x + 63) >> 6;
        double sa = 0, sb = 0;
        for (int i = 0; i < nw; ++i) { sa += sh[i]; sb += sh[8 + i]; }
        a = sa; b = sb;
    }
}

__device__ __forceinline__ void block_sum3_8(double& a, double& b, double& c, double* sh, int nw) {
    a = wave_sum(a);
    b = wave_sum(b);
    c = wave_sum(c);
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    if (l == 0) { sh[w] = a; sh[8 + w] = b; sh[16 + w] = c; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double sa = 0, sb = 0, sc = 0;
        for (int i = 0; i < nw; ++i) { sa += sh[i]; sb += sh[8 + i]; sc += sh[16 + i]; }
        a = sa; b = sb; c = sc;
    }
}

// sum of n <= 64 per-block partials by one wave: lane b fetches partial b (one memory latency instead of n in a
// row), a fixed sequence of lane exchanges (wave_sum) adds them up, every lane gets the total -- the same value in every wave
// of every block, so all blocks of a system still agree on alpha / beta / convergence
__device__ __forceinline__ double wave_total(double v) { return wave_sum(v); }
__device__ __forceinline__ double total_part(const double* part, int n) {
    const int l = threadIdx.x & 63;
    return wave_total(l < n ? part[l] : 0.0);
}
__device__ __forceinline__ cplx total_part(const cplx* part, int n) {
    const int l = threadIdx.x & 63;
    const cplx v = l < n ? part[l] : cplx{0, 0};
    return cplx{wave_total(v.re), wave_total(v.im)};
}

__device__ __forceinline__ cplx sum_partA(const Solver& k, int s) {
    cplx t = cplx{0, 0};
    for (int b = 0; b < k.NB; ++b) t += k.partA[(long)s * MAXNB + b];
    return t;
}

// q = A p (interior nodes), partA = p'q (unconjugated)
__global__ __launch_bounds__(VBLOCK) void k_spmv(Solver k) {
    const int s = blockIdx.y;
    if (!k.active[s]) return;
    __shared__ double sh[32];
    const int mode = s >= k.nFreq;
    const long mo = (long)mode * k.vstride, so = (long)s * k.vstride;
    const double w = k.omega[s];
    const cplx* p = k.p + so;
    cplx* q = k.q + so;
    const long e0 = (long)blockIdx.x * k.chunk, e1 = min(e0 + k.chunk, k.vstride);
    double ar = 0, ai = 0;
    for (long e = e0 + threadIdx.x; e < e1; e += VBLOCK) {
        const int iz = (int)(e / k.NYP), iy = (int)(e - (long)iz * k.NYP);
        if (iz >= 1 && iz <= k.nz - 1 && iy >= 1 && iy <= k.ny - 1) {
            const cplx c = p[e];
            const double dk = k.dK[mo + e], dm = w * k.dM[mo + e];
            cplx acc = cplx{dk * c.re - dm * c.im, dk * c.im + dm * c.re};
            acc += k.cY[mo + e] * p[e + 1];
            acc += k.cY[mo + e - 1] * p[e - 1];
            acc += k.cZ[mo + e] * p[e + k.NYP];
            acc += k.cZ[mo + e - k.NYP] * p[e - k.NYP];
            q[e] = acc;
            ar += c.re * acc.re - c.im * acc.im;
            ai += c.re * acc.im + c.im * acc.re;
        }
    }
    block_sum2(ar, ai, sh);
    if (threadIdx.x == 0) k.partA[(long)s * MAXNB + blockIdx.x] = cplx{ar, ai};
}

// alpha = rho / p'q ; x += alpha p ; r -= alpha q ; partB = |x|^2 over interior nodes
__global__ __launch_bounds__(VBLOCK) void k_update(Solver k) {
    const int s = blockIdx.y;
    if (!k.active[s]) return;
    __shared__ double sh[32];
    const long so = (long)s * k.vstride;
    const cplx al = k.rho[s] / sum_partA(k, s);
    const cplx *p = k.p + so, *q = k.q + so;
    cplx *x = k.x + so, *r = k.r + so;
    const long e0 = (long)blockIdx.x * k.chunk, e1 = min(e0 + k.chunk, k.vstride);
    double xx = 0, dummy = 0;
    for (long e = e0 + threadIdx.x; e < e1; e += VBLOCK) {
        const int iz = (int)(e / k.NYP), iy = (int)(e - (long)iz * k.NYP);
        if (iz >= 1 && iz <= k.nz - 1 && iy >= 1 && iy <= k.ny - 1) {
            cplx xv = x[e] + al * p[e];
            x[e] = xv;
            r[e] -= al * q[e];
            xx += cabs2(xv);
        }
    }
    block_sum2(xx, dummy, sh);
    if (threadIdx.x == 0) {
        k.partB[(long)s * MAXNB + blockIdx.x] = xx;
        if (blockIdx.x == 0) k.alphaBeta[s] = al;
    }
}

// partA = r'z (unconjugated), partB2 = |z|^2
__global__ __launch_bounds__(VBLOCK) void k_dots(Solver k, double* partZZ) {
    const int s = blockIdx.y;
    if (!k.active[s]) return;
    __shared__ double sh[32];
    __shared__ double sh2[32];
    const long so = (long)s * k.vstride;
    const cplx *r = k.r + so, *z = k.z + so;
    const long e0 = (long)blockIdx.x * k.chunk, e1 = min(e0 + k.chunk, k.vstride);
    double ar = 0, ai = 0, zz = 0, dummy = 0;
    for (long e = e0 + threadIdx.x; e < e1; e += VBLOCK) {
        const cplx a = r[e], b = z[e];
        ar += a.re * b.re - a.im * b.im;
        ai += a.re * b.im + a.im * b.re;
        zz += cabs2(b);
    }
    block_sum2(ar, ai, sh);
    block_sum2(zz, dummy, sh2);
    if (threadIdx.x == 0) {
        k.partA[(long)s * MAXNB + blockIdx.x] = cplx{ar, ai};
        partZZ[(long)s * MAXNB + blockIdx.x] = zz;
    }
}

// per-system scalar bookkeeping: convergence test on the error estimate ||z|| <= tol ||x||,
// beta = rho_new / rho_old.  first != 0: initialise (rho = r'z, beta = 0).
__global__ void k_check(Solver k, const double* partZZ, int first, int maxit) {
    __shared__ int cnt;
    if (threadIdx.x == 0) cnt = 0;
    __syncthreads();
    for (int s = threadIdx.x; s < k.S; s += blockDim.x) {
        if (!k.active[s]) continue;
        cplx rz = cplx{0, 0};
        double zz = 0, xx = 0;
        for (int b = 0; b < k.NB; ++b) {
            rz += k.partA[(long)s * MAXNB + b];
            zz += partZZ[(long)s * MAXNB + b];
            if (!first) xx += k.partB[(long)s * MAXNB + b];
        }
        bool on = true;
        if (first == 2) {                                                // restart with a different preconditioner
            k.rho[s] = rz;
            k.alphaBeta[s] = cplx{0, 0};
        } else if (first) {
            k.rho[s] = rz;
            k.alphaBeta[s] = cplx{0, 0};
            k.errEst[s] = 1.0;
            if (zz == 0.0) { on = false; k.errEst[s] = 0.0; }          // zero right-hand side
        } else {
            k.iters[s] += 1;
            k.errEst[s] = sqrt(zz / xx);
            if (zz <= k.tol2 * xx) on = false;
            else {
                k.alphaBeta[s] = rz / k.rho[s];
                k.rho[s] = rz;
                if (k.iters[s] >= maxit) { on = false; k.status[s] = HMCMT_ENOCONV; *k.failHost = HMCMT_ENOCONV; __threadfence_system(); }
            }
        }
        if (!(isfinite(rz.re) && isfinite(rz.im) && isfinite(zz) && isfinite(xx))) {
            on = false; k.status[s] = HMCMT_EBREAKDOWN; *k.failHost = HMCMT_EBREAKDOWN; __threadfence_system();
        }
        if (!on) k.active[s] = 0;
        else atomicAdd(&cnt, 1);
    }
    __syncthreads();
    if (threadIdx.x == 0) { *k.nactive = cnt; *k.nactHost = cnt; }
}

// p = z + beta p   (first: p = z)
__global__ __launch_bounds__(VBLOCK) void k_pupdate(Solver k, int first) {
    const int s = blockIdx.y;
    if (!k.active[s]) return;
    const long so = (long)s * k.vstride;
    const cplx be = k.alphaBeta[s];
    const cplx* z = k.z + so;
    cplx* p = k.p + so;
    const long e0 = (long)blockIdx.x * k.chunk, e1 = min(e0 + k.chunk, k.vstride);
    for (long e = e0 + threadIdx.x; e < e1; e += VBLOCK) p[e] = first ? z[e] : z[e] + be * p[e];
}

// z = r / diag(A)  (Jacobi)
__global__ __launch_bounds__(VBLOCK) void k_jacobi(Solver k) {
    const int s = blockIdx.y;
    if (!k.active[s]) return;
    const int mode = s >= k.nFreq;
    const long mo = (long)mode * k.vstride, so = (long)s * k.vstride;
    const double w = k.omega[s];
    const long e0 = (long)blockIdx.x * k.chunk, e1 = min(e0 + k.chunk, k.vstride);
    for (long e = e0 + threadIdx.x; e < e1; e += VBLOCK) {
        const int iz = (int)(e / k.NYP), iy = (int)(e - (long)iz * k.NYP);
        cplx zv = cplx{0, 0};
        if (iz >= 1 && iz <= k.nz - 1 && iy >= 1 && iy <= k.ny - 1)
            zv = k.r[so + e] / cplx{k.dK[mo + e], w * k.dM[mo + e]};
        k.z[so + e] = zv;
    }
}
