// libhmcmt_hip.so -- gfx950 (MI355X) implementation of the HMCMT2D hot path behind include/hmcmt.h.
//
// Structure:
//   * "item" kernels: one thread per node / cell / receiver / boundary column, bodies in
//     hmcmt_items.h (assembly from sigma, 1-D boundary fields, receiver functionals, adjoint
//     sources, J^T accumulation).
//   * batched COCG over all S = 2*nFreq complex-symmetric 5-point systems at once, with
//       - stencil SpMV on the padded nodal grid (real K shared by all frequencies of a mode,
//         i*omega*D formed on the fly),
//       - fast-diagonalisation preconditioner: two FP64-MFMA transforms with the mesh's y-eigenbasis
//         (v_mfma_f64_16x16x4_f64) around a batched tridiagonal solve in z,
//       - deterministic two-stage reductions (wave shuffles + fixed partial arrays).
//   * no host compute path: every entry point needs a HIP device.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include "../../include/hmcmt.h"
#include "hmcmt_host.h"
#include "hmcmt_items.h"

using namespace hmcmt;

namespace {

constexpr int MAXNB = 64;          // max partial-sum blocks per system (<= 64: one wave sums them, total_part)
constexpr int VBLOCK = 256;        // threads of the vector kernels

// ----------------------------------------------------------------------------------------------
// solver state shared by the COCG kernels
// ----------------------------------------------------------------------------------------------
struct Solver {
    int S, NB, NYP, NZP, ny, nz, nFreq;
    long vstride, chunk;
    const double* omega;
    const double *cY, *cZ, *dK, *dM;      // [2][vstride]
    const double* ofz;                    // [2][NZP]
    const cplx* invp;                     // [S][vstride]
    cplx *x, *r, *p, *q, *z, *y, *t;      // [S][vstride]
    cplx *dinv;                           // [S][vstride] omegaJ / diag(A) on interior nodes, 0 elsewhere
    // mixed-precision FDM stage (options.fdm_precision == 0): bf16 transform operands, fp32 tridiagonal
    float2* t32;                          // [S][vstride] complex64 transform input (or its pre-split bf16 form, see store_t32)
    int splitT;                           // 1: t32 / y32 hold bf16 hi/lo planes instead of complex64
    int twist;                            // = View.twist: the inverse pivots are those of the twisted factorisation
    float2* y32;                          // [S][vstride] complex64
    const float2* invp32;                 // [S][vstride]
    cplx *p2, *r2;                        // second buffers of p and r for the fused kernels
    int RT, NTR;                          // rows per tile / row tiles per system of the fused kernels (NTR <= MAXNB)
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
    int* stallHost;                       // pinned host flag: a system has not improved its error estimate 10-fold in STALL_IT iterations
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
__device__ __forceinline__ double wave_sum(double v) {
    v += dpp_mov_f64<0xB1>(v);          // quad_perm [1,0,3,2]
    v += dpp_mov_f64<0x4E>(v);          // quad_perm [2,3,0,1]
    v += dpp_mov_f64<0x124>(v);         // row_ror:4
    v += dpp_mov_f64<0x128>(v);         // row_ror:8  -> the sum of the lane's row of 16
    return (readlane_f64(v, 0) + readlane_f64(v, 16)) + (readlane_f64(v, 32) + readlane_f64(v, 48));
}

// block-wide deterministic sum of up to 2 doubles; result valid in thread 0
__device__ __forceinline__ void block_sum2(double& a, double& b, double* sh /* [2*4] */) {
    a = wave_sum(a);
    b = wave_sum(b);
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    if (l == 0) { sh[w] = a; sh[4 + w] = b; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const int nw = (blockDim.x + 63) >> 6;
        double sa = 0, sb = 0;
        for (int i = 0; i < nw; ++i) { sa += sh[i]; sb += sh[4 + i]; }
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
    __shared__ double sh[8];
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
    __shared__ double sh[8];
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
    __shared__ double sh[8];
    __shared__ double sh2[8];
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
                if (k.iters[s] >= maxit) { on = false; k.status[s] = HMCMT_ENOCONV; }
            }
        }
        if (!(isfinite(rz.re) && isfinite(rz.im) && isfinite(zz) && isfinite(xx))) {
            on = false; k.status[s] = HMCMT_EBREAKDOWN;
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

// ----------------------------------------------------------------------------------------------
// FDM transform: C[m][n] = sum_k A[m][k] * B[k][n],  A, C complex [M][NYP], B real [NYP][NYP].
//
// One wave = 8 complex rows x NTW column tiles of 16.  The 16-row MFMA tile stacks the rows' real
// parts (tile rows 0-7) and imaginary parts (8-15), so one B fragment feeds both.
// v_mfma_f64_16x16x4_f64 layout (measured, scripts/probe/mfma_f64_layout.hip):
//   A[i = lane%16][k = lane/16], B[k = lane/16][j = lane%16], D[i = 4*r + lane/16][j = lane%16].
// The reduction index is processed 16 at a time with the permutation k = 16*kg + 4*(lane/16) + i
// for MFMA step i = 0..3, so each lane reads 4 consecutive complex of its A row (64 B) and the
// constant B operand is pre-swizzled on the host into fragment order
//   Bsw[((kg*NT + t)*64 + lane)*4 + i] = B[16*kg + 4*(lane/16) + i][16*t + lane%16]
// (two 16-byte loads per tile per 4 MFMAs).  Operands of group kg+1 are fetched into registers
// while group kg is multiplied; there is no LDS and no barrier.
// ----------------------------------------------------------------------------------------------
typedef double d4 __attribute__((ext_vector_type(4)));

template <int NTW>
__device__ __forceinline__ void transform_body(const cplx* __restrict__ A, const double* __restrict__ Bsw,
                                               cplx* __restrict__ C, int M, int NYP, int m0, int t0, int lane) {
    const int NT = NYP >> 4, KG = NYP >> 4;
    // Every workgroup streams the same B; starting each at a different k-group keeps the CUs of an
    // XCD on different L2 channels instead of all requesting the same lines at once.
    const int kg0 = (m0 >> 3) % KG;
    const int li = lane & 15, lk = lane >> 4;
    const bool im = (li >> 3) != 0;
    const int arow = min(m0 + (li & 7), M - 1);
    const d4* Ap = reinterpret_cast<const d4*>(A + (long)arow * NYP + 4 * lk);
    const d4* Bp = reinterpret_cast<const d4*>(Bsw) + (long)t0 * 64 + lane;
    const long bstride = (long)NT * 64;                // d4 per k-group
    d4 acc[NTW];
#pragma unroll
    for (int t = 0; t < NTW; ++t) acc[t] = d4{0, 0, 0, 0};
    d4 a0 = Ap[kg0 * 8], a1 = Ap[kg0 * 8 + 1];
    d4 b[NTW];
#pragma unroll
    for (int t = 0; t < NTW; ++t) b[t] = Bp[kg0 * bstride + t * 64];
    for (int it = 0; it < KG; ++it) {
        int kn = kg0 + it + 1;                          // next k-group (wraps; last one re-reads, harmless)
        if (kn >= KG) kn -= KG;
        const d4 na0 = Ap[kn * 8], na1 = Ap[kn * 8 + 1];
        d4 nb[NTW];
#pragma unroll
        for (int t = 0; t < NTW; ++t) nb[t] = Bp[kn * bstride + t * 64];
        // Pin the software pipeline.  Left alone, LLVM folds the phi of loads back into a load at the
        // top of the iteration and the scheduler emits load -> wait -> MFMA with no overlap.  The two
        // scheduling barriers keep "issue next loads | multiply current | wait for next" in this order.
        __builtin_amdgcn_sched_barrier(0);
        const double av[4] = {im ? a0[1] : a0[0], im ? a0[3] : a0[2], im ? a1[1] : a1[0], im ? a1[3] : a1[2]};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
            for (int t = 0; t < NTW; ++t) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[i], b[t][i], acc[t], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        a0 = na0; a1 = na1;
        asm volatile("" : "+v"(a0), "+v"(a1));
#pragma unroll
        for (int t = 0; t < NTW; ++t) { b[t] = nb[t]; asm volatile("" : "+v"(b[t])); }
    }
    // r = 0,1: real parts of complex rows lk, 4+lk; r = 2,3: their imaginary parts -> every lane owns two
    // complete complex results; 16 lanes write 256 contiguous bytes.
#pragma unroll
    for (int t = 0; t < NTW; ++t) {
        const long col = (long)(t0 + t) * 16 + li;
        if (m0 + lk < M) C[(long)(m0 + lk) * NYP + col] = cplx{acc[t][0], acc[t][2]};
        if (m0 + 4 + lk < M) C[(long)(m0 + 4 + lk) * NYP + col] = cplx{acc[t][1], acc[t][3]};
    }
}

// Workgroup = RG row groups (8 complex rows each) x NW column splits, RG*NW <= 4 waves, so that a
// CU holding one workgroup runs one wave per SIMD (two 2-wave workgroups on a CU land on the same
// SIMD pair and halve the MFMA rate -- measured, scripts/probe/transform_bench.hip).  Column tiles
// are dealt to the NW waves as evenly as possible (first `extra` waves get one more).
__global__ __launch_bounds__(256) void k_transform(const cplx* __restrict__ A, const double* __restrict__ Bsw,
                                                    cplx* __restrict__ C, int M, int NYP, int rowsPerSys,
                                                    const int* __restrict__ active, int NW, int RG) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int rg = wave / NW, nw = wave - rg * NW;
    const int m0 = (blockIdx.x * RG + rg) * 8;
    if (m0 >= M) return;
    if (active) {
        const int s0 = m0 / rowsPerSys, s1 = min(m0 + 7, M - 1) / rowsPerSys;
        if (!active[s0] && !active[s1]) return;
    }
    const int NT = NYP >> 4;
    const int base = NT / NW, extra = NT % NW;
    const int ntl = base + (nw < extra ? 1 : 0);
    const int t0 = nw * base + min(nw, extra);
    switch (ntl) {
        case 1: transform_body<1>(A, Bsw, C, M, NYP, m0, t0, lane); break;
        case 2: transform_body<2>(A, Bsw, C, M, NYP, m0, t0, lane); break;
        case 3: transform_body<3>(A, Bsw, C, M, NYP, m0, t0, lane); break;
        case 4: transform_body<4>(A, Bsw, C, M, NYP, m0, t0, lane); break;
        case 5: transform_body<5>(A, Bsw, C, M, NYP, m0, t0, lane); break;
        case 6: transform_body<6>(A, Bsw, C, M, NYP, m0, t0, lane); break;
        case 7: transform_body<7>(A, Bsw, C, M, NYP, m0, t0, lane); break;
        default: break;
    }
}

// batched tridiagonal solve in z for every (system, eigenmode j), in place on y[s][iz][j]; lanes =
// consecutive j (coalesced rows).  About 75 ns per row at one wave per CU; neither deeper prefetch, more
// waves nor shorter chains change that (all measured).
// With HMCMT_TWIST the twisted factorisation of item_pivot is used (rows 1..mid swept top-down, rows
// n..mid+1 bottom-up as two interleaved chains joined by one 2x2 solve); it measured SLOWER (20 vs 16 us):
// the kernel is bound by instruction issue of its ~100 lone waves, not by dependency latency, so the
// default is the classic sweep (mid = n, bottom chain compiled out).
// Operands of the next block of rows are in flight while the current block is processed.
constexpr int TB = 8;
constexpr int MAXNZP = 1024;

struct alignas(8) c32 { float re, im; };
__device__ __forceinline__ c32 operator*(c32 a, c32 b) { return c32{a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re}; }
__device__ __forceinline__ c32 operator*(float a, c32 b) { return c32{a * b.re, a * b.im}; }
__device__ __forceinline__ c32 operator-(c32 a, c32 b) { return c32{a.re - b.re, a.im - b.im}; }
__device__ __forceinline__ void pin(cplx& a) { asm volatile("" : "+v"(a.re), "+v"(a.im)); }
__device__ __forceinline__ void pin(c32& a) { asm volatile("" : "+v"(a.re), "+v"(a.im)); }
// a - b*x with four single (unpacked) FMAs, dependent depth two: the serial tridiagonal sweeps are latency
// chains, and the packed v_pk_* forms hipcc's SLP pass would pick are slower per dependent step
__device__ __forceinline__ c32 cmsub(c32 a, c32 b, c32 x) {
    float re, im;
    asm("v_fma_f32 %0, -%1, %2, %3" : "=v"(re) : "v"(b.re), "v"(x.re), "v"(a.re));
    asm("v_fma_f32 %0, -%1, %2, %3" : "=v"(im) : "v"(b.re), "v"(x.im), "v"(a.im));
    asm("v_fma_f32 %0, %1, %2, %3" : "=v"(re) : "v"(b.im), "v"(x.im), "v"(re));
    asm("v_fma_f32 %0, -%1, %2, %3" : "=v"(im) : "v"(b.im), "v"(x.re), "v"(im));
    return c32{re, im};
}

template <class CT, class RT, bool TW>                   // TW = false: classic sweep, the bottom/down chain code is compiled out
__device__ __forceinline__ void thomas_twisted(CT* __restrict__ y, const CT* __restrict__ ip, const RT* sof, int n, long NYP) {
    const int mid = twist_mid(n, TW ? 1 : 0), nt = mid, nb = n - mid;
    CT pt = CT{0, 0}, pb = CT{0, 0};
    CT yt[TB], it[TB], yb[TB], ib[TB];
    // ---- phase 1: normalised elimination, top chain rows 1..mid, bottom chain rows n..mid+1
    auto load1 = [&](int k0, CT* a, CT* b, CT* c, CT* d) {
#pragma unroll
        for (int t = 0; t < TB; ++t) {
            const int kt = min(k0 + t, nt - 1), kb = min(k0 + t, max(nb - 1, 0));
            a[t] = y[(long)(1 + kt) * NYP]; b[t] = ip[(long)(1 + kt) * NYP];
            if (TW) { c[t] = y[(long)(n - kb) * NYP]; d[t] = ip[(long)(n - kb) * NYP]; }
        }
    };
    load1(0, yt, it, yb, ib);
    for (int k0 = 0; k0 < nt; k0 += TB) {
        CT nyt[TB], nit[TB], nyb[TB], nib[TB];
        load1(min(k0 + TB, max(nt - 1, 0)), nyt, nit, nyb, nib);
        RT ot[TB], ob[TB];
#pragma unroll
        for (int t = 0; t < TB; ++t) { ot[t] = sof[min(k0 + t, nt - 1)]; ob[t] = TW ? sof[n - min(k0 + t, max(nb - 1, 0))] : RT(0); }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < TB; ++t) {
            const int k = k0 + t;
            if (k < nt) { pt = (yt[t] - ot[t] * pt) * it[t]; y[(long)(1 + k) * NYP] = pt; }
            if (TW && k < nb) { pb = (yb[t] - ob[t] * pb) * ib[t]; y[(long)(n - k) * NYP] = pb; }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < TB; ++t) {
            yt[t] = nyt[t]; it[t] = nit[t]; pin(yt[t]); pin(it[t]);
            if (TW) { yb[t] = nyb[t]; ib[t] = nib[t]; pin(yb[t]); pin(ib[t]); }
        }
    }
    // ---- join: x_mid + c x_{mid+1} = y'_mid ; x_{mid+1} + c' x_mid = y''_{mid+1} ; ip[0] = 1/(1 - c c')
    if (TW && nb > 0) {
        const RT o = sof[mid];
        const CT c = o * ip[(long)mid * NYP], c2 = o * ip[(long)(mid + 1) * NYP];
        pt = (pt - c * pb) * ip[0];
        pb = pb - c2 * pt;
        y[(long)mid * NYP] = pt; y[(long)(mid + 1) * NYP] = pb;
    }
    // ---- phase 2: substitution outwards, up chain rows mid-1..1, down chain rows mid+2..n
    const int nu = nt - 1, nd = nb - 1;
    if (nu <= 0 && nd <= 0) return;
    auto load2 = [&](int k0, CT* a, CT* b, CT* c, CT* d) {
#pragma unroll
        for (int t = 0; t < TB; ++t) {
            const int ku = min(k0 + t, max(nu - 1, 0)), kd = min(k0 + t, max(nd - 1, 0));
            const int ru = max(mid - 1 - ku, 1), rd = min(mid + 2 + kd, n);
            a[t] = y[(long)ru * NYP]; b[t] = ip[(long)ru * NYP];
            if (TW) { c[t] = y[(long)rd * NYP]; d[t] = ip[(long)rd * NYP]; }
        }
    };
    load2(0, yt, it, yb, ib);
    const int nmax = max(nu, nd);
    for (int k0 = 0; k0 < nmax; k0 += TB) {
        CT nyt[TB], nit[TB], nyb[TB], nib[TB];
        load2(min(k0 + TB, max(nmax - 1, 0)), nyt, nit, nyb, nib);
        RT ou[TB], od[TB];
#pragma unroll
        for (int t = 0; t < TB; ++t) {
            const int ru = max(mid - 1 - min(k0 + t, max(nu - 1, 0)), 1), rd = min(mid + 2 + min(k0 + t, max(nd - 1, 0)), n);
            ou[t] = sof[ru]; od[t] = TW ? sof[rd - 1] : RT(0);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < TB; ++t) {
            const int k = k0 + t;
            if (k < nu) { pt = yt[t] - (ou[t] * it[t]) * pt; y[(long)(mid - 1 - k) * NYP] = pt; }
            if (TW && k < nd) { pb = yb[t] - (od[t] * ib[t]) * pb; y[(long)(mid + 2 + k) * NYP] = pb; }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < TB; ++t) {
            yt[t] = nyt[t]; it[t] = nit[t]; pin(yt[t]); pin(it[t]);
            if (TW) { yb[t] = nyb[t]; ib[t] = nib[t]; pin(yb[t]); pin(ib[t]); }
        }
    }
}

__global__ __launch_bounds__(64) void k_thomas(Solver k) {
    const int s = blockIdx.y;
    if (!k.active[s]) return;
    __shared__ double sof[MAXNZP];
    const int mode = s >= k.nFreq;
    for (int i = threadIdx.x; i < k.NZP; i += 64) sof[i] = k.ofz[(long)mode * k.NZP + i];
    __syncthreads();
    const int j = blockIdx.x * 64 + threadIdx.x;
    if (j >= k.ny - 1) return;
    if (k.twist) thomas_twisted<cplx, double, true>(k.y + (long)s * k.vstride + j, k.invp + (long)s * k.vstride + j, sof, k.nz - 1, k.NYP);
    else thomas_twisted<cplx, double, false>(k.y + (long)s * k.vstride + j, k.invp + (long)s * k.vstride + j, sof, k.nz - 1, k.NYP);
}

__global__ __launch_bounds__(64) void k_thomas32(Solver k) {
    const int s = blockIdx.y;
    if (!k.active[s]) return;
    __shared__ float sof[MAXNZP];
    const int mode = s >= k.nFreq;
    for (int i = threadIdx.x; i < k.NZP; i += 64) sof[i] = (float)k.ofz[(long)mode * k.NZP + i];
    __syncthreads();
    const int j = blockIdx.x * 64 + threadIdx.x;
    if (j >= k.ny - 1) return;
    if (k.twist) thomas_twisted<c32, float, true>(reinterpret_cast<c32*>(k.y32) + (long)s * k.vstride + j,
                                                   reinterpret_cast<const c32*>(k.invp32) + (long)s * k.vstride + j, sof, k.nz - 1, k.NYP);
    else thomas_twisted<c32, float, false>(reinterpret_cast<c32*>(k.y32) + (long)s * k.vstride + j,
                                            reinterpret_cast<const c32*>(k.invp32) + (long)s * k.vstride + j, sof, k.nz - 1, k.NYP);
}

// ---- symmetric Jacobi / FDM / Jacobi combination (default preconditioner):
//   z0 = wJ D^-1 r ; z1 = z0 + F (r - A z0) ; z = z1 + wJ D^-1 (r - A z1)
// point Jacobi removes the cell-scale coefficient contrast the laterally averaged FDM background
// cannot see; both factors are complex symmetric, so the product form above is too (COCG needs that).
__device__ __forceinline__ cplx stencil_at(const Solver& k, const cplx* u, long mo, long e, double w) {
    const cplx c = u[e];
    const double dk = k.dK[mo + e], dm = w * k.dM[mo + e];
    cplx acc = cplx{dk * c.re - dm * c.im, dk * c.im + dm * c.re};
    acc += k.cY[mo + e] * u[e + 1];
    acc += k.cY[mo + e - 1] * u[e - 1];
    acc += k.cZ[mo + e] * u[e + k.NYP];
    acc += k.cZ[mo + e - k.NYP] * u[e - k.NYP];
    return acc;
}

__global__ __launch_bounds__(VBLOCK) void k_dinv(Solver k, double wJ) {
    const int s = blockIdx.y;
    const int mode = s >= k.nFreq;
    const long mo = (long)mode * k.vstride, so = (long)s * k.vstride;
    const double w = k.omega[s];
    const long e0 = (long)blockIdx.x * k.chunk, e1 = min(e0 + k.chunk, k.vstride);
    for (long e = e0 + threadIdx.x; e < e1; e += VBLOCK) {
        const int iz = (int)(e / k.NYP), iy = (int)(e - (long)iz * k.NYP);
        cplx d = cplx{0, 0};
        if (iz >= 1 && iz <= k.nz - 1 && iy >= 1 && iy <= k.ny - 1) d = wJ / cplx{k.dK[mo + e], w * k.dM[mo + e]};
        k.dinv[so + e] = d;
    }
}

// t = r - A (dinv .* r)
__global__ __launch_bounds__(VBLOCK) void k_pre(Solver k) {
    const int s = blockIdx.y;
    if (!k.active[s]) return;
    const int mode = s >= k.nFreq;
    const long mo = (long)mode * k.vstride, so = (long)s * k.vstride;
    const double w = k.omega[s];
    const cplx *r = k.r + so, *di = k.dinv + so;
    cplx* t = k.t + so;
    const long e0 = (long)blockIdx.x * k.chunk, e1 = min(e0 + k.chunk, k.vstride);
    for (long e = e0 + threadIdx.x; e < e1; e += VBLOCK) {
        const int iz = (int)(e / k.NYP), iy = (int)(e - (long)iz * k.NYP);
        cplx out = cplx{0, 0};
        if (iz >= 1 && iz <= k.nz - 1 && iy >= 1 && iy <= k.ny - 1) {
            const cplx c = di[e] * r[e];
            const double dk = k.dK[mo + e], dm = w * k.dM[mo + e];
            cplx acc = cplx{dk * c.re - dm * c.im, dk * c.im + dm * c.re};
            acc += k.cY[mo + e] * (di[e + 1] * r[e + 1]);
            acc += k.cY[mo + e - 1] * (di[e - 1] * r[e - 1]);
            acc += k.cZ[mo + e] * (di[e + k.NYP] * r[e + k.NYP]);
            acc += k.cZ[mo + e - k.NYP] * (di[e - k.NYP] * r[e - k.NYP]);
            out = r[e] - acc;
        }
        t[e] = out;
    }
}

// z += dinv .* r
__global__ __launch_bounds__(VBLOCK) void k_mid(Solver k) {
    const int s = blockIdx.y;
    if (!k.active[s]) return;
    const long so = (long)s * k.vstride;
    const long e0 = (long)blockIdx.x * k.chunk, e1 = min(e0 + k.chunk, k.vstride);
    for (long e = e0 + threadIdx.x; e < e1; e += VBLOCK) k.z[so + e] += k.dinv[so + e] * k.r[so + e];
}

// t = z + dinv .* (r - A z) ; partA = r't ; partZZ = |t|^2     (t becomes the preconditioned residual)
__global__ __launch_bounds__(VBLOCK) void k_post(Solver k, double* partZZ) {
    const int s = blockIdx.y;
    if (!k.active[s]) return;
    __shared__ double sh[8];
    __shared__ double sh2[8];
    const int mode = s >= k.nFreq;
    const long mo = (long)mode * k.vstride, so = (long)s * k.vstride;
    const double w = k.omega[s];
    const cplx *r = k.r + so, *z = k.z + so, *di = k.dinv + so;
    cplx* t = k.t + so;
    const long e0 = (long)blockIdx.x * k.chunk, e1 = min(e0 + k.chunk, k.vstride);
    double ar = 0, ai = 0, zz = 0, dummy = 0;
    for (long e = e0 + threadIdx.x; e < e1; e += VBLOCK) {
        const int iz = (int)(e / k.NYP), iy = (int)(e - (long)iz * k.NYP);
        cplx out = cplx{0, 0};
        if (iz >= 1 && iz <= k.nz - 1 && iy >= 1 && iy <= k.ny - 1) {
            const cplx rv = r[e];
            out = z[e] + di[e] * (rv - stencil_at(k, z, mo, e, w));
            ar += rv.re * out.re - rv.im * out.im;
            ai += rv.re * out.im + rv.im * out.re;
            zz += cabs2(out);
        }
        t[e] = out;
    }
    block_sum2(ar, ai, sh);
    block_sum2(zz, dummy, sh2);
    if (threadIdx.x == 0) {
        k.partA[(long)s * MAXNB + blockIdx.x] = cplx{ar, ai};
        partZZ[(long)s * MAXNB + blockIdx.x] = zz;
    }
}

// ----------------------------------------------------------------------------------------------
// Mixed-precision FDM stage.  COCG keeps x, r, p and every inner product in fp64; the preconditioner
// only proposes search directions, and running its separable part with bf16 transform operands
// (fp32 accumulation) and a complex64 tridiagonal solve leaves the iteration counts unchanged
// (measured: identical to within +-1 iteration, same final error).  It moves the transforms from the
// 78 TF FP64 matrix pipe to the 2.5 PF BF16 pipe and shrinks the stage's traffic 2-4x.
//
// v_mfma_f32_16x16x32_bf16 (gfx950; 16 cycles per instruction vs 32 for the older 16x16x16 form) layout:
//   A[i = lane%16][k = 8*(lane/16) + t], B[k = 8*(lane/16) + t][j = lane%16], D[i = 4*(lane/16) + r][j = lane%16]
// (D and the 16x16x16 operand layout measured with scripts/probe/mfma_bf16_layout.hip).
// Tile row 2c+part = part (re/im) of complex row c, so a lane's four results are two complete complex
// numbers.  K is consumed 32 at a time, k = 32*kg + 8*(lane/16) + i, i.e. 8 contiguous complex per lane;
// V is pre-swizzled to
//   Bsw[((kg*NT + t)*64 + lane)*8 + i] = bf16(V[32*kg + 8*(lane/16) + i][16*t + lane%16]), zero for k >= NYP.
// ----------------------------------------------------------------------------------------------
typedef short s4v __attribute__((ext_vector_type(4)));
typedef float f4v __attribute__((ext_vector_type(4)));
typedef unsigned u4v __attribute__((ext_vector_type(4)));
typedef __bf16 bf8v __attribute__((ext_vector_type(8)));

__device__ __forceinline__ unsigned bf16_rn(float x) {           // round-to-nearest-even, finite inputs
    const unsigned u = __float_as_uint(x);
    return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
}
__device__ __forceinline__ unsigned pack_bf16(double re, double im) {
    return bf16_rn((float)re) | (bf16_rn((float)im) << 16);
}

constexpr int KCH = 7;             // k-groups (of 32) whose operands are requested together (all of K up to NYP = 224)
constexpr int LP_NTW = 2;          // column tiles per wave of the mixed-precision transform: many light waves hide latency

__device__ __forceinline__ float bf16_to_f32(unsigned h) { return __uint_as_float(h << 16); }

// A: complex64 rows.  Every value is split in registers into hi = bf16(x), lo = bf16(x - hi) and the
// product is accumulated as Ah*Bh + Ah*Bl + Al*Bh (fp32 accumulators): ~16 mantissa bits, i.e. fp32-class
// accuracy from the bf16 pipe.  (Plain bf16 operands stalled one low-frequency TE system in 32.)
// B: Bhi/Blo fragment arrays.  OUT: 0 = complex64, 1 = fp64 complex, 2 = fp64 complex + dinv*r (fused
// first half of the post-smoother).
constexpr int LP_NRG = 2;          // row groups (of 8 complex rows) a workgroup transforms per pass
constexpr int LP_KC = 4;           // k-groups requested together

// A-operand fragments of LP_NRG row groups starting at row m0, staged in LDS by the whole workgroup in fragment order:
//   ast[((rg*KG + kg)*2 + hl)*64 + lane] = the 8 bf16 (hi or lo) lane `lane` feeds the MFMA for row group rg, k-group kg.
// Every wave of the workgroup multiplies the same rows with its own column tiles, so the rows are fetched (and, for
// complex64 input, split into bf16 hi/lo) once per workgroup instead of once per wave.
template <int FMT>       // FMT 0: A is complex64 (split here); 1: A is pre-split (store_t32)
__device__ __forceinline__ void stage_lp_fragments(u4v* __restrict__ ast, const float2* __restrict__ Ain, int M, int NYP, int m0) {
    const int KG = (NYP + 31) >> 5;
    for (int i = threadIdx.x; i < LP_NRG * KG * 64; i += blockDim.x) {
        const int l = i & 63, kg = (i >> 6) % KG, rg = (i >> 6) / KG;
        const int lj = l & 15, g = l >> 4, part = lj & 1;
        const int arow = min(m0 + 8 * rg + (lj >> 1), M - 1);
        u4v ahu, alu;
        if (FMT) {
            const u4v* hp = reinterpret_cast<const u4v*>(reinterpret_cast<const unsigned short*>(Ain) +
                                                         (long)arow * 4 * NYP + part * NYP + 32 * kg + 8 * g);
            ahu = hp[0]; alu = hp[NYP / 4];                  // the lo planes start 2*NYP bf16 = NYP/4 x 16 B later
        } else {
            const f4v* ap = reinterpret_cast<const f4v*>(Ain + (long)arow * NYP + 32 * kg + 8 * g);   // 8 complex = 64 B
            unsigned hh[8], ll[8];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f4v v = ap[q];
                const float x0 = part ? v[1] : v[0], x1 = part ? v[3] : v[2];
                hh[2 * q] = bf16_rn(x0); ll[2 * q] = bf16_rn(x0 - bf16_to_f32(hh[2 * q]));
                hh[2 * q + 1] = bf16_rn(x1); ll[2 * q + 1] = bf16_rn(x1 - bf16_to_f32(hh[2 * q + 1]));
            }
            ahu = u4v{hh[0] | (hh[1] << 16), hh[2] | (hh[3] << 16), hh[4] | (hh[5] << 16), hh[6] | (hh[7] << 16)};
            alu = u4v{ll[0] | (ll[1] << 16), ll[2] | (ll[3] << 16), ll[4] | (ll[5] << 16), ll[6] | (ll[7] << 16)};
        }
        ast[((rg * KG + kg) * 2 + 0) * 64 + l] = ahu;
        ast[((rg * KG + kg) * 2 + 1) * 64 + l] = alu;
    }
}

template <int NTW, int OUT>
__device__ __forceinline__ void transform_lp_body(const u4v* __restrict__ ast, const u4v* __restrict__ Bhi,
                                                  const u4v* __restrict__ Blo, void* __restrict__ Cout,
                                                  const cplx* __restrict__ dinv, const cplx* __restrict__ rvec,
                                                  int M, int NYP, int m0, int t0, int lane) {
    const int NT = NYP >> 4, KG = (NYP + 31) >> 5;
    const int lj = lane & 15, g = lane >> 4;
    f4v acc[LP_NRG][NTW];
#pragma unroll
    for (int rg = 0; rg < LP_NRG; ++rg)
#pragma unroll
        for (int t = 0; t < NTW; ++t) acc[rg][t] = f4v{0, 0, 0, 0};
    for (int kc = 0; kc < KG; kc += LP_KC) {
        u4v ahs[LP_NRG][LP_KC], als[LP_NRG][LP_KC];
        u4v bh[LP_KC][NTW], bl[LP_KC][NTW];
#pragma unroll
        for (int q = 0; q < LP_KC; ++q) {
            const int kg = min(kc + q, KG - 1);
#pragma unroll
            for (int rg = 0; rg < LP_NRG; ++rg) {
                ahs[rg][q] = ast[((rg * KG + kg) * 2 + 0) * 64 + lane]; als[rg][q] = ast[((rg * KG + kg) * 2 + 1) * 64 + lane];
            }
#pragma unroll
            for (int t = 0; t < NTW; ++t) {
                const long bi = ((long)kg * NT + t0 + t) * 64 + lane;
                bh[q][t] = Bhi[bi]; bl[q][t] = Blo[bi];
            }
        }
#pragma unroll
        for (int q = 0; q < LP_KC; ++q) {
            if (kc + q < KG) {
                // v_mfma_f32_16x16x32_bf16: A[i = lane%16][k = 8*(lane/16) + t], t = 0..7 -- exactly a lane's 8 staged values
#pragma unroll
                for (int t = 0; t < NTW; ++t) {
                    const bf8v bhf = __builtin_bit_cast(bf8v, bh[q][t]), blf = __builtin_bit_cast(bf8v, bl[q][t]);
#pragma unroll
                    for (int rg = 0; rg < LP_NRG; ++rg) {
                        const bf8v ah = __builtin_bit_cast(bf8v, ahs[rg][q]), al = __builtin_bit_cast(bf8v, als[rg][q]);
                        acc[rg][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bhf, acc[rg][t], 0, 0, 0);
                        acc[rg][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, blf, acc[rg][t], 0, 0, 0);
                        acc[rg][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bhf, acc[rg][t], 0, 0, 0);
                    }
                }
            }
        }
    }
    // D rows 4g+r: (re, im) of complex rows 2g and 2g+1 of a group, column 16*(t0+t) + lj
#pragma unroll
    for (int rg = 0; rg < LP_NRG; ++rg)
#pragma unroll
        for (int t = 0; t < NTW; ++t) {
            const long col = (long)(t0 + t) * 16 + lj;
#pragma unroll
            for (int h2 = 0; h2 < 2; ++h2) {
                const int row = m0 + 8 * rg + 2 * g + h2;
                if (row < M) {
                    const long e = (long)row * NYP + col;
                    const float re = acc[rg][t][2 * h2], im = acc[rg][t][2 * h2 + 1];
                    if (OUT == 0) reinterpret_cast<float2*>(Cout)[e] = float2{re, im};
                    else if (OUT == 1) reinterpret_cast<cplx*>(Cout)[e] = cplx{(double)re, (double)im};
                    else reinterpret_cast<cplx*>(Cout)[e] = cplx{(double)re, (double)im} + dinv[e] * rvec[e];
                }
            }
        }
}

template <int OUT, int FMT>
__global__ __launch_bounds__(512) void k_transform_lp(const float2* __restrict__ A, const u4v* __restrict__ Bhi,
                                                       const u4v* __restrict__ Blo, void* __restrict__ C,
                                                       const cplx* __restrict__ dinv, const cplx* __restrict__ rvec,
                                                       int M, int NYP, int rowsPerSys, const int* __restrict__ active, int NW) {
    extern __shared__ __attribute__((aligned(16))) char smem_lp[];
    u4v* ast = reinterpret_cast<u4v*>(smem_lp);
    const int lane = threadIdx.x & 63, nw = threadIdx.x >> 6;
    const int m0 = blockIdx.x * 8 * LP_NRG;                 // one set of LP_NRG row groups per workgroup, all waves on it
    if (m0 >= M) return;
    if (active) {
        const int s0 = m0 / rowsPerSys, s1 = min(m0 + 8 * LP_NRG - 1, M - 1) / rowsPerSys;
        bool any = false;
        for (int sy = s0; sy <= s1; ++sy) any = any || active[sy];
        if (!any) return;
    }
    stage_lp_fragments<FMT>(ast, A, M, NYP, m0);
    __syncthreads();
    const int NT = NYP >> 4;
    const int base = NT / NW, extra = NT % NW;
    const int ntl = base + (nw < extra ? 1 : 0);
    const int t0 = nw * base + min(nw, extra);
    // two column tiles at a time (a wave owns more than two only on meshes wider than 256 nodes)
    for (int tt = 0; tt < ntl; tt += 2) {
        if (ntl - tt >= 2) transform_lp_body<2, OUT>(ast, Bhi, Blo, C, dinv, rvec, M, NYP, m0, t0 + tt, lane);
        else transform_lp_body<1, OUT>(ast, Bhi, Blo, C, dinv, rvec, M, NYP, m0, t0 + tt, lane);
    }
}

// ---- pre-split transform operands (fused forward path).  The bf16 hi/lo split of a transform input is the
// same for every workgroup that reads the row (7 slab workgroups in k_fdm_fwd, 7 waves in k_transform_lp), so
// the producing kernel does it once: a row of NYP complex64 values (8 B each) is stored instead as four
// planes of NYP bf16 -- hi(re), hi(im), lo(re), lo(im) -- in the same 8 NYP bytes.  A lane's MFMA A-operand
// (8 consecutive k of one part) is then one 16-byte load per hi / lo, with no conversion work.
__device__ __forceinline__ void store_t32(const Solver& k, float2* tsys, int row, int iy, float re, float im) {
    if (!k.splitT) { tsys[(long)row * k.NYP + iy] = float2{re, im}; return; }
    unsigned short* b = reinterpret_cast<unsigned short*>(tsys) + (long)row * 4 * k.NYP + iy;
    const unsigned hr = bf16_rn(re), hi = bf16_rn(im);
    b[0] = (unsigned short)hr; b[k.NYP] = (unsigned short)hi;
    b[2 * k.NYP] = (unsigned short)bf16_rn(re - bf16_to_f32(hr));
    b[3 * k.NYP] = (unsigned short)bf16_rn(im - bf16_to_f32(hi));
}

// ----------------------------------------------------------------------------------------------
// Forward half of the mixed-precision FDM stage in ONE kernel: Y = T V for a slab of 16 eigen-modes of one
// system (all rows; split-bf16 MFMA as above), then the complex64 tridiagonal solves of those 16 modes with
// the slab resident in LDS, then the coalesced write of the solved slab.  Replaces k_transform_lp<0> +
// k_thomas32 (one launch, no round trip of Y through global memory, and the serial sweeps read LDS).
// The recurrences are pre-multiplied off the serial chain by the MFMA waves:
//   a = y*ip, b = sof[row-1]*ip, c = sof[row]*ip   (ip = inverse pivot, sof = z off-diagonal)
//   down: x_row = a_row - b_row x_{row-1}  (rows 1..n);  up: x_row = x_row - c_row x_{row+1}  (rows n-1..1)
// so each serial step is one complex multiply-subtract (two dependent FMAs).
// LDS: sof[NZP] floats (padded to 128 B) + 3 slabs [NZP][16] complex64.
// ----------------------------------------------------------------------------------------------
constexpr int FW_TB = 8;           // rows requested ahead of the serial chain
constexpr int FW_PRE = 8;          // inverse pivots per thread requested at kernel entry
#ifndef HMCMT_FW_NTW
#define HMCMT_FW_NTW 2
#endif
constexpr int FW_NTW = HMCMT_FW_NTW; // column tiles per slab: 2 -> 32 modes, ceil(NT/2)*S workgroups (224 at cfg3: one round on 256 CUs)

template <int NTW>                 // column tiles (of 16 modes) per slab
__global__ __launch_bounds__(512) void k_fdm_fwd(Solver k, const float2* __restrict__ A, const u4v* __restrict__ Bhi,
                                                 const u4v* __restrict__ Blo, const float2* __restrict__ ip32,
                                                 float2* __restrict__ Y, long long* stamps = nullptr) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // 1-D grid of nslab*S workgroups.  Workgroups go to the 8 XCDs round-robin by linear id, and each XCD has
    // its own L2: all slabs of a system are placed on ONE XCD so that system's rows are fetched into one L2 once
    const int nslab = ((k.NYP >> 4) + NTW - 1) / NTW;
    int s, slab;
    if ((k.S & 7) == 0) { const int q = blockIdx.x >> 3; s = (q / nslab) * 8 + (blockIdx.x & 7); slab = q % nslab; }
    else { s = blockIdx.x / nslab; slab = blockIdx.x % nslab; }
    if (!k.active[s]) return;
    const int NYP = k.NYP, NZP = k.NZP, n = k.nz - 1;
    constexpr int SW = 16 * NTW;
#define FW_STAMP(i) if (stamps && threadIdx.x == 0) stamps[(long)blockIdx.x * 8 + (i)] = __builtin_amdgcn_s_memtime();
    FW_STAMP(0)
    // LDS: sof[NZP] (padded to 128 B), the join factors sj[SW], then three slabs sa / sb / sc.  A slab consists of
    // one region (classic sweep) or two (twisted factorisation, k.twist): region 0 holds matrix rows 0..mid in
    // order, region 1 holds rows n+1, n, .., mid+1 -- MIRRORED, so that both halves of the factorisation walk
    // their region in the same direction and one instruction stream serves the top chain (lanes 0..SW-1) and the
    // bottom chain (lanes SW..2SW-1) of the sweeping wave.  Every region has 2 FW_TB padding rows in front and
    // behind (the inner FW_TB initialised): the sweeps run whole blocks of FW_TB rows without conditionals.
    const int tw = k.twist, mid = twist_mid(n, tw);
    const int RCAP = tw ? mid + 1 : NZP, RL = RCAP + 4 * FW_TB, nreg = tw ? 2 : 1;
    float* sof = reinterpret_cast<float*>(smem);
    c32* sj = reinterpret_cast<c32*>(smem + (((long)NZP * 4 + 127) & ~127L));
    c32* sa = sj + SW + 2 * FW_TB * SW;                  // -> region 0, row 0
    c32* sb = sa + (long)nreg * RL * SW;
    c32* sc = sb + (long)nreg * RL * SW;
    auto lidx = [&](int row) { return (tw && row > mid) ? RL + (n + 1 - row) : row; };   // slab row of a matrix row
    const int mode = s >= k.nFreq;
    for (int i = threadIdx.x; i < NZP; i += blockDim.x) sof[i] = (float)k.ofz[(long)mode * NZP + i];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwave = blockDim.x >> 6;
    const int NT = NYP >> 4, KG = (NYP + 31) >> 5;
    const int t0 = slab * NTW;                         // first column tile of this slab
    // The slab's V fragments (hi and lo, all k-groups) are needed by every wave: staged once in LDS (in the space of
    // sb / sc, which are not written before the transform is over) instead of 7 times through the vector L1.
    // vst[((kg*NTW + t)*2 + hl)*64 + lane]
    u4v* vst = reinterpret_cast<u4v*>(sb - 2 * FW_TB * SW);
    const bool stageV = (size_t)KG * NTW * 2 * 64 * sizeof(u4v) <= (size_t)2 * nreg * RL * SW * sizeof(c32);
    if (stageV)
        for (int i = threadIdx.x; i < KG * NTW * 2 * 64; i += blockDim.x) {
            const int l = i & 63, hl = (i >> 6) & 1, t = (i >> 7) % NTW, kg = (i >> 7) / NTW;
            const long bi = ((long)kg * NT + min(t0 + t, NT - 1)) * 64 + l;
            vst[i] = hl ? Blo[bi] : Bhi[bi];
        }
    __syncthreads();
    const int lj = lane & 15, g = lane >> 4, part = lj & 1;
    const long so = (long)s * k.vstride;
    const float2* As = A + so;
    // this thread's inverse pivots of the pre-multiplication pass, requested now so that their latency hides
    // behind the transform
    float2 ipv[FW_PRE];
#pragma unroll
    for (int e = 0; e < FW_PRE; ++e) {
        const int idx = threadIdx.x + e * blockDim.x;
        const int row = idx / SW, c = t0 * 16 + (idx % SW);
        ipv[e] = (idx < NZP * SW && row >= 1 && row <= n && c < k.ny - 1) ? ip32[so + (long)row * NYP + c] : float2{0.f, 0.f};
    }
    // Two row groups per pass (this wave's group and the one nwave groups further down): the V fragments of a
    // k-group are loaded once for both, and the four accumulator chains keep the MFMA pipe busier than two.
    constexpr int KC = 4;                                  // k-groups requested together
    for (int m0 = wave * 8; m0 < NZP; m0 += 2 * nwave * 8) {
        const int m1 = m0 + nwave * 8;                     // second row group (may lie beyond the last row: clamped, not stored)
        const int arow[2] = {min(m0 + (lj >> 1), NZP - 1), min(m1 + (lj >> 1), NZP - 1)};
        f4v acc[2][NTW];
#pragma unroll
        for (int rg = 0; rg < 2; ++rg)
#pragma unroll
            for (int t = 0; t < NTW; ++t) acc[rg][t] = f4v{0, 0, 0, 0};
        for (int kc = 0; kc < KG; kc += KC) {
            u4v ahs[2][KC], als[2][KC];                // pre-split input (store_t32): one 16-byte load per hi / lo
            u4v bh[KC][NTW], bl[KC][NTW];
#pragma unroll
            for (int q = 0; q < KC; ++q) {
                const int kg = min(kc + q, KG - 1);
#pragma unroll
                for (int rg = 0; rg < 2; ++rg) {
                    const u4v* hp = reinterpret_cast<const u4v*>(reinterpret_cast<const unsigned short*>(As) +
                                                                 (long)arow[rg] * 4 * NYP + part * NYP + 32 * kg + 8 * g);
                    ahs[rg][q] = hp[0]; als[rg][q] = hp[NYP / 4];
                }
#pragma unroll
                for (int t = 0; t < NTW; ++t) {
                    if (stageV) {
                        bh[q][t] = vst[((kg * NTW + t) * 2 + 0) * 64 + lane]; bl[q][t] = vst[((kg * NTW + t) * 2 + 1) * 64 + lane];
                    } else {
                        const long bi = ((long)kg * NT + min(t0 + t, NT - 1)) * 64 + lane;
                        bh[q][t] = Bhi[bi]; bl[q][t] = Blo[bi];
                    }
                }
            }
#pragma unroll
            for (int q = 0; q < KC; ++q) {
                if (kc + q < KG) {
#pragma unroll
                    for (int t = 0; t < NTW; ++t) {
                        const bf8v bhf = __builtin_bit_cast(bf8v, bh[q][t]), blf = __builtin_bit_cast(bf8v, bl[q][t]);
#pragma unroll
                        for (int rg = 0; rg < 2; ++rg) {
                            const bf8v ah = __builtin_bit_cast(bf8v, ahs[rg][q]), al = __builtin_bit_cast(bf8v, als[rg][q]);
                            acc[rg][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bhf, acc[rg][t], 0, 0, 0);
                            acc[rg][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, blf, acc[rg][t], 0, 0, 0);
                            acc[rg][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bhf, acc[rg][t], 0, 0, 0);
                        }
                    }
                }
            }
        }
        // D rows 4g+r: (re, im) of complex rows 2g and 2g+1 of a group, column 16 t + lj of the slab
#pragma unroll
        for (int rg = 0; rg < 2; ++rg)
#pragma unroll
            for (int t = 0; t < NTW; ++t)
#pragma unroll
                for (int h2 = 0; h2 < 2; ++h2) {
                    const int row = (rg ? m1 : m0) + 2 * g + h2;
                    if (row < NZP) sa[lidx(row) * SW + t * 16 + lj] = c32{acc[rg][t][2 * h2], acc[rg][t][2 * h2 + 1]};
                }
    }
    __syncthreads();
    FW_STAMP(1)
    // Pre-multiply the recurrences (kept apart from the MFMA waves' epilogue on purpose: computing these products
    // right behind the last MFMA gave sporadically wrong values on gfx950).  With ip the inverse
    // pivot and o_r the off-diagonal between rows r and r+1:  a = y*ip, and the coefficient of the elimination
    // sweep p1 / of the substitution sweep p2 is  o_{r-1}*ip / o_r*ip  for a top row (swept downwards, then
    // upwards) and  o_r*ip / o_{r-1}*ip  for a bottom row (swept upwards, then downwards).
    auto premul = [&](int idx, float2 ipf) {
        const int row = idx / SW, j = idx % SW, c = t0 * 16 + j;
        const int l = lidx(row) * SW + j;
        c32 p1 = c32{0, 0}, p2 = c32{0, 0};
        if (row >= 1 && row <= n && c < k.ny - 1) {
            const c32 ip = c32{ipf.x, ipf.y};
            const c32 bb = sof[row - 1] * ip, cc = sof[row] * ip;
            sa[l] = sa[l] * ip;
            const bool bottom = tw && row > mid;
            p1 = bottom ? cc : bb; p2 = bottom ? bb : cc;
        }
        sb[l] = p1; sc[l] = p2;
    };
#pragma unroll
    for (int e = 0; e < FW_PRE; ++e) {
        const int idx = threadIdx.x + e * blockDim.x;
        if (idx < NZP * SW) premul(idx, ipv[e]);
    }
    for (int idx = threadIdx.x + FW_PRE * blockDim.x; idx < NZP * SW; idx += blockDim.x) {
        const int row = idx / SW, c = t0 * 16 + (idx % SW);
        premul(idx, (row >= 1 && row <= n && c < k.ny - 1) ? ip32[so + (long)row * NYP + c] : float2{0.f, 0.f});
    }
    // padding rows: in front of a region zeros (the substitution sweep runs into them: 0 - 0*x = 0); behind a
    // region identity rows for the elimination sweep (a = 0, p1 = -1: x stays), zero p2
    for (int idx = threadIdx.x; idx < nreg * FW_TB * SW; idx += blockDim.x) {
        const int reg = idx / (FW_TB * SW), o = idx % (FW_TB * SW);
        const c32 z = c32{0, 0};
        const long front = (long)reg * RL * SW - (long)FW_TB * SW + o;
        const int last = tw ? (reg == 0 ? mid : n + 1 - (mid + 1)) : NZP - 1;          // last initialised row of the region
        const long back = ((long)reg * RL + last + 1) * SW + o;
        sa[front] = z; sb[front] = z; sc[front] = z;
        sa[back] = z; sb[back] = c32{-1.f, 0.f}; sc[back] = z;
    }
    if (threadIdx.x < SW) {                                 // join factor 1 / (1 - c c') of the two halves (item_pivot)
        const int c = t0 * 16 + threadIdx.x;
        const float2 jf = (tw && c < k.ny - 1) ? ip32[so + c] : float2{1.f, 0.f};
        sj[threadIdx.x] = c32{jf.x, jf.y};
    }
    __syncthreads();
    FW_STAMP(2)
    if (wave == 0 && lane < nreg * SW && t0 * 16 + (lane % SW) < k.ny - 1) {
        // rows are addressed from one moving base with compile-time offsets (no clamps: the padding rows absorb the
        // blocks' overhang), so a step is 4 FMAs + 2 LDS reads + 1 LDS write
        const int half = lane / SW, col = lane % SW;
        const int last = tw ? (half == 0 ? mid : n - mid) : n;      // rows 1..last of this lane's region are real
        const int steps = tw ? mid : n;                             // both halves run the longer count (identity rows)
        c32* ra = sa + (long)half * RL * SW + col;
        const c32* rb = sb + (long)half * RL * SW + col;
        const c32* rc = sc + (long)half * RL * SW + col;
        c32 pt = c32{0, 0};
        // ---- elimination, region rows 1..steps.  Blocks of FW_TB rows, two register sets used alternately: while one
        // block is swept, the next one is on its way from LDS (no register copies between blocks; the prefetch of the
        // block behind the last one reads padding rows)
        {
            c32* pa = ra + SW;
            const c32* pb = rb + SW;
            c32 a0[FW_TB], b0[FW_TB], a1[FW_TB], b1[FW_TB];
            const int nblk = (steps + FW_TB - 1) / FW_TB;
#pragma unroll
            for (int t = 0; t < FW_TB; ++t) { a0[t] = pa[t * SW]; b0[t] = pb[t * SW]; }
            int bk = 0;
            for (; bk + 1 < nblk; bk += 2) {
#pragma unroll
                for (int t = 0; t < FW_TB; ++t) { a1[t] = pa[(FW_TB + t) * SW]; b1[t] = pb[(FW_TB + t) * SW]; }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int t = 0; t < FW_TB; ++t) { pt = cmsub(a0[t], b0[t], pt); pa[t * SW] = pt; }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int t = 0; t < FW_TB; ++t) { a0[t] = pa[(2 * FW_TB + t) * SW]; b0[t] = pb[(2 * FW_TB + t) * SW]; }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int t = 0; t < FW_TB; ++t) { pt = cmsub(a1[t], b1[t], pt); pa[(FW_TB + t) * SW] = pt; }
                __builtin_amdgcn_sched_barrier(0);
                pa += 2 * FW_TB * SW; pb += 2 * FW_TB * SW;
            }
            if (bk < nblk) {
#pragma unroll
                for (int t = 0; t < FW_TB; ++t) { pt = cmsub(a0[t], b0[t], pt); pa[t * SW] = pt; }
            }
        }
        // ---- join of the two halves: x_mid = (y'_mid - c y''_{mid+1}) J ;  x_{mid+1} = y''_{mid+1} - c' x_mid
        pt = ra[last * SW];                                   // (the identity rows left it unchanged)
        if (tw) {
            const c32 p2last = rc[last * SW];
            const float pre = pt.re, pim = pt.im;               // (plain floats: shuffling struct members kept pt in scratch)
            const float ore = __shfl_xor(pre, SW), oim = __shfl_xor(pim, SW);
            const c32 xmid = (c32{pre, pim} - p2last * c32{ore, oim}) * sj[col];       // meaningful in the top half
            const float xre = xmid.re, xim = xmid.im;
            const float mre = __shfl_xor(xre, SW), mim = __shfl_xor(xim, SW);
            const c32 xbot = c32{pre, pim} - p2last * c32{mre, mim};
            pt = half == 0 ? c32{xre, xim} : xbot;
            ra[last * SW] = pt;
        }
        // ---- substitution, region rows last-1 .. 1 (rows in front of 1: zeros in, zeros out), same scheme downwards
        {
            c32* pa = ra + (long)(last - 1) * SW;
            const c32* pc = rc + (long)(last - 1) * SW;
            c32 a0[FW_TB], b0[FW_TB], a1[FW_TB], b1[FW_TB];
            const int nblk = (steps - 1 + FW_TB - 1) / FW_TB;
#pragma unroll
            for (int t = 0; t < FW_TB; ++t) { a0[t] = pa[-t * SW]; b0[t] = pc[-t * SW]; }
            int bk = 0;
            for (; bk + 1 < nblk; bk += 2) {
#pragma unroll
                for (int t = 0; t < FW_TB; ++t) { a1[t] = pa[-(FW_TB + t) * SW]; b1[t] = pc[-(FW_TB + t) * SW]; }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int t = 0; t < FW_TB; ++t) { pt = cmsub(a0[t], b0[t], pt); pa[-t * SW] = pt; }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int t = 0; t < FW_TB; ++t) { a0[t] = pa[-(2 * FW_TB + t) * SW]; b0[t] = pc[-(2 * FW_TB + t) * SW]; }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int t = 0; t < FW_TB; ++t) { pt = cmsub(a1[t], b1[t], pt); pa[-(FW_TB + t) * SW] = pt; }
                __builtin_amdgcn_sched_barrier(0);
                pa -= 2 * FW_TB * SW; pc -= 2 * FW_TB * SW;
            }
            if (bk < nblk) {
#pragma unroll
                for (int t = 0; t < FW_TB; ++t) { pt = cmsub(a0[t], b0[t], pt); pa[-t * SW] = pt; }
            }
        }
    }
    FW_STAMP(3)
    __syncthreads();
    // solved slab -> Y, pre-split for the back transform (store_t32's format).  A thread converts 8 consecutive modes
    // of a row and writes each of the four bf16 planes with one 16-byte store instead of 32 two-byte stores
    // (16.0 -> 15.0 us per launch; the same idea in k_update_fused, through an LDS image of its tile: no gain).
    if (k.splitT) {
        constexpr int NG = SW / 8;
        unsigned short* yb = reinterpret_cast<unsigned short*>(Y + so);
        for (int idx = threadIdx.x; idx < NZP * NG; idx += blockDim.x) {
            const int row = idx / NG, j0 = (idx % NG) * 8, c0 = t0 * 16 + j0;
            if (c0 >= NYP) continue;
            const c32* src = sa + lidx(row) * SW + j0;
            u4v pl[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const c32 v0 = src[2 * q], v1 = src[2 * q + 1];
                const unsigned r0 = bf16_rn(v0.re), i0 = bf16_rn(v0.im), r1 = bf16_rn(v1.re), i1 = bf16_rn(v1.im);
                pl[0][q] = r0 | (r1 << 16);
                pl[1][q] = i0 | (i1 << 16);
                pl[2][q] = bf16_rn(v0.re - bf16_to_f32(r0)) | (bf16_rn(v1.re - bf16_to_f32(r1)) << 16);
                pl[3][q] = bf16_rn(v0.im - bf16_to_f32(i0)) | (bf16_rn(v1.im - bf16_to_f32(i1)) << 16);
            }
            unsigned short* b = yb + (long)row * 4 * NYP + c0;
#pragma unroll
            for (int pp = 0; pp < 4; ++pp) *reinterpret_cast<u4v*>(b + pp * NYP) = pl[pp];
        }
    } else {
        for (int idx = threadIdx.x; idx < NZP * SW; idx += blockDim.x) {
            const int row = idx / SW, j = idx % SW, c = t0 * 16 + j;
            if (c < NYP) { const c32 v = sa[lidx(row) * SW + j]; store_t32(k, Y + so, row, c, v.re, v.im); }
        }
    }
    FW_STAMP(4)
}

// ----------------------------------------------------------------------------------------------
// Back half of the mixed-precision FDM stage fused with BOTH Jacobi halves of the post-smoother:
//   z = V y + dinv .* r   (split-bf16 MFMA, as k_transform_lp<2>)      on a tile of 16 rows kept in LDS
//   t = z + dinv .* (r - A z), partial r't and |t|^2                   on the tile's 14 inner rows
// One workgroup = 14 consecutive interior rows of one system plus one halo row on each side (two MFMA row
// groups); the halo rows are transformed twice (by the neighbouring workgroups too: +14 % transform work) in
// exchange for one launch less per iteration and no round trip of z through global memory.
// Replaces k_transform_lp<2> + k_post on the fused path.
// ----------------------------------------------------------------------------------------------
constexpr int BP_OWN = 14;         // interior rows owned by a workgroup (tile = BP_OWN + 2 = two 8-row MFMA groups)

template <int FMT>
__global__ __launch_bounds__(512) void k_back_post(Solver k, const float2* __restrict__ Y, const u4v* __restrict__ Bhi,
                                                   const u4v* __restrict__ Blo, double* partZZ, int NW, long long* stamps) {
    extern __shared__ __attribute__((aligned(16))) char smem_[];
    const int s = blockIdx.y, bx = blockIdx.x, nwg = gridDim.x;
    const int act = k.active[s];       // tested below, after the first loads are on their way
#define BP_STAMP(i) if (stamps && threadIdx.x == 0) stamps[((long)blockIdx.y * gridDim.x + blockIdx.x) * 8 + (i)] = __builtin_amdgcn_s_memtime();
    BP_STAMP(0)
    const int bd = NW << 6;            // = blockDim.x, from the kernel argument (a scalar; the implicit-argument load is a vector load here)
    __shared__ double sh[24];
    cplx* zt = reinterpret_cast<cplx*>(smem_);             // [16][NYP]
    const int NYP = k.NYP, NZP = k.NZP;
    const int iz0 = 1 + bx * BP_OWN, iz1 = min(iz0 + BP_OWN - 1, k.nz - 1), rbase = iz0 - 1;
    const int mode = s >= k.nFreq;
    const long mo = (long)mode * k.vstride, so = (long)s * k.vstride;
    const double w = k.omega[s];
    const cplx *r = k.r + so, *di = k.dinv + so;
    cplx* t = k.t + so;
    const int nown = (iz1 - iz0 + 1) * NYP;
    // Every phase below is a short dependent chain (global load -> LDS -> barrier -> MFMA -> LDS -> barrier -> stencil),
    // so loads are issued as early as their addresses are known and unconditionally (clamped indices): inside
    // `if (row < NZP)` / `if (interior)` the compiler keeps each load next to its use and the phase costs one
    // memory round trip per element instead of one per batch (s_memtime stamps: epilogue 3.5 -> us, stencil 4.7 -> us).
    constexpr int SU = 4;               // stencil elements per thread and batch
    struct Sten { double dk, dm, cy0, cy1, cz0, cz1; cplx rv, dv; int e, iy; };
    // (uniform base + 32-bit lane offset: one address register per element instead of two per load)
    const double *dKm = k.dK + mo, *dMm = k.dM + mo, *cYm = k.cY + mo, *cZm = k.cZ + mo, *cZu = k.cZ + mo - NYP;
    const float rNYP = 1.0f / (float)NYP;
    auto ld_st = [&](int i, Sten& q) {
        const int ic = min(i, nown - 1), lr = (int)(((float)ic + 0.5f) * rNYP);      // ic / NYP (exact: ic < 4096)
        q.iy = ic - lr * NYP;
        q.e = (iz0 + lr) * NYP + q.iy;
        const unsigned e = (unsigned)q.e;
        q.dk = dKm[e]; q.dm = dMm[e];
        q.cy0 = cYm[e]; q.cy1 = cYm[e - 1u];
        q.cz0 = cZm[e]; q.cz1 = cZu[e];
        q.rv = r[e]; q.dv = di[e];
    };
    Sten st[SU], st2[SU];           // 14 NYP <= 8 x blockDim elements: two batches per thread
    {
        // both row groups of the tile in one pass over k: a wave's V fragments are loaded once for the two groups
        const int lane = threadIdx.x & 63, nw = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // scalar: tile ranges uniform
        const int NT = NYP >> 4, KG = (NYP + 31) >> 5;
        const int base = NT == 2 * NW ? 2 : 1, extra = NT - base * NW;      // NW = ceil(NT / 2) waves: no division
        const int ntl = base + (nw < extra ? 1 : 0);
        const int t0 = nw * base + min(nw, extra);
        const int lj = lane & 15, g = lane >> 4;
        const float2* Ys = Y + so;
        constexpr int KC = 8;
        // epilogue operands dinv, r of the wave's 2 x 2 x 2 accumulator elements per lane
        cplx dv[2][2][2], rv[2][2][2];
        auto ld_dr = [&]() {
#pragma unroll
            for (int rg = 0; rg < 2; ++rg)
#pragma unroll
                for (int h2 = 0; h2 < 2; ++h2) {
                    // row = rbase + 8 rg + h2 (uniform) + 2 g (lane), clamped to the mesh (those elements are zeroed below)
                    const int ru = min(rbase + 8 * rg + h2, NZP - 1);
                    const unsigned lo = (unsigned)(min(2 * g, NZP - 1 - ru) * NYP + lj);
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
                        const long ub = (long)ru * NYP + min(t0 + t, NT - 1) * 16;
                        dv[rg][t][h2] = (di + ub)[lo]; rv[rg][t][h2] = (r + ub)[lo];
                    }
                }
        };
        // The tile's 16 rows of y are the A-operand of every wave: staged once in LDS in fragment order
        // (ast[((rg*KG + kg)*2 + hl)*64 + lane] = what lane `lane` feeds the MFMA for row group rg, k-group kg)
        // instead of 7 times through the vector L1.  nast = 256 KG <= 4 x blockDim elements: one batch of 4 per thread.
        // Wave kg stages k-group kg (the launcher starts NW = KG waves): its 4 fragments (row group, hi/lo).
        u4v* ast = reinterpret_cast<u4v*>(zt + (long)16 * NYP);
        constexpr int SG = 4;
        u4v tmp[SG];
        {
            const int llj = lane & 15, lg = lane >> 4;
            const char* yb = reinterpret_cast<const char*>(Ys) + (long)(32 * nw) * 2;       // uniform part (k-group)
#pragma unroll
            for (int u = 0; u < SG; ++u) {
                const int rg = u >> 1, hl = u & 1;
                const int arow = min(rbase + 8 * rg + (llj >> 1), NZP - 1);
                const unsigned off = (unsigned)(((arow * 4 + (llj & 1) + 2 * hl) * NYP + 8 * lg) * 2);   // bytes (bf16 planes re, im, re_lo, im_lo)
                tmp[u] = *reinterpret_cast<const u4v*>(yb + off);
            }
        }
        // all V fragments of the wave's (at most) two column tiles: KG <= KC k-groups (NYP <= 256, checked by the
        // launcher).  Issued behind the staging loads and in k order: the staging barrier does not wait for them and
        // the MFMAs of k-group q start when fragment q has arrived.
        // Order matters twice.  (1) The vector memory path of a CU serves the requests of all its waves in order, so
        // one wave's V loads would sit in front of another wave's staging loads and the staging barrier would wait for
        // (nearly) all of V: a barrier makes sure every wave has issued its staging loads first.  (2) Every workgroup
        // streams the same V; the 32 workgroups of an XCD (one tile index, 32 systems) start at different k-groups so
        // that they are on different L2 channels instead of all asking for the same lines at once.
        __syncthreads();
        u4v bh[KC][2], bl[KC][2];
        const int rot = s % KG;
        {
            const unsigned loff = (unsigned)lane * 16u;
            const int tl0 = min(t0, NT - 1), tl1 = min(t0 + 1, NT - 1);
            const char* ph = reinterpret_cast<const char*>(Bhi + (long)tl0 * 64) + loff;
            const char* pl = reinterpret_cast<const char*>(Blo + (long)tl0 * 64) + loff;
            const long d1 = (long)(tl1 - tl0) * 1024, stride = (long)NT * 1024;   // bytes: second tile, next k-group
#pragma unroll
            for (int q = 0; q < KC; ++q) {
                if (q < KG) {
                    const int kg = rot + q - (rot + q >= KG ? KG : 0);
                    const char *qh = ph + kg * stride, *ql = pl + kg * stride;
                    bh[q][0] = *reinterpret_cast<const u4v*>(qh); bh[q][1] = *reinterpret_cast<const u4v*>(qh + d1);
                    bl[q][0] = *reinterpret_cast<const u4v*>(ql); bl[q][1] = *reinterpret_cast<const u4v*>(ql + d1);
                }
            }
        }
        BP_STAMP(7)
        if (!act) return;                                  // (uniform; nothing has been stored yet)
#pragma unroll
        for (int u = 0; u < SG; ++u) ast[(((u >> 1) * KG + nw) * 2 + (u & 1)) * 64 + lane] = tmp[u];
        __syncthreads();
        BP_STAMP(1)
        ld_dr();                                             // in flight during the MFMA loop
        f4v acc[2][2];
#pragma unroll
        for (int rg = 0; rg < 2; ++rg)
#pragma unroll
            for (int t = 0; t < 2; ++t) acc[rg][t] = f4v{0, 0, 0, 0};
#pragma unroll
        for (int q = 0; q < KC; ++q) {
            if (q < KG) {
                const int kg = rot + q - (rot + q >= KG ? KG : 0);
                bf8v ah[2], al[2];
#pragma unroll
                for (int rg = 0; rg < 2; ++rg) {
                    ah[rg] = __builtin_bit_cast(bf8v, ast[((rg * KG + kg) * 2 + 0) * 64 + lane]);
                    al[rg] = __builtin_bit_cast(bf8v, ast[((rg * KG + kg) * 2 + 1) * 64 + lane]);
                }
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const bf8v bhf = __builtin_bit_cast(bf8v, bh[q][t]), blf = __builtin_bit_cast(bf8v, bl[q][t]);
#pragma unroll
                    for (int rg = 0; rg < 2; ++rg) {
                        acc[rg][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[rg], bhf, acc[rg][t], 0, 0, 0);
                        acc[rg][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[rg], blf, acc[rg][t], 0, 0, 0);
                        acc[rg][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[rg], bhf, acc[rg][t], 0, 0, 0);
                    }
                }
            }
        }
        BP_STAMP(2)
        // coefficients of the first stencil batch: in flight during the epilogue and the barrier
#pragma unroll
        for (int u = 0; u < SU; ++u) ld_st(threadIdx.x + u * bd, st[u]);
        // z = V y + dinv .* r into the tile (rows beyond the mesh: zero)
#pragma unroll
        for (int rg = 0; rg < 2; ++rg)
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                if (t < ntl) {
                    const int col = (t0 + t) * 16 + lj;
#pragma unroll
                    for (int h2 = 0; h2 < 2; ++h2) {
                        const int lr = 8 * rg + 2 * g + h2;
                        cplx val = cplx{(double)acc[rg][t][2 * h2], (double)acc[rg][t][2 * h2 + 1]} + dv[rg][t][h2] * rv[rg][t][h2];
                        if (rbase + lr >= NZP) val = cplx{0.0, 0.0};
                        zt[(long)lr * NYP + col] = val;
                    }
                }
            }
#pragma unroll
        for (int u = 0; u < SU; ++u) ld_st(threadIdx.x + (SU + u) * bd, st2[u]);
    }
    BP_STAMP(3)
    __syncthreads();
    BP_STAMP(4)
    double ar = 0, ai = 0, zz = 0;
    auto stencil = [&](int i, const Sten& q) {
        if (i < nown) {
            const int l = q.e - (rbase * NYP);                 // tile-local index: tile row 0 = mesh row rbase
            const cplx c = zt[l];
            cplx acc = cplx{q.dk * c.re - w * q.dm * c.im, q.dk * c.im + w * q.dm * c.re};
            acc += q.cy0 * zt[l + 1];
            acc += q.cy1 * zt[l - 1];
            acc += q.cz0 * zt[l + NYP];
            acc += q.cz1 * zt[l - NYP];
            cplx out = c + q.dv * (q.rv - acc);
            if (q.iy < 1 || q.iy > k.ny - 1) out = cplx{0, 0};
            ar += q.rv.re * out.re - q.rv.im * out.im;
            ai += q.rv.re * out.im + q.rv.im * out.re;
            zz += cabs2(out);
            t[q.e] = out;
        }
    };
#pragma unroll
    for (int u = 0; u < SU; ++u) stencil(threadIdx.x + u * bd, st[u]);
#pragma unroll
    for (int u = 0; u < SU; ++u) stencil(threadIdx.x + (SU + u) * bd, st2[u]);
    // the two boundary rows of t stay zero (the stencil kernels read them as halo rows)
    if (bx == 0) for (int i = threadIdx.x; i < NYP; i += bd) t[i] = cplx{0, 0};
    if (iz1 == k.nz - 1) for (int i = threadIdx.x; i < NYP; i += bd) t[(long)k.nz * NYP + i] = cplx{0, 0};
    BP_STAMP(5)
    block_sum3_8(ar, ai, zz, sh, NW);
    if (threadIdx.x == 0) {
        k.partA[(long)s * MAXNB + bx] = cplx{ar, ai};
        partZZ[(long)s * MAXNB + bx] = zz;
    }
    // the consumers add up k.NB partial sums per system: clear the slots this launch does not use
    if (bx == 0)
        for (int b = nwg + threadIdx.x; b < k.NB; b += bd) {
            k.partA[(long)s * MAXNB + b] = cplx{0, 0};
            partZZ[(long)s * MAXNB + b] = 0.0;
        }
    BP_STAMP(6)
}

// pre-split planes -> complex64 (hi + lo), tests only
__global__ void k_unsplit(Solver k, const float2* __restrict__ src, float2* __restrict__ dst) {
    const long n = (long)k.S * k.vstride;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long)gridDim.x * blockDim.x) {
        const long row = e / k.NYP;
        const int iy = (int)(e - row * k.NYP);
        const unsigned short* b = reinterpret_cast<const unsigned short*>(src) + row * 4 * k.NYP + iy;
        dst[e] = float2{bf16_to_f32(b[0]) + bf16_to_f32(b[2 * k.NYP]), bf16_to_f32(b[k.NYP]) + bf16_to_f32(b[3 * k.NYP])};
    }
}

// complex64 copy of a vector (plain FDM: the transform input is r itself)
__global__ __launch_bounds__(VBLOCK) void k_to_c64(Solver k, const cplx* src) {
    const int s = blockIdx.y;
    if (!k.active[s]) return;
    const long so = (long)s * k.vstride;
    const long e0 = (long)blockIdx.x * k.chunk, e1 = min(e0 + k.chunk, k.vstride);
    for (long e = e0 + threadIdx.x; e < e1; e += VBLOCK) {
        const int iz = (int)(e / k.NYP), iy = (int)(e - (long)iz * k.NYP);
        store_t32(k, k.t32 + so, iz, iy, (float)src[so + e].re, (float)src[so + e].im);
    }
}

// t = r - A (dinv .* r), written as complex64 for the mixed-precision transform
__global__ __launch_bounds__(VBLOCK) void k_pre_c64(Solver k) {
    const int s = blockIdx.y;
    if (!k.active[s]) return;
    const int mode = s >= k.nFreq;
    const long mo = (long)mode * k.vstride, so = (long)s * k.vstride;
    const double w = k.omega[s];
    const cplx *r = k.r + so, *di = k.dinv + so;
    float2* t = k.t32 + so;
    const long e0 = (long)blockIdx.x * k.chunk, e1 = min(e0 + k.chunk, k.vstride);
    for (long e = e0 + threadIdx.x; e < e1; e += VBLOCK) {
        const int iz = (int)(e / k.NYP), iy = (int)(e - (long)iz * k.NYP);
        cplx out = cplx{0, 0};
        if (iz >= 1 && iz <= k.nz - 1 && iy >= 1 && iy <= k.ny - 1) {
            const cplx c = di[e] * r[e];
            const double dk = k.dK[mo + e], dm = w * k.dM[mo + e];
            cplx acc = cplx{dk * c.re - dm * c.im, dk * c.im + dm * c.re};
            acc += k.cY[mo + e] * (di[e + 1] * r[e + 1]);
            acc += k.cY[mo + e - 1] * (di[e - 1] * r[e - 1]);
            acc += k.cZ[mo + e] * (di[e + k.NYP] * r[e + k.NYP]);
            acc += k.cZ[mo + e - k.NYP] * (di[e - k.NYP] * r[e - k.NYP]);
            out = r[e] - acc;
        }
        store_t32(k, t, iz, iy, (float)out.re, (float)out.im);
    }
}


// ----------------------------------------------------------------------------------------------
// Fused COCG iteration of the default path (Jacobi/FDM/Jacobi preconditioner, mixed precision):
//   k_spmv_fused   : scalar bookkeeping (convergence test on the error estimate, beta), p = z + beta p
//                    recomputed on each node's 5-point halo, q = A p, partial p'q
//   k_update_fused : alpha, x += alpha p, r' = r - alpha q recomputed on the halo, Jacobi pre-smoothing
//                    t = r' - A (dinv .* r') written as complex64 for the transform, partial |x|^2
// Every block of a system reduces that system's partial sums itself (same order -> same value), so no
// separate scalar kernel and no grid synchronisation is needed; p and r are double-buffered because
// blocks read their neighbours' old values while writing new ones.  Block 0 of each system owns the
// per-system records (rho by parity, iteration count, error estimate, active flag, active counter).
// ----------------------------------------------------------------------------------------------
// Both kernels work on tiles of RT interior rows of one system: the tile plus one halo row above and below
// is staged in LDS (dynamic, (RT+2)*NYP complex [+ RT*NYP]), so every global value is read once.
//
// k_spmv_fused is a chain of short phases (scalars -> stage tile -> barrier -> stencil -> reduce), each a memory round
// trip long: its loads are issued as early as their addresses are known, in batches of SB elements per thread,
// unconditionally (clamped indices) and apart from their use -- the per-system partial sums, rho and the first staging
// batch go out together before the active flag is even tested, the stencil coefficients of the first batch before
// the staging barrier (11.0 -> 10.3 us; the same treatment of k_update_fused, which moves twice the bytes and sits
// at 5 TB/s, changed nothing, and neither did computing its dinv from dK, dM instead of loading it).
constexpr int SB = 4;                  // elements per thread and batch
constexpr int STALL_IT = 30;           // mixed-precision stagnation watch: iterations allowed per 10-fold drop of the error estimate
struct StenCo { double dk, dm, cy0, cy1, cz0, cz1; };

__device__ __forceinline__ int div_small(int i, float rcp) { return (int)(((float)i + 0.5f) * rcp); }   // i / n for i < 2^20, rcp = 1/n

__global__ __launch_bounds__(VBLOCK) void k_spmv_fused(Solver k, const double* partZZ, const cplx* pin_, cplx* pout, int it, int maxit) {
    const int s = blockIdx.y;
    extern __shared__ __attribute__((aligned(16))) char smem_[];
    cplx* pn = reinterpret_cast<cplx*>(smem_);            // [(RT+2)][NYP]
    __shared__ double sh[8];
    const bool first = it == 1;
    const int act = k.active[s];
    const int ln = threadIdx.x & 63;
    // the system's partial sums (lane b fetches partial b), rho of the previous iteration
    const cplx paL = ln < k.NB ? k.partA[(long)s * MAXNB + ln] : cplx{0, 0};
    const double pzL = ln < k.NB ? partZZ[(long)s * MAXNB + ln] : 0.0;
    const double pbL = ln < k.NTR ? k.partB[(long)s * MAXNB + ln] : 0.0;
    const cplx rhoPrev = k.rho2[(long)((it - 1) & 1) * k.S + s];
    const int mode = s >= k.nFreq;
    const long mo = (long)mode * k.vstride, so = (long)s * k.vstride;
    const double w = k.omega[s];
    const cplx *z = k.z + so, *pi = pin_ + so;
    cplx *po = pout + so, *q = k.q + so;
    const int NYP = k.NYP, iz0 = 1 + blockIdx.x * k.RT, iz1 = min(iz0 + k.RT - 1, k.nz - 1);   // own rows iz0..iz1
    const float rNYP = 1.0f / (float)NYP;
    // rows iz0-1 .. iz1+1 of the new direction (z, p vanish on boundary / pad nodes: no masking needed)
    const int nrows = iz1 - iz0 + 3, ntot = nrows * NYP, ebase = (iz0 - 1) * NYP;
    cplx zv[SB], pv[SB];
    auto ld_stage = [&](int i0) {
#pragma unroll
        for (int u = 0; u < SB; ++u) {
            const unsigned e = (unsigned)(ebase + min(i0 + u * VBLOCK, ntot - 1));
            zv[u] = z[e];
            pv[u] = first ? cplx{0, 0} : pi[e];
        }
    };
    ld_stage(threadIdx.x);
    if (!act) return;
    const cplx rz = cplx{wave_sum(paL.re), wave_sum(paL.im)};
    const double zz = wave_sum(pzL), xx = wave_sum(pbL);
    bool on = true;
    int st = 0;
    if (first) { if (zz == 0.0) on = false; }
    else if (zz <= k.tol2 * xx) on = false;
    else if (it - 1 >= maxit) { on = false; st = HMCMT_ENOCONV; }
    if (!(isfinite(rz.re) && isfinite(rz.im) && isfinite(zz) && isfinite(xx))) { on = false; st = HMCMT_EBREAKDOWN; }
    const cplx be = first ? cplx{0, 0} : rz / rhoPrev;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        k.rho2[(long)(it & 1) * k.S + s] = rz;
        k.iters[s] = it - 1;
        const double est = first ? (zz == 0.0 ? 0.0 : 1.0) : sqrt(zz / xx);
        k.errEst[s] = est;
        if (st) k.status[s] = st;
        // stagnation watch (the host restarts the stragglers with the fp64 preconditioner when it fires)
        if (first || est < 0.1 * k.errRef[s]) { k.errRef[s] = est; k.errRefIt[s] = it; }
        else if (on && it - k.errRefIt[s] > k.stallIt) *k.stallHost = 1;
    }
    if (!on) {
        // every block of this system takes the same decision; block 0 records it (a block that starts late and
        // already sees the cleared flag returns just the same)
        if (blockIdx.x == 0 && threadIdx.x == 0) { k.active[s] = 0; if (atomicSub(k.nactive, 1) == 1) *k.nactHost = 0; }
        return;
    }
    if (k.cntActive && blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(k.cntActive, 1ull);   // (roofline accounting only)
    for (int i0 = threadIdx.x; i0 < ntot; i0 += SB * VBLOCK) {
        if (i0 != (int)threadIdx.x) ld_stage(i0);
#pragma unroll
        for (int u = 0; u < SB; ++u) {
            const int i = i0 + u * VBLOCK;
            if (i < ntot) {
                const cplx v = first ? zv[u] : zv[u] + be * pv[u];
                pn[i] = v;
                if (i >= NYP && i < ntot - NYP) po[ebase + i] = v;
            }
        }
    }
    const int nown = (iz1 - iz0 + 1) * NYP, obase = iz0 * NYP;
    const double *dKm = k.dK + mo, *dMm = k.dM + mo, *cYm = k.cY + mo, *cZm = k.cZ + mo, *cZu = k.cZ + mo - NYP;
    StenCo co[SB];
    auto ld_co = [&](int i0) {
#pragma unroll
        for (int u = 0; u < SB; ++u) {
            const unsigned e = (unsigned)(obase + min(i0 + u * VBLOCK, nown - 1));
            co[u].dk = dKm[e]; co[u].dm = dMm[e];
            co[u].cy0 = cYm[e]; co[u].cy1 = cYm[e - 1u];
            co[u].cz0 = cZm[e]; co[u].cz1 = cZu[e];
        }
    };
    ld_co(threadIdx.x);
    __syncthreads();
    double ar = 0, ai = 0;
    for (int i0 = threadIdx.x; i0 < nown; i0 += SB * VBLOCK) {
        if (i0 != (int)threadIdx.x) ld_co(i0);
#pragma unroll
        for (int u = 0; u < SB; ++u) {
            const int i = i0 + u * VBLOCK;
            const int iy = i - div_small(i, rNYP) * NYP;
            if (i < nown && iy >= 1 && iy <= k.ny - 1) {
                const int l = i + NYP;
                const cplx pc = pn[l];
                const double dm = w * co[u].dm;
                cplx acc = cplx{co[u].dk * pc.re - dm * pc.im, co[u].dk * pc.im + dm * pc.re};
                acc += co[u].cy0 * pn[l + 1];
                acc += co[u].cy1 * pn[l - 1];
                acc += co[u].cz0 * pn[l + NYP];
                acc += co[u].cz1 * pn[l - NYP];
                q[obase + i] = acc;
                ar += pc.re * acc.re - pc.im * acc.im;
                ai += pc.re * acc.im + pc.im * acc.re;
            }
        }
    }
    block_sum2(ar, ai, sh);
    if (threadIdx.x == 0) k.partPQ[(long)s * MAXNB + blockIdx.x] = cplx{ar, ai};
}

__global__ __launch_bounds__(VBLOCK) void k_update_fused(Solver k, const cplx* pcur, const cplx* rin, cplx* rout, int it) {
    const int s = blockIdx.y;
    if (!k.active[s]) return;
    extern __shared__ __attribute__((aligned(16))) char smem_[];
    const int NYP = k.NYP, iz0 = 1 + blockIdx.x * k.RT, iz1 = min(iz0 + k.RT - 1, k.nz - 1);
    const int nrows = iz1 - iz0 + 3;
    cplx* cs = reinterpret_cast<cplx*>(smem_);            // [(RT+2)][NYP]  dinv .* r'
    cplx* rs = cs + (long)(k.RT + 2) * NYP;               // [RT][NYP]      r' of the own rows
    __shared__ double sh[8];
    const cplx pq = total_part(k.partPQ + (long)s * MAXNB, k.NTR);
    const cplx al = k.rho2[(long)(it & 1) * k.S + s] / pq;
    const int mode = s >= k.nFreq;
    const long mo = (long)mode * k.vstride, so = (long)s * k.vstride;
    const double w = k.omega[s];
    const cplx *p = pcur + so, *q = k.q + so, *ri = rin + so, *di = k.dinv + so;
    cplx *x = k.x + so, *ro = rout + so;
    float2* t = k.t32 + so;
    double xx = 0, dummy = 0;
    // r' = r - alpha q and dinv .* r' on rows iz0-1 .. iz1+1 (r, q, dinv vanish outside the interior: no masking)
    for (int i = threadIdx.x; i < nrows * NYP; i += VBLOCK) {
        const int lr = i / NYP, iy = i - lr * NYP;
        const long e = (long)(iz0 - 1 + lr) * NYP + iy;
        const cplx rn = ri[e] - al * q[e];
        cs[i] = di[e] * rn;
        if (lr >= 1 && lr <= nrows - 2) {
            rs[i - NYP] = rn;
            ro[e] = rn;
            const cplx xv = x[e] + al * p[e];             // p vanishes outside the interior
            x[e] = xv;
            xx += cabs2(xv);
        }
    }
    __syncthreads();
    const int nown = (iz1 - iz0 + 1) * NYP;
    for (int i = threadIdx.x; i < nown; i += VBLOCK) {
        const int lr = i / NYP, iy = i - lr * NYP;
        const long e = (long)(iz0 + lr) * NYP + iy;
        cplx out = cplx{0, 0};
        if (iy >= 1 && iy <= k.ny - 1) {
            const int l = (lr + 1) * NYP + iy;
            const cplx c = cs[l];
            const double dk = k.dK[mo + e], dm = w * k.dM[mo + e];
            cplx acc = cplx{dk * c.re - dm * c.im, dk * c.im + dm * c.re};
            acc += k.cY[mo + e] * cs[l + 1];
            acc += k.cY[mo + e - 1] * cs[l - 1];
            acc += k.cZ[mo + e] * cs[l + NYP];
            acc += k.cZ[mo + e - NYP] * cs[l - NYP];
            out = rs[i] - acc;
        }
        store_t32(k, t, iz0 + lr, iy, (float)out.re, (float)out.im);
    }
    block_sum2(xx, dummy, sh);
    if (threadIdx.x == 0) {
        k.partB[(long)s * MAXNB + blockIdx.x] = xx;
        if (blockIdx.x == 0) k.alphaBeta[s] = al;
    }
}

// warm start: r <- r - A x over interior nodes, x including whatever sits on its boundary nodes
// (forward: Dirichlet values, so with r = 0 on entry this is the reference's rhs -Aio*bc minus Aii*x0)
// sysOn != nullptr: workgroup (0,0) also does k_solve_begin's bookkeeping for the solve that follows (one launch less
// on the critical path in front of each solve)
__global__ __launch_bounds__(VBLOCK) void k_resid0(Solver k, cplx* x, int zero_r, const int* __restrict__ sysOn) {
    const int s = blockIdx.y;
    if (sysOn && blockIdx.x == 0 && blockIdx.y == 0) {
        for (int t = threadIdx.x; t < k.S * MAXNB; t += VBLOCK) k.partB[t] = 0.0;
        for (int t = threadIdx.x; t < k.S; t += VBLOCK) { k.active[t] = sysOn[t]; k.iters[t] = 0; k.status[t] = 0; }
        if (threadIdx.x == 0) {
            int n = 0;
            for (int q = 0; q < k.S; ++q) n += sysOn[q];
            *k.nactive = n;
            *k.nactHost = n;
        }
    }
    const int mode = s >= k.nFreq;
    const long mo = (long)mode * k.vstride, so = (long)s * k.vstride;
    const double w = k.omega[s];
    const cplx* u = x + so;
    cplx* r = k.r + so;
    const long e0 = (long)blockIdx.x * k.chunk, e1 = min(e0 + k.chunk, k.vstride);
    for (long e = e0 + threadIdx.x; e < e1; e += VBLOCK) {
        const int iz = (int)(e / k.NYP), iy = (int)(e - (long)iz * k.NYP);
        cplx out = cplx{0, 0};
        if (iz >= 1 && iz <= k.nz - 1 && iy >= 1 && iy <= k.ny - 1)
            out = (zero_r ? cplx{0, 0} : r[e]) - stencil_at(k, u, mo, e, w);
        r[e] = out;
    }
}

// k_resid0 + k_pre_c64 in one launch (the start of a solve on the default path): on a tile of RT rows
//   r = b - A x0        on the tile's rows and one halo row on each side (x0 staged with two halo rows)
//   t = r - A (dinv r)  on the tile's rows, written in the transform's input format
// so the residual is not re-read by a second kernel and one launch disappears in front of each solve.  The residual is
// written to a SECOND buffer (rout): the halo rows' b are read from rin while the neighbouring workgroups write theirs.
// Workgroup (0,0) also does k_solve_begin's bookkeeping.
__global__ __launch_bounds__(VBLOCK) void k_resid_pre(Solver k, const cplx* x, const cplx* rin, cplx* rout, int zero_r,
                                                      const int* __restrict__ sysOn) {
    const int s = blockIdx.y;
    if (blockIdx.x == 0 && blockIdx.y == 0) {
        for (int t = threadIdx.x; t < k.S * MAXNB; t += VBLOCK) k.partB[t] = 0.0;
        for (int t = threadIdx.x; t < k.S; t += VBLOCK) { k.active[t] = sysOn[t]; k.iters[t] = 0; k.status[t] = 0; }
        if (threadIdx.x == 0) {
            int n = 0;
            for (int q = 0; q < k.S; ++q) n += sysOn[q];
            *k.nactive = n;
            *k.nactHost = n;
        }
    }
    extern __shared__ __attribute__((aligned(16))) char smem_[];
    const int NYP = k.NYP, iz0 = 1 + blockIdx.x * k.RT, iz1 = min(iz0 + k.RT - 1, k.nz - 1), nown = iz1 - iz0 + 1;
    cplx* xs = reinterpret_cast<cplx*>(smem_);            // [(RT+4)][NYP]  x0 rows iz0-2 .. iz1+2; later dinv .* r (rows iz0-1 ..)
    cplx* rs = xs + (long)(k.RT + 4) * NYP;               // [(RT+2)][NYP]  r rows iz0-1 .. iz1+1
    const int mode = s >= k.nFreq;
    const long mo = (long)mode * k.vstride, so = (long)s * k.vstride;
    const double w = k.omega[s];
    const float rNYP = 1.0f / (float)NYP;
    const cplx* u = x + so;
    for (int i = threadIdx.x; i < (nown + 4) * NYP; i += VBLOCK) {
        const int lr = div_small(i, rNYP), g = iz0 - 2 + lr;
        xs[i] = (g >= 0 && g <= k.nz) ? u[(long)g * NYP + (i - lr * NYP)] : cplx{0, 0};
    }
    __syncthreads();
    for (int i = threadIdx.x; i < (nown + 2) * NYP; i += VBLOCK) {
        const int lr = div_small(i, rNYP), iy = i - lr * NYP, g = iz0 - 1 + lr;
        const long e = (long)g * NYP + iy;
        cplx out = cplx{0, 0};
        if (g >= 1 && g <= k.nz - 1 && iy >= 1 && iy <= k.ny - 1) {
            const int l = i + NYP;                         // the same node in xs (one more halo row in front)
            const cplx c = xs[l];
            const double dk = k.dK[mo + e], dm = w * k.dM[mo + e];
            cplx acc = cplx{dk * c.re - dm * c.im, dk * c.im + dm * c.re};
            acc += k.cY[mo + e] * xs[l + 1];
            acc += k.cY[mo + e - 1] * xs[l - 1];
            acc += k.cZ[mo + e] * xs[l + NYP];
            acc += k.cZ[mo + e - NYP] * xs[l - NYP];
            out = (zero_r ? cplx{0, 0} : rin[so + e]) - acc;
        }
        rs[i] = out;
        if (lr >= 1 && lr <= nown) rout[so + e] = out;
    }
    // the two boundary rows of r and of t (zeros)
    if (blockIdx.x == 0 || iz1 == k.nz - 1) {
        const int row = blockIdx.x == 0 ? 0 : k.nz;
        for (int iy = threadIdx.x; iy < NYP; iy += VBLOCK) {
            rout[so + (long)row * NYP + iy] = cplx{0, 0};
            store_t32(k, k.t32 + so, row, iy, 0.f, 0.f);
        }
        if (blockIdx.x == 0 && iz1 == k.nz - 1)            // (a single tile: both rows)
            for (int iy = threadIdx.x; iy < NYP; iy += VBLOCK) {
                rout[so + (long)k.nz * NYP + iy] = cplx{0, 0};
                store_t32(k, k.t32 + so, k.nz, iy, 0.f, 0.f);
            }
    }
    __syncthreads();
    const cplx* di = k.dinv + so;
    for (int i = threadIdx.x; i < (nown + 2) * NYP; i += VBLOCK) xs[i] = di[(long)(iz0 - 1) * NYP + i] * rs[i];
    __syncthreads();
    for (int i = threadIdx.x; i < nown * NYP; i += VBLOCK) {
        const int lr = div_small(i, rNYP), iy = i - lr * NYP;
        const long e = (long)(iz0 + lr) * NYP + iy;
        cplx out = cplx{0, 0};
        if (iy >= 1 && iy <= k.ny - 1) {
            const int l = i + NYP;
            const cplx c = xs[l];
            const double dk = k.dK[mo + e], dm = w * k.dM[mo + e];
            cplx acc = cplx{dk * c.re - dm * c.im, dk * c.im + dm * c.re};
            acc += k.cY[mo + e] * xs[l + 1];
            acc += k.cY[mo + e - 1] * xs[l - 1];
            acc += k.cZ[mo + e] * xs[l + NYP];
            acc += k.cZ[mo + e - NYP] * xs[l - NYP];
            out = rs[l] - acc;
        }
        store_t32(k, k.t32 + so, iz0 + lr, iy, (float)out.re, (float)out.im);
    }
}

// start of a solve: every requested system active, records cleared (one launch instead of five copies/memsets)
__global__ void k_solve_begin(Solver k, const int* __restrict__ sysOn) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < k.S * MAXNB) k.partB[t] = 0.0;
    if (t < k.S) { k.active[t] = sysOn[t]; k.iters[t] = 0; k.status[t] = 0; }
    if (t == 0) {
        int n = 0;
        for (int s = 0; s < k.S; ++s) n += sysOn[s];
        *k.nactive = n;
        *k.nactHost = n;
    }
}

// end of a solve: per-system records of this solve kind into the packed read-back buffer
// rec = [2 kinds][S] iters (int) | [2][S] status (int) | [2][S] error estimate (double)
__global__ void k_solve_end(Solver k, int kind, int* __restrict__ recI, double* __restrict__ recE) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= k.S) return;
    recI[kind * k.S + s] = k.iters[s];
    recI[(2 + kind) * k.S + s] = k.status[s];
    recE[kind * k.S + s] = k.errEst[s];
}

// ---- initial guess extrapolated along the model path (options.warm_start == 2) ----
// The fields are smooth functions of the model, and a leapfrog trajectory moves the model along an almost
// straight line at almost constant speed.  With the last three models on that line at "times"
// tau = -1-gamma, -1, 0 (unit = the last step; gamma, alpha = projections of the previous / the new step on
// the last one) the guess is the Lagrange extrapolation of the last three fields to tau = alpha:
//     x0 = w2 x_k + w1 x_{k-1} + w0 x_{k-2}      (uniform steps: 3, -3, 1)
// when the three steps are nearly collinear, else the linear one  x0 = x_k + alpha (x_k - x_{k-1})
// (e.g. across a momentum refresh).  State per solve kind, all on the device (no host round trip):
// hist = [m_k | m_{k-1} | ... | m_{k-NP+1}], ext = {w_k, ..., w_{k-NP+1}, keep, count}; every further step that is
// nearly collinear with the last one and of comparable length adds a point (and an order) to the Lagrange
// extrapolation, up to EXT_NP fields.  A repeated model (getHamiltonian after the last leapfrog step) keeps the
// history untouched.
constexpr int EXT_NP = 6;          // fields kept per solve kind: the current one + EXT_NP-1 earlier ones (Lagrange order <= EXT_NP-1)
constexpr int EXT_NBLK = 32;       // blocks of the partial-sum pass
constexpr int EXT_NS = 2 * EXT_NP; // partial sums per block: <d_j,d1> (j = 0..NP-1), <d_j,d_j> (j = 0, 2..NP-1), <m_k,m_k>
constexpr int EXT_KEEP = EXT_NP, EXT_COUNT = EXT_NP + 1, EXT_PART = EXT_NP + 2;   // ext = {w_0..w_{NP-1}, keep, count, partial sums...}

// pass 1: per-block partial sums over the model history hist = [m_k | m_{k-1} | ... | m_{k-NP+1}]: steps
// d0 = m_new - m_k, d_j = m_{k-j+1} - m_{k-j};  a[j] = <d_j,d1> (j < NP), a[NP] = <d0,d0>, a[NP+j-1] = <d_j,d_j>
// (j = 2..NP-1), a[2NP-1] = <m_k,m_k>   ->  part[block][EXT_NS]
__global__ __launch_bounds__(256) void k_extrap_sums(const double* __restrict__ mNew, const double* __restrict__ hist, int nAC,
                                                      double* __restrict__ part) {
    __shared__ double sh[EXT_NS][4];
    double a[EXT_NS];
#pragma unroll
    for (int q = 0; q < EXT_NS; ++q) a[q] = 0.0;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < nAC; i += EXT_NBLK * 256) {
        double m[EXT_NP], d[EXT_NP];
#pragma unroll
        for (int j = 0; j < EXT_NP; ++j) m[j] = hist[(long)j * nAC + i];
        d[0] = mNew[i] - m[0];
#pragma unroll
        for (int j = 1; j < EXT_NP; ++j) d[j] = m[j - 1] - m[j];
#pragma unroll
        for (int j = 0; j < EXT_NP; ++j) a[j] += d[j] * d[1];
        a[EXT_NP] += d[0] * d[0];
#pragma unroll
        for (int j = 2; j < EXT_NP; ++j) a[EXT_NP + j - 1] += d[j] * d[j];
        a[EXT_NS - 1] += m[0] * m[0];
    }
    const int w = threadIdx.x >> 6;
#pragma unroll
    for (int q = 0; q < EXT_NS; ++q) {
        a[q] = wave_sum(a[q]);
        if ((threadIdx.x & 63) == 0) sh[q][w] = a[q];
    }
    __syncthreads();
    if (threadIdx.x < EXT_NS) part[blockIdx.x * EXT_NS + threadIdx.x] = sh[threadIdx.x][0] + sh[threadIdx.x][1] + sh[threadIdx.x][2] + sh[threadIdx.x][3];
}

// pass 2 (one wave): the extrapolation weights from the partial sums
__global__ __launch_bounds__(64) void k_extrap_weights(const double* __restrict__ part, double* ext, int maxNp) {
    double a[EXT_NS];
#pragma unroll
    for (int q = 0; q < EXT_NS; ++q) a[q] = wave_sum(threadIdx.x < EXT_NBLK ? part[threadIdx.x * EXT_NS + q] : 0.0);
    if (threadIdx.x != 0) return;
    const int count = (int)ext[EXT_COUNT];
    const double d1d1 = a[1], d0d0 = a[EXT_NP], mkmk = a[EXT_NS - 1];
    const bool keep = count >= 1 && d0d0 <= 1e-28 * mkmk;
    double wts[EXT_NP];                                       // weights of x_k, x_{k-1}, ...
    for (int i = 0; i < EXT_NP; ++i) wts[i] = i == 0 ? 1.0 : 0.0;
    if (!keep && count >= 2 && d1d1 > 0) {
        const double alpha = fmin(2.0, fmax(-1.0, a[0] / d1d1));
        wts[0] = 1.0 + alpha; wts[1] = -alpha;
        // "times" of the models on the line through the last step: 0, -1, -1-g2, -1-g2-g3, ...; a further point is used
        // while its step is nearly collinear with the last one and of comparable length
        double tau[EXT_NP];
        tau[0] = 0.0; tau[1] = -1.0;
        int np = 2;
        if (d0d0 > 0 && a[0] / sqrt(d0d0 * d1d1) > 0.95 && alpha > 0.5 && alpha < 2.0) {
            for (int j = 2; j < EXT_NP; ++j) {
                const double djdj = a[EXT_NP + j - 1], g = a[j] / d1d1;
                if (!(count >= j + 1 && djdj > 0 && a[j] / sqrt(djdj * d1d1) > 0.95 && g > 0.5 && g < 2.0)) break;
                tau[j] = tau[j - 1] - g;
                np = j + 1;
            }
        }
        np = min(maxNp, np);
        if (np > 2) {
            for (int i = 0; i < EXT_NP; ++i) {
                double l = i < np ? 1.0 : 0.0;
                for (int j = 0; j < np; ++j)
                    if (j != i && i < np) l *= (alpha - tau[j]) / (tau[i] - tau[j]);
                wts[i] = l;
            }
        }
    }
    for (int i = 0; i < EXT_NP; ++i) ext[i] = wts[i];
    ext[EXT_KEEP] = keep ? 1.0 : 0.0;
    if (!keep) ext[EXT_COUNT] = (double)min(count + 1, EXT_NP);
}

// pass 3: the model history moves on (unless the model is a repeat)
__global__ __launch_bounds__(256) void k_extrap_shift(const double* __restrict__ mNew, double* hist, int nAC, const double* __restrict__ ext) {
    if (ext[EXT_KEEP] != 0.0) return;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < nAC) {
#pragma unroll
        for (int j = EXT_NP - 1; j >= 1; --j) hist[(long)j * nAC + i] = hist[(long)(j - 1) * nAC + i];
        hist[i] = mNew[i];
    }
}

// x <- sum_j w_j x_{k-j}, history shifted (... <- xp1 <- xp0 <- old x), on interior nodes (runs beside
// k_bc_forward, which writes X's boundary nodes); xp = [EXT_NP-1][S*vstride]
__global__ __launch_bounds__(VBLOCK) void k_extrap(Solver k, cplx* x, cplx* xp, const double* __restrict__ ext) {
    if (ext[EXT_KEEP] != 0.0) return;
    double w[EXT_NP];
#pragma unroll
    for (int j = 0; j < EXT_NP; ++j) w[j] = ext[j];
    const long so = (long)blockIdx.y * k.vstride, hs = (long)k.S * k.vstride;
    const long e0 = (long)blockIdx.x * k.chunk, e1 = min(e0 + k.chunk, k.vstride);
    for (long e = e0 + threadIdx.x; e < e1; e += VBLOCK) {
        const int iz = (int)(e / k.NYP), iy = (int)(e - (long)iz * k.NYP);
        if (iz < 1 || iz > k.nz - 1 || iy < 1 || iy > k.ny - 1) continue;
        cplx q[EXT_NP];
        q[0] = x[so + e];
#pragma unroll
        for (int j = 1; j < EXT_NP; ++j) q[j] = xp[(long)(j - 1) * hs + so + e];
        cplx acc = w[0] * q[0];
#pragma unroll
        for (int j = 1; j < EXT_NP; ++j) acc += w[j] * q[j];
#pragma unroll
        for (int j = EXT_NP - 1; j >= 1; --j) xp[(long)(j - 1) * hs + so + e] = q[j - 1];
        x[so + e] = acc;
    }
}

// true residual norm check: partB = |b - A x|^2 with b passed separately (verify option)
__global__ __launch_bounds__(VBLOCK) void k_trueres(Solver k, const cplx* b, const cplx* x, double* partRes, double* partBn) {
    const int s = blockIdx.y;
    __shared__ double sh[8];
    const int mode = s >= k.nFreq;
    const long mo = (long)mode * k.vstride, so = (long)s * k.vstride;
    const double w = k.omega[s];
    const cplx* p = x + so;
    const long e0 = (long)blockIdx.x * k.chunk, e1 = min(e0 + k.chunk, k.vstride);
    double rr = 0, bb = 0;
    for (long e = e0 + threadIdx.x; e < e1; e += VBLOCK) {
        const int iz = (int)(e / k.NYP), iy = (int)(e - (long)iz * k.NYP);
        if (iz >= 1 && iz <= k.nz - 1 && iy >= 1 && iy <= k.ny - 1) {
            const cplx c = p[e];
            const double dk = k.dK[mo + e], dm = w * k.dM[mo + e];
            cplx acc = cplx{dk * c.re - dm * c.im, dk * c.im + dm * c.re};
            // boundary entries of x hold Dirichlet values: they belong to the right-hand side
            if (iy + 1 <= k.ny - 1) acc += k.cY[mo + e] * p[e + 1];
            if (iy - 1 >= 1) acc += k.cY[mo + e - 1] * p[e - 1];
            if (iz + 1 <= k.nz - 1) acc += k.cZ[mo + e] * p[e + k.NYP];
            if (iz - 1 >= 1) acc += k.cZ[mo + e - k.NYP] * p[e - k.NYP];
            rr += cabs2(b[so + e] - acc);
            bb += cabs2(b[so + e]);
        }
    }
    block_sum2(rr, bb, sh);
    if (threadIdx.x == 0) { partRes[(long)s * MAXNB + blockIdx.x] = rr; partBn[(long)s * MAXNB + blockIdx.x] = bb; }
}

// ----------------------------------------------------------------------------------------------
// item kernels
// ----------------------------------------------------------------------------------------------
#define TID1 (blockIdx.x * blockDim.x + threadIdx.x)

__global__ void k_null() {}
__global__ void k_sigma(View v) { int c = TID1; if (c < v.nCell) item_sigma(v, c); }
// lateral means of one cell row per wave (deterministic shuffle reduction)
__global__ __launch_bounds__(64) void k_rowmean(View v) {
    const int kz = blockIdx.x;
    double sa = 0.0, sl = 0.0;
    for (int ky = threadIdx.x; ky < v.ny; ky += 64) {
        const double s = v.sigma[(long)kz * v.ny + ky];
        sa += s; sl += log(s);
    }
    sa = wave_sum(sa); sl = wave_sum(sl);
    if (threadIdx.x == 0) { v.sigMeanA[kz] = sa / v.ny; v.sigMeanG[kz] = exp(sl / v.ny); }
}
__global__ void k_coef(View v, int te_doK, int te_doM, int tm_doK, int tm_doM) {
    int e = TID1;
    if (e >= v.NZP * (v.ny + 1)) return;
    int iz = e / (v.ny + 1), iy = e % (v.ny + 1);
    if (te_doK || te_doM) item_coef(v, 0, iy, iz, te_doK, te_doM);
    if (tm_doK || tm_doM) item_coef(v, 1, iy, iz, tm_doK, tm_doM);
}
__global__ void k_fdm_z(View v) { int e = TID1; if (e < 2 * v.NZP) item_fdm_z(v, e / v.NZP, e % v.NZP); }
// Inverse pivots of the FDM tridiagonals, with what used to be two more launches in front of and behind it: every
// workgroup computes its mode's four z-coefficient rows (item_fdm_z: a hundred values) straight into LDS -- the
// serial loop reads them from there, not from global memory -- and the first workgroup of each mode also stores
// them for the solver's kernels; the complex64 copy of the pivots (ip32 != nullptr) is written along.
__global__ __launch_bounds__(64) void k_pivot(View v, float2* ip32) {
    extern __shared__ __attribute__((aligned(16))) char smem_pv[];
    double* tab = reinterpret_cast<double*>(smem_pv);
    const int j = blockIdx.x * blockDim.x + threadIdx.x, s = blockIdx.y, mode = s >= v.nFreq;
    const bool store = blockIdx.x == 0 && s == mode * v.nFreq;
    for (int i = threadIdx.x; i < v.NZP; i += blockDim.x) {
        double a, b, c, d;
        fdm_z_values(v, mode, i, a, b, c, d);
        tab[i] = a; tab[v.NZP + i] = b; tab[2 * v.NZP + i] = c; tab[3 * v.NZP + i] = d;
        if (store) {
            const long o = (long)mode * v.NZP + i;
            v.mzq[o] = a; v.dgz[o] = b; v.ofz[o] = c; v.mzs[o] = d;
        }
    }
    __syncthreads();
    if (j < v.ny - 1)
        item_pivot_tab(v, s, j, tab, tab + v.NZP, tab + 2 * v.NZP, tab + 3 * v.NZP,
                       ip32 ? reinterpret_cast<float*>(ip32 + (long)s * v.vstride + j) : nullptr);
}
__global__ __launch_bounds__(64) void k_bc_layers(View v) {       // grid z: frequencies
    int col = blockIdx.x * blockDim.x + threadIdx.x, j = blockIdx.y, f = blockIdx.z;
    if (col <= v.ny) item_bc_layers_f(v, f, j, col);
}
// One thread per boundary column.  The two edge columns need the whole 1-D field (left / right boundary values):
// their lane writes it to LDS inside the recurrence (same code path as every other lane) and the wave copies it
// out afterwards.
// One lane = one boundary column of one FREQUENCY: the layered-earth recurrences are the same for the two
// polarisations, so one pass yields both systems' values.  Only the two edge columns need the fields under every layer;
// their lanes store the amplitudes per layer in LDS and the workgroup evaluates the outputs afterwards in parallel
// (fwd_outputs: ~30 fp64 instructions per layer that would otherwise sit in every wave's serial loop).
__global__ __launch_bounds__(64) void k_bc_forward(View v) {
    extern __shared__ __attribute__((aligned(16))) char smem_bc[];
    cplx* amp = reinterpret_cast<cplx*>(smem_bc);         // [slot 0: column 0, slot 1: column ny][nz][eu, ed]
    __shared__ FwdTop top[2];
    __shared__ int deadAt[2];
    const int col0 = blockIdx.x * blockDim.x, col = col0 + threadIdx.x, f = blockIdx.y;
    const bool onE = v.sysOn[f] != 0, onH = v.sysOn[v.nFreq + f] != 0;
    if (!onE && !onH) return;
    cplx* XE = v.X + (long)f * v.vstride;
    cplx* XH = v.X + (long)(v.nFreq + f) * v.vstride;
    const long ls = v.ny + 1, qs = (long)v.nz * ls;
    const bool has0 = col0 == 0, hasN = col0 <= v.ny && v.ny < col0 + (int)blockDim.x;
    if (col <= v.ny) {
        if (onE) XE[nidx(v, col, 0)] = cplx{1.0, 0.0};    // top row incl. corners
        if (onH) XH[nidx(v, col, 0)] = cplx{1.0, 0.0};
        const cplx* T = v.fwdTab + (long)f * FWD_NQ * qs + col;
        // ONE instantiation of the recurrence for every lane (a separate call for the edge lanes would make their
        // wave run the whole chain twice, once per divergent path: that was the kernel's critical path)
        const bool isEdge = col == 0 || col == v.ny;
        const int slot = col == 0 ? 0 : 1;
        cplx* ea = amp + (long)slot * 2 * v.nz;
        int dAt = v.nz;                                   // first layer behind the overflow cut-off
        FwdTop tp;
        cplx lastE, lastH;
        bc1d_forward_core(v.omega[f], v.nz, T, qs, ls, [&](int i, cplx eu, cplx ed, cplx, bool dead) {
            if (isEdge) { ea[2 * i] = eu; ea[2 * i + 1] = ed; if (dead && dAt > i) dAt = i; }
        }, tp, lastE, lastH);
        if (isEdge) { top[slot] = tp; deadAt[slot] = dAt; }
        else {
            if (onE) XE[nidx(v, col, v.nz)] = lastE;
            if (onH) XH[nidx(v, col, v.nz)] = lastH;
        }
    }
    if (has0 || hasN) {
        __syncthreads();
        for (int i = threadIdx.x; i < v.nz; i += blockDim.x) {
#pragma unroll
            for (int slot = 0; slot < 2; ++slot) {
                if (slot == 0 ? !has0 : !hasN) continue;
                const int ecol = slot == 0 ? 0 : v.ny;
                const cplx* T = v.fwdTab + (long)f * FWD_NQ * qs + ecol;
                const cplx kj = T[(long)(i + 1 < v.nz ? i + 1 : v.nz - 1) * ls];      // k of the layer below (the last layer: its own)
                cplx oE, oH;
                fwd_outputs(top[slot], amp[((long)slot * v.nz + i) * 2], amp[((long)slot * v.nz + i) * 2 + 1], kj, i >= deadAt[slot], oE, oH);
                if (onE) XE[nidx(v, ecol, 1 + i)] = oE;
                if (onH) XH[nidx(v, ecol, 1 + i)] = oH;
            }
        }
    }
}
__global__ __launch_bounds__(64) void k_sens_layers(View v) {
    int j = blockIdx.x * blockDim.x + threadIdx.x, prof = blockIdx.y, s = blockIdx.z;
    if (j <= v.nz) item_sens_layers(v, s, prof, j);
}
__global__ __launch_bounds__(64) void k_sens_profile(View v) {
    int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e < 3 * v.S) item_sens_profile(v, e / 3, e % 3);
}
__global__ void k_rhs(View v) {
    int e = blockIdx.x * blockDim.x + threadIdx.x, s = blockIdx.y;
    if (e >= v.NZP * (v.ny + 1)) return;
    item_rhs(v, s, e % (v.ny + 1), e / (v.ny + 1));
}
__global__ __launch_bounds__(64) void k_rx(View v, int wantDeriv) {
    int e = TID1;
    if (e < v.S * v.nRx) item_rx(v, e / v.nRx, e % v.nRx, wantDeriv != 0);
}
__global__ void k_resid(View v) { int p = TID1; if (p < v.nData) item_resid(v, p); }
__global__ void k_misfit(View v, double* out) {
    __shared__ double sh[8];
    double a = 0, b = 0;
    for (int p = threadIdx.x; p < v.nData; p += blockDim.x) a += v.misfitPart[p];
    block_sum2(a, b, sh);
    if (threadIdx.x == 0) *out = a;
}
// Between the two solves: per (system, receiver) the impedance (+ its derivatives), then the residual / misfit terms of
// the data that address this receiver and their sum of conj(W'W r) -- one launch instead of three in a row on the
// critical path (a datum belongs to exactly one (system, receiver), so there is no cross-thread dependency; the
// misfit itself, a reduction over all data, is not needed by the adjoint half and is summed after the sources).
__global__ __launch_bounds__(64) void k_rxall(View v, int wantGrad) {        // (64: registers instead of 200 B of spills)
    const int e = TID1;
    if (e >= v.S * v.nRx) return;
    const int s = e / v.nRx, r = e % v.nRx;
    item_rx(v, s, r, wantGrad != 0);
    cplx c = cplx{0, 0};
    for (int t = v.srStart[e]; t < v.srStart[e + 1]; ++t) {
        const int p = v.srList[t];
        item_resid(v, p);
        c += v.vbar[p];
    }
    if (wantGrad) v.rxCoef[e] = c;
}
__global__ void k_rxcoef(View v) { int e = TID1; if (e < v.S * v.nRx) item_rxcoef(v, e / v.nRx, e % v.nRx); }
// adjoint sources; workgroup (0,0) also adds up the misfit terms (a reduction nothing on the device waits for: no
// launch of its own on the critical path between the solves)
__global__ __launch_bounds__(128) void k_src(View v, double* misfitOut, int nsrc) {
    // blocks x < nsrc: the sources; the blocks behind them: the receiver-layer Q-terms of the gradient (item_qterm
    // needs nothing from the adjoint solve: here they cost no launch in the gradient tail)
    const int s = blockIdx.y;
    if ((int)blockIdx.x >= nsrc) {
        const int ky = (blockIdx.x - nsrc) * blockDim.x + threadIdx.x;
        if (ky < v.ny) item_qterm(v, s, ky);
        return;
    }
    int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e < 2 * (v.ny + 1)) item_src(v, s, e / (v.ny + 1), e % (v.ny + 1));
    if (blockIdx.x == 0 && blockIdx.y == 0) {
        __shared__ double sh[2];
        double a = 0;
        for (int p = threadIdx.x; p < v.nData; p += 128) a += v.misfitPart[p];
        a = wave_sum(a);
        if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = a;
        __syncthreads();
        if (threadIdx.x == 0) *misfitOut = sh[0] + sh[1];
    }
}
__global__ void k_wb(View v) {
    int e = blockIdx.x * blockDim.x + threadIdx.x, s = blockIdx.y;
    if (e < v.nz) item_wside(v, s, e + 1);
    else if (e < v.nz + v.ny) item_colw(v, s, e - v.nz);
}
__global__ __launch_bounds__(64) void k_bcsens_pre(View v) {
    int c = blockIdx.x * blockDim.x + threadIdx.x, prof = blockIdx.y, s = blockIdx.z;
    if (c < v.nz) item_bcsens_pre(v, s, prof, c);
}
__global__ void k_bcsens_contract(View v) {
    int c = blockIdx.x * blockDim.x + threadIdx.x, prof = blockIdx.y, s = blockIdx.z;
    if (c < v.nz) item_bcsens_contract(v, s, prof, c);
}
__global__ void k_gradcell(View v) {
    int c = blockIdx.x * blockDim.x + threadIdx.x, mode = blockIdx.y, grp = blockIdx.z;
    if (c < v.nCell) item_gradcell_group(v, mode, grp, c);
}
__global__ __launch_bounds__(64) void k_qterm(View v) {
    int ky = blockIdx.x * blockDim.x + threadIdx.x, s = blockIdx.y;
    if (ky < v.ny) item_qterm(v, s, ky);
}
// final assembly with four lanes per active cell (a latency-bound loop over the systems: 4x the threads), each
// taking every fourth system / partial sum; the four partial sums are added in lane order
__global__ void k_gradfinal(View v) {
    const int t = TID1, a = t >> 2, l = t & 3;
    double g = 0.0;
    if (a < v.nAC) {
        const int cell = v.act[a];
        const int ky = cell % v.ny, kz = cell / v.ny;
        for (int q = l; q < 2 * GRAD_NG; q += 4) g += v.gPartG[(long)q * v.nCell + cell];
#pragma unroll 4
        for (int s = l; s < v.S; s += 4) g += gradfinal_sys(v, s, ky, kz);
        if (kz == v.zid)
            for (int s = l; s < v.S; s += 4) g += v.qPart[(long)s * v.ny + ky];
    }
    // lanes 4a .. 4a+3 are neighbours in a wave (the grid is a multiple of 64 threads)
    const double g1 = __shfl_down(g, 1, 4), g2 = __shfl_down(g, 2, 4), g3 = __shfl_down(g, 3, 4);
    if (a < v.nAC && l == 0) v.grad[a] = exp(v.m[a]) * (((g + g1) + g2) + g3);
}

// copy padded nodal layout -> reference layout [(ny+1)*(nz+1)] per frequency
__global__ void k_unpad(View v, const cplx* src, cplx* dst, int s0) {
    int e = blockIdx.x * blockDim.x + threadIdx.x, f = blockIdx.y;
    const int nn = (v.ny + 1) * (v.nz + 1);
    if (e >= nn) return;
    int iz = e / (v.ny + 1), iy = e % (v.ny + 1);
    dst[(long)f * nn + e] = src[(long)(s0 + f) * v.vstride + nidx(v, iy, iz)];
}

// ----------------------------------------------------------------------------------------------
// leapfrog vector kernels (proposeLeapfrog, HMCSampler.jl:206-269; diagonal mass)
// ----------------------------------------------------------------------------------------------
struct LfView {
    int n;
    const double *mref, *invM, *wmVal;
    const long long *wmRow, *wmCol;
    double *m, *p, *g;            // model, momentum, data gradient (in) / total gradient (out)
    double *part;                 // [LFNB] partial maxima / sums
    double *scal;                 // [0] mnorm
    int* flag;                    // non-zero: non-finite value met
};
constexpr int LFNB = 64;

// g <- g + lambda*Wm*(m - mref) ; p <- p - c*dt*g      (HMCSampler.jl:223-228, 255-263)
__global__ void k_lf_momentum(LfView L, double lambda, double cdt) {
    const int a = TID1;
    if (a >= L.n) return;
    double acc = 0.0;
    for (long long t = L.wmRow[a]; t < L.wmRow[a + 1]; ++t) {
        const long long j = L.wmCol[t];
        acc += L.wmVal[t] * (L.m[j] - L.mref[j]);
    }
    const double g = L.g[a] + lambda * acc;       // L.g stays the data gradient (hmcmt_leapfrog memoises it)
    L.p[a] -= cdt * g;
}
// partial max |dt*invM*p|   (HMCSampler.jl:237-240)
__global__ __launch_bounds__(256) void k_lf_dmmax(LfView L, double dt) {
    __shared__ double sh[4];
    double mx = 0.0;
    for (int a = blockIdx.x * 256 + threadIdx.x; a < L.n; a += 256 * LFNB) mx = fmax(mx, fabs(dt * L.invM[a] * L.p[a]));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmax(mx, __shfl_down(mx, o, 64));
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = mx;
    __syncthreads();
    if (threadIdx.x == 0) L.part[blockIdx.x] = fmax(fmax(sh[0], sh[1]), fmax(sh[2], sh[3]));
}
// m += dm (clamped to max |dm| = 3), reflect at the ln-sigma bounds, flip momentum (:241-247, :515-559)
__global__ void k_lf_step(LfView L, double dt, double lo, double hi) {
    const int a = TID1;
    if (a >= L.n) return;
    double mx = 0.0;
    for (int b = 0; b < LFNB; ++b) mx = fmax(mx, L.part[b]);
    double dm = dt * L.invM[a] * L.p[a];
    if (mx > 3.0) dm = dm / mx * 3.0;
    double m = L.m[a] + dm, p = L.p[a];
    if (!isfinite(m)) { atomicExch(L.flag, 1); return; }
    for (int it = 0; it < 500 && !(m <= hi && m >= lo); ++it) {
        if (m < lo) { m = 2.0 * lo - m; p = -p; }
        if (m > hi) { m = 2.0 * hi - m; p = -p; }
    }
    L.m[a] = m; L.p[a] = p;
}
// mnorm = 0.5*lambda*(m-mref)' Wm (m-mref)   (HMCSampler.jl:389-391): partial sums, then block 0 finishes
__global__ __launch_bounds__(256) void k_lf_mnorm(LfView L, double lambda) {
    __shared__ double sh[8];
    double acc = 0.0, dummy = 0.0;
    for (int a = blockIdx.x * 256 + threadIdx.x; a < L.n; a += 256 * LFNB) {
        double row = 0.0;
        for (long long t = L.wmRow[a]; t < L.wmRow[a + 1]; ++t) { const long long j = L.wmCol[t]; row += L.wmVal[t] * (L.m[j] - L.mref[j]); }
        acc += (L.m[a] - L.mref[a]) * row;
    }
    block_sum2(acc, dummy, sh);
    if (threadIdx.x == 0) L.part[blockIdx.x] = acc;
}
__global__ void k_lf_mnorm_final(LfView L, double lambda) {
    double acc = 0.0;
    for (int b = 0; b < LFNB; ++b) acc += L.part[b];
    L.scal[0] = 0.5 * lambda * acc;
}

}  // namespace

// ----------------------------------------------------------------------------------------------
// context
// ----------------------------------------------------------------------------------------------
struct hmcmt_ctx {
    HostProblem hp;
    View v{};
    Solver sv{};
    hmcmt_options opt{};
    hmcmt_stats stats{};
    int device = 0;
    hipStream_t stream = nullptr;
    hipStream_t side = nullptr;       // sigma-only sensitivity tables run beside the forward solve
    hipStream_t side2 = nullptr;      // inverse pivots of the FDM tridiagonals run beside the boundary-value kernels
    hipEvent_t evModel = nullptr, evSens = nullptr, evExtF = nullptr, evExtA = nullptr, evPiv = nullptr, evPoll = nullptr, evRec = nullptr;
    bool solveBegun = false;                 // k_resid0 / k_resid_pre has done k_solve_begin's work for the next solve
    bool preDone = false;                    // k_resid_pre has done the first pre-smoothing pass of the next solve
    bool statsPending = false, pendingAdj = false;   // records of an asynchronous evaluation not read yet
    std::vector<void*> allocs;
    std::string err;
    // device scalars / buffers not in View
    u4v *d_Vb = nullptr, *d_Vtb = nullptr;        // bf16 fragment-order copies of V, V' (hi parts)
    u4v *d_Vbl = nullptr, *d_Vtbl = nullptr;      // ... lo parts: V = hi + lo to ~16 mantissa bits
    float2* d_invp32 = nullptr;
    double *d_m = nullptr, *d_V = nullptr, *d_Vt = nullptr, *d_partZZ = nullptr, *d_misfit = nullptr;
    double *d_partRes = nullptr, *d_partBn = nullptr;
    cplx* d_b = nullptr;                  // copy of the right-hand side (verify)
    cplx* d_fieldsOut = nullptr;
    // pinned host staging
    int* h_nactive = nullptr;
    int* h_stall = nullptr;               // pinned, mapped: Solver::stallHost
    hipEvent_t evPoll2[2] = {nullptr, nullptr};   // convergence polls look one iteration back (see solve())
    double* h_rec = nullptr;              // packed per-solve records: [2][S] iters, [2][S] status (int), [2][S] err (double)
    double* h_stage = nullptr;            // m / grad / pred / misfit staging
    size_t stageDoubles = 0;
    std::vector<int> itersLast;           // [2*S]
    double* d_recHost = nullptr;          // device address of h_rec (pinned, mapped): k_solve_end writes it directly
    bool solveDone[2] = {true, true};
    int lastItFwd = 0, lastItAdj = 0;
    bool haveModel = false;
    bool haveFwd = false, haveAdj = false;   // previous fields usable as initial guesses
    cplx* d_prevField[2] = {nullptr, nullptr};   // the two previous solutions (warm_start == 2), per solve kind: [2][S*vstride]
    double* d_mHist[2] = {nullptr, nullptr};     // [3][nAC] model history per solve kind
    double* d_ext[2] = {nullptr, nullptr};       // {w0, w1, w2, keep, count}
    double jacobiW = 0.8;                    // damping of the point-Jacobi halves (HMCMT_JACOBI_W; 0.7 in round 1: 0.8 saves 3-8 % of the iterations on structured models, costs 6-25 % on white-noise models of std >= 1)
    int extrapNp = EXT_NP;                   // fields used by the initial-guess extrapolation (HMCMT_EXTRAP_POINTS = 2..EXT_NP)
    bool fusedFwd = true;                    // forward transform + tridiagonal solve in one kernel (HMCMT_FUSED_FWD=0: separate)
    View sideView; const double* sideM = nullptr;   // deferred side-stream launches of the adjoint half (launch_adjoint_side)
    bool sidePending = false, sideExtrap = false, sideSens = false;
    bool fusedBack = true;                   // back transform + post-smoother in one kernel (HMCMT_FUSED_BACK=0: separate)
    long long* backStamps = nullptr;         // HMCMT_BACK_STAMPS: per-block s_memtime stamps of k_back_post (debug entry only)
    size_t maxLdsBack = 64 * 1024;
    bool twistOn = true;                     // HMCMT_TWIST=0: classic one-sided sweeps in the fused kernel as well
    bool fusedFwdForce = false;              // HMCMT_FUSED_FWD=2: also where the heuristic prefers the separate kernels
    size_t maxLds = 64 * 1024;               // dynamic LDS the fused kernels may request
    bool lpFallback = false;                 // this solve has switched its stragglers to the fp64 preconditioner
    // profiling
    unsigned profMask = 0;            // bit c: time category c with HIP events
    int profEvery = 1;                // ... in every profEvery-th evaluation only (the brackets cost ~20 % if always on)
    long long evalCount = 0;
    std::vector<hipEvent_t> evPool;
    std::vector<int> evCat;
    size_t evUsed = 0;
    double profOverheadMs = 0.0;      // event-bracket overhead of one launch (null-kernel calibration)
    double profMs[HMCMT_NCAT] = {0};
    long long profN[HMCMT_NCAT] = {0};
    unsigned long long* d_cnt = nullptr;   // device counter behind Solver::cntActive
    long long profStartSys = 0, profEvals = 0, profSolves = 0;   // sampled: systems active at the start of a solve (summed), evaluations, solves
    int nSysOn = 0;
    // leapfrog / prior
    double *d_mref = nullptr, *d_invM = nullptr, *d_wmVal = nullptr, *d_p = nullptr, *d_mcur = nullptr, *d_g = nullptr;
    long long *d_wmRow = nullptr, *d_wmCol = nullptr;
    double *d_lfPart = nullptr, *d_lfScal = nullptr;
    int* d_lfFlag = nullptr;
    int* h_lfFlag = nullptr;                 // pinned copy of d_lfFlag (read after a synchronisation)
    double* d_gStart = nullptr;              // data gradient at the start model of the last trajectory (a rejection restarts there)
    bool havePrior = false, lfHaveGrad = false, lfFlagPending = false;
    // results of the last two host-API evaluations, keyed by the model: a sampler re-evaluates the model it has
    // just evaluated (getHamiltonian at the proposal, HMCSampler.jl:364; the first gradient of the next trajectory,
    // :217) or, after a rejection, the start model of the trajectory before -- those calls cost a memcmp
    struct Memo { std::vector<double> m, pred, grad; double misfit = 0; bool valid = false, hasGrad = false; hmcmt_stats stats{}; };
    Memo memo[2];
    int memoNext = 0;
    long long memoHits = 0;
};

static std::string g_createError;

#define HIPCHK(call)                                                                                   \
    do {                                                                                               \
        hipError_t e_ = (call);                                                                        \
        if (e_ != hipSuccess) {                                                                        \
            ctx->err = std::string(#call) + ": " + hipGetErrorString(e_);                              \
            return (e_ == hipErrorOutOfMemory) ? HMCMT_ENOMEM : HMCMT_EHIP;                            \
        }                                                                                              \
    } while (0)

namespace {

template <class T>
int dalloc(hmcmt_ctx* ctx, T** p, size_t n, bool zero = true) {
    void* q = nullptr;
    size_t bytes = std::max<size_t>(n, 1) * sizeof(T);
    HIPCHK(hipMalloc(&q, bytes));
    ctx->allocs.push_back(q);
    if (zero) HIPCHK(hipMemsetAsync(q, 0, bytes, ctx->stream));
    *p = (T*)q;
    return 0;
}
template <class T>
int dupload(hmcmt_ctx* ctx, T** p, const std::vector<T>& h) {
    int rc = dalloc(ctx, p, h.size(), false);
    if (rc) return rc;
    if (!h.empty()) HIPCHK(hipMemcpy(*p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
    return 0;
}

struct ProfScope {
    hmcmt_ctx* c; int cat; size_t idx;
    ProfScope(hmcmt_ctx* ctx, int cat_) : c(ctx), cat(cat_), idx((size_t)-1) {
        if (!((c->profMask >> cat) & 1u) || (c->evalCount % c->profEvery) != 0) return;
        if (c->lpFallback) return;          // the fp64 restart of a straggling solve runs other kernels: not part of the sampled population
        if (c->evUsed + 2 > c->evPool.size()) {
            for (int i = 0; i < 512; ++i) { hipEvent_t e; if (hipEventCreate(&e) != hipSuccess) return; c->evPool.push_back(e); c->evCat.push_back(0); }
        }
        idx = c->evUsed; c->evUsed += 2;
        c->evCat[idx] = cat;
        hipEventRecord(c->evPool[idx], c->stream);
    }
    ~ProfScope() { if (idx != (size_t)-1) hipEventRecord(c->evPool[idx + 1], c->stream); }
};

void prof_collect(hmcmt_ctx* c) {
    if (c->evUsed == 0) return;
    hipStreamSynchronize(c->stream);
    for (size_t i = 0; i + 1 < c->evUsed; i += 2) {
        float ms = 0;
        if (c->evCat[i] < 0) continue;                    // (a launch known to have been empty: solve())
        if (hipEventElapsedTime(&ms, c->evPool[i], c->evPool[i + 1]) == hipSuccess) {
            c->profMs[c->evCat[i]] += std::max(0.0, (double)ms - c->profOverheadMs);
            c->profN[c->evCat[i]] += 1;
        }
    }
    c->evUsed = 0;
}

inline dim3 grid1(int n, int b) { return dim3((n + b - 1) / b); }

int launch_transform(hmcmt_ctx* ctx, const cplx* A, const double* Bsw, cplx* C, const int* active) {
    const View& v = ctx->v;
    const int M = v.S * v.NZP, NT = v.NYP / 16;
    const int groups = (M + 7) / 8;
    const int NW = std::min(4, (NT + 6) / 7);
    if ((NT + NW - 1) / NW > 7) { ctx->err = "mesh too wide for the transform kernel (ny+1 > 448)"; return HMCMT_EINVAL; }
    const int RG = std::max(1, 4 / NW);
    ProfScope ps(ctx, 0);
    hipLaunchKernelGGL(k_transform, dim3((groups + RG - 1) / RG), dim3(64 * NW * RG), 0, ctx->stream, A, Bsw, C, M,
                       v.NYP, v.NZP, active, NW, RG);
    return 0;
}

template <int OUT>
int launch_transform_lp(hmcmt_ctx* ctx, const float2* A, bool transposed, void* C, const int* active) {
    const View& v = ctx->v;
    const int M = v.S * v.NZP, NT = v.NYP / 16;
    const int groups = (M + 8 * LP_NRG - 1) / (8 * LP_NRG);
    // waves of LP_NTW column tiles; a workgroup holds all NW waves of RG row-group sets (<= 8 waves)
    const int NW = std::min(8, (NT + LP_NTW - 1) / LP_NTW);      // wider meshes: a wave loops over its tiles
    ProfScope ps(ctx, 0);
    const dim3 grid(groups), block(64 * NW);
    const size_t lds = (size_t)LP_NRG * ((v.NYP + 31) / 32) * 2 * 64 * sizeof(u4v);       // staged A fragments
    const u4v *bh = transposed ? ctx->d_Vtb : ctx->d_Vb, *bl = transposed ? ctx->d_Vtbl : ctx->d_Vbl;
    if (ctx->sv.splitT)        // the input was written pre-split (store_t32 / k_fdm_fwd)
        hipLaunchKernelGGL((k_transform_lp<OUT, 1>), grid, block, lds, ctx->stream, A, bh, bl, C, ctx->sv.dinv, ctx->sv.r, M, v.NYP,
                           v.NZP, active, NW);
    else
        hipLaunchKernelGGL((k_transform_lp<OUT, 0>), grid, block, lds, ctx->stream, A, bh, bl, C, ctx->sv.dinv, ctx->sv.r, M, v.NYP,
                           v.NZP, active, NW);
    return 0;
}

// forward half of the mixed-precision FDM stage: y32 = tridiag^-1 (t32 V); fused kernel when its LDS slabs fit
// back half of the FDM stage + post-smoother: fused kernel on the pre-split path when its 16-row tile fits LDS
int launch_back_post(hmcmt_ctx* ctx) {
    Solver& k = ctx->sv;
    const int NT = k.NYP / 16, NW = std::min(8, (NT + LP_NTW - 1) / LP_NTW);
    const size_t lds = (size_t)16 * k.NYP * sizeof(cplx) + (size_t)2 * ((k.NYP + 31) / 32) * 2 * 64 * 16;   // z tile + staged A fragments
    dim3 vg(k.NB, k.S), vb(VBLOCK);
    if (k.splitT && ctx->fusedBack && lds <= ctx->maxLdsBack && k.NYP <= 256) {      // the kernel holds all of a wave's V fragments: 8 k-groups, 2 tiles
        const int nwg = (k.nz - 1 + BP_OWN - 1) / BP_OWN;
        ProfScope ps(ctx, 0);
        hipLaunchKernelGGL(k_back_post<1>, dim3(nwg, k.S), dim3(64 * NW), lds, ctx->stream, k, k.y32, ctx->d_Vtb, ctx->d_Vtbl,
                           ctx->d_partZZ, NW, ctx->backStamps);
        return 0;
    }
    int rc;
    if ((rc = launch_transform_lp<2>(ctx, k.y32, true, k.z, k.active))) return rc;   // z = F t + dinv r
    { ProfScope ps(ctx, 7); hipLaunchKernelGGL(k_post, vg, vb, 0, ctx->stream, k, ctx->d_partZZ); }
    return 0;
}

// slab width (in 16-mode tiles) of the fused forward kernel for this problem, 0 = use the separate kernels
size_t fdm_fwd_lds(const Solver& k, int ntw, int twist) {
    const int n = k.nz - 1, mid = twist_mid(n, twist);
    const size_t rl = (size_t)(twist ? mid + 1 : k.NZP) + 4 * FW_TB, nreg = twist ? 2 : 1, sw = 16 * (size_t)ntw;
    return (((size_t)k.NZP * sizeof(float) + 127) & ~(size_t)127) + (sw + 2 * FW_TB * sw + 3 * nreg * rl * sw + 2 * FW_TB * sw) * sizeof(c32);
}
int fdm_fwd_ntw(const hmcmt_ctx* ctx) {
    const Solver& k = ctx->sv;
    // The fused kernel pays off while every slab workgroup of a launch is resident at once and re-reading a
    // system's rows per slab is cheap: 32-mode slabs that fit LDS on meshes up to 256 nodes wide (measured:
    // 21 vs 29 us at 200x100 cells; at 400x200 the separate kernels win, 146 vs 219 us).  HMCMT_FUSED_FWD=2
    // forces it (16-mode slabs if need be), =0 disables it.
    if (!ctx->fusedFwd) return 0;
    const int tw = ctx->twistOn ? 1 : 0;
    if (fdm_fwd_lds(k, FW_NTW, tw) <= ctx->maxLds && (k.NYP <= 256 || ctx->fusedFwdForce)) return FW_NTW;
    if (ctx->fusedFwdForce && fdm_fwd_lds(k, 1, tw) <= ctx->maxLds) return 1;
    return 0;
}

int launch_fdm_fwd(hmcmt_ctx* ctx) {
    Solver& k = ctx->sv;
    auto ldsFor = [&](int ntw) { return fdm_fwd_lds(k, ntw, k.twist); };
    const int ntw = k.splitT ? fdm_fwd_ntw(ctx) : 0;       // (k.splitT is set from fdm_fwd_ntw: the operand format goes with the path)
    if (ntw) {
        const int G = (k.NZP + 7) / 8, per = (G + 7) / 8, nw = (G + per - 1) / per;
        const dim3 grid(((k.NYP / 16 + ntw - 1) / ntw) * k.S), block(64 * nw);
        ProfScope ps(ctx, 1);
        if (ntw == 1)
            hipLaunchKernelGGL(k_fdm_fwd<1>, grid, block, ldsFor(1), ctx->stream, k, k.t32, ctx->d_Vb, ctx->d_Vbl, ctx->d_invp32, k.y32, (long long*)nullptr);
        else
            hipLaunchKernelGGL(k_fdm_fwd<FW_NTW>, grid, block, ldsFor(FW_NTW), ctx->stream, k, k.t32, ctx->d_Vb, ctx->d_Vbl, ctx->d_invp32, k.y32, (long long*)nullptr);
        return 0;
    }
    int rc;
    if ((rc = launch_transform_lp<0>(ctx, k.t32, false, k.y32, k.active))) return rc;
    const dim3 tg((k.ny - 1 + 63) / 64, k.S);
    { ProfScope ps(ctx, 1); hipLaunchKernelGGL(k_thomas32, tg, dim3(64), 0, ctx->stream, k); }
    return 0;
}

// z = P^-1 r for the active systems and the partial sums of r'z, |z|^2 (partA / d_partZZ)
int apply_precond(hmcmt_ctx* ctx) {
    Solver& k = ctx->sv;
    dim3 vg(k.NB, k.S), vb(VBLOCK);
    if (ctx->opt.precond == HMCMT_PRECOND_JACOBI) {
        { ProfScope ps(ctx, 3); hipLaunchKernelGGL(k_jacobi, vg, vb, 0, ctx->stream, k); }
        { ProfScope ps(ctx, 3); hipLaunchKernelGGL(k_dots, vg, vb, 0, ctx->stream, k, ctx->d_partZZ); }
        return 0;
    }
    const bool smooth = ctx->opt.precond == HMCMT_PRECOND_FDM_JACOBI;
    const dim3 tg((k.ny - 1 + 63) / 64, k.S);
    int rc;
    if (ctx->opt.fdm_precision == 0 && !ctx->lpFallback) {
        // mixed precision: split-bf16 operands / fp32 accumulation in the transforms, complex64 tridiagonal
        if (smooth && ctx->preDone) ctx->preDone = false;                        // (k_resid_pre has written t)
        else if (smooth) { ProfScope ps(ctx, 2); hipLaunchKernelGGL(k_pre_c64, vg, vb, 0, ctx->stream, k); }
        else { ProfScope ps(ctx, 3); hipLaunchKernelGGL(k_to_c64, vg, vb, 0, ctx->stream, k, k.r); }
        if ((rc = launch_fdm_fwd(ctx))) return rc;
        if (smooth) {
            if ((rc = launch_back_post(ctx))) return rc;                                     // z = F t + dinv r, then the second Jacobi half
            std::swap(k.z, k.t);
        } else {
            if ((rc = launch_transform_lp<1>(ctx, k.y32, true, k.z, k.active))) return rc;
            ProfScope ps(ctx, 3);
            hipLaunchKernelGGL(k_dots, vg, vb, 0, ctx->stream, k, ctx->d_partZZ);
        }
        return 0;
    }
    if (smooth) { ProfScope ps(ctx, 2); hipLaunchKernelGGL(k_pre, vg, vb, 0, ctx->stream, k); }
    if ((rc = launch_transform(ctx, smooth ? k.t : k.r, ctx->d_V, k.y, k.active))) return rc;
    { ProfScope ps(ctx, 1); hipLaunchKernelGGL(k_thomas, tg, dim3(64), 0, ctx->stream, k); }
    if ((rc = launch_transform(ctx, k.y, ctx->d_Vt, k.z, k.active))) return rc;
    if (smooth) {
        { ProfScope ps(ctx, 3); hipLaunchKernelGGL(k_mid, vg, vb, 0, ctx->stream, k); }
        { ProfScope ps(ctx, 7); hipLaunchKernelGGL(k_post, vg, vb, 0, ctx->stream, k, ctx->d_partZZ); }
        std::swap(k.z, k.t);                            // the smoothed result is the preconditioned residual
    } else {
        ProfScope ps(ctx, 3);
        hipLaunchKernelGGL(k_dots, vg, vb, 0, ctx->stream, k, ctx->d_partZZ);
    }
    return 0;
}

void launch_adjoint_side(hmcmt_ctx* ctx);
int collect_pending(hmcmt_ctx* ctx);

// Solves A x = r for all systems (x zero on interior on entry; r destroyed).  kind 0 forward, 1 adjoint.
int solve(hmcmt_ctx* ctx, cplx* x, int kind) {
    Solver& k = ctx->sv;
    const View& v = ctx->v;
    k.x = x;
    k.tol2 = ctx->opt.tol * ctx->opt.tol;
    const int S = k.S;
    dim3 vg(k.NB, S), vb(VBLOCK);
    const size_t vecBytes = (size_t)S * k.vstride * sizeof(cplx);
    if (ctx->opt.verify) HIPCHK(hipMemcpyAsync(ctx->d_b, k.r, vecBytes, hipMemcpyDeviceToDevice, ctx->stream));
    // all systems of the requested modes start active (device copy: no host round trip)
    if (!ctx->solveBegun)           // (otherwise done by the residual kernel in front of this solve)
        hipLaunchKernelGGL(k_solve_begin, dim3((S * MAXNB + 255) / 256), dim3(256), 0, ctx->stream, k, ctx->v.sysOn);
    ctx->solveBegun = false;
    if (k.cntActive) { ++ctx->profSolves; ctx->profStartSys += ctx->nSysOn; }
    int& guess = kind == 0 ? ctx->lastItFwd : ctx->lastItAdj;
    int nextCheck = guess > 2 ? guess : 4;
    const int every = ctx->opt.check_every > 0 ? ctx->opt.check_every : 2;
    int it = 0;
    bool done = false;
    ctx->lpFallback = false;
    const int lpCap = 60;               // (classic loop only) mixed-precision safety net: stragglers continue with the fp64 preconditioner
    const dim3 tg((k.ny - 1 + 63) / 64, S);
    const bool fused = ctx->opt.precond == HMCMT_PRECOND_FDM_JACOBI && ctx->opt.fdm_precision == 0;
    cplx* const r_entry = k.r;
    if (fused) {
        { int prc = apply_precond(ctx); if (prc) return prc; }          // z = P^-1 r and the partial sums of r'z, |z|^2
        cplx* pb[2] = {k.p, k.p2};
        cplx* rb[2] = {k.r, k.r2};
        int rcur = 0;
        // Convergence polls.  The device keeps the number of active systems (and the stagnation flag) in mapped pinned
        // memory; k_spmv_fused of iteration `it` updates them with the decision on the state after iteration it-1.  From
        // the iteration count of the previous call on, the host looks at them once per iteration -- but ONE ITERATION BACK:
        // it waits for the event behind k_spmv_fused(it-1) after queuing all of iteration `it`, so seven launches (~90 us)
        // are queued when it wakes up and the device never idles at a poll; when the solve is over the launches queued
        // behind the deciding one find every system inactive and exit at once (~2 us each).
        *ctx->h_stall = 0;
        bool stalled = false;
        while (!done && !stalled && it < ctx->opt.maxit + 1) {
            ++it;
            // decide convergence of the state after iteration it-1, p = z + beta p, q = A p
            { ProfScope ps(ctx, 2); hipLaunchKernelGGL(k_spmv_fused, dim3(k.NTR, S), vb, (size_t)(k.RT + 2) * k.NYP * sizeof(cplx), ctx->stream, k, ctx->d_partZZ, pb[(it - 1) & 1], pb[it & 1], it, ctx->opt.maxit); }
            const bool rec = it >= nextCheck || it - 1 == ctx->opt.maxit;
            if (rec) HIPCHK(hipEventRecord(ctx->evPoll2[it & 1], ctx->stream));
            { ProfScope ps(ctx, 3); hipLaunchKernelGGL(k_update_fused, dim3(k.NTR, S), vb, (size_t)(2 * k.RT + 2) * k.NYP * sizeof(cplx), ctx->stream, k, pb[it & 1], rb[rcur], rb[rcur ^ 1], it); }
            rcur ^= 1;
            k.r = rb[rcur];
            int prc;
            if ((prc = launch_fdm_fwd(ctx))) return prc;
            if ((prc = launch_back_post(ctx))) return prc;
            std::swap(k.z, k.t);
            if (it - 1 >= nextCheck || it - 1 == ctx->opt.maxit) {        // the event of iteration it-1 exists
                const bool last = it - 1 == ctx->opt.maxit;
                HIPCHK(hipEventSynchronize(ctx->evPoll2[(last ? it : it - 1) & 1]));
                if (*(volatile int*)ctx->h_nactive == 0) { done = true; break; }
                if (*(volatile int*)ctx->h_stall) stalled = true;
            }
            // (a dozen API calls, ~150 us of host time: issued once the queue is five iterations deep -- at two the main
            // queue ran dry for 86 us of every evaluation)
            if (kind == 0 && it == 5) launch_adjoint_side(ctx);
        }
        if (!done) {
            // stragglers (or the iteration cap): read the counter once more, then hand over to the classic loop
            HIPCHK(hipStreamSynchronize(ctx->stream));
            if (*(volatile int*)ctx->h_nactive == 0) done = true;
        }
        k.p = pb[it & 1];                // current search direction (only needed by the classic restart below)
        k.p2 = pb[(it - 1) & 1];
        if (done) it = std::max(0, it - 1);
    }
    if (!done) {
        const bool restart = fused;      // coming from the fused loop: restart COCG with the fp64 preconditioner
        if (restart) { ctx->lpFallback = true; ++ctx->stats.fallback_solves; }
        // z = P^-1 r ; rho = r'z ; p = z
        { int prc = apply_precond(ctx); if (prc) return prc; }
        { ProfScope ps(ctx, 3); hipLaunchKernelGGL(k_check, dim3(1), dim3(128), 0, ctx->stream, k, ctx->d_partZZ, restart ? 2 : 1, ctx->opt.maxit); }
        { ProfScope ps(ctx, 3); hipLaunchKernelGGL(k_pupdate, vg, vb, 0, ctx->stream, k, 1); }
        while (!done && it < ctx->opt.maxit) {
            ++it;
            if (it == lpCap + 1 && ctx->opt.fdm_precision == 0 && ctx->opt.precond != HMCMT_PRECOND_JACOBI && !ctx->lpFallback) {
                ctx->lpFallback = true;     // restart COCG for the still-active systems: z = P64^-1 r, p = z
                ++ctx->stats.fallback_solves;
                { int prc = apply_precond(ctx); if (prc) return prc; }
                { ProfScope ps(ctx, 3); hipLaunchKernelGGL(k_check, dim3(1), dim3(128), 0, ctx->stream, k, ctx->d_partZZ, 2, ctx->opt.maxit); }
                { ProfScope ps(ctx, 3); hipLaunchKernelGGL(k_pupdate, vg, vb, 0, ctx->stream, k, 1); }
            }
            { ProfScope ps(ctx, 2); hipLaunchKernelGGL(k_spmv, vg, vb, 0, ctx->stream, k); }
            { ProfScope ps(ctx, 3); hipLaunchKernelGGL(k_update, vg, vb, 0, ctx->stream, k); }
            { int prc = apply_precond(ctx); if (prc) return prc; }
            { ProfScope ps(ctx, 3); hipLaunchKernelGGL(k_check, dim3(1), dim3(128), 0, ctx->stream, k, ctx->d_partZZ, 0, ctx->opt.maxit); }
            { ProfScope ps(ctx, 3); hipLaunchKernelGGL(k_pupdate, vg, vb, 0, ctx->stream, k, 0); }
            if (it >= nextCheck || it == ctx->opt.maxit) {
                HIPCHK(hipStreamSynchronize(ctx->stream));
                if (*(volatile int*)ctx->h_nactive == 0) done = true;
                nextCheck = it + every;
            }
        }
    }
    // the fused loop ping-pongs r between the caller's buffer and r2: leave the struct as it was found
    if (k.r != r_entry) { k.r2 = k.r; k.r = r_entry; }
    guess = it;        // (launched iterations; collect_stats replaces it with the iterations actually needed)
    ctx->solveDone[kind] = done;
    // iteration counts / status / error estimates stay on the device; evaluate() reads both solves back at once
    hipLaunchKernelGGL(k_solve_end, dim3((S + 63) / 64), dim3(64), 0, ctx->stream, k, kind, (int*)ctx->d_recHost, ctx->d_recHost + 2 * S);
    if (ctx->opt.verify) {
        hipLaunchKernelGGL(k_trueres, vg, vb, 0, ctx->stream, k, ctx->d_b, x, ctx->d_partRes, ctx->d_partBn);
        std::vector<double> pr((size_t)S * MAXNB), pb((size_t)S * MAXNB);
        HIPCHK(hipMemcpyAsync(pr.data(), ctx->d_partRes, pr.size() * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(hipMemcpyAsync(pb.data(), ctx->d_partBn, pb.size() * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(hipStreamSynchronize(ctx->stream));
        for (int s = 0; s < S; ++s) {
            double rr = 0, bb = 0;
            for (int b = 0; b < k.NB; ++b) { rr += pr[(size_t)s * MAXNB + b]; bb += pb[(size_t)s * MAXNB + b]; }
            if (bb > 0) ctx->stats.true_res_max = std::max(ctx->stats.true_res_max, std::sqrt(rr / bb));
        }
    }
    (void)v;
    return 0;
}

// weights of the initial-guess extrapolation for solve kind kd (side stream): partial sums, weights, history shift
void launch_extrap_weights(hmcmt_ctx* ctx, const double* d_m, int kd) {
    const int nAC = ctx->v.nAC;
    double* part = ctx->d_ext[kd] + EXT_PART;
    hipLaunchKernelGGL(k_extrap_sums, dim3(EXT_NBLK), dim3(256), 0, ctx->side, d_m, ctx->d_mHist[kd], nAC, part);
    hipLaunchKernelGGL(k_extrap_weights, dim3(1), dim3(64), 0, ctx->side, part, ctx->d_ext[kd], ctx->extrapNp);
    hipLaunchKernelGGL(k_extrap_shift, dim3((nAC + 255) / 256), dim3(256), 0, ctx->side, d_m, ctx->d_mHist[kd], nAC, ctx->d_ext[kd]);
}

// side stream, beside the forward solve: the adjoint initial guess and the sigma-only sensitivity tables (joined
// before the adjoint residual / before k_bcsens).  Called when the main queue is well filled, so that the host's
// dozen API calls do not leave the device idle.
void launch_adjoint_side(hmcmt_ctx* ctx) {
    if (!ctx->sidePending) return;
    ctx->sidePending = false;
    const View& v = ctx->sideView;
    const int S = v.S;
    if (ctx->sideExtrap) {
        launch_extrap_weights(ctx, ctx->sideM, 1);
        hipLaunchKernelGGL(k_extrap, dim3(ctx->sv.NB, S), dim3(VBLOCK), 0, ctx->side, ctx->sv, v.Lam, ctx->d_prevField[1], ctx->d_ext[1]);
        hipEventRecord(ctx->evExtA, ctx->side);
    }
    if (ctx->sideSens) {
        hipStreamWaitEvent(ctx->side, ctx->evPiv, 0);          // (the lateral means come from the second side stream)
        hipLaunchKernelGGL(k_sens_layers, dim3((v.nz + 1 + 63) / 64, 3, S), dim3(64), 0, ctx->side, v);
        hipLaunchKernelGGL(k_sens_profile, dim3((3 * S + 63) / 64), dim3(64), 0, ctx->side, v);
        // the serial half of dBC^T w (78 us on a handful of CUs) depends on sigma only: here, not after the adjoint solve
        hipLaunchKernelGGL(k_bcsens_pre, dim3((v.nz + 63) / 64, 3, S), dim3(64), 0, ctx->side, v);
        hipEventRecord(ctx->evSens, ctx->side);
    }
}

// the whole hot path on device buffers
int evaluate(hmcmt_ctx* ctx, const double* d_m, bool wantGrad, double* d_pred, double* d_misfit, double* d_grad) {
    View v = ctx->v;
    v.m = d_m;
    if (d_pred) v.pred = reinterpret_cast<cplx*>(d_pred);
    if (d_grad) v.grad = d_grad;
    hipStream_t st = ctx->stream;
    const int S = v.S;
    { int prc = collect_pending(ctx); if (prc) return prc; }      // (an earlier asynchronous evaluation's records / status)
    ctx->stats = hmcmt_stats{};
    ctx->stats.nsystems = S;
    ++ctx->evalCount;
    ctx->sv.cntActive = (ctx->profMask && ctx->evalCount % ctx->profEvery == 0) ? ctx->d_cnt : nullptr;
    if (ctx->sv.cntActive) ++ctx->profEvals;
    const int nodes = v.NZP * (v.ny + 1);
    const size_t vecBytes = (size_t)S * v.vstride * sizeof(cplx);
    // initial guesses (options.warm_start): verify checks against the cold right-hand side
    const bool warmF = ctx->opt.warm_start && ctx->haveFwd && !ctx->opt.verify;
    const bool warmA = ctx->opt.warm_start && ctx->haveAdj && !ctx->opt.verify;
    const bool extrap = ctx->opt.warm_start == 2 && !ctx->opt.verify;
    // start of a solve on the default path: residual and first pre-smoothing pass in one launch (k_resid_pre)
    const size_t startLds = (size_t)(2 * ctx->sv.RT + 6) * v.NYP * sizeof(cplx);
    const bool fusedStart = ctx->opt.precond == HMCMT_PRECOND_FDM_JACOBI && ctx->opt.fdm_precision == 0 && !ctx->opt.verify &&
                            startLds <= (size_t)150 * 1024 && !getenv("HMCMT_NO_FUSED_START");
    {
        ProfScope ps(ctx, 4);
        hipLaunchKernelGGL(k_sigma, grid1(v.nCell, 256), dim3(256), 0, st, v);
        // initial guesses: zero on a cold start, otherwise the previous fields, optionally extrapolated
        if (!warmF) {
            HIPCHK(hipMemsetAsync(v.X, 0, vecBytes, st));
            HIPCHK(hipMemsetAsync(ctx->d_ext[0], 0, EXT_PART * sizeof(double), st));
        }
        if (wantGrad && !warmA) {
            HIPCHK(hipMemsetAsync(v.Lam, 0, vecBytes, st));
            HIPCHK(hipMemsetAsync(ctx->d_ext[1], 0, EXT_PART * sizeof(double), st));
        }
        HIPCHK(hipEventRecord(ctx->evModel, st));
        // Three chains start from sigma and meet at the forward residual:
        //   main    boundary-value tables and the serial 1-D recurrences (0.1 ms: the critical one)
        //   side2   lateral means -> FDM background -> inverse pivots of its tridiagonals (serial, 60-85 us)
        //   side    stencil coefficients, Jacobi diagonal, the extrapolated forward guess
        // The host issues them in this order (after the previous evaluation's synchronisation the order of the API
        // calls is the schedule); the side-stream work of the adjoint half follows from inside the forward solve.
        hipLaunchKernelGGL(k_bc_layers, dim3((v.ny + 1 + 63) / 64, v.nz, v.nFreq), dim3(64), 0, st, v);
        hipLaunchKernelGGL(k_bc_forward, dim3((v.ny + 1 + 63) / 64, v.nFreq), dim3(64), 4 * (size_t)v.nz * sizeof(cplx), st, v);
        const bool pivots = ctx->opt.precond != HMCMT_PRECOND_JACOBI;
        HIPCHK(hipStreamWaitEvent(ctx->side2, ctx->evModel, 0));
        hipLaunchKernelGGL(k_rowmean, dim3(v.nz), dim3(64), 0, ctx->side2, v);
        if (pivots)
            hipLaunchKernelGGL(k_pivot, dim3((v.ny - 1 + 63) / 64, S), dim3(64), 4 * (size_t)v.NZP * sizeof(double), ctx->side2, v,
                               ctx->opt.fdm_precision == 0 ? ctx->d_invp32 : (float2*)nullptr);
        else
            hipLaunchKernelGGL(k_fdm_z, grid1(2 * v.NZP, 64), dim3(64), 0, ctx->side2, v);
        HIPCHK(hipEventRecord(ctx->evPiv, ctx->side2));
        HIPCHK(hipStreamWaitEvent(ctx->side, ctx->evModel, 0));
        hipLaunchKernelGGL(k_coef, grid1(nodes, 256), dim3(256), 0, ctx->side, v, 0, 1, 1, 0);
        if (ctx->opt.precond == HMCMT_PRECOND_FDM_JACOBI)
            hipLaunchKernelGGL(k_dinv, dim3(ctx->sv.NB, S), dim3(VBLOCK), 0, ctx->side, ctx->sv, ctx->jacobiW);
        if (extrap) {
            // (interior nodes only -- k_bc_forward owns the boundary nodes of X)
            launch_extrap_weights(ctx, d_m, 0);
            hipLaunchKernelGGL(k_extrap, dim3(ctx->sv.NB, S), dim3(VBLOCK), 0, ctx->side, ctx->sv, v.X, ctx->d_prevField[0], ctx->d_ext[0]);
        }
        HIPCHK(hipEventRecord(ctx->evExtF, ctx->side));
        HIPCHK(hipStreamWaitEvent(st, ctx->evExtF, 0));
        // r = -Aio*bc - Aii*x0 with x0 = previous solution (or 0): one stencil pass over X
        if (fusedStart) {
            hipLaunchKernelGGL(k_resid_pre, dim3(ctx->sv.NTR, S), dim3(VBLOCK), startLds, st, ctx->sv, v.X, ctx->sv.r, ctx->sv.r2, 1, ctx->v.sysOn);
            std::swap(ctx->sv.r, ctx->sv.r2);             // (the residual went to the second buffer; swapped back after the solve)
            ctx->solveBegun = ctx->preDone = true;
        } else {
            hipLaunchKernelGGL(k_resid0, dim3(ctx->sv.NB, S), dim3(VBLOCK), 0, st, ctx->sv, v.X, 1, ctx->opt.verify ? nullptr : ctx->v.sysOn);
            ctx->solveBegun = !ctx->opt.verify;
        }
        HIPCHK(hipStreamWaitEvent(st, ctx->evPiv, 0));
        // (the adjoint half's side-stream work -- its initial guess, the sigma-only sensitivity tables -- is launched
        // from inside the forward solve, once the main queue holds two iterations: launch_adjoint_side)
        ctx->sideView = v; ctx->sideM = d_m; ctx->sideExtrap = extrap && wantGrad; ctx->sideSens = wantGrad;
        ctx->sidePending = wantGrad;
    }
    int rc = solve(ctx, v.X, 0);
    if (fusedStart) std::swap(ctx->sv.r, ctx->sv.r2);
    if (!wantGrad) HIPCHK(hipEventRecord(ctx->evRec, st));          // behind the last k_solve_end
    launch_adjoint_side(ctx);            // (no-op when the solve has already done it)
    ctx->haveFwd = (rc == 0 && ctx->solveDone[0]);
    if (rc) return rc;
    {
        ProfScope ps(ctx, 5);
        hipLaunchKernelGGL(k_rxall, grid1(S * v.nRx, 64), dim3(64), 0, st, v, wantGrad ? 1 : 0);
        if (!wantGrad) hipLaunchKernelGGL(k_misfit, dim3(1), dim3(256), 0, st, v, d_misfit ? d_misfit : ctx->d_misfit);
    }
    if (wantGrad) {
        {
            ProfScope ps(ctx, 5);
            HIPCHK(hipMemsetAsync(v.R, 0, vecBytes, st));       // (k_src assigns the two receiver rows and all of srcB)
            const int nsrc = (2 * (v.ny + 1) + 127) / 128;
            hipLaunchKernelGGL(k_src, dim3(nsrc + (v.ny + 127) / 128, S), dim3(128), 0, st, v, d_misfit ? d_misfit : ctx->d_misfit, nsrc);
            if (extrap) HIPCHK(hipStreamWaitEvent(st, ctx->evExtA, 0));
            if (warmA && fusedStart) {
                hipLaunchKernelGGL(k_resid_pre, dim3(ctx->sv.NTR, S), dim3(VBLOCK), startLds, st, ctx->sv, v.Lam, ctx->sv.r, ctx->sv.r2, 0, ctx->v.sysOn);
                std::swap(ctx->sv.r, ctx->sv.r2);
                ctx->solveBegun = ctx->preDone = true;
            } else if (warmA) {
                hipLaunchKernelGGL(k_resid0, dim3(ctx->sv.NB, S), dim3(VBLOCK), 0, st, ctx->sv, v.Lam, 0, ctx->v.sysOn);
                ctx->solveBegun = true;
            }
        }
        rc = solve(ctx, v.Lam, 1);
        if (warmA && fusedStart) std::swap(ctx->sv.r, ctx->sv.r2);
        HIPCHK(hipEventRecord(ctx->evRec, st));                        // behind the last k_solve_end
        ctx->haveAdj = (rc == 0 && ctx->solveDone[1]);
        if (rc) return rc;
        ProfScope ps(ctx, 6);
        hipLaunchKernelGGL(k_wb, dim3((v.nz + v.ny + 127) / 128, S), dim3(128), 0, st, v);
        HIPCHK(hipStreamWaitEvent(st, ctx->evSens, 0));         // sensitivity tables, boundary values, dBC (side stream)
        hipLaunchKernelGGL(k_bcsens_contract, dim3((v.nz + 127) / 128, 2, S), dim3(128), 0, st, v);
        hipLaunchKernelGGL(k_gradcell, dim3((v.nCell + 127) / 128, 2, GRAD_NG), dim3(128), 0, st, v);
        hipLaunchKernelGGL(k_gradfinal, grid1(4 * v.nAC, 128), dim3(128), 0, st, v);
    }
    HIPCHK(hipGetLastError());
    ctx->haveModel = true;
    return 0;
}

// the per-solve records (written by k_solve_end into mapped pinned memory) -> statistics; the caller has waited for them
void parse_stats(hmcmt_ctx* ctx, bool withAdjoint) {
    const int S = ctx->v.S, nk = withAdjoint ? 2 : 1;
    const int* h_iters = reinterpret_cast<const int*>(ctx->h_rec);
    const int* h_status = h_iters + 2 * S;
    const double* h_err = ctx->h_rec + 2 * S;
    for (int kind = 0; kind < nk; ++kind) {
        int mx = 0, sum = 0;
        for (int s = 0; s < S; ++s) {
            const int itv = h_iters[kind * S + s];
            ctx->itersLast[kind * S + s] = itv;
            mx = std::max(mx, itv); sum += itv;
            if (h_status[kind * S + s] != 0 && ctx->stats.status == 0) ctx->stats.status = h_status[kind * S + s];
            if (h_err[kind * S + s] > ctx->stats.err_est_max) ctx->stats.err_est_max = h_err[kind * S + s];
        }
        if (kind == 0) { ctx->stats.iters_fwd_max = mx; ctx->stats.iters_fwd_sum = sum; }
        else { ctx->stats.iters_adj_max = mx; ctx->stats.iters_adj_sum = sum; }
        // first convergence poll of the next evaluation: where this one actually finished (the loop itself only
        // knows how many iterations it launched, which includes the empty ones behind the last poll)
        if (ctx->solveDone[kind] && mx > 0) (kind == 0 ? ctx->lastItFwd : ctx->lastItAdj) = mx;
        if (!ctx->solveDone[kind] && ctx->stats.status == 0) ctx->stats.status = HMCMT_ENOCONV;
    }
    if (ctx->stats.status != 0) {
        // a failed evaluation (breakdown, non-finite values, iteration cap) must not seed the next one: its fields may
        // hold Inf/NaN, and so may the extrapolation history -- the next call starts cold
        ctx->haveFwd = ctx->haveAdj = false;
        ctx->lastItFwd = ctx->lastItAdj = 0;
    }
    if (!withAdjoint) for (int s = 0; s < S; ++s) ctx->itersLast[S + s] = 0;
}

// after an evaluation: wait for the whole stream, then the statistics
int collect_stats(hmcmt_ctx* ctx, bool withAdjoint) {
    HIPCHK(hipStreamSynchronize(ctx->stream));
    parse_stats(ctx, withAdjoint);
    ctx->statsPending = false;
    return 0;
}

int finish_status(hmcmt_ctx* ctx);

// An asynchronous evaluation (hmcmt_grad_device_async) leaves its records unread: they are picked up -- waiting only
// for the event behind its last k_solve_end, not for its gradient tail -- before the next evaluation is issued
// (its first poll wants the iteration counts, and k_solve_end will overwrite the records) or by hmcmt_wait.
int collect_pending(hmcmt_ctx* ctx) {
    if (!ctx->statsPending) return 0;
    HIPCHK(hipEventSynchronize(ctx->evRec));
    parse_stats(ctx, ctx->pendingAdj);
    ctx->statsPending = false;
    return finish_status(ctx);
}

int finish_status(hmcmt_ctx* ctx) {
    if (ctx->stats.status == HMCMT_ENOCONV) { ctx->err = "iterative solve did not converge within maxit"; return HMCMT_ENOCONV; }
    if (ctx->stats.status == HMCMT_EBREAKDOWN) { ctx->err = "Krylov breakdown or non-finite values (NaN/Inf model?)"; return HMCMT_EBREAKDOWN; }
    return 0;
}

}  // namespace

// ----------------------------------------------------------------------------------------------
// C ABI
// ----------------------------------------------------------------------------------------------
extern "C" {

void hmcmt_default_options(hmcmt_options* o) {
    if (!o) return;
    o->precond = HMCMT_PRECOND_FDM_JACOBI;
    o->maxit = 2000;
    o->tol = 1e-11;
    o->check_every = 2;
    o->verify = 0;
    o->warm_start = 2;
    o->fdm_precision = 0;
}

const char* hmcmt_last_error(const hmcmt_ctx* ctx) { return ctx ? ctx->err.c_str() : g_createError.c_str(); }

int hmcmt_destroy(hmcmt_ctx* ctx) {
    if (!ctx) return HMCMT_EINVAL;
    hipSetDevice(ctx->device);
    if (ctx->stream) hipStreamSynchronize(ctx->stream);
    if (ctx->side) hipStreamSynchronize(ctx->side);      // (side-stream work of an evaluation nobody waited for)
    if (ctx->side2) hipStreamSynchronize(ctx->side2);
    for (void* p : ctx->allocs) hipFree(p);
    for (hipEvent_t e : ctx->evPool) hipEventDestroy(e);
    if (ctx->h_nactive) hipHostFree(ctx->h_nactive);
    if (ctx->h_stall) hipHostFree(ctx->h_stall);
    for (auto& e : ctx->evPoll2) if (e) hipEventDestroy(e);
    if (ctx->h_rec) hipHostFree(ctx->h_rec);
    if (ctx->h_stage) hipHostFree(ctx->h_stage);
    if (ctx->h_lfFlag) hipHostFree(ctx->h_lfFlag);
    if (ctx->evModel) hipEventDestroy(ctx->evModel);
    if (ctx->evSens) hipEventDestroy(ctx->evSens);
    if (ctx->evExtF) hipEventDestroy(ctx->evExtF);
    if (ctx->evPiv) hipEventDestroy(ctx->evPiv);
    if (ctx->evPoll) hipEventDestroy(ctx->evPoll);
    if (ctx->evRec) hipEventDestroy(ctx->evRec);
    if (ctx->side2) hipStreamDestroy(ctx->side2);
    if (ctx->evExtA) hipEventDestroy(ctx->evExtA);
    if (ctx->side) hipStreamDestroy(ctx->side);
    if (ctx->stream) hipStreamDestroy(ctx->stream);
    delete ctx;
    return 0;
}

static int create_impl(hmcmt_ctx* ctx, int32_t device_id) {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { ctx->err = "no HIP device available (this library has no host compute path)"; return HMCMT_ENODEV; }
    if (device_id < 0 || device_id >= ndev) { ctx->err = "device_id out of range"; return HMCMT_ENODEV; }
    ctx->device = device_id;
    if (ctx->hp.NZP > MAXNZP) { ctx->err = "nz too large for the tridiagonal kernel (nz+1 > 1024)"; return HMCMT_EINVAL; }
    HIPCHK(hipSetDevice(device_id));
    HIPCHK(hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
    HIPCHK(hipStreamCreateWithFlags(&ctx->side, hipStreamNonBlocking));
    HIPCHK(hipEventCreateWithFlags(&ctx->evModel, hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&ctx->evSens, hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&ctx->evExtF, hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&ctx->evPiv, hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&ctx->evPoll, hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&ctx->evRec, hipEventDisableTiming));
    HIPCHK(hipStreamCreate(&ctx->side2));
    {
        const char* e = getenv("HMCMT_FUSED_FWD");
        ctx->fusedFwd = !(e && e[0] == '0');
        ctx->fusedFwdForce = e && e[0] == '2';
        if (const char* et = getenv("HMCMT_TWIST")) ctx->twistOn = et[0] != '0';
        if (const char* eb = getenv("HMCMT_FUSED_BACK")) ctx->fusedBack = eb[0] != '0';
        // (this kernel also has a few hundred bytes of static LDS: ask for less than the full 160 KB)
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(k_back_post<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024) == hipSuccess)
            ctx->maxLdsBack = 152 * 1024;
        else (void)hipGetLastError();
        if (const char* ew = getenv("HMCMT_JACOBI_W")) ctx->jacobiW = std::min(1.2, std::max(0.1, atof(ew)));
        if (const char* ep = getenv("HMCMT_EXTRAP_POINTS")) ctx->extrapNp = std::max(2, std::min(EXT_NP, atoi(ep)));
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(k_fdm_fwd<FW_NTW>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess &&
            hipFuncSetAttribute(reinterpret_cast<const void*>(k_fdm_fwd<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess)
            ctx->maxLds = 160 * 1024;
        else (void)hipGetLastError();
        // the stencil kernels' tiles can pass 64 KB on wide meshes
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(k_update_fused), hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024) != hipSuccess) (void)hipGetLastError();
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(k_spmv_fused), hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024) != hipSuccess) (void)hipGetLastError();
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(k_resid_pre), hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024) != hipSuccess) (void)hipGetLastError();
    }
    HIPCHK(hipEventCreateWithFlags(&ctx->evExtA, hipEventDisableTiming));
    const HostProblem& h = ctx->hp;
    View& v = ctx->v;
    v.ny = h.ny; v.nz = h.nz; v.NYP = h.NYP; v.NZP = h.NZP; v.nFreq = h.nFreq; v.S = h.S; v.nRx = h.nRx;
    v.nData = h.nData; v.nAC = h.nAC; v.nCell = h.nCell; v.zid = h.zid; v.vstride = (long)h.NZP * h.NYP;
    const size_t VS = (size_t)v.vstride, S = (size_t)h.S;
    int rc;
#define UP(field, vec) { decltype(vec)::value_type* p_ = nullptr; if ((rc = dupload(ctx, &p_, vec))) return rc; v.field = p_; }
    UP(yLen, h.yLen) UP(zLen, h.zLen) UP(omega, h.omega) UP(lam, h.lam) UP(sysOn, h.sysOn)
    UP(cell2act, h.cell2act) UP(bg, h.bg) UP(act, h.act)
    UP(rxIdn, h.rxIdn) UP(rxDy1, h.rxDy1) UP(rxDy2, h.rxDy2) UP(rxKL, h.rxKL) UP(rxKR, h.rxKR) UP(rxWL, h.rxWL) UP(rxWR, h.rxWR)
    UP(predSys, h.predSys) UP(predRx, h.predRx) UP(datSys, h.datSys) UP(datRx, h.datRx) UP(predKind, h.predKind) UP(datKind, h.datKind)
    UP(obs, h.obs) UP(dataW, h.dataW) UP(srStart, h.srStart) UP(srList, h.srList)
#undef UP
    {
        // fragment-order copies of V and V' (see k_transform)
        const int NT = h.NYP / 16, KG = h.NYP / 16;
        auto swz = [&](const std::vector<double>& B) {
            std::vector<double> o((size_t)h.NYP * h.NYP);
            for (int kg = 0; kg < KG; ++kg)
                for (int t = 0; t < NT; ++t)
                    for (int lane = 0; lane < 64; ++lane)
                        for (int i = 0; i < 4; ++i)
                            o[(((size_t)kg * NT + t) * 64 + lane) * 4 + i] =
                                B[(size_t)(16 * kg + 4 * (lane / 16) + i) * h.NYP + 16 * t + lane % 16];
            return o;
        };
        if ((rc = dupload(ctx, &ctx->d_V, swz(h.Vpad)))) return rc;
        if ((rc = dupload(ctx, &ctx->d_Vt, swz(h.Vtpad)))) return rc;
        // bf16 fragment order of k_transform_lp (k consumed 32 at a time, zero beyond NYP)
        const int KG32 = (h.NYP + 31) / 32;
        auto bf = [](double x) -> unsigned short {
            float f = (float)x; unsigned u; std::memcpy(&u, &f, 4);
            return (unsigned short)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
        };
        auto bf2f = [](unsigned short hval) -> float { unsigned u = (unsigned)hval << 16; float f; std::memcpy(&f, &u, 4); return f; };
        auto swzb = [&](const std::vector<double>& B, bool lo) {
            std::vector<unsigned short> o((size_t)KG32 * NT * 64 * 8, 0);
            for (int kg = 0; kg < KG32; ++kg)
                for (int t = 0; t < NT; ++t)
                    for (int lane = 0; lane < 64; ++lane)
                        for (int i = 0; i < 8; ++i) {
                            const int k = 32 * kg + 8 * (lane / 16) + i;
                            if (k >= h.NYP) continue;
                            const double x = B[(size_t)k * h.NYP + 16 * t + lane % 16];
                            const unsigned short hi = bf(x);
                            o[(((size_t)kg * NT + t) * 64 + lane) * 8 + i] = lo ? bf(x - (double)bf2f(hi)) : hi;
                        }
            return o;
        };
        unsigned short* tmp = nullptr;
        if ((rc = dupload(ctx, &tmp, swzb(h.Vpad, false)))) return rc;
        ctx->d_Vb = reinterpret_cast<u4v*>(tmp);
        if ((rc = dupload(ctx, &tmp, swzb(h.Vpad, true)))) return rc;
        ctx->d_Vbl = reinterpret_cast<u4v*>(tmp);
        if ((rc = dupload(ctx, &tmp, swzb(h.Vtpad, false)))) return rc;
        ctx->d_Vtb = reinterpret_cast<u4v*>(tmp);
        if ((rc = dupload(ctx, &tmp, swzb(h.Vtpad, true)))) return rc;
        ctx->d_Vtbl = reinterpret_cast<u4v*>(tmp);
    }
#define DA(ptr, n) if ((rc = dalloc(ctx, &(ptr), (n)))) return rc;
    DA(v.sigma, h.nCell) DA(v.sigMeanA, h.nz) DA(v.sigMeanG, h.nz)
    DA(v.cY, 2 * VS) DA(v.cZ, 2 * VS) DA(v.dK, 2 * VS) DA(v.dM, 2 * VS)
    DA(v.mzq, 2 * h.NZP) DA(v.dgz, 2 * h.NZP) DA(v.ofz, 2 * h.NZP) DA(v.mzs, 2 * h.NZP)
    DA(v.invp, S * VS) DA(v.X, S * VS) DA(v.Lam, S * VS) DA(v.R, S * VS)
    DA(v.Zrx, S * h.nRx) DA(v.rxN0, S * h.nRx) DA(v.rxD, S * h.nRx * 11) DA(v.rxCoef, S * h.nRx)
    DA(v.pred, h.nData) DA(v.vbar, h.nData) DA(v.misfitPart, h.nData)
    DA(v.srcB, S * 4) DA(v.wL, S * h.nz) DA(v.wR, S * h.nz) DA(v.colw, S * h.ny)
    DA(v.gL, S * h.nz) DA(v.gR, S * h.nz) DA(v.gMn, S * h.nz) DA(v.dBC, S * 2 * (size_t)h.nz * h.nz) DA(v.bcsL, S * h.nz) DA(v.bcsR, S * h.nz) DA(v.bcsB, S)
    DA(v.fwdTab, S * FWD_NQ * (size_t)h.nz * (h.ny + 1)) DA(v.sensTab, S * 15 * (size_t)(h.nz + 1))
    DA(v.sensEu, S * 3 * (size_t)(h.nz + 1)) DA(v.sensEd, S * 3 * (size_t)(h.nz + 1)) DA(v.sensMix, S * 12 * (size_t)h.nz)
    DA(v.sensDz1, S * 3 * (size_t)h.nz) DA(v.sensZ1, S * 3) DA(v.sensDead, S * 3)
    DA(v.qPart, S * h.ny) DA(v.gPart, 2 * (size_t)h.nCell) DA(v.gPartG, 2 * GRAD_NG * (size_t)h.nCell) DA(v.grad, h.nAC)
    DA(ctx->d_m, h.nAC) DA(ctx->d_misfit, 1) DA(ctx->d_cnt, 1)
    for (int q : h.sysOn) ctx->nSysOn += q;
    for (int kd = 0; kd < 2; ++kd) { DA(ctx->d_prevField[kd], (EXT_NP - 1) * S * VS) DA(ctx->d_mHist[kd], EXT_NP * (size_t)h.nAC) DA(ctx->d_ext[kd], EXT_PART + EXT_NS * EXT_NBLK) }
    Solver& k = ctx->sv;
    k.S = h.S; k.NYP = h.NYP; k.NZP = h.NZP; k.ny = h.ny; k.nz = h.nz; k.nFreq = h.nFreq; k.vstride = v.vstride;
    k.NB = std::max(1, std::min(32, (1024 + h.S - 1) / h.S));
    k.chunk = (v.vstride + k.NB - 1) / k.NB;
    // rows per tile of the stencil kernels: about 512 workgroups per launch, at most 8 rows (measured at cfg3:
    // 4 rows 484, 6 rows 489, 8 rows 492, 12 rows 456 steps/s), at least what MAXNB partial sums per system allow
    {
        // ... and few enough that two workgroups of k_update_fused fit a CU's LDS ((2 RT + 2) rows of NYP complex128)
        const int ldsRows = std::max(1, (int)((80 * 1024 / ((size_t)h.NYP * sizeof(cplx)) - 2) / 2));
        const int want = std::min(std::min(8, ldsRows), std::max(2, (int)(((long)(h.nz - 1) * h.S + 511) / 512)));
        const char* e = getenv("HMCMT_RT");
        k.RT = std::max(e ? atoi(e) : want, (h.nz - 1 + MAXNB - 1) / MAXNB);
    }
    k.NTR = (h.nz - 1 + k.RT - 1) / k.RT;
    k.omega = v.omega; k.cY = v.cY; k.cZ = v.cZ; k.dK = v.dK; k.dM = v.dM; k.ofz = v.ofz; k.invp = v.invp;
    k.r = v.R;
    DA(k.p, S * VS) DA(k.q, S * VS) DA(k.z, S * VS) DA(k.y, S * VS) DA(k.t, S * VS) DA(k.dinv, S * VS)
    DA(k.t32, S * VS + 64) DA(k.y32, S * VS + 64) DA(ctx->d_invp32, S * VS)
    DA(k.p2, S * VS) DA(k.r2, S * VS) DA(k.partPQ, S * MAXNB) DA(k.rho2, 2 * S)
    k.invp32 = ctx->d_invp32;
    DA(k.partA, S * MAXNB) DA(k.partB, S * MAXNB) DA(ctx->d_partZZ, S * MAXNB)
    DA(ctx->d_partRes, S * MAXNB) DA(ctx->d_partBn, S * MAXNB)
    DA(k.rho, S) DA(k.alphaBeta, S) DA(k.active, S) DA(k.iters, S) DA(k.status, S) DA(k.nactive, 1) DA(k.errEst, S) DA(k.errRef, S) DA(k.errRefIt, S)
    DA(ctx->d_b, S * VS)
    DA(ctx->d_fieldsOut, (size_t)h.nFreq * (h.ny + 1) * (h.nz + 1))
#undef DA
    HIPCHK(hipHostMalloc((void**)&ctx->h_nactive, sizeof(int), hipHostMallocMapped));
    HIPCHK(hipHostGetDevicePointer((void**)&k.nactHost, ctx->h_nactive, 0));
    HIPCHK(hipHostMalloc((void**)&ctx->h_stall, sizeof(int), hipHostMallocMapped));
    *ctx->h_stall = 0;
    HIPCHK(hipHostGetDevicePointer((void**)&k.stallHost, ctx->h_stall, 0));
    k.stallIt = STALL_IT;
    if (const char* es = getenv("HMCMT_STALL_IT")) k.stallIt = std::max(1, atoi(es));   // (tests force the fp64 restart with a short window)
    for (auto& e : ctx->evPoll2) HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    HIPCHK(hipHostMalloc((void**)&ctx->h_rec, sizeof(double) * 4 * h.S, hipHostMallocMapped));
    HIPCHK(hipHostGetDevicePointer((void**)&ctx->d_recHost, ctx->h_rec, 0));
    ctx->stageDoubles = (size_t)h.nAC * 4 + (size_t)h.nData * 2 + 16;
    HIPCHK(hipHostMalloc((void**)&ctx->h_stage, sizeof(double) * ctx->stageDoubles));
    ctx->itersLast.assign(2 * h.S, 0);
    // constant halves of the stencils: TE stiffness (mesh only), TM mass (mesh only)
    const int nodes = v.NZP * (v.ny + 1);
    hipLaunchKernelGGL(k_coef, grid1(nodes, 256), dim3(256), 0, ctx->stream, v, 1, 0, 0, 1);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return 0;
}

static const char* options_error(const hmcmt_options* o) {
    if (o->precond < HMCMT_PRECOND_JACOBI || o->precond > HMCMT_PRECOND_FDM_JACOBI) return "unknown preconditioner";
    if (o->fdm_precision != 0 && o->fdm_precision != 1) return "fdm_precision must be 0 (bf16/fp32) or 1 (fp64)";
    if (!(o->tol > 0) || o->maxit < 1) return "tol must be > 0 and maxit >= 1";
    if (o->warm_start < 0 || o->warm_start > 2) return "warm_start must be 0, 1 or 2";
    return nullptr;
}

int hmcmt_create(hmcmt_ctx** out, int32_t device_id, int64_t ny, int64_t nz, const double* yLen,
                 const double* zLen, const double* origin, int64_t nFreq, const double* freqs, int64_t nRx,
                 const double* rxY, const double* rxZ, int64_t nComp, const int64_t* compMode, int64_t nData,
                 const int64_t* freqID, const int64_t* rxID, const int64_t* dtID, const uint8_t* dataID,
                 const double* obs, const double* dataW, int64_t nAC, const int64_t* activeIdx,
                 const double* bgModel, const hmcmt_options* opts) {
    if (!out) { g_createError = "null ctx pointer"; return HMCMT_EINVAL; }
    *out = nullptr;
    if (!yLen || !zLen || !origin || !freqs || !rxY || !rxZ || !compMode || !dataID || !activeIdx || !bgModel ||
        (nData > 0 && (!freqID || !rxID || !dtID || !obs || !dataW))) {
        g_createError = "null input array"; return HMCMT_EINVAL;
    }
    if (opts) if (const char* e = options_error(opts)) { g_createError = e; return HMCMT_EINVAL; }
    hmcmt_ctx* ctx = new hmcmt_ctx();
    hmcmt_default_options(&ctx->opt);
    if (opts) ctx->opt = *opts;
    if (!ctx->hp.build(ny, nz, yLen, zLen, origin, nFreq, freqs, nRx, rxY, rxZ, nComp, compMode, nData, freqID,
                       rxID, dtID, dataID, obs, dataW, nAC, activeIdx, bgModel)) {
        g_createError = ctx->hp.error;
        delete ctx;
        return HMCMT_EINVAL;
    }
    int rc = create_impl(ctx, device_id);
    if (rc) {
        g_createError = ctx->err;
        hmcmt_destroy(ctx);
        return rc;
    }
    ctx->sv.splitT = fdm_fwd_ntw(ctx) > 0;
    ctx->sv.twist = ctx->v.twist = ctx->sv.splitT && ctx->twistOn;     // the fused forward kernel sweeps both ways at once
    *out = ctx;
    return 0;
}

int hmcmt_set_options(hmcmt_ctx* ctx, const hmcmt_options* o) {
    if (!ctx || !o) return HMCMT_EINVAL;
    if (const char* e = options_error(o)) { ctx->err = e; return HMCMT_EINVAL; }
    ctx->opt = *o;
    ctx->lastItFwd = ctx->lastItAdj = 0;
    ctx->haveFwd = ctx->haveAdj = false;
    ctx->memo[0].valid = ctx->memo[1].valid = false;
    return 0;
}

int hmcmt_get_stats(const hmcmt_ctx* ctx, hmcmt_stats* out) {
    if (!ctx || !out) return HMCMT_EINVAL;
    *out = ctx->stats;
    return 0;
}

int hmcmt_get_iters(const hmcmt_ctx* ctx, int32_t* iters) {
    if (!ctx || !iters) return HMCMT_EINVAL;
    for (size_t i = 0; i < ctx->itersLast.size(); ++i) iters[i] = ctx->itersLast[i];
    return 0;
}

int hmcmt_profile_counters(hmcmt_ctx* ctx, int64_t* out) {
    if (!ctx || !out) return HMCMT_EINVAL;
    HIPCHK(hipSetDevice(ctx->device));
    prof_collect(ctx);
    unsigned long long c = 0;
    HIPCHK(hipMemcpy(&c, ctx->d_cnt, sizeof c, hipMemcpyDeviceToHost));
    out[0] = (int64_t)c; out[1] = ctx->profStartSys; out[2] = ctx->profEvals; out[3] = ctx->profSolves;
    return 0;
}

int hmcmt_dims(const hmcmt_ctx* ctx, int32_t* o) {
    if (!ctx || !o) return HMCMT_EINVAL;
    o[0] = ctx->v.NYP; o[1] = ctx->v.NZP; o[2] = ctx->v.S; o[3] = ctx->v.ny; o[4] = ctx->v.nz; o[5] = ctx->v.zid; o[6] = ctx->sv.NB;
    return 0;
}

int hmcmt_grad_device(hmcmt_ctx* ctx, const double* d_m, double* d_pred, double* d_misfit, double* d_grad) {
    if (!ctx || !d_m || !d_grad) return HMCMT_EINVAL;
    HIPCHK(hipSetDevice(ctx->device));
    int rc = evaluate(ctx, d_m, true, d_pred, d_misfit, d_grad);
    if (rc) return rc;
    if ((rc = collect_stats(ctx, true))) return rc;      // includes the stream synchronisation
    prof_collect(ctx);
    return finish_status(ctx);
}

int hmcmt_grad_device_async(hmcmt_ctx* ctx, const double* d_m, double* d_pred, double* d_misfit, double* d_grad) {
    if (!ctx || !d_m || !d_grad) return HMCMT_EINVAL;
    HIPCHK(hipSetDevice(ctx->device));
    int rc = evaluate(ctx, d_m, true, d_pred, d_misfit, d_grad);
    if (rc) return rc;
    ctx->statsPending = true;
    ctx->pendingAdj = true;
    return 0;
}

static int leapfrog_flag(hmcmt_ctx* ctx);

int hmcmt_wait(hmcmt_ctx* ctx) {
    if (!ctx) return HMCMT_EINVAL;
    HIPCHK(hipSetDevice(ctx->device));
    const int rc = collect_pending(ctx);
    HIPCHK(hipStreamSynchronize(ctx->stream));
    prof_collect(ctx);
    const int rf = leapfrog_flag(ctx);
    return rc ? rc : rf;
}

int hmcmt_forward_device(hmcmt_ctx* ctx, const double* d_m, double* d_pred, double* d_misfit) {
    if (!ctx || !d_m) return HMCMT_EINVAL;
    HIPCHK(hipSetDevice(ctx->device));
    int rc = evaluate(ctx, d_m, false, d_pred, d_misfit, nullptr);
    if (rc) return rc;
    if ((rc = collect_stats(ctx, false))) return rc;
    prof_collect(ctx);
    return finish_status(ctx);
}

static hmcmt_ctx::Memo* memo_find(hmcmt_ctx* ctx, const double* m, bool needGrad) {
    const size_t bytes = sizeof(double) * ctx->v.nAC;
    for (auto& e : ctx->memo)
        if (e.valid && (!needGrad || e.hasGrad) && std::memcmp(e.m.data(), m, bytes) == 0) return &e;
    return nullptr;
}
static void memo_store(hmcmt_ctx* ctx, const double* m, const double* pred, double misfit, const double* grad) {
    const int nAC = ctx->v.nAC, nData = ctx->v.nData;
    hmcmt_ctx::Memo* e = memo_find(ctx, m, false);
    if (e && e->hasGrad && !grad) return;                 // keep the richer entry of the same model
    if (!e) { e = &ctx->memo[ctx->memoNext]; ctx->memoNext ^= 1; }
    e->m.assign(m, m + nAC);
    e->pred.assign(pred, pred + 2 * nData);
    e->misfit = misfit;
    e->hasGrad = grad != nullptr;
    if (grad) e->grad.assign(grad, grad + nAC);
    e->stats = ctx->stats;
    e->valid = true;
}

static int host_eval(hmcmt_ctx* ctx, const double* m, double* pred, double* misfit, double* grad, bool wantGrad) {
    if (!ctx || !m || (wantGrad && !grad)) return HMCMT_EINVAL;
    HIPCHK(hipSetDevice(ctx->device));
    const int nAC = ctx->v.nAC, nData = ctx->v.nData;
    for (int i = 0; i < nAC; ++i)
        if (!std::isfinite(m[i])) { ctx->err = "non-finite model value"; return HMCMT_EBREAKDOWN; }
    if (!ctx->opt.verify)
        if (const hmcmt_ctx::Memo* e = memo_find(ctx, m, wantGrad)) {     // this very model was evaluated a moment ago
            if (pred) std::memcpy(pred, e->pred.data(), sizeof(cplx) * nData);
            if (misfit) *misfit = e->misfit;
            if (wantGrad) std::memcpy(grad, e->grad.data(), sizeof(double) * nAC);
            ctx->stats = e->stats;                                           // (this call itself iterated nothing)
            ctx->stats.iters_fwd_max = ctx->stats.iters_adj_max = ctx->stats.iters_fwd_sum = ctx->stats.iters_adj_sum = 0;
            ++ctx->memoHits;
            return 0;
        }
    std::memcpy(ctx->h_stage, m, sizeof(double) * nAC);
    HIPCHK(hipMemcpyAsync(ctx->d_m, ctx->h_stage, sizeof(double) * nAC, hipMemcpyHostToDevice, ctx->stream));
    int rc = evaluate(ctx, ctx->d_m, wantGrad, nullptr, nullptr, nullptr);
    if (rc) return rc;
    double* hs = ctx->h_stage + nAC;
    HIPCHK(hipMemcpyAsync(hs, ctx->v.pred, sizeof(cplx) * nData, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipMemcpyAsync(hs + 2 * nData, ctx->d_misfit, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    if (wantGrad) HIPCHK(hipMemcpyAsync(hs + 2 * nData + 1, ctx->v.grad, sizeof(double) * nAC, hipMemcpyDeviceToHost, ctx->stream));
    if ((rc = collect_stats(ctx, wantGrad))) return rc;   // includes the stream synchronisation
    prof_collect(ctx);
    if (pred) std::memcpy(pred, hs, sizeof(cplx) * nData);
    if (misfit) *misfit = hs[2 * nData];
    if (wantGrad) std::memcpy(grad, hs + 2 * nData + 1, sizeof(double) * nAC);
    rc = finish_status(ctx);
    if (rc == 0) memo_store(ctx, m, hs, hs[2 * nData], wantGrad ? hs + 2 * nData + 1 : nullptr);
    return rc;
}

int hmcmt_grad(hmcmt_ctx* ctx, const double* m, double* pred, double* misfit, double* grad) {
    return host_eval(ctx, m, pred, misfit, grad, true);
}
int hmcmt_forward(hmcmt_ctx* ctx, const double* m, double* pred, double* misfit) {
    return host_eval(ctx, m, pred, misfit, nullptr, false);
}

int hmcmt_get_fields(hmcmt_ctx* ctx, int32_t adjoint, double* exTE, double* hxTM) {
    if (!ctx) return HMCMT_EINVAL;
    if (!ctx->haveModel) { ctx->err = "no evaluation has been run yet"; return HMCMT_EINVAL; }
    HIPCHK(hipSetDevice(ctx->device));
    const View& v = ctx->v;
    const int nn = (v.ny + 1) * (v.nz + 1);
    const cplx* src = adjoint ? v.Lam : v.X;
    for (int mode = 0; mode < 2; ++mode) {
        double* dst = mode == 0 ? exTE : hxTM;
        if (!dst) continue;
        hipLaunchKernelGGL(k_unpad, dim3((nn + 255) / 256, v.nFreq), dim3(256), 0, ctx->stream, v, src, ctx->d_fieldsOut, mode * v.nFreq);
        HIPCHK(hipMemcpyAsync(dst, ctx->d_fieldsOut, sizeof(cplx) * (size_t)nn * v.nFreq, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(hipStreamSynchronize(ctx->stream));
    }
    return 0;
}

int hmcmt_profile(hmcmt_ctx* ctx, int32_t enable) {
    if (!ctx) return HMCMT_EINVAL;
    ctx->profMask = 0;
    ctx->evUsed = 0;
    ctx->profOverheadMs = 0.0;
    if (enable) {
        // An event pair around a launch also times the dispatch latency of that launch.  Calibrate it
        // with a null kernel on the same stream (median of 33) and subtract it from every sample.
        HIPCHK(hipSetDevice(ctx->device));
        // steady state of a busy queue: N null launches with an event after each, one sync at the end
        const int N = 48;
        std::vector<hipEvent_t> ev(N + 1);
        for (auto& e : ev) HIPCHK(hipEventCreate(&e));
        hipLaunchKernelGGL(k_null, dim3(1), dim3(64), 0, ctx->stream);
        HIPCHK(hipEventRecord(ev[0], ctx->stream));
        for (int i = 0; i < N; ++i) {
            hipLaunchKernelGGL(k_null, dim3(1), dim3(64), 0, ctx->stream);
            HIPCHK(hipEventRecord(ev[i + 1], ctx->stream));
        }
        HIPCHK(hipStreamSynchronize(ctx->stream));
        std::vector<float> t;
        for (int i = 0; i < N; ++i) { float ms = 0; HIPCHK(hipEventElapsedTime(&ms, ev[i], ev[i + 1])); t.push_back(ms); }
        std::sort(t.begin(), t.end());
        // The null kernel itself runs ~1.5 us (rocprofv3).  Of the remaining launch+event period about 60 %
        // also precedes a LONG kernel inside its bracket (the rest overlaps its execution): factor fitted
        // once against the rocprofv3 average of k_transform (profiles/r01_bench_cfg3_kernel_stats.csv).
        ctx->profOverheadMs = std::max(0.0, 0.6 * ((double)t[t.size() / 2] - 0.0015));
        for (auto& e : ev) hipEventDestroy(e);
    }
    ctx->profMask = (unsigned)enable;
    for (int i = 0; i < HMCMT_NCAT; ++i) { ctx->profMs[i] = 0; ctx->profN[i] = 0; }
    ctx->profStartSys = ctx->profEvals = ctx->profSolves = 0;
    HIPCHK(hipMemsetAsync(ctx->d_cnt, 0, sizeof(unsigned long long), ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return 0;
}

int hmcmt_profile_every(hmcmt_ctx* ctx, int32_t n) {
    if (!ctx || n < 1) return HMCMT_EINVAL;
    ctx->profEvery = n;
    return 0;
}

int hmcmt_profile_read(hmcmt_ctx* ctx, double* ms, int64_t* launches) {
    if (!ctx || !ms || !launches) return HMCMT_EINVAL;
    prof_collect(ctx);
    for (int i = 0; i < HMCMT_NCAT; ++i) { ms[i] = ctx->profMs[i]; launches[i] = ctx->profN[i]; }
    return 0;
}

static int set_all_active(hmcmt_ctx* ctx) {
    std::vector<int> one(ctx->v.S, 1);
    HIPCHK(hipMemcpy(ctx->sv.active, one.data(), sizeof(int) * one.size(), hipMemcpyHostToDevice));
    return 0;
}

int hmcmt_debug_transform(hmcmt_ctx* ctx, int32_t which, const double* A, double* C) {
    if (!ctx || !A || !C) return HMCMT_EINVAL;
    HIPCHK(hipSetDevice(ctx->device));
    const size_t bytes = (size_t)ctx->v.S * ctx->v.vstride * sizeof(cplx);
    HIPCHK(hipMemcpy(ctx->sv.p, A, bytes, hipMemcpyHostToDevice));
    if (which >= 2) {
        // mixed-precision kernel: q = fp32(A) * V (which == 2) or V' (which == 3), split-bf16 operands, fp32 accumulation
        int rc = set_all_active(ctx);
        if (rc) return rc;
        hipLaunchKernelGGL(k_to_c64, dim3(ctx->sv.NB, ctx->v.S), dim3(VBLOCK), 0, ctx->stream, ctx->sv, ctx->sv.p);
        if ((rc = launch_transform_lp<1>(ctx, ctx->sv.t32, which == 3, ctx->sv.q, nullptr))) return rc;
        HIPCHK(hipStreamSynchronize(ctx->stream));
        HIPCHK(hipMemcpy(C, ctx->sv.q, bytes, hipMemcpyDeviceToHost));
        return 0;
    }
    launch_transform(ctx, ctx->sv.p, which ? ctx->d_Vt : ctx->d_V, ctx->sv.q, nullptr);
    HIPCHK(hipStreamSynchronize(ctx->stream));
    HIPCHK(hipMemcpy(C, ctx->sv.q, bytes, hipMemcpyDeviceToHost));
    return 0;
}

int hmcmt_debug_spmv(hmcmt_ctx* ctx, const double* p, double* q) {
    if (!ctx || !p || !q) return HMCMT_EINVAL;
    if (!ctx->haveModel) { ctx->err = "no evaluation has been run yet"; return HMCMT_EINVAL; }
    HIPCHK(hipSetDevice(ctx->device));
    const size_t bytes = (size_t)ctx->v.S * ctx->v.vstride * sizeof(cplx);
    HIPCHK(hipMemcpy(ctx->sv.p, p, bytes, hipMemcpyHostToDevice));
    HIPCHK(hipMemset(ctx->sv.q, 0, bytes));
    int rc = set_all_active(ctx);
    if (rc) return rc;
    hipLaunchKernelGGL(k_spmv, dim3(ctx->sv.NB, ctx->v.S), dim3(VBLOCK), 0, ctx->stream, ctx->sv);
    HIPCHK(hipStreamSynchronize(ctx->stream));
    HIPCHK(hipMemcpy(q, ctx->sv.q, bytes, hipMemcpyDeviceToHost));
    return 0;
}

int hmcmt_debug_precond(hmcmt_ctx* ctx, const double* r, double* z) {
    if (!ctx || !r || !z) return HMCMT_EINVAL;
    if (!ctx->haveModel) { ctx->err = "no evaluation has been run yet"; return HMCMT_EINVAL; }
    HIPCHK(hipSetDevice(ctx->device));
    const size_t bytes = (size_t)ctx->v.S * ctx->v.vstride * sizeof(cplx);
    HIPCHK(hipMemcpy(ctx->sv.r, r, bytes, hipMemcpyHostToDevice));
    int rc = set_all_active(ctx);
    if (rc) return rc;
    apply_precond(ctx);
    HIPCHK(hipStreamSynchronize(ctx->stream));
    HIPCHK(hipMemcpy(z, ctx->sv.z, bytes, hipMemcpyDeviceToHost));
    return 0;
}

// forward half of the mixed-precision FDM stage on a caller-supplied vector: out[0..n) = fused kernel,
// out[n..2n) = separate transform + tridiagonal kernels (complex64 pairs widened to double), n = S*vstride
int hmcmt_debug_fdm_fwd(hmcmt_ctx* ctx, const double* t, double* out) {
    if (!ctx || !t || !out) return HMCMT_EINVAL;
    if (!ctx->haveModel) { ctx->err = "no evaluation has been run yet"; return HMCMT_EINVAL; }
    HIPCHK(hipSetDevice(ctx->device));
    Solver& k = ctx->sv;
    const size_t n = (size_t)ctx->v.S * ctx->v.vstride;
    HIPCHK(hipMemcpy(k.r, t, n * sizeof(cplx), hipMemcpyHostToDevice));
    int rc = set_all_active(ctx);
    if (rc) return rc;
    std::vector<float2> h(n);
    if (getenv("HMCMT_FWD_STAMPS")) {
        const int gx = (k.NYP / 16 + FW_NTW - 1) / FW_NTW, nb = gx * k.S;
        long long* d_st = nullptr;
        HIPCHK(hipMalloc((void**)&d_st, sizeof(long long) * 8 * nb));
        HIPCHK(hipMemset(d_st, 0, sizeof(long long) * 8 * nb));
        const size_t lds = fdm_fwd_lds(k, FW_NTW, k.twist);
        const int G = (k.NZP + 7) / 8, per = (G + 7) / 8, nw = (G + per - 1) / per;
        hipLaunchKernelGGL(k_to_c64, dim3(k.NB, k.S), dim3(VBLOCK), 0, ctx->stream, k, k.r);
        for (int rep = 0; rep < 3; ++rep)
            hipLaunchKernelGGL(k_fdm_fwd<FW_NTW>, dim3(gx * k.S), dim3(64 * nw), lds, ctx->stream, k, k.t32, ctx->d_Vb, ctx->d_Vbl,
                               ctx->d_invp32, k.y32, d_st);
        HIPCHK(hipStreamSynchronize(ctx->stream));
        std::vector<long long> st(8 * (size_t)nb);
        HIPCHK(hipMemcpy(st.data(), d_st, sizeof(long long) * 8 * nb, hipMemcpyDeviceToHost));
        hipFree(d_st);
        double d[4] = {0, 0, 0, 0};
        long long tmin = st[0], tmax = st[4];
        for (int b = 0; b < nb; ++b) {
            for (int i = 0; i < 4; ++i) d[i] += double(st[8 * b + i + 1] - st[8 * b + i]) / nb;
            tmin = std::min(tmin, st[8 * b]); tmax = std::max(tmax, st[8 * b + 4]);
        }
        fprintf(stderr, "k_fdm_fwd stamps (s_memtime ticks, mean over %d blocks): transform %.0f premul %.0f chain %.0f write %.0f | first start -> last end %lld\n",
                nb, d[0], d[1], d[2], d[3], tmax - tmin);
    }
    const bool keep = ctx->fusedFwd;
    const int keepSplit = k.splitT;
    for (int pass = 0; pass < 2; ++pass) {
        ctx->fusedFwd = pass == 0;
        k.splitT = pass == 0 && fdm_fwd_ntw(ctx) > 0;      // the operand format goes with the path
        hipLaunchKernelGGL(k_to_c64, dim3(k.NB, k.S), dim3(VBLOCK), 0, ctx->stream, k, k.r);
        HIPCHK(hipMemsetAsync(k.y32, 0xff, n * sizeof(float2), ctx->stream));
        rc = launch_fdm_fwd(ctx);
        const float2* res = k.y32;
        if (!rc && k.splitT) {                              // pre-split result -> complex64 (hi + lo), into t32
            hipLaunchKernelGGL(k_unsplit, dim3(512), dim3(256), 0, ctx->stream, k, k.y32, k.t32);
            res = k.t32;
        }
        ctx->fusedFwd = keep;
        k.splitT = keepSplit;
        if (rc) return rc;
        HIPCHK(hipMemcpyAsync(h.data(), res, n * sizeof(float2), hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(hipStreamSynchronize(ctx->stream));
        for (size_t i = 0; i < n; ++i) { out[2 * (pass * n + i)] = h[i].x; out[2 * (pass * n + i) + 1] = h[i].y; }
    }
    return 0;
}

// back half of the FDM stage + post-smoother on caller-supplied vectors: out[0..n) = fused kernel (k_back_post),
// out[n..2n) = k_transform_lp<2> + k_post, n = S*vstride complex; sums[0..2) / [2..4) = sum over systems of the
// r't partials (re, im) of the two paths, sums[4] / [5] = of the |t|^2 partials
int hmcmt_debug_back_post(hmcmt_ctx* ctx, const double* y, const double* r, double* out, double* sums) {
    if (!ctx || !y || !r || !out || !sums) return HMCMT_EINVAL;
    if (!ctx->haveModel) { ctx->err = "no evaluation has been run yet"; return HMCMT_EINVAL; }
    HIPCHK(hipSetDevice(ctx->device));
    Solver& k = ctx->sv;
    const size_t n = (size_t)ctx->v.S * ctx->v.vstride;
    int rc = set_all_active(ctx);
    if (rc) return rc;
    const bool keep = ctx->fusedBack;
    std::vector<cplx> pa((size_t)k.S * MAXNB);
    std::vector<double> pz((size_t)k.S * MAXNB);
    for (int pass = 0; pass < 2; ++pass) {
        // y -> y32 in the operand format of the path (through t32, the buffer k_to_c64 writes), r -> k.r
        HIPCHK(hipMemcpy(k.r, y, n * sizeof(cplx), hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k_to_c64, dim3(k.NB, k.S), dim3(VBLOCK), 0, ctx->stream, k, k.r);
        HIPCHK(hipMemcpyAsync(k.y32, k.t32, n * sizeof(float2), hipMemcpyDeviceToDevice, ctx->stream));
        HIPCHK(hipMemcpyAsync(k.r, r, n * sizeof(cplx), hipMemcpyHostToDevice, ctx->stream));
        HIPCHK(hipMemsetAsync(k.t, 0xff, n * sizeof(cplx), ctx->stream));
        ctx->fusedBack = pass == 0;
        rc = launch_back_post(ctx);
        if (!rc && pass == 0 && getenv("HMCMT_BACK_STAMPS")) {
            const int nb = ((k.nz - 1 + BP_OWN - 1) / BP_OWN) * k.S;
            long long* d_st = nullptr;
            HIPCHK(hipMalloc((void**)&d_st, sizeof(long long) * 8 * nb));
            HIPCHK(hipMemset(d_st, 0, sizeof(long long) * 8 * nb));
            ctx->backStamps = d_st;
            for (int rep = 0; rep < 3 && !rc; ++rep) rc = launch_back_post(ctx);
            ctx->backStamps = nullptr;
            HIPCHK(hipStreamSynchronize(ctx->stream));
            std::vector<long long> st(8 * (size_t)nb);
            HIPCHK(hipMemcpy(st.data(), d_st, sizeof(long long) * 8 * nb, hipMemcpyDeviceToHost));
            hipFree(d_st);
            double d[7] = {0, 0, 0, 0, 0, 0, 0};
            long long tmin = st[0], tmax = st[6];
            for (int b = 0; b < nb; ++b) {
                for (int i = 0; i < 6; ++i) d[i] += double(st[8 * b + i + 1] - st[8 * b + i]) / nb;
                d[6] += double(st[8 * b + 7] - st[8 * b]) / nb;
                tmin = std::min(tmin, st[8 * b]); tmax = std::max(tmax, st[8 * b + 6]);
            }
            fprintf(stderr, "k_back_post stamps (s_memtime ticks, mean over %d blocks): stage %.0f mfma(first pair) %.0f epilogue+rest %.0f barrier %.0f stencil %.0f reduce %.0f | start -> loads issued %.0f | first start -> last end %lld\n",
                    nb, d[0], d[1], d[2], d[3], d[4], d[5], d[6], tmax - tmin);
        }
        ctx->fusedBack = keep;
        if (rc) return rc;
        HIPCHK(hipMemcpyAsync(out + 2 * pass * n, k.t, n * sizeof(cplx), hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(hipMemcpyAsync(pa.data(), k.partA, pa.size() * sizeof(cplx), hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(hipMemcpyAsync(pz.data(), ctx->d_partZZ, pz.size() * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(hipStreamSynchronize(ctx->stream));
        double are = 0, aim = 0, zz = 0;
        for (int s = 0; s < k.S; ++s)
            for (int b = 0; b < k.NB; ++b) { are += pa[(size_t)s * MAXNB + b].re; aim += pa[(size_t)s * MAXNB + b].im; zz += pz[(size_t)s * MAXNB + b]; }
        sums[2 * pass] = are; sums[2 * pass + 1] = aim; sums[4 + pass] = zz;
    }
    return 0;
}

int hmcmt_set_prior(hmcmt_ctx* ctx, const double* mref, const int64_t* rowptr, const int64_t* colind,
                    const double* val, const double* invM) {
    if (!ctx || !mref || !rowptr || !colind || !val || !invM) return HMCMT_EINVAL;
    HIPCHK(hipSetDevice(ctx->device));
    const int n = ctx->v.nAC;
    const int64_t nnz = rowptr[n];
    if (rowptr[0] != 0 || nnz < 0) { ctx->err = "Wm row pointer must be 0-based CSR"; return HMCMT_EINVAL; }
    for (int64_t t = 0; t < nnz; ++t)
        if (colind[t] < 0 || colind[t] >= n) { ctx->err = "Wm column index out of range"; return HMCMT_EINVAL; }
    std::vector<double> v_mref(mref, mref + n), v_invM(invM, invM + n), v_val(val, val + nnz);
    std::vector<long long> v_row(rowptr, rowptr + n + 1), v_col(colind, colind + nnz);
    int rc;
    HIPCHK(hipStreamSynchronize(ctx->stream));              // (a trajectory may still be reading the previous prior)
    // a repeated call replaces the previous prior: its buffers are released, not kept until hmcmt_destroy
    for (void* old : {(void*)ctx->d_mref, (void*)ctx->d_invM, (void*)ctx->d_wmVal, (void*)ctx->d_wmRow, (void*)ctx->d_wmCol}) {
        if (!old) continue;
        auto it = std::find(ctx->allocs.begin(), ctx->allocs.end(), old);
        if (it != ctx->allocs.end()) ctx->allocs.erase(it);
        hipFree(old);
    }
    ctx->d_mref = ctx->d_invM = ctx->d_wmVal = nullptr; ctx->d_wmRow = ctx->d_wmCol = nullptr;
    ctx->havePrior = false;
    if ((rc = dupload(ctx, &ctx->d_mref, v_mref))) return rc;
    if ((rc = dupload(ctx, &ctx->d_invM, v_invM))) return rc;
    if ((rc = dupload(ctx, &ctx->d_wmVal, v_val))) return rc;
    if ((rc = dupload(ctx, &ctx->d_wmRow, v_row))) return rc;
    if ((rc = dupload(ctx, &ctx->d_wmCol, v_col))) return rc;
    if (!ctx->d_p) {
        if ((rc = dalloc(ctx, &ctx->d_p, (size_t)n))) return rc;
        if ((rc = dalloc(ctx, &ctx->d_mcur, (size_t)n))) return rc;
        if ((rc = dalloc(ctx, &ctx->d_g, (size_t)n))) return rc;
        if ((rc = dalloc(ctx, &ctx->d_lfPart, (size_t)LFNB))) return rc;
        if ((rc = dalloc(ctx, &ctx->d_lfScal, (size_t)4))) return rc;
        if ((rc = dalloc(ctx, &ctx->d_lfFlag, (size_t)1))) return rc;
        if ((rc = dalloc(ctx, &ctx->d_gStart, (size_t)n))) return rc;
        HIPCHK(hipHostMalloc((void**)&ctx->h_lfFlag, sizeof(int), hipHostMallocDefault));
        *ctx->h_lfFlag = 0;
    }
    ctx->lfHaveGrad = false;
    HIPCHK(hipStreamSynchronize(ctx->stream));
    ctx->havePrior = true;
    return 0;
}

// One trajectory on device vectors d_m, d_p (updated in place).  startGrad: 0 evaluate the gradient at the start
// model, 1 ctx->d_g holds it (the start model is the end model of the previous trajectory), 2 ctx->d_gStart holds it
// (the start model is the start model of the previous trajectory: a rejected proposal).  Returns with everything
// enqueued on the context's stream and the solver status of every evaluation collected.
static int leapfrog_core(hmcmt_ctx* ctx, double* d_m, double* d_p, double dt, int32_t L, double regParam, double lnSigMin,
                         double lnSigMax, int startGrad, double* d_pred, double* d_misfit, int* evalsOut) {
    const int n = ctx->v.nAC;
    hipStream_t st = ctx->stream;
    HIPCHK(hipMemsetAsync(ctx->d_lfFlag, 0, sizeof(int), st));
    LfView lf{n, ctx->d_mref, ctx->d_invM, ctx->d_wmVal, ctx->d_wmRow, ctx->d_wmCol, d_m, d_p, ctx->d_g,
              ctx->d_lfPart, ctx->d_lfScal, ctx->d_lfFlag};
    const dim3 g1((n + 127) / 128), b1(128);
    int evals = 0;
    int rc = 0;
    if (startGrad == 0) {
        rc = evaluate(ctx, d_m, true, d_pred, d_misfit, ctx->d_g);
        if (rc) return rc;
        if ((rc = collect_stats(ctx, true))) return rc;
        if ((rc = finish_status(ctx))) return rc;
    } else if (startGrad == 2) {
        HIPCHK(hipMemcpyAsync(ctx->d_g, ctx->d_gStart, sizeof(double) * n, hipMemcpyDeviceToDevice, st));
    }
    if (startGrad != 2) HIPCHK(hipMemcpyAsync(ctx->d_gStart, ctx->d_g, sizeof(double) * n, hipMemcpyDeviceToDevice, st));
    ++evals;                                                 // counted as the reference counts it (hmcprior.nfevals, :217)
    hipLaunchKernelGGL(k_lf_momentum, g1, b1, 0, st, lf, regParam, 0.5 * dt);
    for (int k = 1; k <= L; ++k) {
        hipLaunchKernelGGL(k_lf_dmmax, dim3(LFNB), dim3(256), 0, st, lf, dt);
        hipLaunchKernelGGL(k_lf_step, g1, b1, 0, st, lf, dt, lnSigMin, lnSigMax);
        rc = evaluate(ctx, d_m, true, d_pred, d_misfit, ctx->d_g);      // (reports a failure of the step before)
        if (rc) return rc;
        // asynchronous, as hmcmt_grad_device_async: the next step's launches overlap this step's gradient tail
        ctx->statsPending = true;
        ctx->pendingAdj = true;
        ++evals;
        hipLaunchKernelGGL(k_lf_momentum, g1, b1, 0, st, lf, regParam, (k < L ? 1.0 : 0.5) * dt);
    }
    if ((rc = collect_pending(ctx))) return rc;
    hipLaunchKernelGGL(k_lf_mnorm, dim3(LFNB), dim3(256), 0, st, lf, regParam);
    hipLaunchKernelGGL(k_lf_mnorm_final, dim3(1), dim3(1), 0, st, lf, regParam);
    HIPCHK(hipMemcpyAsync(ctx->h_lfFlag, ctx->d_lfFlag, sizeof(int), hipMemcpyDeviceToHost, st));
    ctx->lfFlagPending = true;
    if (evalsOut) *evalsOut = evals;
    return 0;
}

// after a synchronisation: did a trajectory meet a non-finite model value?
static int leapfrog_flag(hmcmt_ctx* ctx) {
    if (!ctx->lfFlagPending) return 0;
    ctx->lfFlagPending = false;
    if (*(volatile int*)ctx->h_lfFlag) {
        *ctx->h_lfFlag = 0;
        ctx->haveFwd = ctx->haveAdj = false;
        ctx->err = "non-finite model value during the trajectory";
        return HMCMT_EBREAKDOWN;
    }
    return 0;
}

int hmcmt_leapfrog_device(hmcmt_ctx* ctx, double* d_m, double* d_p, double dt, int32_t L, double regParam,
                          double lnSigMin, double lnSigMax, int32_t start_grad, double* d_pred, double* d_misfit,
                          double* d_mnorm, int32_t* nfevals) {
    if (!ctx || !d_m || !d_p) return HMCMT_EINVAL;
    if (!ctx->havePrior) { ctx->err = "hmcmt_set_prior has not been called"; return HMCMT_EINVAL; }
    if (L < 1 || !(dt > 0) || !(lnSigMax > lnSigMin)) { ctx->err = "need L >= 1, dt > 0, lnSigMax > lnSigMin"; return HMCMT_EINVAL; }
    if (start_grad < 0 || start_grad > 2) { ctx->err = "start_grad must be 0, 1 or 2"; return HMCMT_EINVAL; }
    if (start_grad != 0 && !ctx->lfHaveGrad) { ctx->err = "start_grad != 0 needs a previous trajectory on this context"; return HMCMT_EINVAL; }
    HIPCHK(hipSetDevice(ctx->device));
    int evals = 0;
    ctx->lfHaveGrad = false;
    int rc = leapfrog_core(ctx, d_m, d_p, dt, L, regParam, lnSigMin, lnSigMax, start_grad, d_pred, d_misfit, &evals);
    if (rc) return rc;
    if (d_mnorm) HIPCHK(hipMemcpyAsync(d_mnorm, ctx->d_lfScal, sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
    HIPCHK(hipGetLastError());
    ctx->lfHaveGrad = true;
    if (nfevals) *nfevals = evals;
    return 0;
}

int hmcmt_leapfrog(hmcmt_ctx* ctx, const double* m0, const double* p0, double dt, int32_t L, double regParam,
                   double lnSigMin, double lnSigMax, double* m1, double* p1, double* pred, double* misfit,
                   double* mnorm, int32_t* nfevals) {
    if (!ctx || !m0 || !p0 || !m1 || !p1) return HMCMT_EINVAL;
    if (!ctx->havePrior) { ctx->err = "hmcmt_set_prior has not been called"; return HMCMT_EINVAL; }
    if (L < 1 || !(dt > 0) || !(lnSigMax > lnSigMin)) { ctx->err = "need L >= 1, dt > 0, lnSigMax > lnSigMin"; return HMCMT_EINVAL; }
    HIPCHK(hipSetDevice(ctx->device));
    const int n = ctx->v.nAC, nData = ctx->v.nData;
    for (int i = 0; i < n; ++i)
        if (!std::isfinite(m0[i]) || !std::isfinite(p0[i])) { ctx->err = "non-finite model or momentum"; return HMCMT_EBREAKDOWN; }
    hipStream_t st = ctx->stream;
    std::memcpy(ctx->h_stage, m0, sizeof(double) * n);
    std::memcpy(ctx->h_stage + n, p0, sizeof(double) * n);
    HIPCHK(hipMemcpyAsync(ctx->d_mcur, ctx->h_stage, sizeof(double) * n, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(ctx->d_p, ctx->h_stage + n, sizeof(double) * n, hipMemcpyHostToDevice, st));
    int evals = 0, startGrad = 0;
    if (const hmcmt_ctx::Memo* e = ctx->opt.verify ? nullptr : memo_find(ctx, m0, true)) {
        // the gradient at the start model is known (end of the previous trajectory, or its start after a rejection)
        std::memcpy(ctx->h_stage + 2 * n, e->grad.data(), sizeof(double) * n);
        HIPCHK(hipMemcpyAsync(ctx->d_g, ctx->h_stage + 2 * n, sizeof(double) * n, hipMemcpyHostToDevice, st));
        ++ctx->memoHits;
        startGrad = 1;
    }
    ctx->lfHaveGrad = false;
    int rc = leapfrog_core(ctx, ctx->d_mcur, ctx->d_p, dt, L, regParam, lnSigMin, lnSigMax, startGrad, nullptr, nullptr, &evals);
    if (rc) return rc;
    double* hs = ctx->h_stage;
    HIPCHK(hipMemcpyAsync(hs, ctx->d_mcur, sizeof(double) * n, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(hs + n, ctx->d_p, sizeof(double) * n, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(hs + 2 * n, ctx->v.pred, sizeof(cplx) * nData, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(hs + 2 * n + 2 * nData, ctx->d_misfit, sizeof(double), hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(hs + 2 * n + 2 * nData + 1, ctx->d_lfScal, sizeof(double), hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(hs + 2 * n + 2 * nData + 2, ctx->d_g, sizeof(double) * n, hipMemcpyDeviceToHost, st));   // for the memo
    HIPCHK(hipStreamSynchronize(st));
    HIPCHK(hipGetLastError());
    prof_collect(ctx);
    if ((rc = leapfrog_flag(ctx))) return rc;
    ctx->lfHaveGrad = true;
    memo_store(ctx, hs, hs + 2 * n, hs[2 * n + 2 * nData], hs + 2 * n + 2 * nData + 2);   // the end model's data gradient
    std::memcpy(m1, hs, sizeof(double) * n);
    std::memcpy(p1, hs + n, sizeof(double) * n);
    if (pred) std::memcpy(pred, hs + 2 * n, sizeof(cplx) * nData);
    if (misfit) *misfit = hs[2 * n + 2 * nData];
    if (mnorm) *mnorm = hs[2 * n + 2 * nData + 1];
    if (nfevals) *nfevals = evals;
    return 0;
}

}  // extern "C"
