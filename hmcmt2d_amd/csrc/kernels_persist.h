// kernels_persist.h -- part of libhmcmt_hip.so; included by hmcmt_hip.hip INSIDE its anonymous namespace (one translation unit).
//
// The whole COCG solve of the default path (Jacobi / FDM / Jacobi preconditioner, mixed precision) as ONE persistent
// kernel (round 4).  The launch-per-phase form (k_spmv_fused -> k_update_fused ->
// k_fdm_fwd -> k_back_post, kernels_fused.h / kernels_fdm.h) moves every vector of every system through the fabric four
// times per iteration and pays four launch ramps; but every dependency of an iteration is per SYSTEM, a system's vectors
// (x, r, p: 0.9 MB) fit the registers of a few CUs, and a barrier among workgroups of ONE XCD costs ~1 us
// (scripts/probe/xcd_barrier.hip).  Here a system is solved by G = ceil((nz-1)/14) workgroups of one XCD:
//
//   * workgroup j of a system OWNS the interior node rows 1+14j .. 14+14j; thread (iy, c) owns the column iy of 7 of them
//     and keeps r (fp64) of those nodes in registers for the whole solve, together with the float stencil coefficients of
//     its 12 tile rows; x stays in memory (touched once per iteration by its owner);
//   * the tile of a workgroup is its 14 rows + 5 halo rows on each side = 24 rows = three 8-row MFMA groups.  The back
//     transform of the FDM stage produces V y on all 24 rows, and everything between two FDM stages -- post-sweeps, p = z +
//     beta p, q = A p, r -= alpha q, pre-sweeps -- is recomputed on the halo rows, shrinking by one row per stencil
//     (z3 +-5, z4 +-4, z5 +-3, p +-3, q +-2, r' +-2, z1 +-2, z2 +-1, t own): NO halo exchange has a synchronisation of
//     its own.  What the halo rows need from their owners (r', the pre-smoothed iterate z2, the old direction p, all
//     complex64) is published through the XCD's L2 before the FDM stage's first synchronisation and picked up behind its
//     second;
//   * four synchronisations per iteration among the G workgroups of the system (R1: rho, |z|, |x| -> beta and the
//     convergence decision; R2: p'q -> alpha; T1: rows -> mode slabs of the forward transform; T2: solved slabs -> rows),
//     each an atomic add in the L2 + a poll by one lane (agent scope), payload by plain stores (they stay in the XCD's
//     L2) drained with s_waitcnt vmcnt(0), picked up by loads that bypass the L1 (sc1).  R1 is split: arrive behind the
//     first post-sweep, wait in front of the p update, the second post-sweep in between.
//   * workgroups b, b + 8, b + 16, .. share an XCD (round-robin dispatch; NOT a HIP guarantee): every group checks it at
//     kernel start with XCC_ID and gives up -- systems untouched, the host runs the launch-per-phase loop -- if it does
//     not hold.  All spins are bounded.
// The arithmetic follows the launch-per-phase kernels (same preconditioner, same stopping rule on the error estimate,
// same stagnation watch, z and p rounded to complex64, x, r, q and all inner products fp64) except that the smoother
// works from a complex64 copy of r throughout, its diagonal is formed from the float couplings, and the halo rows' q is
// fp32: iteration counts agree within +-1, results to the solver tolerance (tests/test_gpu_persist.py).
// Reference: the solves at MTFwdSolver/mt2DTE.jl:47-55, mt2DTM.jl:46-54, MTSensitivity/compJacTMatVec.jl:220-229, 291-300.
#pragma once

constexpr int PS_OWN = 14;                        // interior rows owned by a workgroup
constexpr int PS_HALO = 5;                        // rows recomputed on each side
constexpr int PS_ROWS = PS_OWN + 2 * PS_HALO;     // 24 = three MFMA row groups
constexpr int PS_J = PS_ROWS / 2;                 // tile rows per thread
constexpr int PS_NO = PS_OWN / 2;                 // own rows per thread: j = PS_HALO .. PS_J - 1
constexpr int PS_DONE = 0x7fffffff;               // progress word: the kernel has ended
constexpr unsigned PS_SPIN_LIMIT = 1u << 22;      // polls (~1 us each) before a wait gives up

struct PersistArgs {
    unsigned* sync;            // [groups][32] per group: [0] barrier counter, [1] arrivals of the placement check, [2] OR of 1 << XCC_ID
    unsigned* exitCnt;         // workgroups that have left the kernel
    int* fail;                 // device word: 1 = a group's workgroups are not on one XCD, 2 = a wait timed out
    int* placeHost;            // pinned host word: set when a group's workgroups are not on one XCD (the host then runs the launch-per-phase loop)
    int G, slots, maxit, precondOnly;
    float wJ;                  // damping of the Jacobi sweeps (the factor k_coef_all folds into Solver::dinv)
    const u4v *Vb, *Vtb;       // bf16 fragment-order copies of V, V'
    double* partZZ;            // [S][MAXNB]
    float2 *pubR, *pubZ, *pubP;   // [S][vstride] complex64: r', the pre-smoothed iterate (z2; one sweep: z1), p of the own rows
    float2* yhat;              // [S][vstride] complex64 rows of the forward transform
    float2* ysol;              // [S][vstride] solved slabs, pre-split bf16 planes (store_t32's format)
    const float2* ip32;        // inverse pivots (complex64)
    float2* zout;              // precondOnly: z = P^-1 r
    long long* stamps;         // [workgroup][16] s_memtime stamps of one iteration's phases (HMCMT_STAMPS=persist)
};

__device__ __forceinline__ unsigned ps_xcc_id() { unsigned v; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v)); return v & 0xf; }
// loads of values another workgroup of this launch has written: the L1 of this CU is never refreshed by other CUs' stores
__device__ __forceinline__ float2 ps_ld_f2(const float2* p) {
    const unsigned long long v = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return float2{__uint_as_float((unsigned)v), __uint_as_float((unsigned)(v >> 32))};
}
__device__ __forceinline__ c32 ps_ld_c32(const float2* p) { const float2 v = ps_ld_f2(p); return c32{v.x, v.y}; }
__device__ __forceinline__ double ps_ld_f64(const double* p) {
    return __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
__device__ __forceinline__ bool ps_wait(unsigned* cnt, unsigned target, int* fail) {
    for (unsigned spins = 0;; ++spins) {
        if (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= target) return true;
        __builtin_amdgcn_s_sleep(1);
        if ((spins & 0x3ff) == 0x3ff) {
            if (__hip_atomic_load(fail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return false;
            if (spins > PS_SPIN_LIMIT) { __hip_atomic_store(fail, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return false; }
        }
    }
}
__device__ __forceinline__ c32 operator+(c32 a, c32 b) { return c32{a.re + b.re, a.im + b.im}; }
// element i of an array addressed as UNIFORM base + 32-bit BYTE offset (one address register per access; with 64-bit element
// addresses the loop-invariant address of every row of every array was hoisted and spilled: 2.2 KB of scratch per lane)
template <class T>
__device__ __forceinline__ T* ps_at(T* base, unsigned i) { return reinterpret_cast<T*>(reinterpret_cast<char*>(base) + i * (unsigned)sizeof(T)); }
template <class T>
__device__ __forceinline__ const T* ps_at(const T* base, unsigned i) { return reinterpret_cast<const T*>(reinterpret_cast<const char*>(base) + i * (unsigned)sizeof(T)); }

// float stencil coefficients of a thread's 12 tile rows: lateral couplings, omega * mass, and the coupling between its
// rows j and j + 1 (cV[11]: to the first row of the other half of the tile); the diagonal is minus the sum of the four
// couplings (Appendix E.1 of SURVEY.md: K_ii = -(c_E + c_W + c_S + c_N))
struct PsCo { float cE[PS_J], cW[PS_J], dmw[PS_J], cV[PS_J]; };

// (A u) on the thread's rows JLO <= j < JHI: vertical neighbours from its registers, lateral ones (and the inner neighbour
// of its last row) from the tile in LDS.  A row that two workgroups compute (an own row of one, a halo row of the other)
// must come out BIT FOR BIT the same in both -- the halo rows' p enters the owners' fp64 q = A p, and x += alpha p, r -=
// alpha q stay consistent only if every workgroup uses the same p -- so the terms are ordered by MESH direction (north,
// south), not by the thread's direction (outer, inner; the two halves of a tile are mirrored).
template <int JLO, int JHI = PS_J, class F>
__device__ __forceinline__ void ps_apply(const PsCo& co, const c32 (&u)[PS_J], const c32* __restrict__ T, int t0i, int es, int tini, int c, F&& f) {
#pragma unroll
    for (int j = JLO; j < JHI; ++j) {
        const int ti = t0i + j * es;
        const c32 ue = T[ti + 1], uw = T[ti - 1];
        const c32 ui = j + 1 < PS_J ? u[j + 1 < PS_J ? j + 1 : j] : T[tini];
        const c32 uo = u[j - 1], uc = u[j];
        const c32 un = c ? ui : uo, us = c ? uo : ui;
        const float cn = c ? co.cV[j] : co.cV[j - 1], cs = c ? co.cV[j - 1] : co.cV[j];
        float ce = co.cE[j];
        asm volatile("" : "+v"(ce));               // (see ps_dinv)
        const float dk = -((ce + co.cW[j]) + (cn + cs));
        f(j, c32{((dk * uc.re - co.dmw[j] * uc.im) + (ce * ue.re + co.cW[j] * uw.re)) + (cn * un.re + cs * us.re),
                 ((dk * uc.im + co.dmw[j] * uc.re) + (ce * ue.im + co.cW[j] * uw.im)) + (cn * un.im + cs * us.im)});
    }
}
// damped inverse diagonal wJ / (dk + i omega dm) of row j (the same sum as in ps_apply)
__device__ __forceinline__ c32 ps_dinv(const PsCo& co, int j, float wJ, int c) {
    const float va = co.cV[j], vb = co.cV[j > 0 ? j - 1 : 0];
    const float cn = c ? va : vb, cs = c ? vb : va;
    float ce = co.cE[j], dm = co.dmw[j];
    asm volatile("" : "+v"(ce), "+v"(dm));      // (recomputed at every use: 36 loop-invariant values would otherwise be hoisted and spilled)
    const float dk = -((ce + co.cW[j]) + (cn + cs));
    const float inv = wJ * __builtin_amdgcn_rcpf(dk * dk + dm * dm);
    return c32{dk * inv, -dm * inv};
}

// block-wide deterministic sums of three doubles (NW waves); result in every thread
template <int NWV>
__device__ __forceinline__ void ps_block_sum3(double& a, double& b, double& c, double* sh) {
    a = wave_sum(a); b = wave_sum(b); c = wave_sum(c);
    const int w = threadIdx.x >> 6;
    __syncthreads();                                   // (sh may still be read from the previous reduction)
    if ((threadIdx.x & 63) == 0) { sh[w] = a; sh[8 + w] = b; sh[16 + w] = c; }
    __syncthreads();
    double sa = 0, sb = 0, sc = 0;
#pragma unroll
    for (int i = 0; i < NWV; ++i) { sa += sh[i]; sb += sh[8 + i]; sc += sh[16 + i]; }
    a = sa; b = sb; c = sc;
}

// ---- the tridiagonal solves of one 32-mode slab of one system: k_fdm_fwd's LDS scheme (pre-multiplied recurrences, twisted
// factorisation, mirrored bottom half, padded regions: kernels_fdm.h) fed from the rows the G workgroups have transformed
template <int NT>
__device__ __forceinline__ void ps_slab_solve(const Solver& k, char* smem, int s, int slab, const float2* __restrict__ yhat,
                                              float2* __restrict__ ysol, const float2* __restrict__ ip32) {
    constexpr int NTW = 2, SW = 16 * NTW;
    const int NYP = k.NYP, NZP = k.NZP, n = k.nz - 1;
    const int tw = k.twist, mid = twist_mid(n, tw);
    const int RCAP = tw ? mid + 1 : NZP, RL = RCAP + 4 * FW_TB, nreg = tw ? 2 : 1;
    float* sof = reinterpret_cast<float*>(smem);
    c32* sj = reinterpret_cast<c32*>(smem + (((long)NZP * 4 + 127) & ~127L));
    c32* sa = sj + SW + 2 * FW_TB * SW;                  // -> region 0, row 0
    c32* sb = sa + (long)nreg * RL * SW;
    c32* sc = sb + (long)nreg * RL * SW;
    auto lidx = [&](int row) { return (tw && row > mid) ? RL + (n + 1 - row) : row; };   // slab row of a matrix row
    const int mode = s >= k.nFreq;
    const long so = (long)s * k.vstride;
    const int t0 = slab * NTW;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < NZP; i += NT) sof[i] = (float)k.ofz[(long)mode * NZP + i];
    __syncthreads();
    // rows of the slab: a = y * ip, and the pre-multiplied coefficients of the two sweeps
    constexpr int PB = 4;
    for (int b0 = 0; b0 < NZP * SW; b0 += PB * NT) {
        const int i0 = b0 + threadIdx.x;
        float2 yv[PB], ipf[PB];
        bool ok[PB];
#pragma unroll
        for (int u = 0; u < PB; ++u) {
            const int idx = min(i0 + u * NT, NZP * SW - 1);
            const int row = idx / SW, c = t0 * 16 + (idx % SW);
            ok[u] = row >= 1 && row <= n && c < k.ny - 1;
            const unsigned e = (unsigned)((ok[u] ? row : 1) * NYP + (ok[u] ? c : 1));
            yv[u] = ps_ld_f2(ps_at(yhat + so, e));
            ipf[u] = *ps_at(ip32 + so, e);
        }
#pragma unroll
        for (int u = 0; u < PB; ++u) {
            const int idx = i0 + u * NT;
            if (idx < NZP * SW) {
                const int row = idx / SW, j = idx % SW;
                const int l = lidx(row) * SW + j;
                c32 av = c32{0, 0}, p1 = c32{0, 0}, p2 = c32{0, 0};
                if (ok[u]) {
                    const c32 ip = c32{ipf[u].x, ipf[u].y};
                    const c32 bb = sof[row - 1] * ip, cc = sof[row] * ip;
                    av = c32{yv[u].x, yv[u].y} * ip;
                    const bool bottom = tw && row > mid;
                    p1 = bottom ? cc : bb; p2 = bottom ? bb : cc;
                }
                sa[l] = av; sb[l] = p1; sc[l] = p2;
            }
        }
    }
    // padding rows: in front of a region zeros; behind a region identity rows for the elimination sweep (a = 0, p1 = -1), zero p2
    for (int idx = threadIdx.x; idx < nreg * FW_TB * SW; idx += NT) {
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
    if (wave == 0 && lane < nreg * SW && t0 * 16 + (lane % SW) < k.ny - 1) {
        const int half = lane / SW, col = lane % SW;
        const int last = tw ? (half == 0 ? mid : n - mid) : n;      // rows 1..last of this lane's region are real
        const int steps = tw ? mid : n;                             // both halves run the longer count (identity rows)
        c32* ra = sa + (long)half * RL * SW + col;
        const c32* rb = sb + (long)half * RL * SW + col;
        const c32* rc = sc + (long)half * RL * SW + col;
        c32 pt = c32{0, 0};
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
        pt = ra[last * SW];                                   // (the identity rows left it unchanged)
        if (tw) {
            const c32 p2last = rc[last * SW];
            const float pre = pt.re, pim = pt.im;
            const float ore = __shfl_xor(pre, SW), oim = __shfl_xor(pim, SW);
            const c32 xmid = (c32{pre, pim} - p2last * c32{ore, oim}) * sj[col];       // meaningful in the top half
            const float xre = xmid.re, xim = xmid.im;
            const float mre = __shfl_xor(xre, SW), mim = __shfl_xor(xim, SW);
            const c32 xbot = c32{pre, pim} - p2last * c32{mre, mim};
            pt = half == 0 ? c32{xre, xim} : xbot;
            ra[last * SW] = pt;
        }
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
    __syncthreads();
    // solved slab -> ysol, pre-split for the back transform (store_t32's format), 16-byte stores
    {
        constexpr int NG = SW / 8;
        unsigned short* yb = reinterpret_cast<unsigned short*>(ysol + so);
        for (int idx = threadIdx.x; idx < NZP * NG; idx += NT) {
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
    }
    __syncthreads();
}

// LDS of the persistent kernel: 1 KB of scratch + max(the slab scheme, planes of 24 rows + two tiles of 24 rows)
__host__ __device__ inline size_t ps_tiles_bytes(int NYP) {
    return (((size_t)PS_ROWS * 4 * NYP * 2 + 64 + 255) & ~(size_t)255) + 2 * (((size_t)PS_ROWS * NYP * 8 + 255) & ~(size_t)255) + 512;
}

#define PS_STAMP(i) if (a.stamps && tid == 0 && it == 3) a.stamps[(long)blockIdx.x * 16 + (i)] = __builtin_amdgcn_s_memtime();

template <int CW, int SW>
__global__ __launch_bounds__(2 * CW) void k_cocg_persist(Solver k, PersistArgs a) {
    constexpr int NT = 2 * CW, NWV = NT / 64;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double* sh = reinterpret_cast<double*>(smem);                           // [24] block reductions
    volatile int* sflag = reinterpret_cast<volatile int*>(smem + 256);      // [0] give up
    char* arena = smem + 1024;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int G = a.G;
    const int xcd = blockIdx.x & 7, lq = blockIdx.x >> 3, slot = lq / G, jwg = lq - slot * G;
    unsigned* sy = a.sync + 32 * (xcd * a.slots + slot);
    unsigned epoch = 0;                 // synchronisations of this group so far (the same in all its threads)
    int it = 0;
    if (tid == 0) sflag[0] = 0;
    __syncthreads();
    // ---- placement check: the G workgroups of the group must share an XCD (their hand-offs go through ITS L2)
    if (tid == 0) {
        __hip_atomic_fetch_or(sy + 2, 1u << ps_xcc_id(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_fetch_add(sy + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (!ps_wait(sy + 1, (unsigned)G, a.fail)) sflag[0] = 2;
        else if (__popc(__hip_atomic_load(sy + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) != 1) {
            sflag[0] = 1;
            __hip_atomic_store(a.fail, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            *a.placeHost = 1;
        }
    }
    __syncthreads();
    bool alive = sflag[0] == 0;

    auto sys_arrive = [&]() {           // ONE thread, behind ITS OWN payload stores (or behind a drained workgroup barrier)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_fetch_add(sy, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    auto sys_wait = [&]() -> bool {     // all threads
        ++epoch;
        if (tid == 0 && !ps_wait(sy, (unsigned)G * epoch, a.fail)) sflag[0] = 2;
        __syncthreads();
        return sflag[0] == 0;
    };
    auto sys_sync = [&]() -> bool {     // all threads: every wave's stores drained, then arrive + wait
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) sys_arrive();
        return sys_wait();
    };

    // ---- geometry of this thread: column iy of tile rows j = 0..11; half c = 1 is MIRRORED (j = 0 is the outermost halo
    // row of either half, j >= 5 are own rows, j = 11 borders the other half)
    const int c = __builtin_amdgcn_readfirstlane(tid / CW);
    const int iy = tid & (CW - 1);
    const int NYP = k.NYP, ny = k.ny, nz = k.nz;
    const bool colOK = iy < NYP;
    const int iyc = min(iy, NYP - 1);
    const int iz0 = 1 + PS_OWN * jwg, R0 = iz0 - PS_HALO;
    const int gb = c ? R0 + PS_ROWS - 1 : R0, gs = c ? -1 : 1;          // mesh row of thread-row j: gb + gs * j
    const int tb = c ? PS_ROWS - 1 : 0;                                 // tile row of thread-row j: tb + gs * j
    const int tin = c ? PS_J - 1 : PS_J;                                // tile row of the inner neighbour of j = 11
    unsigned inM = 0;                                                   // bit j: node (row j, iy) is an interior node of the mesh
#pragma unroll
    for (int j = 0; j < PS_J; ++j) {
        const int g = gb + gs * j;
        if (g >= 1 && g <= nz - 1 && iy >= 1 && iy <= ny - 1) inM |= 1u << j;
    }
    auto isIn = [&](int j) { return (inM >> j) & 1u; };
    // element offset of (row j, iy) in a system's [NZP][NYP] arrays: ONE 32-bit lane offset serves every array (uniform base
    // pointer + offset: 64-bit per-row addresses of a dozen arrays were what the register allocator spilled).  eo: the node itself
    // (valid where the row is in the mesh), ei: the node if it is an interior one, else a harmless interior node (unconditional loads)
    int e0 = gb * NYP + iy;
    const int es = gs * NYP;
    auto eo = [&](int j) -> unsigned { return (unsigned)(e0 + j * es); };
    auto ei = [&](int j) -> unsigned { return isIn(j) ? (unsigned)(e0 + j * es) : (unsigned)(NYP + 1); };
    int t0i = tb * NYP + iy, tini = tin * NYP + iy;                     // the same for the tiles in LDS: t0i + j * es; inner neighbour of j = 11
    // LDS carve: planes [24][4][NYP] bf16 (+ 64 B that the last k-group over-reads), two tiles [24][NYP] complex64
    unsigned short* PL = reinterpret_cast<unsigned short*>(arena);
    c32* T0 = reinterpret_cast<c32*>(arena + (((size_t)PS_ROWS * 4 * NYP * 2 + 64 + 255) & ~(size_t)255));
    c32* T1 = T0 + ((((size_t)PS_ROWS * NYP * 8 + 255) & ~(size_t)255) / 8);
    // MFMA work split: column tiles of 16 over the waves, at most two per wave (NYP <= 32 NWV)
    const int NTc = NYP >> 4, KG = (NYP + 31) >> 5;
    const int tbase = NTc / NWV, textra = NTc - tbase * NWV;
    const int ntl = tbase + (wave < textra ? 1 : 0), t0w = wave * tbase + min(wave, textra);
    const int tl0 = min(t0w, NTc - 1), tl1 = min(t0w + 1, NTc - 1);
    const int lj = lane & 15, g4 = lane >> 4;
    const int nslab = (NTc + 1) / 2;

    for (int round = 0; alive; ++round) {
        const int s = xcd + 8 * (slot + a.slots * round);
        if (s >= k.S) break;
        if (!k.active[s]) continue;
        const int mode = s >= k.nFreq;
        const long mo = (long)mode * k.vstride, so = (long)s * k.vstride;
        const double w = k.omega[s];
        const float wf = (float)w;
        float2 *pubR = a.pubR + so, *pubZ = a.pubZ + so, *pubP = a.pubP + so;
        cplx *rsys = k.r + so, *xsys = k.x + so;
        // ---- coefficients of the 12 tile rows (registers for the whole solve)
        PsCo co;
        {
            const float4* cf = k.cf32 + 2 * mo;
            float vin[PS_J + 1], vout[PS_J + 1];
#pragma unroll
            for (int j = 0; j <= PS_J; ++j) {
                const int g = gb + gs * j, gc = min(max(g, 0), nz);
                const unsigned e = (unsigned)(gc * NYP + iyc);
                const float4 ca = cf[2u * e], cb = cf[2u * e + 1u];
                const bool rowIn = g >= 0 && g <= nz;
                if (j < PS_J) { co.cE[j] = rowIn ? ca.z : 0.f; co.cW[j] = rowIn ? ca.w : 0.f; co.dmw[j] = rowIn ? wf * ca.y : 0.f; }
                vin[j] = rowIn ? (c ? cb.y : cb.x) : 0.f;
                vout[j] = rowIn ? (c ? cb.x : cb.y) : 0.f;
            }
#pragma unroll
            for (int j = 0; j < PS_J; ++j) co.cV[j] = vin[j] != 0.f ? vin[j] : vout[j + 1];
        }
        // ---- state: r of the own rows (fp64), r of the halo rows (complex64, refreshed from the owners every iteration)
        cplx r64[PS_NO];
        c32 rh[PS_HALO];
        c32 tt[PS_NO];
#pragma unroll
        for (int q = 0; q < PS_NO; ++q) {
            const int j = PS_HALO + q, g = gb + gs * j;
            r64[q] = *ps_at(rsys, ei(j));
            if (!isIn(j)) r64[q] = cplx{0, 0};
            tt[q] = c32{0, 0};
            if (colOK && g >= 1 && g <= nz - 1) {
                *ps_at(pubR, eo(j)) = float2{(float)r64[q].re, (float)r64[q].im};
                *ps_at(pubP, eo(j)) = float2{0.f, 0.f};
            }
        }
        if (!sys_sync()) { alive = false; break; }
#pragma unroll
        for (int j = 0; j < PS_HALO; ++j) {
            rh[j] = ps_ld_c32(ps_at(pubR, ei(j)));
            if (!isIn(j)) rh[j] = c32{0, 0};
        }
        auto rr = [&](int j) -> c32 { return j < PS_HALO ? rh[j < PS_HALO ? j : 0] : c32{(float)r64[j >= PS_HALO ? j - PS_HALO : 0].re, (float)r64[j >= PS_HALO ? j - PS_HALO : 0].im}; };

        cplx rhoPrev = cplx{0, 0}, rhoCur = cplx{0, 0};
        double errRef = 0.0;
        int errRefIt = 0;
        bool stalled = false;
        int st = 0;
        double est = 0.0;
        it = 0;
        for (;;) {
            asm volatile("" : "+v"(e0), "+v"(t0i), "+v"(tini), "+v"(inM));   // (row offsets and masks are re-derived per iteration instead of living in 60 registers)
            PS_STAMP(0)
            // ================= pre-smoother: t = G^2-smoothed residual on the own rows =================
            c32 u1[PS_J];
            constexpr int JZ1 = SW == 2 ? 3 : 4;
#pragma unroll
            for (int j = 0; j < PS_J; ++j) {
                u1[j] = (j >= JZ1 && isIn(j)) ? ps_dinv(co, j, a.wJ, c) * rr(j) : c32{0, 0};
                if (colOK) T0[t0i + j * es] = u1[j];
            }
            __syncthreads();
            double p1r = 0, p1i = 0, dum = 0;
            if constexpr (SW == 2) {
                c32 u2[PS_J];
#pragma unroll
                for (int j = 0; j < 4; ++j) u2[j] = c32{0, 0};
                ps_apply<4>(co, u1, T0, t0i, es, tini, c, [&](int j, c32 av) __attribute__((always_inline)) {
                    u2[j] = isIn(j) ? u1[j] + (k.w2 * ps_dinv(co, j, a.wJ, c)) * (rr(j) - av) : c32{0, 0};
                });
#pragma unroll
                for (int j = 0; j < PS_J; ++j) if (colOK) T1[t0i + j * es] = u2[j];
                __syncthreads();
                ps_apply<5>(co, u2, T1, t0i, es, tini, c, [&](int j, c32 av) __attribute__((always_inline)) {
                    const int q = j - PS_HALO, g = gb + gs * j;
                    const c32 rv = rr(j);
                    tt[q] = isIn(j) ? rv - av : c32{0, 0};
                    const double sr = (double)rv.re + (double)tt[q].re, si = (double)rv.im + (double)tt[q].im;     // (r' + t) .* z2
                    p1r += sr * u2[j].re - si * u2[j].im; p1i += sr * u2[j].im + si * u2[j].re;
                    if (colOK && g >= 1 && g <= nz - 1) *ps_at(pubZ, eo(j)) = float2{u2[j].re, u2[j].im};
                });
            } else {
                ps_apply<5>(co, u1, T0, t0i, es, tini, c, [&](int j, c32 av) __attribute__((always_inline)) {
                    const int q = j - PS_HALO, g = gb + gs * j;
                    tt[q] = isIn(j) ? rr(j) - av : c32{0, 0};
                    if (colOK && g >= 1 && g <= nz - 1) *ps_at(pubZ, eo(j)) = float2{u1[j].re, u1[j].im};
                });
            }
            PS_STAMP(1)
            // ================= forward transform of the own rows: t -> bf16 hi/lo planes in LDS -> MFMA -> yhat =================
            // (T0 / T1 are not touched: PL is a region of its own)
            if (colOK) {
#pragma unroll
                for (int q = 0; q < PS_NO; ++q) {
                    const int rho = tb + gs * (PS_HALO + q) - PS_HALO;         // row of the 16-row operand: tile row - 5
                    unsigned short* b = PL + (long)rho * 4 * NYP + iy;
                    const unsigned hr = bf16_rn(tt[q].re), hi = bf16_rn(tt[q].im);
                    b[0] = (unsigned short)hr; b[NYP] = (unsigned short)hi;
                    b[2 * NYP] = (unsigned short)bf16_rn(tt[q].re - bf16_to_f32(hr));
                    b[3 * NYP] = (unsigned short)bf16_rn(tt[q].im - bf16_to_f32(hi));
                }
            }
            for (int i = tid; i < 2 * 4 * NYP / 2 + 16; i += NT) reinterpret_cast<unsigned*>(PL)[PS_OWN * 4 * NYP / 2 + i] = 0u;   // rows 14, 15 and the over-read pad
            __syncthreads();
            {
                f4v acc[2][2];
#pragma unroll
                for (int rg = 0; rg < 2; ++rg)
#pragma unroll
                    for (int t = 0; t < 2; ++t) acc[rg][t] = f4v{0, 0, 0, 0};
                constexpr int KC = 4;
                for (int kc = 0; kc < KG; kc += KC) {
                    u4v bh[KC][2];
#pragma unroll
                    for (int q = 0; q < KC; ++q) {
                        const int kg = min(kc + q, KG - 1);
                        bh[q][0] = *ps_at(a.Vb, (unsigned)((kg * NTc + tl0) * 64 + lane));
                        bh[q][1] = *ps_at(a.Vb, (unsigned)((kg * NTc + tl1) * 64 + lane));
                    }
#pragma unroll
                    for (int q = 0; q < KC; ++q) {
                        if (kc + q < KG) {
                            const int kg = kc + q;
#pragma unroll
                            for (int rg = 0; rg < 2; ++rg) {
                                const unsigned short* ap = PL + ((long)(8 * rg + (lj >> 1)) * 4 + (lj & 1)) * NYP + 32 * kg + 8 * g4;
                                const bf8v ah = __builtin_bit_cast(bf8v, *reinterpret_cast<const u4v*>(ap));
                                const bf8v al = __builtin_bit_cast(bf8v, *reinterpret_cast<const u4v*>(ap + 2 * NYP));
#pragma unroll
                                for (int t = 0; t < 2; ++t) {
                                    const bf8v bhf = __builtin_bit_cast(bf8v, bh[q][t]);
                                    acc[rg][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bhf, acc[rg][t], 0, 0, 0);
                                    acc[rg][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bhf, acc[rg][t], 0, 0, 0);
                                }
                            }
                        }
                    }
                }
                float2* yh = a.yhat + so;
#pragma unroll
                for (int rg = 0; rg < 2; ++rg)
#pragma unroll
                    for (int t = 0; t < 2; ++t)
#pragma unroll
                        for (int h2 = 0; h2 < 2; ++h2) {
                            const int rho = 8 * rg + 2 * g4 + h2, g = iz0 + rho;
                            if (t < ntl && rho < PS_OWN && g <= nz - 1)
                                *ps_at(yh, (unsigned)(g * NYP + (t0w + t) * 16 + lj)) = float2{acc[rg][t][2 * h2], acc[rg][t][2 * h2 + 1]};
                        }
            }
            if constexpr (SW == 2) {
                ps_block_sum3<NWV>(p1r, p1i, dum, sh);
                if (tid == 0) k.partR[(long)s * MAXNB + jwg] = cplx{p1r, p1i};
            }
            PS_STAMP(2)
            if (!sys_sync()) { alive = false; break; }                         // T1: every row of yhat is in the L2
            PS_STAMP(3)
            // ================= tridiagonal solves of this workgroup's mode slabs =================
            for (int slab = jwg; slab < nslab; slab += G) ps_slab_solve<NT>(k, arena, s, slab, a.yhat, a.ysol, a.ip32);
            PS_STAMP(4)
            if (!sys_sync()) { alive = false; break; }                         // T2: every solved slab is in the L2
            PS_STAMP(5)
            ++it;
            // ================= back transform of the 24 tile rows: planes -> LDS, MFMA, V y -> T0 =================
            {
                const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float2*>(a.ysol + so), 0, (int)(k.vstride * 8), 0x00020000);
                const int rowU = NYP / 2, n16 = PS_ROWS * rowU;             // 16-byte units per row / in the tile
                u4v tmp[6];
#pragma unroll
                for (int u = 0; u < 6; ++u) {
                    const int i = min(tid + u * NT, n16 - 1);
                    const int row = i / rowU, g = R0 + row;
                    const int gc = min(max(g, 0), nz);
                    tmp[u] = __builtin_amdgcn_raw_buffer_load_b128(rs, (gc * rowU + (i - row * rowU)) * 16, 0, 16);
                    if (g < 0 || g > nz) tmp[u] = u4v{0u, 0u, 0u, 0u};
                }
#pragma unroll
                for (int u = 0; u < 6; ++u) {
                    const int i = tid + u * NT;
                    if (i < n16) reinterpret_cast<u4v*>(PL)[i] = tmp[u];
                }
                if (tid < 16) reinterpret_cast<unsigned*>(PL)[PS_ROWS * 4 * NYP / 2 + tid] = 0u;
            }
            __syncthreads();
            c32 zz2[PS_J];            // the iterate the FDM correction is added to: z2 (two sweeps) / z1 (one), from its owners
            {
                f4v acc[3][2];
#pragma unroll
                for (int rg = 0; rg < 3; ++rg)
#pragma unroll
                    for (int t = 0; t < 2; ++t) acc[rg][t] = f4v{0, 0, 0, 0};
                constexpr int KC = 4;
                for (int kc = 0; kc < KG; kc += KC) {
                    u4v bh[KC][2];
#pragma unroll
                    for (int q = 0; q < KC; ++q) {
                        const int kg = min(kc + q, KG - 1);
                        bh[q][0] = *ps_at(a.Vtb, (unsigned)((kg * NTc + tl0) * 64 + lane));
                        bh[q][1] = *ps_at(a.Vtb, (unsigned)((kg * NTc + tl1) * 64 + lane));
                    }
#pragma unroll
                    for (int q = 0; q < KC; ++q) {
                        if (kc + q < KG) {
                            const int kg = kc + q;
#pragma unroll
                            for (int rg = 0; rg < 3; ++rg) {
                                const unsigned short* ap = PL + ((long)(8 * rg + (lj >> 1)) * 4 + (lj & 1)) * NYP + 32 * kg + 8 * g4;
                                const bf8v ah = __builtin_bit_cast(bf8v, *reinterpret_cast<const u4v*>(ap));
                                const bf8v al = __builtin_bit_cast(bf8v, *reinterpret_cast<const u4v*>(ap + 2 * NYP));
#pragma unroll
                                for (int t = 0; t < 2; ++t) {
                                    const bf8v bhf = __builtin_bit_cast(bf8v, bh[q][t]);
                                    acc[rg][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bhf, acc[rg][t], 0, 0, 0);
                                    acc[rg][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bhf, acc[rg][t], 0, 0, 0);
                                }
                            }
                        }
                    }
                }
                // (requested here: in flight during the epilogue and the barrier)
#pragma unroll
                for (int j = 0; j < PS_J; ++j) {
                    zz2[j] = ps_ld_c32(ps_at(pubZ, ei(j)));
                    if (j < PS_HALO) rh[j] = ps_ld_c32(ps_at(pubR, ei(j)));              // the owners' r' (no drift of the local copies)
                }
#pragma unroll
                for (int j = 0; j < PS_J; ++j)
                    if (!isIn(j)) { zz2[j] = c32{0, 0}; if (j < PS_HALO) rh[j] = c32{0, 0}; }
#pragma unroll
                for (int rg = 0; rg < 3; ++rg)
#pragma unroll
                    for (int t = 0; t < 2; ++t)
#pragma unroll
                        for (int h2 = 0; h2 < 2; ++h2) {
                            const int tau = 8 * rg + 2 * g4 + h2;
                            if (t < ntl) T0[tau * NYP + (t0w + t) * 16 + lj] = c32{acc[rg][t][2 * h2], acc[rg][t][2 * h2 + 1]};
                        }
            }
            __syncthreads();
            PS_STAMP(6)
            // ================= post-smoother =================
            c32 z3[PS_J], zf[PS_J];
            double ar = 0, ai = 0, zzs = 0;
#pragma unroll
            for (int j = 0; j < PS_J; ++j) {
                const int g = gb + gs * j;
                const c32 uv = colOK ? T0[t0i + j * es] : c32{0, 0};
                z3[j] = isIn(j) ? uv + zz2[j] : c32{0, 0};
                if (SW == 2 && j >= PS_HALO && isIn(j)) {              // second part of the rho identity: t .* (V y) on the own rows
                    const double tr = tt[j - PS_HALO].re, ti = tt[j - PS_HALO].im;
                    ar += tr * (double)uv.re - ti * (double)uv.im; ai += tr * (double)uv.im + ti * (double)uv.re;
                }
                if (colOK) T1[t0i + j * es] = z3[j];
            }
            __syncthreads();
#pragma unroll
            for (int j = 0; j < 1; ++j) zf[j] = c32{0, 0};
            ps_apply<1>(co, z3, T1, t0i, es, tini, c, [&](int j, c32 av) __attribute__((always_inline)) {
                zf[j] = c32{0, 0};
                if (isIn(j)) {
                    const c32 d = ps_dinv(co, j, a.wJ, c);
                    zf[j] = z3[j] + (SW == 2 ? k.w2 * d : d) * (rr(j) - av);
                }
                if (j >= PS_HALO) {
                    zzs += (double)zf[j].re * zf[j].re + (double)zf[j].im * zf[j].im;
                    if (SW == 1) {
                        const cplx rv = r64[j >= PS_HALO ? j - PS_HALO : 0];
                        ar += rv.re * (double)zf[j].re - rv.im * (double)zf[j].im; ai += rv.re * (double)zf[j].im + rv.im * (double)zf[j].re;
                    }
                }
            });
            ps_block_sum3<NWV>(ar, ai, zzs, sh);
            if (tid == 0) {
                k.partA[(long)s * MAXNB + jwg] = cplx{ar, ai};
                a.partZZ[(long)s * MAXNB + jwg] = zzs;
                sys_arrive();                                                  // R1, first half
            }
            if constexpr (SW == 2) {
                // second post-sweep, while the partial sums travel: z5 = z4 + D (r - A z4) on rows j >= 2
#pragma unroll
                for (int j = 0; j < PS_J; ++j) if (colOK) T0[t0i + j * es] = zf[j];
                __syncthreads();
#pragma unroll
                for (int j = 0; j < 2; ++j) z3[j] = c32{0, 0};
                ps_apply<2>(co, zf, T0, t0i, es, tini, c, [&](int j, c32 av) __attribute__((always_inline)) {
                    z3[j] = isIn(j) ? zf[j] + ps_dinv(co, j, a.wJ, c) * (rr(j) - av) : c32{0, 0};     // (z3 reused: the preconditioned residual z)
                });
            } else {
#pragma unroll
                for (int j = 0; j < PS_J; ++j) z3[j] = zf[j];
            }
            if (a.precondOnly) {
#pragma unroll
                for (int q = 0; q < PS_NO; ++q) {
                    const int j = PS_HALO + q, g = gb + gs * j;
                    if (colOK && g >= 1 && g <= nz - 1) *ps_at(a.zout + so, eo(j)) = float2{z3[j].re, z3[j].im};
                }
                if (!sys_wait()) alive = false;
                break;
            }
            PS_STAMP(7)
            c32 pold[PS_J];                                                    // the old direction, from its owners: in flight during the wait
#pragma unroll
            for (int j = 0; j < PS_J; ++j) pold[j] = ps_ld_c32(ps_at(pubP, ei(j)));      // (masked where it is used)
            if (!sys_wait()) { alive = false; break; }                         // R1, second half
            PS_STAMP(8)
            // ================= scalars: rho, error estimate, convergence, beta =================
            cplx rz;
            double zz, xx;
            {
                const long sl = (long)s * MAXNB + min(lane, G - 1);
                double pr_ = ps_ld_f64(&k.partA[sl].re), pi_ = ps_ld_f64(&k.partA[sl].im);
                if (SW == 2) { pr_ += ps_ld_f64(&k.partR[sl].re); pi_ += ps_ld_f64(&k.partR[sl].im); }
                double pz_ = ps_ld_f64(a.partZZ + sl), pb_ = ps_ld_f64(k.partB + sl);
                if (lane >= G) { pr_ = 0; pi_ = 0; pz_ = 0; pb_ = 0; }
                rz = cplx{wave_sum(pr_), wave_sum(pi_)};
                zz = wave_sum(pz_); xx = wave_sum(pb_);
            }
            const bool first = it == 1;
            bool on = true;
            st = 0;
            if (first) { if (zz == 0.0) on = false; }
            else if (zz <= k.tol2 * xx) on = false;
            else if (it - 1 >= a.maxit) { on = false; st = HMCMT_ENOCONV; }
            if (!(isfinite(rz.re) && isfinite(rz.im) && isfinite(zz) && isfinite(xx))) { on = false; st = HMCMT_EBREAKDOWN; }
            est = first ? (zz == 0.0 ? 0.0 : 1.0) : sqrt(zz / xx);
            if (first || est < 0.1 * errRef) { errRef = est; errRefIt = it; }
            else if (on && it - errRefIt > k.stallIt) { stalled = true; on = false; }
            if (!on) break;
            const cplx be = first ? cplx{0, 0} : rz / rhoPrev;
            rhoPrev = rz; rhoCur = rz;
            // ================= p = z + beta p (rounded to complex64), q = A p, p'q =================
            constexpr int JP = SW == 2 ? 2 : 1;
            c32 pn[PS_J];
#pragma unroll
            for (int j = 0; j < PS_J; ++j) {
                pn[j] = c32{0, 0};
                if (j >= JP && isIn(j)) {
                    const cplx v = cplx{(double)z3[j].re, (double)z3[j].im} + be * cplx{(double)pold[j].re, (double)pold[j].im};
                    pn[j] = c32{(float)v.re, (float)v.im};
                }
                if (colOK) T1[t0i + j * es] = pn[j];
            }
            __syncthreads();
            c32 qh[PS_HALO];                                                   // the halo rows' q: fp32
#pragma unroll
            for (int j = 0; j < PS_HALO; ++j) qh[j] = c32{0, 0};
            ps_apply<JP + 1, PS_HALO>(co, pn, T1, t0i, es, tini, c, [&](int j, c32 av) __attribute__((always_inline)) { qh[j] = av; });
            // own rows: fp64 coefficients, in two batches (registers); q itself waits in LDS for alpha (each thread reads back
            // what it wrote: the planes / first tile are free here; 28 registers less across the wait)
            cplx* Qs = reinterpret_cast<cplx*>(arena);
            double pqr = 0, pqi = 0, dum2 = 0;
            const double *dKm = k.dK + mo, *dMm = k.dM + mo, *cYm = k.cY + mo, *cZm = k.cZ + mo;
            auto qrows = [&](auto QLO, auto QHI) {
                constexpr int q0 = decltype(QLO)::value, q1 = decltype(QHI)::value;
                double dk64[q1 - q0], dm64[q1 - q0], ce64[q1 - q0], cw64[q1 - q0], ci64[q1 - q0], co64[q1 - q0];
#pragma unroll
                for (int q = q0; q < q1; ++q) {
                    const int j = PS_HALO + q;
                    const unsigned e = ei(j);
                    dk64[q - q0] = *ps_at(dKm, e); dm64[q - q0] = w * *ps_at(dMm, e);
                    ce64[q - q0] = *ps_at(cYm, e); cw64[q - q0] = *ps_at(cYm, e - 1u);
                    const double cs = *ps_at(cZm, e), cn = *ps_at(cZm, e - (unsigned)NYP);
                    ci64[q - q0] = c ? cn : cs; co64[q - q0] = c ? cs : cn;
                }
#pragma unroll
                for (int q = q0; q < q1; ++q) {
                    const int j = PS_HALO + q;
                    const int ti = t0i + j * es;
                    cplx qv = cplx{0, 0};
                    if (isIn(j)) {
                        const c32 pe = T1[ti + 1], pw = T1[ti - 1];
                        const c32 pi = j + 1 < PS_J ? pn[j + 1 < PS_J ? j + 1 : j] : T1[tini];
                        const c32 po = pn[j - 1], pc = pn[j];
                        cplx acc = cplx{dk64[q - q0] * (double)pc.re - dm64[q - q0] * (double)pc.im, dk64[q - q0] * (double)pc.im + dm64[q - q0] * (double)pc.re};
                        acc += ce64[q - q0] * cplx{(double)pe.re, (double)pe.im};
                        acc += cw64[q - q0] * cplx{(double)pw.re, (double)pw.im};
                        acc += ci64[q - q0] * cplx{(double)pi.re, (double)pi.im};
                        acc += co64[q - q0] * cplx{(double)po.re, (double)po.im};
                        qv = acc;
                        pqr += (double)pc.re * acc.re - (double)pc.im * acc.im;
                        pqi += (double)pc.re * acc.im + (double)pc.im * acc.re;
                    }
                    if (colOK) Qs[ti - PS_HALO * NYP] = qv;
                }
            };
            qrows(std::integral_constant<int, 0>{}, std::integral_constant<int, 4>{});
            __builtin_amdgcn_sched_barrier(0);
            qrows(std::integral_constant<int, 4>{}, std::integral_constant<int, PS_NO>{});
            // x of the own rows: requested here, used behind R2
            cplx xv[PS_NO];
#pragma unroll
            for (int q = 0; q < PS_NO; ++q) {
                const int j = PS_HALO + q, g = gb + gs * j;
                xv[q] = *ps_at(xsys, (colOK && g >= 1 && g <= nz - 1) ? eo(j) : (unsigned)(NYP + 1));
            }
            ps_block_sum3<NWV>(pqr, pqi, dum2, sh);
            if (tid == 0) { k.partPQ[(long)s * MAXNB + jwg] = cplx{pqr, pqi}; sys_arrive(); }
            PS_STAMP(9)
            if (!sys_wait()) { alive = false; break; }                         // R2
            PS_STAMP(10)
            // ================= alpha; x += alpha p, r -= alpha q; publish r', p =================
            cplx al;
            {
                const long sl = (long)s * MAXNB + min(lane, G - 1);
                double pr_ = ps_ld_f64(&k.partPQ[sl].re), pi_ = ps_ld_f64(&k.partPQ[sl].im);
                if (lane >= G) { pr_ = 0; pi_ = 0; }
                al = rhoCur / cplx{wave_sum(pr_), wave_sum(pi_)};
            }
            const c32 alf = c32{(float)al.re, (float)al.im};
            double xxs = 0, dum3 = 0, dum4 = 0;
#pragma unroll
            for (int q = 0; q < PS_NO; ++q) {
                const int j = PS_HALO + q, g = gb + gs * j;
                cplx xn = xv[q];
                if (isIn(j)) {
                    xn = xv[q] + al * cplx{(double)pn[j].re, (double)pn[j].im};
                    *ps_at(xsys, eo(j)) = xn;
                    r64[q] -= al * Qs[t0i + j * es - PS_HALO * NYP];
                }
                if (colOK && g >= 1 && g <= nz - 1) {
                    xxs += cabs2(xn);                                          // (rows 1 .. nz-1, all columns: as the launch-per-phase kernels)
                    *ps_at(pubR, eo(j)) = float2{(float)r64[q].re, (float)r64[q].im};
                    *ps_at(pubP, eo(j)) = float2{pn[j].re, pn[j].im};
                }
            }
#pragma unroll
            for (int j = 0; j < PS_HALO; ++j)
                if (j >= JP + 1) rh[j] = isIn(j) ? rh[j] - alf * qh[j] : c32{0, 0};
            ps_block_sum3<NWV>(xxs, dum3, dum4, sh);
            if (tid == 0) k.partB[(long)s * MAXNB + jwg] = xxs;
            PS_STAMP(11)
        }
        if (!alive || a.precondOnly) { if (a.precondOnly) continue; break; }
        // ---- the system has left the iteration: records (workgroup 0 of the group), r back to memory (a stalled or capped
        // system is continued by the host's classic loop with the fp64 preconditioner)
#pragma unroll
        for (int q = 0; q < PS_NO; ++q) {
            const int j = PS_HALO + q;
            if (isIn(j)) *ps_at(rsys, eo(j)) = r64[q];
        }
        if (jwg == 0 && tid == 0) {
            k.iters[s] = it - 1;
            k.errEst[s] = est;
            if (st) { k.status[s] = st; *k.failHost = st; }
            if (stalled) *k.stallHost = 1;
            else { k.active[s] = 0; if (atomicSub(k.nactive, 1) == 1) *k.nactHost = 0; }
        }
    }
    // ---- exit: the last workgroup to leave tells the host
    __syncthreads();
    if (tid == 0) {
        if (sflag[0] == 2) *k.failHost = HMCMT_EHIP;       // a wait timed out: the solve is void
        __threadfence_system();
        const unsigned nLeft = __hip_atomic_fetch_add(a.exitCnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (nLeft == gridDim.x - 1) {
            __threadfence_system();
            *(volatile int*)k.progHost = PS_DONE;
        }
    }
}
