// kernels_fused.h -- part of libhmcmt_hip.so; included by hmcmt_hip.hip INSIDE its anonymous namespace (one translation unit).
// The fused COCG iteration of the default path (k_spmv_fused, k_update_fused), the solve start (k_resid0, k_resid_pre),
// per-solve bookkeeping, the initial-guess extrapolation along the model path, and the true-residual check.
#pragma once

// ----------------------------------------------------------------------------------------------
// Fused COCG iteration of the default path (Jacobi/FDM/Jacobi preconditioner, mixed precision):
//   k_spmv_fused   : scalar bookkeeping (convergence test on the error estimate, beta), p = z + beta p
//                    recomputed on each node's 5-point halo, q = A p, partial p'q
//   k_update_fused : alpha, x += alpha p, r' = r - alpha q recomputed on the halo, Jacobi pre-smoothing
//                    t = r' - A (dinv .* r') written as complex64 for the transform, partial |x|^2
// Both exist in a second form (<2>) for the smoother with TWO damped Jacobi sweeps on each side of the FDM stage
// (Solver::sweeps, chosen per solve by the host: pick_sweeps): k_update_fused<2> does both pre-sweeps, k_spmv_fused<2>
// the second post-sweep (the first one is k_back_post<., 2>'s).
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
#define PH_STAMP(kid, i) if (k.stamps && k.stampKernel == (kid) && threadIdx.x == 0 && (long)blockIdx.x + (long)gridDim.x * blockIdx.y < 4096) k.stamps[((long)blockIdx.x + (long)gridDim.x * blockIdx.y) * 8 + (i)] = __builtin_amdgcn_s_memtime();
constexpr int SB = 4;                  // elements per thread and batch
constexpr int STALL_IT = 30;           // mixed-precision stagnation watch: iterations allowed per 10-fold drop of the error estimate
struct StenCo { double dk, dm, cy0, cy1, cz0, cz1; };

__device__ __forceinline__ int div_small(int i, float rcp) { return (int)(((float)i + 0.5f) * rcp); }   // i / n for i < 2^20, rcp = 1/n

// z and p are STORED as complex64 (8 B instead of 16 per value on the two busiest streams of the iteration): z is the
// output of a preconditioner whose FDM stage is fp32-class already, and a search direction rounded to fp32 is still a
// valid direction as long as x += alpha p, q = A p and r -= alpha q all use the SAME rounded p -- they do: p is rounded
// here before the stencil, the update kernel reads the stored value.  x, r, q and all inner products stay fp64.
// (CPU prototype on the headline systems: iteration counts +-2, final error 1e-12 either way.)
// SW = 2 (two sweeps per side, Solver::merged2): the preconditioned residual arrives as z4, the iterate after the FIRST
// post-sweep (k_back_post<., 2>), with rho = r'z already known from the identity of Solver::partR and |z4|^2 as the
// error estimate; the second post-sweep  z = z4 + dinv .* (r - A z4)  is done here, on the tile's rows and one halo row
// on each side (z4 staged with two), in fp32 like the rest of the smoother -- no launch of its own (k_post2: 8 us).
// Workgroup -> (row tile, system) of the two stencil kernels of the iteration: the plain 2-D grid.  Workgroups are dealt to the
// 8 XCDs round-robin by linear id, each XCD with its own 4 MB L2.  Two XCD-aware placements were measured in round 3
// (profiles/r03_xmap.md) and removed in round 4: a system's tiles on one XCD (shared halo rows: the same time and the same PMC
// bytes -- an XCD's workgroups stream ~20 MB through its 4 MB L2, the halo rows are served by the Infinity Cache either way), and
// the 16 frequencies of a (tile, polarisation) on one XCD (shared stencil coefficients: 2-5 % SLOWER -- 32 workgroups asking one L2
// for the same lines in the same microsecond queue up on its channels).  The persistent kernel (kernels_persist.h) is where
// placement pays: there a system's workgroups exchange through their XCD's L2 by design.
__device__ __forceinline__ bool tile_map(const Solver& k, int ntiles, int& tile, int& s) {
    (void)k; (void)ntiles;
    tile = blockIdx.x; s = blockIdx.y;
    return true;
}

template <int SW, int NT>             // NT: threads per workgroup (chosen by the host: launch_spmv)
__global__ __launch_bounds__(NT) void k_spmv_fused(Solver k, const double* partZZ, const float2* pin_, float2* pout, int it, int maxit) {
    extern __shared__ __attribute__((aligned(16))) char smem_[];
    cplx* pn = reinterpret_cast<cplx*>(smem_);            // [(RT+2)][NYP]
    __shared__ double sh[32];
    tick_begin(k.ticks, TK_SPMV);
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) *(volatile int*)k.progHost = it;   // "iteration it-1 is complete"
    // rows per tile: the two-sweep form may run on taller tiles of its own (Solver::RTS >= RT, fewer halo rows per own row;
    // its partial sums then fill the first slots of the k.NTR the consumer adds up, the rest are cleared below)
    const int RTt = SW == 2 ? k.RTS : k.RT;
    const int ntiles = (k.nz - 1 + RTt - 1) / RTt;
    int tile, s;
    if (!tile_map(k, ntiles, tile, s)) return;
    PH_STAMP(2, 0)
    const bool first = it == 1;
    const int act = k.active[s];
    const int ln = threadIdx.x & 63;
    // the system's partial sums (lane b fetches partial b), rho of the previous iteration
    cplx paL = ln < k.NB ? k.partA[(long)s * MAXNB + ln] : cplx{0, 0};
    if (SW == 2 && ln < k.NTR) paL += k.partR[(long)s * MAXNB + ln];
    const double pzL = ln < k.NB ? partZZ[(long)s * MAXNB + ln] : 0.0;
    const double pbL = k.partB[(long)s * MAXNB + ln];      // all MAXNB = 64 slots: zeroed at the start of a solve, filled by the row tiles of
                                                           // k_update_fused (<= NTR) or by the slabs of k_fdm_fwd (Solver::xInFwd; their number is independent of NTR)
    const cplx rhoPrev = k.rho2[(long)((it - 1) & 1) * k.S + s];
    const int mode = s >= k.nFreq;
    const long mo = (long)mode * k.vstride, so = (long)s * k.vstride;
    const double w = k.omega[s];
    const float2 *z = (SW == 2 ? k.z4_32 : k.z32) + so, *pi = pin_ + so;
    float2* po = pout + so;
    cplx* q = k.q + so;
    const int NYP = k.NYP, iz0 = 1 + tile * RTt, iz1 = min(iz0 + RTt - 1, k.nz - 1);   // own rows iz0..iz1
    const float rNYP = 1.0f / (float)NYP;
    // rows iz0-1 .. iz1+1 of the new direction (z, p vanish on boundary / pad nodes: no masking needed)
    const int nrows = iz1 - iz0 + 3, ntot = nrows * NYP, ebase = (iz0 - 1) * NYP;
    float2 zv[SB], pv[SB];
    auto ld_stage = [&](int i0) {
#pragma unroll
        for (int u = 0; u < SB; ++u) {
            const unsigned e = (unsigned)(ebase + min(i0 + u * NT, ntot - 1));
            zv[u] = z[e];
            pv[u] = first ? float2{0.f, 0.f} : pi[e];
        }
    };
    // SW = 2: the first batch of the z4 rows (iz0-2 .. iz1+2) and of the second post-sweep's operands, requested here as
    // well -- unconditionally, clamped, apart from their use.  (Until round 3 the z4 staging loop loaded each element
    // inside `if (row in the mesh)` right in front of its LDS store: one memory round trip per element, 4.5 per thread
    // in a row -- 11 500 of the kernel's 24 000 ticks, HMCMT_STAMPS=spmv.)
    constexpr int ZB = 5;                                          // z4 elements per thread and batch
    constexpr int UB2 = 4;
    const int nz4 = (nrows + 2) * NYP;
    float2 zq[ZB];
    bool zin[ZB];
    auto ldz = [&](int i0) {
#pragma unroll
        for (int u = 0; u < ZB; ++u) {
            const int i = min(i0 + u * NT, nz4 - 1);
            const int lr = div_small(i, rNYP), row = iz0 - 2 + lr;
            zin[u] = row >= 0 && row <= k.nz;
            zq[u] = z[(unsigned)(min(max(row, 0), k.nz) * NYP + (i - lr * NYP))];
        }
    };
    const cplx* rr = k.r + so;
    const float2* di = k.dinv32 + so;
    const float4* cf = k.cf32 + 2 * mo;
    float4 ca[UB2], cb[UB2];
    cplx rv[UB2];
    float2 pv2[UB2], dv[UB2];
    bool in[UB2];
    auto ld2 = [&](int i0) {
#pragma unroll
        for (int u = 0; u < UB2; ++u) {
            const int i = min(i0 + u * NT, ntot - 1);
            const int lr = div_small(i, rNYP), iy = i - lr * NYP, row = iz0 - 1 + lr;
            in[u] = row >= 1 && row <= k.nz - 1 && iy >= 1 && iy <= k.ny - 1;
            const unsigned e = (unsigned)(row * NYP + iy);
            ca[u] = cf[2u * e]; cb[u] = cf[2u * e + 1u]; rv[u] = rr[e]; dv[u] = di[e];
            pv2[u] = first ? float2{0.f, 0.f} : pi[e];
        }
    };
    if (!act) return;
    if (SW == 1) ld_stage(threadIdx.x);
    if (SW == 2) { ldz(threadIdx.x); ld2(threadIdx.x); }        // (behind the test: the workgroups of a converged system must not pull 60 KB each through the fabric)
    const cplx rz = cplx{wave_sum(paL.re), wave_sum(paL.im)};
    const double zz = wave_sum(pzL), xx = wave_sum(pbL);
    bool on = true;
    int st = 0;
    if (first) { if (zz == 0.0) on = false; }
    else if (zz <= k.tol2 * xx) on = false;
    else if (it - 1 >= maxit) { on = false; st = HMCMT_ENOCONV; }
    if (!(isfinite(rz.re) && isfinite(rz.im) && isfinite(zz) && isfinite(xx))) { on = false; st = HMCMT_EBREAKDOWN; }
    const cplx be = first ? cplx{0, 0} : rz / rhoPrev;
    if (tile == 0 && threadIdx.x == 0) {
        k.rho2[(long)(it & 1) * k.S + s] = rz;
        k.iters[s] = it - 1;
        const double est = first ? (zz == 0.0 ? 0.0 : 1.0) : sqrt(zz / xx);
        k.errEst[s] = est;
        if (st) { k.status[s] = st; *k.failHost = st; __threadfence_system(); }     // (in front of the counter the host polls: it reads this word right behind it)
        // stagnation watch (the host restarts the stragglers with the fp64 preconditioner when it fires)
        if (first || est < 0.1 * k.errRef[s]) { k.errRef[s] = est; k.errRefIt[s] = it; }
        else if (on && it - k.errRefIt[s] > k.stallIt) *k.stallHost = 1;
    }
    if (!on) {
        // every block of this system takes the same decision; block 0 records it (a block that starts late and
        // already sees the cleared flag returns just the same)
        if (tile == 0 && threadIdx.x == 0) { k.active[s] = 0; if (atomicSub(k.nactive, 1) == 1) *k.nactHost = 0; }
        return;
    }
    if (k.cntActive && tile == 0 && threadIdx.x == 0) atomicAdd(k.cntActive, 1ull);   // (roofline accounting only)
    PH_STAMP(2, 1)
    if constexpr (SW == 2) {
        // z4 on rows iz0-2 .. iz1+2 (outside the mesh: zero) -> LDS, then z = z4 + dinv .* (r - A z4) on rows iz0-1 .. iz1+1
        c32* z4s = reinterpret_cast<c32*>(pn + (long)(RTt + 2) * NYP);      // [(RT+4)][NYP]
        for (int i0 = threadIdx.x; i0 < nz4; i0 += ZB * NT) {
            if (i0 != (int)threadIdx.x) ldz(i0);
#pragma unroll
            for (int u = 0; u < ZB; ++u) {
                const int i = i0 + u * NT;
                if (i < nz4) z4s[i] = zin[u] ? c32{zq[u].x, zq[u].y} : c32{0.f, 0.f};
            }
        }
        const float wf = (float)w;
        PH_STAMP(2, 2)
        __syncthreads();                                         // z4s complete
        PH_STAMP(2, 3)
        for (int i0 = threadIdx.x; i0 < ntot; i0 += UB2 * NT) {
            if (i0 != (int)threadIdx.x) ld2(i0);
#pragma unroll
            for (int u = 0; u < UB2; ++u) {
                const int i = i0 + u * NT;
                if (i < ntot) {
                    cplx zc = cplx{0, 0};
                    if (in[u]) {
                        const int l = i + NYP;                        // the same node in z4s
                        const c32 c = z4s[l];
                        const float dm = wf * ca[u].y;
                        const c32 az = c32{ca[u].x * c.re - dm * c.im + ca[u].z * z4s[l + 1].re + ca[u].w * z4s[l - 1].re + cb[u].x * z4s[l + NYP].re + cb[u].y * z4s[l - NYP].re,
                                           ca[u].x * c.im + dm * c.re + ca[u].z * z4s[l + 1].im + ca[u].w * z4s[l - 1].im + cb[u].x * z4s[l + NYP].im + cb[u].y * z4s[l - NYP].im};
                        const cplx res = cplx{rv[u].re - (double)az.re, rv[u].im - (double)az.im};
                        zc = cplx{(double)c.re, (double)c.im} + cplx{(double)dv[u].x, (double)dv[u].y} * res;
                    }
                    const cplx v = first ? zc : zc + be * cplx{(double)pv2[u].x, (double)pv2[u].y};
                    const float2 vf = float2{(float)v.re, (float)v.im};
                    pn[i] = cplx{(double)vf.x, (double)vf.y};          // the stencil sees the value that is stored
                    if (i >= NYP && i < ntot - NYP) po[ebase + i] = vf;
                }
            }
        }
    } else
    for (int i0 = threadIdx.x; i0 < ntot; i0 += SB * NT) {
        if (i0 != (int)threadIdx.x) ld_stage(i0);
#pragma unroll
        for (int u = 0; u < SB; ++u) {
            const int i = i0 + u * NT;
            if (i < ntot) {
                const cplx zc = cplx{(double)zv[u].x, (double)zv[u].y};
                const cplx v = first ? zc : zc + be * cplx{(double)pv[u].x, (double)pv[u].y};
                const float2 vf = float2{(float)v.re, (float)v.im};
                pn[i] = cplx{(double)vf.x, (double)vf.y};          // the stencil sees the value that is stored
                if (i >= NYP && i < ntot - NYP) po[ebase + i] = vf;
            }
        }
    }
    const int nown = (iz1 - iz0 + 1) * NYP, obase = iz0 * NYP;
    const double *dKm = k.dK + mo, *dMm = k.dM + mo, *cYm = k.cY + mo, *cZm = k.cZ + mo, *cZu = k.cZ + mo - NYP;
    StenCo co[SB];
    auto ld_co = [&](int i0) {
#pragma unroll
        for (int u = 0; u < SB; ++u) {
            const unsigned e = (unsigned)(obase + min(i0 + u * NT, nown - 1));
            co[u].dk = dKm[e]; co[u].dm = dMm[e];
            co[u].cy0 = cYm[e]; co[u].cy1 = cYm[e - 1u];
            co[u].cz0 = cZm[e]; co[u].cz1 = cZu[e];
        }
    };
    ld_co(threadIdx.x);
    PH_STAMP(2, 4)
    __syncthreads();
    PH_STAMP(2, 5)
    double ar = 0, ai = 0;
    for (int i0 = threadIdx.x; i0 < nown; i0 += SB * NT) {
        if (i0 != (int)threadIdx.x) ld_co(i0);
#pragma unroll
        for (int u = 0; u < SB; ++u) {
            const int i = i0 + u * NT;
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
    if (threadIdx.x == 0) k.partPQ[(long)s * MAXNB + tile] = cplx{ar, ai};
    if (SW == 2 && tile == 0)
        for (int b = ntiles + threadIdx.x; b < k.NTR; b += NT) k.partPQ[(long)s * MAXNB + b] = cplx{0, 0};
    PH_STAMP(2, 6)
    tick_end(k.ticks, TK_SPMV);
}

// SW = 2 (two damped Jacobi sweeps on each side of the FDM stage, k.sweeps == 2): the pre-smoother becomes
//   z1 = D r', t1 = r' - A z1, z2 = z1 + D t1 (= D (r' + t1)), t = r' - A z2      (D = dinv)
// on the tile's rows, with r' and z1 recomputed on TWO halo rows on each side and z2 on one; z2 is also written out
// (complex64): the back transform adds it to the FDM correction (k_back_post<., 2>).  The tile's intermediate
// values live in LDS as complex64 -- they are internals of the preconditioner, whose output t goes to 16-bit planes
// anyway -- so the three arrays of (RT + 4), (RT + 2), (RT + 2) rows fit where the two fp64 arrays of SW = 1 do.
// startOnly (SW = 2): the pre-smoothing of the residual of a solve's first preconditioner application -- alpha = 0,
// x and r stay untouched (k_resid_pre does this for one sweep).
template <int SW, int NT, int UBX>    // NT: threads per workgroup; UBX: elements per thread and batch of the two-sweep form
__global__ __launch_bounds__(NT) void k_update_fused(Solver k, const float2* pcur, const cplx* rin, cplx* rout, int it, int startOnly) {
    const int RTt = SW == 2 ? k.RT2 : k.RT;
    tick_begin(k.ticks, TK_UPDATE);
    const int ntiles = (k.nz - 1 + RTt - 1) / RTt;
    int tile, s;
    if (!tile_map(k, ntiles, tile, s)) return;
    if (!k.active[s]) return;
    extern __shared__ __attribute__((aligned(16))) char smem_[];
    const int NYP = k.NYP, iz0 = 1 + tile * RTt, iz1 = min(iz0 + RTt - 1, k.nz - 1);
    const int nrows = iz1 - iz0 + 3;
    if constexpr (SW == 2) {
        c32* z1s = reinterpret_cast<c32*>(smem_);             // [(RT+4)][NYP]  z1 = dinv .* r'   rows iz0-2 .. iz1+2
        c32* rs = z1s + (long)(RTt + 4) * NYP;                // [(RT+2)][NYP]  r'                rows iz0-1 .. iz1+1
        c32* z2s = rs + (long)(RTt + 2) * NYP;               // [(RT+2)][NYP]  z2                rows iz0-1 .. iz1+1
        __shared__ double sh2[32];
        cplx al = cplx{0, 0};
        if (!startOnly) {
            const cplx pq = total_part(k.partPQ + (long)s * MAXNB, k.NTR);
            al = k.rho2[(long)(it & 1) * k.S + s] / pq;
        }
        const int mode = s >= k.nFreq;
        const long mo = (long)mode * k.vstride, so = (long)s * k.vstride;
        const float w = (float)k.omega[s];
        const float rNYP = 1.0f / (float)NYP;
        const float2* p = pcur + so;
        const cplx *q = k.q + so, *ri = rin + so;
        const float2* di = k.dinv32 + so;
        cplx *x = k.x + so, *ro = rout + so;
        float2 *t = k.t32 + so, *z2o = k.zs32 + so, *t2o = k.t2_32 + so;
        double xx = 0, dummy = 0, p1r = 0, p1i = 0;
        PH_STAMP(1, 0)
        // Every phase is a short chain (loads -> arithmetic -> LDS -> barrier): the loads of a batch of UB elements per
        // thread are issued together, unconditionally (clamped addresses), apart from their use -- inside `if (valid)` the
        // compiler keeps each load next to its use and a phase costs one memory round trip per element (48 VGPRs,
        // 20.6 us) instead of one per batch.
        constexpr int UB = UBX;
        const int nA = (nrows + 2) * NYP, nB = nrows * NYP, nown = (iz1 - iz0 + 1) * NYP;
        // phase A: r' and z1 on rows iz0-2 .. iz1+2 (rows outside the mesh: zero)
        for (int i0 = threadIdx.x; i0 < nA; i0 += UB * NT) {
            cplx rv[UB], qv[UB], xv[UB];
            float2 pv[UB], dv[UB];
            long ee[UB];
            int lrs[UB];
            bool ok[UB];
#pragma unroll
            for (int u = 0; u < UB; ++u) {
                const int i = min(i0 + u * NT, nA - 1);
                const int lr = div_small(i, rNYP), iy = i - lr * NYP, row = iz0 - 2 + lr;
                ok[u] = i0 + u * NT < nA && row >= 0 && row <= k.nz;
                lrs[u] = lr;
                ee[u] = (long)min(max(row, 0), k.nz) * NYP + iy;
                rv[u] = ri[ee[u]]; dv[u] = di[ee[u]];
                if (!startOnly) {
                    qv[u] = q[ee[u]];
                    if (!k.xInFwd && lr >= 2 && lr <= nrows - 1) { xv[u] = x[ee[u]]; pv[u] = p[ee[u]]; }     // (own rows only: a predicated load)
                }
            }
#pragma unroll
            for (int u = 0; u < UB; ++u) {
                const int i = i0 + u * NT;
                if (i < nA) {
                    cplx rn = cplx{0, 0}, z1 = cplx{0, 0};
                    if (ok[u]) {
                        rn = startOnly ? rv[u] : rv[u] - al * qv[u];
                        z1 = cplx{(double)dv[u].x, (double)dv[u].y} * rn;
                    }
                    z1s[i] = c32{(float)z1.re, (float)z1.im};
                    const int lr = lrs[u];
                    if (lr >= 1 && lr <= nrows) rs[i - NYP] = c32{(float)rn.re, (float)rn.im};
                    if (!startOnly && lr >= 2 && lr <= nrows - 1) {
                        ro[ee[u]] = rn;
                        if (!k.xInFwd) {
                            const cplx xn = xv[u] + al * cplx{(double)pv[u].x, (double)pv[u].y};
                            x[ee[u]] = xn;
                            xx += cabs2(xn);
                        }
                    }
                }
            }
        }
        auto sten = [&](const c32* u, int l, const float4& ca, const float4& cb) -> c32 {   // (A u) at tile index l
            const c32 c = u[l];
            const float dm = w * ca.y;
            return c32{ca.x * c.re - dm * c.im + ca.z * u[l + 1].re + ca.w * u[l - 1].re + cb.x * u[l + NYP].re + cb.y * u[l - NYP].re,
                       ca.x * c.im + dm * c.re + ca.z * u[l + 1].im + ca.w * u[l - 1].im + cb.x * u[l + NYP].im + cb.y * u[l - NYP].im};
        };
        // phase B: z2 = z1 + D (r' - A z1) on rows iz0-1 .. iz1+1; the first batch's coefficients are requested before the barrier
        const float4* cf = k.cf32 + 2 * mo;
        {
            float4 ca[UB], cb[UB];
            float2 dv[UB];
            bool in[UB];
            auto ldB = [&](int i0) {
#pragma unroll
                for (int u = 0; u < UB; ++u) {
                    const int i = min(i0 + u * NT, nB - 1);
                    const int lr = div_small(i, rNYP), iy = i - lr * NYP, row = iz0 - 1 + lr;
                    in[u] = i0 + u * NT < nB && row >= 1 && row <= k.nz - 1 && iy >= 1 && iy <= k.ny - 1;
                    const long e = (long)row * NYP + iy;
                    ca[u] = cf[2 * e]; cb[u] = cf[2 * e + 1]; dv[u] = di[e];
                }
            };
            ldB(threadIdx.x);
            PH_STAMP(1, 1)
            __syncthreads();                                         // z1s, rs complete
            PH_STAMP(1, 2)
            for (int i0 = threadIdx.x; i0 < nB; i0 += UB * NT) {
                if (i0 != (int)threadIdx.x) ldB(i0);
#pragma unroll
                for (int u = 0; u < UB; ++u) {
                    const int i = i0 + u * NT;
                    if (i < nB) {
                        c32 z2 = c32{0.f, 0.f};
                        if (in[u]) {
                            const int l = i + NYP;                       // the same node in z1s
                            const c32 t1 = rs[i] - sten(z1s, l, ca[u], cb[u]);
                            const c32 dt = c32{k.w2 * dv[u].x, k.w2 * dv[u].y} * t1;
                            z2 = c32{z1s[l].re + dt.re, z1s[l].im + dt.im};
                        }
                        z2s[i] = z2;
                    }
                }
            }
        }
        // phase C: t = r' - A z2 on the own rows; z2 of the own rows goes out too
        {
            float4 ca[UB], cb[UB];
            auto ldC = [&](int i0) {
#pragma unroll
                for (int u = 0; u < UB; ++u) {
                    const long e = (long)iz0 * NYP + min(i0 + u * NT, nown - 1);
                    ca[u] = cf[2 * e]; cb[u] = cf[2 * e + 1];
                }
            };
            ldC(threadIdx.x);
            PH_STAMP(1, 3)
            __syncthreads();                                         // z2s complete
            PH_STAMP(1, 4)
            for (int i0 = threadIdx.x; i0 < nown; i0 += UB * NT) {
                if (i0 != (int)threadIdx.x) ldC(i0);
#pragma unroll
                for (int u = 0; u < UB; ++u) {
                    const int i = i0 + u * NT;
                    if (i < nown) {
                        const int lr = div_small(i, rNYP), iy = i - lr * NYP;
                        const int l = i + NYP;                         // the same node in rs / z2s
                        c32 out = c32{0.f, 0.f};
                        if (iy >= 1 && iy <= k.ny - 1) out = rs[l] - sten(z2s, l, ca[u], cb[u]);
                        store_t32(k, t, iz0 + lr, iy, out.re, out.im);
                        const c32 z2v = z2s[l];
                        z2o[(long)iz0 * NYP + i] = float2{z2v.re, z2v.im};
                        t2o[(long)iz0 * NYP + i] = float2{out.re, out.im};
                        const double sr = (double)rs[l].re + (double)out.re, si = (double)rs[l].im + (double)out.im;     // (r' + t) .* z2
                        p1r += sr * z2v.re - si * z2v.im; p1i += sr * z2v.im + si * z2v.re;
                    }
                }
            }
        }
        __shared__ double sh3[32];
        PH_STAMP(1, 5)
        block_sum2(p1r, p1i, sh3);
        if (threadIdx.x == 0) k.partR[(long)s * MAXNB + tile] = cplx{p1r, p1i};
        if (tile == 0)                                                 // (fewer tiles than the k.NTR partial sums the consumers add up)
            for (int b = ntiles + threadIdx.x; b < k.NTR; b += NT) {
                k.partR[(long)s * MAXNB + b] = cplx{0, 0};
                if (!startOnly && !k.xInFwd) k.partB[(long)s * MAXNB + b] = 0.0;
            }
        if (startOnly) return;
        if (k.xInFwd) {                                                // (x and |x|^2: k_fdm_fwd's idle waves)
            if (threadIdx.x == 0 && tile == 0) k.alphaBeta[s] = al;
            PH_STAMP(1, 6)
            return;
        }
        block_sum2(xx, dummy, sh2);
        if (threadIdx.x == 0) {
            k.partB[(long)s * MAXNB + tile] = xx;
            if (tile == 0) k.alphaBeta[s] = al;
        }
        PH_STAMP(1, 6)
        return;
    }
    cplx* cs = reinterpret_cast<cplx*>(smem_);            // [(RT+2)][NYP]  dinv .* r'
    cplx* rs = cs + (long)(k.RT + 2) * NYP;               // [RT][NYP]      r' of the own rows
    __shared__ double sh[32];
    const cplx pq = total_part(k.partPQ + (long)s * MAXNB, k.NTR);
    const cplx al = k.rho2[(long)(it & 1) * k.S + s] / pq;
    const int mode = s >= k.nFreq;
    const long mo = (long)mode * k.vstride, so = (long)s * k.vstride;
    const double w = k.omega[s];
    const float2* p = pcur + so;
    const cplx *q = k.q + so, *ri = rin + so;
    const float2* di = k.dinv32 + so;      // (complex64 since round 3, as in the two-sweep form: the smoother's output goes to bf16 planes anyway)
    cplx *x = k.x + so, *ro = rout + so;
    float2* t = k.t32 + so;
    double xx = 0, dummy = 0;
    // r' = r - alpha q and dinv .* r' on rows iz0-1 .. iz1+1 (r, q, dinv vanish outside the interior: no masking)
    for (int i = threadIdx.x; i < nrows * NYP; i += NT) {
        const int lr = i / NYP, iy = i - lr * NYP;
        const long e = (long)(iz0 - 1 + lr) * NYP + iy;
        const cplx rn = ri[e] - al * q[e];
        const float2 d2 = di[e];
        cs[i] = cplx{(double)d2.x, (double)d2.y} * rn;
        if (lr >= 1 && lr <= nrows - 2) {
            rs[i - NYP] = rn;
            ro[e] = rn;
            if (!k.xInFwd) {
                const float2 pf = p[e];
                const cplx xv = x[e] + al * cplx{(double)pf.x, (double)pf.y};     // p vanishes outside the interior
                x[e] = xv;
                xx += cabs2(xv);
            }
        }
    }
    // The smoother's stencil coefficients as the packed floats the two-sweep form reads (Solver::cf32; its output goes to bf16
    // planes), a batch of UC nodes per thread requested together, unconditionally, and the first batch BEFORE the barrier
    // (round 3: until then six fp64 loads per node sat inside `if (interior)` next to their use -- a memory round trip per
    // node, three in a row per thread).
    constexpr int UC = 3;
    const int nown = (iz1 - iz0 + 1) * NYP;
    const float4* cf = k.cf32 + 2 * mo + 2 * (long)iz0 * NYP;
    float4 ca[UC], cb[UC];
    auto ldc = [&](int i0) {
#pragma unroll
        for (int u = 0; u < UC; ++u) { const unsigned io = (unsigned)min(i0 + u * NT, nown - 1); ca[u] = cf[2u * io]; cb[u] = cf[2u * io + 1u]; }
    };
    ldc(threadIdx.x);
    __syncthreads();
    for (int i0 = threadIdx.x; i0 < nown; i0 += UC * NT) {
        if (i0 != (int)threadIdx.x) ldc(i0);
#pragma unroll
        for (int u = 0; u < UC; ++u) {
            const int i = i0 + u * NT;
            if (i < nown) {
                const int lr = i / NYP, iy = i - lr * NYP;
                cplx out = cplx{0, 0};
                if (iy >= 1 && iy <= k.ny - 1) {
                    const int l = (lr + 1) * NYP + iy;
                    const cplx c = cs[l];
                    const double dk = (double)ca[u].x, dm = w * (double)ca[u].y;
                    cplx acc = cplx{dk * c.re - dm * c.im, dk * c.im + dm * c.re};
                    acc += (double)ca[u].z * cs[l + 1];
                    acc += (double)ca[u].w * cs[l - 1];
                    acc += (double)cb[u].x * cs[l + NYP];
                    acc += (double)cb[u].y * cs[l - NYP];
                    out = rs[i] - acc;
                }
                store_t32(k, t, iz0 + lr, iy, (float)out.re, (float)out.im);
            }
        }
    }
    if (k.xInFwd) {                                                    // (x and |x|^2: k_fdm_fwd's idle waves)
        if (threadIdx.x == 0 && tile == 0) k.alphaBeta[s] = al;
        return;
    }
    block_sum2(xx, dummy, sh);
    if (threadIdx.x == 0) {
        k.partB[(long)s * MAXNB + tile] = xx;
        if (tile == 0) k.alphaBeta[s] = al;
    }
    tick_end(k.ticks, TK_UPDATE);
}

// warm start: r <- r - A x over interior nodes, x including whatever sits on its boundary nodes
// (forward: Dirichlet values, so with r = 0 on entry this is the reference's rhs -Aio*bc minus Aii*x0)
// sysOn != nullptr: workgroup (0,0) also does k_solve_begin's bookkeeping for the solve that follows (one launch less
// on the critical path in front of each solve)
// zero_r: 0 the right-hand side is read from r at every node; 1 it is zero (the forward problem: the sources are the Dirichlet
// values in x); 2 + row: it is read on node rows row, row + 1 only and zero elsewhere -- the adjoint sources live on the two
// node rows of the receiver layer (item_src), so the buffer need not be cleared in front of k_src (a 11 MB fill and an API
// call between the solves) nor read here outside those rows
// onlyActive: systems whose active flag is off are left alone (behind a persistent launch that was to form the residual itself and
// left some systems untouched: kernels_persist.h, PsLaunch::resid)
__global__ __launch_bounds__(VBLOCK) void k_resid0(Solver k, cplx* x, int zero_r, const int* __restrict__ sysOn, int onlyActive) {
    const int s = blockIdx.y;
    if (onlyActive && !k.active[s]) return;
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
    // RB nodes per thread and pass with every operand requested up front, at clamped addresses (inside `if (interior)` the
    // compiler keeps a load next to its use: one memory round trip per node and thread, lesson (4) of DESIGN.md section 5)
    constexpr int RB = 4;
    for (long eb = e0 + threadIdx.x; eb < e1; eb += (long)RB * VBLOCK) {
        double dk[RB], dm[RB], cy0[RB], cy1[RB], cz0[RB], cz1[RB];
        cplx uc[RB], ue[RB], uw[RB], us[RB], un[RB], bv[RB];
#pragma unroll
        for (int b = 0; b < RB; ++b) {
            const long e = min(eb + (long)b * VBLOCK, e1 - 1);
            const int iz = (int)(e / k.NYP), iy = (int)(e - (long)iz * k.NYP);
            const long ec = (long)min(max(iz, 1), k.nz - 1) * k.NYP + min(max(iy, 1), k.ny - 1);
            dk[b] = k.dK[mo + ec]; dm[b] = k.dM[mo + ec];
            cy0[b] = k.cY[mo + ec]; cy1[b] = k.cY[mo + ec - 1]; cz0[b] = k.cZ[mo + ec]; cz1[b] = k.cZ[mo + ec - k.NYP];
            uc[b] = u[ec]; ue[b] = u[ec + 1]; uw[b] = u[ec - 1]; us[b] = u[ec + k.NYP]; un[b] = u[ec - k.NYP];
            const int izc = min(max(iz, 1), k.nz - 1);
            bv[b] = (zero_r == 0 || (zero_r >= 2 && (unsigned)(izc - (zero_r - 2)) < 2u)) ? r[ec] : cplx{0, 0};
        }
#pragma unroll
        for (int b = 0; b < RB; ++b) {
            const long e = eb + (long)b * VBLOCK;
            if (e >= e1) continue;
            const int iz = (int)(e / k.NYP), iy = (int)(e - (long)iz * k.NYP);
            cplx out = cplx{0, 0};
            if (iz >= 1 && iz <= k.nz - 1 && iy >= 1 && iy <= k.ny - 1) {
                const double dmw = w * dm[b];
                cplx acc = cplx{dk[b] * uc[b].re - dmw * uc[b].im, dk[b] * uc[b].im + dmw * uc[b].re};
                acc += cy0[b] * ue[b];
                acc += cy1[b] * uw[b];
                acc += cz0[b] * us[b];
                acc += cz1[b] * un[b];
                out = bv[b] - acc;
            }
            r[e] = out;
        }
    }
}

// k_resid0 + k_pre_c64 in one launch (the start of a solve on the default path): on a tile of RT rows
//   r = b - A x0        on the tile's rows and one halo row on each side (x0 staged with two halo rows)
//   t = r - A (dinv r)  on the tile's rows, written in the transform's input format
// so the residual is not re-read by a second kernel and one launch disappears in front of each solve.  The residual is
// written to a SECOND buffer (rout): the halo rows' b are read from rin while the neighbouring workgroups write theirs.
// Workgroup (0,0) also does k_solve_begin's bookkeeping.
template <int NT>              // threads: one workgroup per CU at the headline size, so 512-1024 (16 waves) hide what 256 cannot
__global__ __launch_bounds__(NT) void k_resid_pre(Solver k, const cplx* x, const cplx* rin, cplx* rout, int zero_r,
                                                      const int* __restrict__ sysOn) {
    const int s = blockIdx.y;
    tick_begin(k.ticks, zero_r == 1 ? TK_RESID_F : TK_RESID_A);
    if (blockIdx.x == 0 && blockIdx.y == 0) {
        for (int t = threadIdx.x; t < k.S * MAXNB; t += NT) k.partB[t] = 0.0;
        for (int t = threadIdx.x; t < k.S; t += NT) { k.active[t] = sysOn[t]; k.iters[t] = 0; k.status[t] = 0; }
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
    for (int i = threadIdx.x; i < (nown + 4) * NYP; i += NT) {
        const int lr = div_small(i, rNYP), g = iz0 - 2 + lr;
        xs[i] = (g >= 0 && g <= k.nz) ? u[(long)g * NYP + (i - lr * NYP)] : cplx{0, 0};
    }
    __syncthreads();
    constexpr int RB = 4;             // nodes per thread and pass, operands requested up front at clamped addresses (as k_resid0)
    const int n2 = (nown + 2) * NYP;
    for (int ib = threadIdx.x; ib < n2; ib += RB * NT) {
        double dk[RB], dm[RB], cy0[RB], cy1[RB], cz0[RB], cz1[RB];
        cplx bv[RB];
#pragma unroll
        for (int b = 0; b < RB; ++b) {
            const int i = min(ib + b * NT, n2 - 1);
            const int lr = div_small(i, rNYP), iy = i - lr * NYP, g = iz0 - 1 + lr;
            const long ec = (long)min(max(g, 1), k.nz - 1) * NYP + min(max(iy, 1), k.ny - 1);
            dk[b] = k.dK[mo + ec]; dm[b] = k.dM[mo + ec];
            cy0[b] = k.cY[mo + ec]; cy1[b] = k.cY[mo + ec - 1]; cz0[b] = k.cZ[mo + ec]; cz1[b] = k.cZ[mo + ec - NYP];
            const int gc = min(max(g, 1), k.nz - 1);
            bv[b] = (zero_r == 0 || (zero_r >= 2 && (unsigned)(gc - (zero_r - 2)) < 2u)) ? rin[so + ec] : cplx{0, 0};
        }
#pragma unroll
        for (int b = 0; b < RB; ++b) {
            const int i = ib + b * NT;
            if (i >= n2) continue;
            const int lr = div_small(i, rNYP), iy = i - lr * NYP, g = iz0 - 1 + lr;
            const long e = (long)g * NYP + iy;
            cplx out = cplx{0, 0};
            if (g >= 1 && g <= k.nz - 1 && iy >= 1 && iy <= k.ny - 1) {
                const int l = i + NYP;                         // the same node in xs (one more halo row in front)
                const cplx c = xs[l];
                const double dmw = w * dm[b];
                cplx acc = cplx{dk[b] * c.re - dmw * c.im, dk[b] * c.im + dmw * c.re};
                acc += cy0[b] * xs[l + 1];
                acc += cy1[b] * xs[l - 1];
                acc += cz0[b] * xs[l + NYP];
                acc += cz1[b] * xs[l - NYP];
                out = bv[b] - acc;
            }
            rs[i] = out;
            if (lr >= 1 && lr <= nown) rout[so + e] = out;
        }
    }
    // the two boundary rows of r and of t (zeros)
    if (blockIdx.x == 0 || iz1 == k.nz - 1) {
        const int row = blockIdx.x == 0 ? 0 : k.nz;
        for (int iy = threadIdx.x; iy < NYP; iy += NT) {
            rout[so + (long)row * NYP + iy] = cplx{0, 0};
            store_t32(k, k.t32 + so, row, iy, 0.f, 0.f);
        }
        if (blockIdx.x == 0 && iz1 == k.nz - 1)            // (a single tile: both rows)
            for (int iy = threadIdx.x; iy < NYP; iy += NT) {
                rout[so + (long)k.nz * NYP + iy] = cplx{0, 0};
                store_t32(k, k.t32 + so, k.nz, iy, 0.f, 0.f);
            }
    }
    __syncthreads();
    // t = r - A (dinv r) with the smoother's operands as every later application of this solve reads them (k_update_fused<1>):
    // the complex64 Jacobi diagonal and the packed float coefficients (Solver::cf32), two 16-byte loads per node
    const float2* di = k.dinv32 + so;
    for (int i = threadIdx.x; i < n2; i += NT) {
        const float2 d = di[(long)(iz0 - 1) * NYP + i];
        xs[i] = cplx{(double)d.x, (double)d.y} * rs[i];
    }
    __syncthreads();
    const float4* cfm = k.cf32 + 2 * mo;
    const int n4 = nown * NYP;
    for (int ib = threadIdx.x; ib < n4; ib += RB * NT) {
        float4 ca[RB], cb[RB];
#pragma unroll
        for (int b = 0; b < RB; ++b) {
            const int i = min(ib + b * NT, n4 - 1);
            const int lr = div_small(i, rNYP), iy = i - lr * NYP;
            const long ec = (long)(iz0 + lr) * NYP + min(max(iy, 1), k.ny - 1);
            ca[b] = cfm[2 * ec]; cb[b] = cfm[2 * ec + 1];
        }
#pragma unroll
        for (int b = 0; b < RB; ++b) {
            const int i = ib + b * NT;
            if (i >= n4) continue;
            const int lr = div_small(i, rNYP), iy = i - lr * NYP;
            cplx out = cplx{0, 0};
            if (iy >= 1 && iy <= k.ny - 1) {
                const int l = i + NYP;
                const cplx c = xs[l];
                const double dk = ca[b].x, dm = w * (double)ca[b].y;
                cplx acc = cplx{dk * c.re - dm * c.im, dk * c.im + dm * c.re};
                acc += (double)ca[b].z * xs[l + 1];
                acc += (double)ca[b].w * xs[l - 1];
                acc += (double)cb[b].x * xs[l + NYP];
                acc += (double)cb[b].y * xs[l - NYP];
                out = rs[l] - acc;
            }
            store_t32(k, k.t32 + so, iz0 + lr, iy, (float)out.re, (float)out.im);
        }
    }
    tick_end(k.ticks, zero_r == 1 ? TK_RESID_F : TK_RESID_A);
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
    tick_begin(k.ticks, TK_SOLVE_END);
    if (s >= k.S) return;
    recI[kind * k.S + s] = k.iters[s];
    recI[(2 + kind) * k.S + s] = k.status[s];
    recE[kind * k.S + s] = k.errEst[s];
    tick_end(k.ticks, TK_SOLVE_END);
}

// ---- initial guess extrapolated along the model path (options.warm_start == 2) ----
// The fields are smooth functions of the model, and a leapfrog trajectory moves the model along an almost
// straight line at almost constant speed.  With the last three models on that line at "times"
// tau = -1-gamma, -1, 0 (unit = the last step; gamma, alpha = projections of the previous / the new step on
// the last one) the guess is the Lagrange extrapolation of the last three fields to tau = alpha:
//     x0 = w2 x_k + w1 x_{k-1} + w0 x_{k-2}      (uniform steps: 3, -3, 1)
// when the three steps are nearly collinear, else the linear one  x0 = x_k + alpha (x_k - x_{k-1})
// (e.g. across a momentum refresh).  State per solve kind, all on the device (no host round trip):
// hist = the last NP models (a ring), ext = {w_k, ..., w_{k-NP+1}, keep, count, ring heads}; every further step that is
// nearly collinear with the last one and of comparable length adds a point (and an order) to the Lagrange
// extrapolation, up to EXT_NP fields.  A repeated model (getHamiltonian after the last leapfrog step) keeps the
// history untouched.
constexpr int EXT_NP = 6;          // fields kept per solve kind: the current one + EXT_NP-1 earlier ones (Lagrange order <= EXT_NP-1)
constexpr int EXT_NBLK = 32;       // blocks of the partial-sum pass
constexpr int EXT_NS = 2 * EXT_NP; // partial sums per block: <d_j,d1> (j = 0..NP-1), <d_j,d_j> (j = 0, 2..NP-1), <m_k,m_k>
constexpr int EXT_KEEP = EXT_NP, EXT_COUNT = EXT_NP + 1, EXT_HEAD = EXT_NP + 2, EXT_MHEAD = EXT_NP + 3, EXT_PART = EXT_NP + 4;
// ext = {w_0..w_{NP-1}, keep, count, head of the field ring, head of the model ring, partial sums [NBLK][NS], ticket}
constexpr int EXT_TICKET = EXT_PART + EXT_NS * EXT_NBLK, EXT_LEN = EXT_TICKET + 1;

// The weights from the summed partials (one thread).
// coldUnlessSmooth (the adjoint solve): without at least three collinear points the weights are all zero -- the adjoint's
// right-hand side is the weighted residual, which changes by O(1) from step to step, so its previous solution is no
// better a guess than zero (measured on the sampler's trajectories: 0-4 iterations WORSE than a cold start) unless the
// model path is smooth enough for the extrapolation proper.  Returns keep (the model is a repeat: nothing moves).
__device__ bool extrap_weights(const double* a, double* ext, int maxNp, int coldUnlessSmooth) {
    const int count = (int)ext[EXT_COUNT];
    const double d1d1 = a[1], d0d0 = a[EXT_NP], mkmk = a[EXT_NS - 1];
    const bool keep = count >= 1 && d0d0 <= 1e-28 * mkmk;
    double wts[EXT_NP];                                       // weights of x_k, x_{k-1}, ...
    for (int i = 0; i < EXT_NP; ++i) wts[i] = i == 0 ? 1.0 : 0.0;
    if (coldUnlessSmooth && !keep) wts[0] = 0.0;
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
        if (coldUnlessSmooth && np <= 2) wts[0] = wts[1] = 0.0;
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
    if (!keep) {
        ext[EXT_COUNT] = (double)min(count + 1, EXT_NP);
        // both histories are rings: the heads move back by one, the model ring's onto the slot k_extrap_prepare has just
        // filled with the new model, the field ring's onto its oldest entry, which receives the solution that is about to
        // be replaced by the new guess (k_extrap)
        ext[EXT_HEAD] = (double)(((int)ext[EXT_HEAD] + EXT_NP - 2) % (EXT_NP - 1));
        ext[EXT_MHEAD] = (double)(((int)ext[EXT_MHEAD] + EXT_NP) % (EXT_NP + 1));
    }
    return keep;
}

// One launch (round 1: three in a row in front of every k_extrap -- 45 us of serial side-stream time that the forward
// residual waited for): per-block partial sums over the model history -- a ring hist[slot][nAC] of NP+1 slots, the model
// j evaluations back in slot (head + j) mod (NP+1) -- of the steps d0 = m_new - m_k, d_j = m_{k-j+1} - m_{k-j}:
//     a[j] = <d_j,d1> (j < NP), a[NP] = <d0,d0>, a[NP+j-1] = <d_j,d_j> (j = 2..NP-1), a[2NP-1] = <m_k,m_k>;
// every block also stores its part of the new model in the one slot nobody reads, the slot in front of the head; the
// block that takes the last ticket adds the partial sums up, derives the weights and -- unless the model is a repeat --
// moves the head onto that slot.
__global__ __launch_bounds__(256) void k_extrap_prepare(const double* __restrict__ mNew, double* hist, int nAC, double* ext,
                                                         int maxNp, int coldUnlessSmooth, long long* ticks) {
    __shared__ double sh[EXT_NS][4];
    __shared__ int lastFlag;
    if (!coldUnlessSmooth) tick_begin(ticks, TK_EXTW);
    const int hm = (int)ext[EXT_MHEAD];
    double* fresh = hist + (long)((hm + EXT_NP) % (EXT_NP + 1)) * nAC;
    double a[EXT_NS];
#pragma unroll
    for (int q = 0; q < EXT_NS; ++q) a[q] = 0.0;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < nAC; i += EXT_NBLK * 256) {
        double m[EXT_NP], d[EXT_NP];
#pragma unroll
        for (int j = 0; j < EXT_NP; ++j) m[j] = hist[(long)((hm + j) % (EXT_NP + 1)) * nAC + i];
        const double mn = mNew[i];
        fresh[i] = mn;
        d[0] = mn - m[0];
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
    double* part = ext + EXT_PART;
    if (threadIdx.x < EXT_NS) part[blockIdx.x * EXT_NS + threadIdx.x] = sh[threadIdx.x][0] + sh[threadIdx.x][1] + sh[threadIdx.x][2] + sh[threadIdx.x][3];
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned* ticket = reinterpret_cast<unsigned*>(ext + EXT_TICKET);
        const unsigned t = atomicAdd(ticket, 1u);
        lastFlag = t == EXT_NBLK - 1;
        if (lastFlag) *ticket = 0;
    }
    __syncthreads();
    if (!lastFlag || threadIdx.x >= 64) return;
    __threadfence();
    const volatile double* vp = part + (threadIdx.x < EXT_NBLK ? threadIdx.x : 0) * EXT_NS;
    double v[EXT_NS];
#pragma unroll
    for (int q = 0; q < EXT_NS; ++q) v[q] = vp[q];                    // (all twelve requested before the first is used)
#pragma unroll
    for (int q = 0; q < EXT_NS; ++q) a[q] = wave_sum(threadIdx.x < EXT_NBLK ? v[q] : 0.0);
    if (threadIdx.x == 0) extrap_weights(a, ext, maxNp, coldUnlessSmooth);
    if (!coldUnlessSmooth) tick_end(ticks, TK_EXTW);
}

// (Round 3 also tried the whole job in ONE workgroup, as an extra block of the launch that forms sigma -- both need nothing but
// the new model: 26-34 us for 5 000 parameters even with every load of a pass requested up front, against 15 us for this
// kernel on the side stream; the launch in front of the boundary-value kernel then ends 30 us later.  Removed.)

// x <- sum_j w_j x_{k-j} on interior nodes (runs beside k_bc_forward, which writes X's boundary nodes).  The previous
// solutions live in a RING of EXT_NP-1 slots xp[slot][S*vstride]: with the head h already moved by k_extrap_weights, the
// solution j evaluations back (j = 1..NP-1, before this call) is slot (h + j) mod (NP-1), and the current solution goes
// to slot h -- the oldest one, read (if its weight is non-zero) before it is overwritten by the same thread.  Only the
// fields with a non-zero weight are read: 2 on a path that is not smooth, none for a cold adjoint start (round 1
// shifted the whole history through memory: 12 vector passes of 11.5 MB per call instead of 2-8).
__global__ __launch_bounds__(VBLOCK) void k_extrap(Solver k, cplx* x, cplx* xp, const double* __restrict__ ext) {
    if (x == k.xTickF) tick_begin(k.ticks, TK_EXT);
    if (ext[EXT_KEEP] != 0.0) return;
    double w[EXT_NP];
#pragma unroll
    for (int j = 0; j < EXT_NP; ++j) w[j] = ext[j];
    const int h = (int)ext[EXT_HEAD];
    const long so = (long)blockIdx.y * k.vstride, hs = (long)k.S * k.vstride;
    const long e0 = (long)blockIdx.x * k.chunk, e1 = min(e0 + k.chunk, k.vstride);
    for (long e = e0 + threadIdx.x; e < e1; e += VBLOCK) {
        const int iz = (int)(e / k.NYP), iy = (int)(e - (long)iz * k.NYP);
        if (iz < 1 || iz > k.nz - 1 || iy < 1 || iy > k.ny - 1) continue;
        const cplx q0 = x[so + e];
        cplx acc = w[0] * q0;
#pragma unroll
        for (int j = 1; j < EXT_NP; ++j)
            if (w[j] != 0.0) acc += w[j] * xp[(long)((h + j) % (EXT_NP - 1)) * hs + so + e];
        xp[(long)h * hs + so + e] = q0;
        x[so + e] = acc;
    }
    if (x == k.xTickF) tick_end(k.ticks, TK_EXT);
}

// true residual norm check: partB = |b - A x|^2 with b passed separately (verify option, and the production guard on the
// error-estimate stopping rule).  b == nullptr: the forward problem -- the right-hand side is what the Dirichlet values in x's
// boundary nodes contribute, b_i = - sum over the boundary neighbours of i of c_ib x_b (mt2DTE.jl:38-44), formed here.
__global__ __launch_bounds__(VBLOCK) void k_trueres(Solver k, const cplx* b, const cplx* x, double* partRes, double* partBn) {
    const int s = blockIdx.y;
    __shared__ double sh[32];
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
            cplx bnd = cplx{0, 0};
            if (iy + 1 <= k.ny - 1) acc += k.cY[mo + e] * p[e + 1]; else bnd -= k.cY[mo + e] * p[e + 1];
            if (iy - 1 >= 1) acc += k.cY[mo + e - 1] * p[e - 1]; else bnd -= k.cY[mo + e - 1] * p[e - 1];
            if (iz + 1 <= k.nz - 1) acc += k.cZ[mo + e] * p[e + k.NYP]; else bnd -= k.cZ[mo + e] * p[e + k.NYP];
            if (iz - 1 >= 1) acc += k.cZ[mo + e - k.NYP] * p[e - k.NYP]; else bnd -= k.cZ[mo + e - k.NYP] * p[e - k.NYP];
            const cplx bv = b ? b[so + e] : bnd;
            rr += cabs2(bv - acc);
            bb += cabs2(bv);
        }
    }
    block_sum2(rr, bb, sh);
    if (threadIdx.x == 0) { partRes[(long)s * MAXNB + blockIdx.x] = rr; partBn[(long)s * MAXNB + blockIdx.x] = bb; }
}
