// kernels_fdm.h -- part of libhmcmt_hip.so; included by hmcmt_hip.hip INSIDE its anonymous namespace (one translation unit).
// The fast-diagonalisation stage: fp64 MFMA transforms and tridiagonal solves (fdm_precision = 1, the restart path),
// the mixed-precision stage (split-bf16 MFMA transforms, complex64 tridiagonal sweeps), the fused forward kernel
// k_fdm_fwd and the fused back transform + post-smoother k_back_post.
#pragma once

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

// float copies of the stencil coefficients, packed per node (Solver::cf32); grid (NB, 2 modes)
__global__ __launch_bounds__(VBLOCK) void k_coef32(Solver k, float4* __restrict__ out) {
    const int mode = blockIdx.y;
    const long mo = (long)mode * k.vstride;
    const long e0 = (long)blockIdx.x * k.chunk, e1 = min(e0 + k.chunk, k.vstride);
    for (long e = e0 + threadIdx.x; e < e1; e += VBLOCK) {
        const int iz = (int)(e / k.NYP), iy = (int)(e - (long)iz * k.NYP);
        float4 a = float4{0.f, 0.f, 0.f, 0.f}, b = a;
        if (iz >= 1 && iz <= k.nz - 1 && iy >= 1 && iy <= k.ny - 1) {
            a = float4{(float)k.dK[mo + e], (float)k.dM[mo + e], (float)k.cY[mo + e], (float)k.cY[mo + e - 1]};
            b = float4{(float)k.cZ[mo + e], (float)k.cZ[mo + e - k.NYP], 0.f, 0.f};
        }
        out[2 * (mo + e)] = a; out[2 * (mo + e) + 1] = b;
    }
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
        k.dinv32[so + e] = float2{(float)d.re, (float)d.im};
    }
}

// k_coef + k_dinv + k_coef32 in one launch (three in a row on the side stream the forward residual waits for): thread
// (node, system) forms the model-dependent half of its mode's stencil at the node from the four cells around it -- TM: the
// couplings and the stiffness diagonal from 1/sigma, TE: the mass from sigma; the other half is constant and read -- and from
// it the system's Jacobi diagonal; the threads of the first system of each mode also store the coefficient arrays and
// their packed float copy (Solver::cf32).  grid (NB, S)
// sysStride = nFreq with a grid of (NB, 2): the two writers only -- an evaluation whose solves run in the persistent kernel, which forms
// the Jacobi diagonal from the coefficients and never reads Solver::dinv (the other systems' diagonals are 64 x 24 B per node of
// stores at the stress size, 110 us in front of the forward solve); k_dinv makes up for them if a solve leaves that kernel after all.
__global__ __launch_bounds__(VBLOCK) void k_coef_all(View v, Solver k, double wJ, float4* __restrict__ cf, int sysStride) {
    const int s = blockIdx.y * sysStride, mode = s >= k.nFreq;
    tick_begin(v.ticks, TK_COEF);
    const bool writer = s == mode * k.nFreq;
    const long mo = (long)mode * k.vstride, so = (long)s * k.vstride;
    const double w = k.omega[s];
    const long per = (k.vstride + gridDim.x - 1) / gridDim.x;          // (gridDim.x = NB: Solver::chunk; the writers-only launch spreads the nodes wider)
    const long e0 = (long)blockIdx.x * per, e1 = min(e0 + per, k.vstride);
    for (long e = e0 + threadIdx.x; e < e1; e += VBLOCK) {
        const int iz = (int)(e / k.NYP), iy = (int)(e - (long)iz * k.NYP);
        const bool rowI = iz >= 1 && iz <= k.nz - 1, colI = iy >= 1 && iy <= k.ny - 1, interior = rowI && colI;
        double cy = 0.0, cz = 0.0, cym = 0.0, czm = 0.0, dk = 0.0, dm = 0.0;
        if (mode == 1) {
            if (rowI && iy <= k.ny - 1) cy = coupY(v, 1, iy, iz);
            if (colI && iz <= k.nz - 1) cz = coupZ(v, 1, iy, iz);
            if (interior) {
                cym = coupY(v, 1, iy - 1, iz); czm = coupZ(v, 1, iy, iz - 1);
                dk = -(cy + cym + cz + czm);
                dm = v.dM[mo + e];
            }
        } else if (interior) {
            dk = v.dK[e];
            const double ya = v.yLen[iy - 1], yb = v.yLen[iy], za = v.zLen[iz - 1], zb = v.zLen[iz];
            dm = 0.25 * (ya * za * cells(v, 0, iy - 1, iz - 1) + yb * za * cells(v, 0, iy, iz - 1) +
                         ya * zb * cells(v, 0, iy - 1, iz) + yb * zb * cells(v, 0, iy, iz));
            if (writer) { cy = v.cY[e]; cym = v.cY[e - 1]; cz = v.cZ[e]; czm = v.cZ[e - k.NYP]; }
        }
        cplx d = cplx{0, 0};
        if (interior) d = wJ / cplx{dk, w * dm};
        k.dinv[so + e] = d;
        k.dinv32[so + e] = float2{(float)d.re, (float)d.im};
        if (writer) {
            if (iy <= k.ny) {
                if (mode == 1) { v.cY[mo + e] = cy; v.cZ[mo + e] = cz; v.dK[mo + e] = dk; }
                else v.dM[e] = dm;
            }
            float4 a = float4{0.f, 0.f, 0.f, 0.f}, b = a;
            if (interior) { a = float4{(float)dk, (float)dm, (float)cy, (float)cym}; b = float4{(float)cz, (float)czm, 0.f, 0.f}; }
            cf[2 * (mo + e)] = a; cf[2 * (mo + e) + 1] = b;
        }
    }
    tick_end(v.ticks, TK_COEF);
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

// ---- EXPERIMENT (HMCMT_SWEEPS=2, fp64 classic path only): two damped Jacobi sweeps on each side of the FDM stage
// mode 0: out = dinv .* (r + a)            (second pre-sweep from the first one's residual a = r - A dinv r)
// mode 1: out = r - A a                    (residual of the pre-smoothed iterate)
// mode 2: out = a + dinv .* (r - A a)      (one post-sweep)
// mode 3: out += a
__global__ __launch_bounds__(VBLOCK) void k_sweep_exp(Solver k, const cplx* ain, cplx* outv, int mode) {
    const int s = blockIdx.y;
    if (!k.active[s]) return;
    const int md = s >= k.nFreq;
    const long mo = (long)md * k.vstride, so = (long)s * k.vstride;
    const double w = k.omega[s];
    const cplx *r = k.r + so, *di = k.dinv + so, *a = ain + so;
    cplx* o = outv + so;
    const long e0 = (long)blockIdx.x * k.chunk, e1 = min(e0 + k.chunk, k.vstride);
    for (long e = e0 + threadIdx.x; e < e1; e += VBLOCK) {
        const int iz = (int)(e / k.NYP), iy = (int)(e - (long)iz * k.NYP);
        cplx out = cplx{0, 0};
        if (iz >= 1 && iz <= k.nz - 1 && iy >= 1 && iy <= k.ny - 1) {
            if (mode == 0) out = di[e] * (r[e] + a[e]);
            else if (mode == 1) out = r[e] - stencil_at(k, a, mo, e, w);
            else if (mode == 2) out = a[e] + di[e] * (r[e] - stencil_at(k, a, mo, e, w));
            else out = o[e] + a[e];
        }
        o[e] = out;
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
// z32out != nullptr (the fused COCG loop on meshes too wide for k_back_post): the result is stored as complex64 there
// instead of k.t, and the partial sums are those of the stored (rounded) values.
__global__ __launch_bounds__(VBLOCK) void k_post(Solver k, double* partZZ, float2* z32out) {
    const int s = blockIdx.y;
    if (!k.active[s]) return;
    __shared__ double sh[32];
    __shared__ double sh2[32];
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
            if (z32out) out = cplx{(double)(float)out.re, (double)(float)out.im};
            ar += rv.re * out.re - rv.im * out.im;
            ai += rv.re * out.im + rv.im * out.re;
            zz += cabs2(out);
        }
        if (z32out) z32out[so + e] = float2{(float)out.re, (float)out.im};
        else t[e] = out;
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
// two floats -> their bf16 roundings (to nearest even) packed in one word, and the bf16 roundings of what the first rounding left:
// v_cvt_pk_bf16_f32 (gfx950) -- 5 instructions for the hi/lo split of a pair against ~20 with the integer rounding of bf16_rn
// (same results on finite values)
typedef __bf16 bf2v __attribute__((ext_vector_type(2)));
typedef float f2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned bf16_pk(float a, float b) { return __builtin_bit_cast(unsigned, __builtin_convertvector(f2v{a, b}, bf2v)); }
__device__ __forceinline__ void bf16_split_pk(float a, float b, unsigned& hi, unsigned& lo) {
    hi = bf16_pk(a, b);
    lo = bf16_pk(a - __uint_as_float(hi << 16), b - __uint_as_float(hi & 0xffff0000u));
}

// A: complex64 rows.  Every value is split in registers into hi = bf16(x), lo = bf16(x - hi) and the
// product is accumulated as Ah*Bh + Al*Bh [+ Ah*Bl with HMCMT_VLO] (fp32 accumulators): the input to ~16 mantissa
// bits from the bf16 pipe.  (Plain bf16 operands stalled one low-frequency TE system in 32.)
// B: Bhi/Blo fragment arrays.  OUT: 0 = complex64, 1 = fp64 complex, 2 = fp64 complex + dinv*r (fused
// first half of the post-smoother).
// The eigenvector matrix V is used in PLAIN bf16 (HMCMT_VLO = 0, the default): the term Ah*Bl and the lo fragments of V
// are dropped.  Rounding V is not rounding the data: M = V_hi T^-1 V_hi' is still a fixed, linear, complex-symmetric
// operator -- exactly what COCG needs -- and V_hi = V (I + F) with |F| ~ 2^-9 moves the preconditioned spectrum by
// a per cent while the layered FDM background itself is off by the lateral contrast.  Measured: the same iteration
// counts on every trajectory of the bench (18/25 near the true model, 39/46 from the rough state), a third of the
// MFMAs and half of the V stream gone.  The INPUT rows keep their hi/lo split: rounding them is a nonlinearity of
// 2^-9 per application, which is what stalled a system when both operands were plain bf16.
#ifndef HMCMT_VLO
#define HMCMT_VLO 0
#endif
constexpr bool V_LO = HMCMT_VLO != 0;
constexpr int LP_NRG = HMCMT_LP_NRG;   // row groups (of 8 complex rows) a workgroup transforms per pass
constexpr int LP_KC = HMCMT_LP_KC;     // k-groups requested together

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
                bh[q][t] = Bhi[bi];
                if (V_LO) bl[q][t] = Blo[bi];
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
                        if (V_LO) acc[rg][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, blf, acc[rg][t], 0, 0, 0);
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
                                                 float2* __restrict__ Y, long long* stamps = nullptr, const float2* __restrict__ pX = nullptr) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    tick_begin(k.ticks, TK_FWD);
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
    double* xsh = reinterpret_cast<double*>(sj + (SW + 2 * FW_TB * SW + 3 * (long)nreg * RL * SW + 2 * FW_TB * SW));   // [8], behind the slabs
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
        for (int i = threadIdx.x; i < KG * NTW * (V_LO ? 2 : 1) * 64; i += blockDim.x) {
            const int l = i & 63, hl = V_LO ? (i >> 6) & 1 : 0, tk = V_LO ? i >> 7 : i >> 6, t = tk % NTW, kg = tk / NTW;
            const long bi = ((long)kg * NT + min(t0 + t, NT - 1)) * 64 + l;
            vst[((kg * NTW + t) * 2 + hl) * 64 + l] = hl ? Blo[bi] : Bhi[bi];
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
                        bh[q][t] = vst[((kg * NTW + t) * 2 + 0) * 64 + lane];
                        if (V_LO) bl[q][t] = vst[((kg * NTW + t) * 2 + 1) * 64 + lane];
                    } else {
                        const long bi = ((long)kg * NT + min(t0 + t, NT - 1)) * 64 + lane;
                        bh[q][t] = Bhi[bi];
                        if (V_LO) bl[q][t] = Blo[bi];
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
                            if (V_LO) acc[rg][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, blf, acc[rg][t], 0, 0, 0);
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
    else if (pX && wave > 0) {
        // x += alpha p and |x|^2 of this slab's share of the system's interior rows, by the waves that have nothing to do
        // while wave 0 sweeps (round 3).  The update kernel's load phase runs at the fabric's bandwidth: the
        // 40 B per unknown of the x update (x in and out, p in) were 37 % of what it moves there -- here they travel while
        // this kernel's memory pipes idle, behind the serial sweeps.  alpha is the update kernel's (Solver::alphaBeta),
        // p the direction k_spmv_fused stored for this iteration; p vanishes on boundary and pad nodes, so x keeps its
        // Dirichlet values; the norm runs over rows 1 .. nz-1, all columns, as k_update_fused's did.  The per-wave sums go
        // to 64 bytes BEHIND the slabs (fdm_fwd_lds): a static __shared__ array would move the slabs by 64 bytes, and that
        // alone doubles the sweeps' time (LDS bank pattern: 6 250 -> 12 150 ticks).
        const cplx al = k.alphaBeta[s];
        const int total = (k.nz - 1) * NYP, per = (total + nslab - 1) / nslab;
        const int lo = slab * per, hi = min(lo + per, total);
        const int tid = threadIdx.x - 64, nt = blockDim.x - 64;
        cplx* xs = k.x + so + NYP;
        const float2* ps = pX + so + NYP;
        constexpr int XB = 5;          // (nine -- one batch for a slab's 3 150 nodes over 384 threads -- measures the same)
        double xx = 0;
        for (int e0 = lo + tid; e0 < hi; e0 += XB * nt) {
            cplx xv[XB];
            float2 pv[XB];
#pragma unroll
            for (int u = 0; u < XB; ++u) { const unsigned e = (unsigned)min(e0 + u * nt, hi - 1); xv[u] = xs[e]; pv[u] = ps[e]; }
#pragma unroll
            for (int u = 0; u < XB; ++u) {
                const int e = e0 + u * nt;
                if (e < hi) {
                    const cplx xn = xv[u] + al * cplx{(double)pv[u].x, (double)pv[u].y};
                    xs[e] = xn;
                    xx += cabs2(xn);
                }
            }
        }
        xx = wave_sum(xx);
        if (lane == 0) xsh[wave] = xx;
    }
    FW_STAMP(3)
    __syncthreads();
    if (pX && threadIdx.x == 0) {
        double t = 0;
        for (int wv = 1; wv < nwave; ++wv) t += xsh[wv];
        k.partB[(long)s * MAXNB + slab] = t;
    }
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
    tick_end(k.ticks, TK_FWD);
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

// SW = 2 (two sweeps per side): the first term added to the FDM correction is the pre-smoothed iterate z2 that
// k_update_fused<2> stored (instead of dinv .* r), and the result -- the iterate after the FIRST post-sweep -- goes to
// z4_32 together with the second part of the rho identity (Solver::partR) and |z4|^2; the second post-sweep runs inside
// k_spmv_fused<2> (or, for the test hook that needs z in memory, as k_post2).
template <int FMT, int SW = 1>
__global__ __launch_bounds__(512) void k_back_post(Solver k, const float2* __restrict__ Y, const u4v* __restrict__ Bhi,
                                                   const u4v* __restrict__ Blo, double* partZZ, int NW, long long* stamps) {
    extern __shared__ __attribute__((aligned(16))) char smem_[];
    tick_begin(k.ticks, TK_BACK);
    const int s = blockIdx.y, bx = blockIdx.x, nwg = gridDim.x;
    const int act = k.active[s];
    // (round 3) tested HERE: rounds 1-2 tested it after the staging and V-fragment loads had been issued -- one memory round
    // trip less for an active workgroup, but on a real chain 40 % of a launch's workgroups belong to converged systems and
    // each of them pulled its 26 KB tile of y and a whole V through the memory system before returning
    if (!act) return;
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
    float2* t = (SW == 2 ? k.z4_32 : k.z32) + so;   // the preconditioned residual leaves as complex64 (the FDM stage is fp32-class anyway)
    const float2 *z2i = k.zs32 + so, *t2i = k.t2_32 + so;
    double p2r = 0, p2i = 0;           // SW = 2: sum over the own rows of t .* (V y), the second part of the rho identity (Solver::partR)
    const int nown = (iz1 - iz0 + 1) * NYP;
    // Every phase below is a short dependent chain (global load -> LDS -> barrier -> MFMA -> LDS -> barrier -> stencil),
    // so loads are issued as early as their addresses are known and unconditionally (clamped indices): inside
    // `if (row < NZP)` / `if (interior)` the compiler keeps each load next to its use and the phase costs one
    // memory round trip per element instead of one per batch (s_memtime stamps: epilogue 3.5 -> us, stencil 4.7 -> us).
    constexpr int SU = 4;               // stencil elements per thread and batch
    struct Sten { float dk, dm, cy0, cy1, cz0, cz1; cplx rv, dv; int e, iy; };   // (coefficients: the packed floats of Solver::cf32 since round 3 --
    // two 16-byte loads instead of six 8-byte ones, 32 B per node instead of 48; the post-sweep's output is complex64, and the
    // pre-sweeps of the two-sweep form have read these floats all along: now both sides of the FDM stage smooth with the same operator)
    // (uniform base + 32-bit lane offset: one address register per element instead of two per load)
    const float4* cfm = k.cf32 + 2 * mo;
    const float rNYP = 1.0f / (float)NYP;
    auto ld_st = [&](int i, Sten& q) {
        const int ic = min(i, nown - 1), lr = (int)(((float)ic + 0.5f) * rNYP);      // ic / NYP (exact: ic < 4096)
        q.iy = ic - lr * NYP;
        q.e = (iz0 + lr) * NYP + q.iy;
        const unsigned e = (unsigned)q.e;
        { const float4 a = cfm[2u * e], b = cfm[2u * e + 1u]; q.dk = a.x; q.dm = a.y; q.cy0 = a.z; q.cy1 = a.w; q.cz0 = b.x; q.cz1 = b.y; }
        q.rv = r[e];
        { const float2 d2 = (k.dinv32 + so)[e]; q.dv = cplx{(double)d2.x, (double)d2.y}; }      // (complex64 for one sweep too since round 3)
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
        float2 zq[2][2][2], tq[2][2][2];
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
                        if (SW == 2) { zq[rg][t][h2] = (z2i + ub)[lo]; tq[rg][t][h2] = (t2i + ub)[lo]; }
                        else { const float2 d2 = (k.dinv32 + so + ub)[lo]; dv[rg][t][h2] = cplx{(double)d2.x, (double)d2.y}; rv[rg][t][h2] = (r + ub)[lo]; }
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
                    if (V_LO) { bl[q][0] = *reinterpret_cast<const u4v*>(ql); bl[q][1] = *reinterpret_cast<const u4v*>(ql + d1); }
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
                        if (V_LO) acc[rg][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[rg], blf, acc[rg][t], 0, 0, 0);
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
                        cplx val = cplx{(double)acc[rg][t][2 * h2], (double)acc[rg][t][2 * h2 + 1]};
                        if (SW == 2) {
                            if (lr >= 1 && rbase + lr <= iz1 && col >= 1 && col <= k.ny - 1) {         // own rows, interior columns
                                const double tr = tq[rg][t][h2].x, ti = tq[rg][t][h2].y;
                                p2r += tr * val.re - ti * val.im; p2i += tr * val.im + ti * val.re;
                            }
                            val += cplx{(double)zq[rg][t][h2].x, (double)zq[rg][t][h2].y};
                        }
                        else val += dv[rg][t][h2] * rv[rg][t][h2];
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
            const double dk = (double)q.dk, dm = w * (double)q.dm;
            cplx acc = cplx{dk * c.re - dm * c.im, dk * c.im + dm * c.re};
            acc += (double)q.cy0 * zt[l + 1];
            acc += (double)q.cy1 * zt[l - 1];
            acc += (double)q.cz0 * zt[l + NYP];
            acc += (double)q.cz1 * zt[l - NYP];
            cplx out = c + (SW == 2 ? (double)k.w2 : 1.0) * (q.dv * (q.rv - acc));
            if (q.iy < 1 || q.iy > k.ny - 1) out = cplx{0, 0};
            const float2 of = float2{(float)out.re, (float)out.im};
            out = cplx{(double)of.x, (double)of.y};            // the sums are those of the value that is stored
            ar += q.rv.re * out.re - q.rv.im * out.im;
            ai += q.rv.re * out.im + q.rv.im * out.re;
            zz += cabs2(out);
            t[q.e] = of;
        }
    };
#pragma unroll
    for (int u = 0; u < SU; ++u) stencil(threadIdx.x + u * bd, st[u]);
#pragma unroll
    for (int u = 0; u < SU; ++u) stencil(threadIdx.x + (SU + u) * bd, st2[u]);
    // the two boundary rows of t stay zero (the stencil kernels read them as halo rows)
    if (bx == 0) for (int i = threadIdx.x; i < NYP; i += bd) t[i] = float2{0.f, 0.f};
    if (iz1 == k.nz - 1) for (int i = threadIdx.x; i < NYP; i += bd) t[(long)k.nz * NYP + i] = float2{0.f, 0.f};
    BP_STAMP(5)
    if (SW == 2) { ar = p2r; ai = p2i; }       // (rho comes from the identity; zz stays |z4|^2, the error estimate of the two-sweep path)
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
    tick_end(k.ticks, TK_BACK);
}

// Back half of the two-sweep smoother on meshes too wide for k_back_post (separate back transform k_transform_lp<1>):
//   z3 = F t + z2 (F t = k.z from the transform, z2 from k_update_fused<2>),  z4 = z3 + dinv .* (r - A z3)  -> z4_32,
// with the partial sums of t .* (F t) (second part of the rho identity, Solver::partR) and |z4|^2.  z3 is formed on the
// fly at the five stencil points.
__global__ __launch_bounds__(VBLOCK) void k_post_w2(Solver k, double* partZZ) {
    const int s = blockIdx.y;
    if (!k.active[s]) return;
    __shared__ double sh[32];
    __shared__ double sh2[32];
    const int mode = s >= k.nFreq;
    const long mo = (long)mode * k.vstride, so = (long)s * k.vstride;
    const double w = k.omega[s];
    const cplx *r = k.r + so, *ft = k.z + so;
    const float2 *z2 = k.zs32 + so, *t2 = k.t2_32 + so, *di = k.dinv32 + so;
    float2* z4 = k.z4_32 + so;
    const long e0 = (long)blockIdx.x * k.chunk, e1 = min(e0 + k.chunk, k.vstride);
    double ar = 0, ai = 0, zz = 0, dummy = 0;
    auto z3 = [&](long e) { const float2 a = z2[e]; const cplx f = ft[e]; return cplx{f.re + (double)a.x, f.im + (double)a.y}; };
    for (long e = e0 + threadIdx.x; e < e1; e += VBLOCK) {
        const int iz = (int)(e / k.NYP), iy = (int)(e - (long)iz * k.NYP);
        float2 of = float2{0.f, 0.f};
        if (iz >= 1 && iz <= k.nz - 1 && iy >= 1 && iy <= k.ny - 1) {
            const cplx c = z3(e), f = ft[e];
            const double dk = k.dK[mo + e], dm = w * k.dM[mo + e];
            cplx acc = cplx{dk * c.re - dm * c.im, dk * c.im + dm * c.re};
            acc += k.cY[mo + e] * z3(e + 1);
            acc += k.cY[mo + e - 1] * z3(e - 1);
            acc += k.cZ[mo + e] * z3(e + k.NYP);
            acc += k.cZ[mo + e - k.NYP] * z3(e - k.NYP);
            const float2 d = di[e], tv = t2[e];
            const cplx out = c + (double)k.w2 * (cplx{(double)d.x, (double)d.y} * (r[e] - acc));
            of = float2{(float)out.re, (float)out.im};
            ar += (double)tv.x * f.re - (double)tv.y * f.im;
            ai += (double)tv.x * f.im + (double)tv.y * f.re;
            zz += (double)of.x * of.x + (double)of.y * of.y;
        }
        z4[e] = of;
    }
    block_sum2(ar, ai, sh);
    block_sum2(zz, dummy, sh2);
    if (threadIdx.x == 0) {
        k.partA[(long)s * MAXNB + blockIdx.x] = cplx{ar, ai};
        partZZ[(long)s * MAXNB + blockIdx.x] = zz;
    }
}

// ----------------------------------------------------------------------------------------------
// Second post-sweep of the two-sweep smoother (k.sweeps == 2):  z = z4 + dinv .* (r - A z4)  on tiles of RT rows
// with one halo row of z4 staged in LDS, the result stored as complex64 where k_spmv_fused reads it, and the
// partial sums r'z, |z|^2 of the stored values (those k_back_post<., 1> produces on the one-sweep path).
// ----------------------------------------------------------------------------------------------
__global__ __launch_bounds__(VBLOCK) void k_post2(Solver k, double* partZZ) {
    const int s = blockIdx.y;
    if (!k.active[s]) return;
    extern __shared__ __attribute__((aligned(16))) char smem_p2[];
    c32* zs = reinterpret_cast<c32*>(smem_p2);            // [(RT+2)][NYP]
    __shared__ double sh[32];
    __shared__ double sh2[32];
    const int NYP = k.NYP, iz0 = 1 + blockIdx.x * k.RT, iz1 = min(iz0 + k.RT - 1, k.nz - 1);
    const int nrows = iz1 - iz0 + 3;
    const int mode = s >= k.nFreq;
    const long mo = (long)mode * k.vstride, so = (long)s * k.vstride;
    const float wf = (float)k.omega[s], rNYP = 1.0f / (float)NYP;
    const float2* z4 = k.z4_32 + so;
    const cplx *r = k.r + so, *di = k.dinv + so;
    float2* zo = k.z32 + so;
    for (int i = threadIdx.x; i < nrows * NYP; i += VBLOCK) {
        const float2 v = z4[(long)(iz0 - 1) * NYP + i];      // (rows 0 and nz of z4 are zero: k_back_post keeps them so)
        zs[i] = c32{v.x, v.y};
    }
    __syncthreads();
    double ar = 0, ai = 0, zz = 0, dummy = 0;
    const int nown = (iz1 - iz0 + 1) * NYP;
    for (int i = threadIdx.x; i < nown; i += VBLOCK) {
        const int lr = (int)(((float)i + 0.5f) * rNYP), iy = i - lr * NYP;      // i / NYP (exact: i < 2^20)
        const long e = (long)(iz0 + lr) * NYP + iy;
        const int l = i + NYP;
        float2 of = float2{0.f, 0.f};
        if (iy >= 1 && iy <= k.ny - 1) {
            const float4 ca = k.cf32[2 * (mo + e)], cb = k.cf32[2 * (mo + e) + 1];
            const c32 cc = zs[l];
            const float dm = wf * ca.y;
            const c32 af = c32{ca.x * cc.re - dm * cc.im + ca.z * zs[l + 1].re + ca.w * zs[l - 1].re + cb.x * zs[l + NYP].re + cb.y * zs[l - NYP].re,
                               ca.x * cc.im + dm * cc.re + ca.z * zs[l + 1].im + ca.w * zs[l - 1].im + cb.x * zs[l + NYP].im + cb.y * zs[l - NYP].im};
            const cplx c = cplx{(double)cc.re, (double)cc.im}, acc = cplx{(double)af.re, (double)af.im};
            const cplx rv = r[e];
            const cplx out = c + di[e] * (rv - acc);
            of = float2{(float)out.re, (float)out.im};
            const cplx o = cplx{(double)of.x, (double)of.y};     // the sums are those of the value that is stored
            ar += rv.re * o.re - rv.im * o.im;
            ai += rv.re * o.im + rv.im * o.re;
            zz += cabs2(o);
        }
        zo[e] = of;
    }
    // the two boundary rows of z stay zero (k_spmv_fused reads them as halo rows)
    if (blockIdx.x == 0) for (int i = threadIdx.x; i < NYP; i += VBLOCK) zo[i] = float2{0.f, 0.f};
    if (iz1 == k.nz - 1) for (int i = threadIdx.x; i < NYP; i += VBLOCK) zo[(long)k.nz * NYP + i] = float2{0.f, 0.f};
    block_sum2(ar, ai, sh);
    block_sum2(zz, dummy, sh2);
    if (threadIdx.x == 0) {
        k.partA[(long)s * MAXNB + blockIdx.x] = cplx{ar, ai};
        partZZ[(long)s * MAXNB + blockIdx.x] = zz;
    }
    // the consumers add up k.NB partial sums per system: clear the slots this launch does not use
    if (blockIdx.x == 0)
        for (int b = gridDim.x + threadIdx.x; b < k.NB; b += VBLOCK) {
            k.partA[(long)s * MAXNB + b] = cplx{0, 0};
            partZZ[(long)s * MAXNB + b] = 0.0;
        }
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
