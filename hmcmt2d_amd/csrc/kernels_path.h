// kernels_path.h -- part of libhmcmt_hip.so; included by hmcmt_hip.hip INSIDE its anonymous namespace (one translation unit).
// The kernels around the solves: assembly from sigma, 1-D boundary fields and their sensitivities, receiver
// functionals and adjoint sources, J^T v accumulation (bodies in hmcmt_items.h), and the leapfrog vector kernels.
#pragma once

// ----------------------------------------------------------------------------------------------
// item kernels
// ----------------------------------------------------------------------------------------------
#define TID1 (blockIdx.x * blockDim.x + threadIdx.x)

__global__ void k_null() {}
// spins for `ticks` of the constant-rate wall clock (hipDeviceAttributeWallClockRate): the known-duration kernel the
// event-bracket overhead is calibrated with (hmcmt_profile)
__global__ void k_spin(long long ticks) {
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(1);
}
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
struct LfView {
    int n;
    const double *mref, *invM, *wmVal;
    const long long *wmRow, *wmCol;
    double *m, *p, *g;            // model, momentum, data gradient (in) / total gradient (out)
    double *part;                 // [LFNB] partial maxima / sums
    double *scal;                 // [0] mnorm
    int* flag;                    // non-zero: non-finite value met
    long long* ticks;             // HMCMT_TICKS
};
constexpr int LFNB = 64;
// m += dm (clamped to max |dm| = 3), reflect at the ln-sigma bounds, flip momentum (HMCSampler.jl:241-247, :515-559) for
// parameter a; mx = the maximum of the partial step bounds.  Returns the new value (the old one on a non-finite step, flagged).
__device__ __forceinline__ double lf_step_one(const LfView& L, int a, double dt, double lo, double hi, double mx) {
    double dm = dt * L.invM[a] * L.p[a];
    if (mx > 3.0) dm = dm / mx * 3.0;
    double m = L.m[a] + dm, p = L.p[a];
    if (!isfinite(m)) { atomicExch(L.flag, 1); return L.m[a]; }
    for (int it = 0; it < 500 && !(m <= hi && m >= lo); ++it) {
        if (m < lo) { m = 2.0 * lo - m; p = -p; }
        if (m > hi) { m = 2.0 * hi - m; p = -p; }
    }
    L.m[a] = m; L.p[a] = p;
    return m;
}
__device__ __forceinline__ double lf_step_bound(const LfView& L) {
    double mx = 0.0;
    for (int b = 0; b < LFNB; ++b) mx = fmax(mx, L.part[b]);
    return mx;
}
// a position update that k_sigma_rows performs on its way (the device-resident leapfrog: one launch less per step)
struct LfStep { int on; LfView L; double dt, lo, hi; };
// a momentum update that k_gradfinal performs on its way (one launch less per step): gen / done make it happen once per
// parameter and evaluation however often the gradient's last kernel runs (a repeated evaluation, speculative followers)
struct LfMom { int on; LfView L; double lambda, cdt, dt; int gen; int* done; };
// k_sigma + k_rowmean: one wave per cell row computes the row's conductivities and, from the values it has just formed,
// the lateral means (same summation order as k_rowmean) -- the FDM background (k_pivot) then needs nothing but this launch
__global__ __launch_bounds__(64) void k_sigma_rows(View v, LfStep step) {
    tick_begin(v.ticks, TK_SIGMA);
    const int kz = blockIdx.x;
    const double mx = step.on ? lf_step_bound(step.L) : 0.0;
    double sa = 0.0, sl = 0.0;
    for (int ky = threadIdx.x; ky < v.ny; ky += 64) {
        const long cell = (long)kz * v.ny + ky;
        const int a = v.cell2act[cell];
        double ma = 0.0;
        if (a >= 0) ma = step.on ? lf_step_one(step.L, a, step.dt, step.lo, step.hi, mx) : v.m[a];      // (every parameter is one cell's)
        const double s = v.bg[cell] + (a >= 0 ? exp(ma) : 0.0);
        v.sigma[cell] = s;
        sa += s; sl += log(s);
    }
    sa = wave_sum(sa); sl = wave_sum(sl);
    if (threadIdx.x == 0) { v.sigMeanA[kz] = sa / v.ny; v.sigMeanG[kz] = exp(sl / v.ny); }
    tick_end(v.ticks, TK_SIGMA);
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
    tick_begin(v.ticks, TK_PIVOT);
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
    tick_end(v.ticks, TK_PIVOT);
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
// k_bc_layers + k_bc_forward in ONE launch for meshes whose columns fit the chip in one round (round 3): a workgroup takes
// CW boundary columns of one frequency, builds their per-layer terms in LDS -- first the three of the impedance
// recurrence, then (in the same space) the five of the amplitude propagation: nz*CW independent items of transcendental
// fp64 work over 256 threads -- and lanes 0..CW-1 run the serial recurrences from LDS.  The table never goes through
// global memory (k_bc_forward fetched it from the Infinity Cache, eight layers per round trip: 26 round trips in a row
// were its 45 us), and one launch is gone from the critical chain in front of the forward residual.
// LDS: slab[5][nz][CW] complex, amplitudes of the edge column(s) [nslot][nz][2].
__global__ __launch_bounds__(256) void k_bc_fused(View v, int CW, int nslot) {
    extern __shared__ __attribute__((aligned(16))) char smem_bcf[];
    tick_begin(v.ticks, TK_BC);
    const int nz = v.nz;
    const long ls = CW, qs = (long)nz * CW;
    cplx* tab = reinterpret_cast<cplx*>(smem_bcf);
    cplx* amp = tab + 5 * qs;
    __shared__ FwdTop top[2];
    __shared__ int deadAt[2];
    const int col0 = blockIdx.x * CW, f = blockIdx.y;
    const bool onE = v.sysOn[f] != 0, onH = v.sysOn[v.nFreq + f] != 0;
    if (!onE && !onH) return;
    const double omega = v.omega[f];
    cplx* XE = v.X + (long)f * v.vstride;
    cplx* XH = v.X + (long)(v.nFreq + f) * v.vstride;
    const bool has0 = col0 == 0, hasN = col0 <= v.ny && v.ny < col0 + CW;
    const int col = col0 + threadIdx.x;
    const bool serial = (int)threadIdx.x < CW && col <= v.ny;
    const bool isEdge = serial && (col == 0 || col == v.ny);
    const int slot = (col == 0 || nslot == 1) ? 0 : 1;    // (one slot: the two edge columns are in different workgroups)
    if (serial) {
        if (onE) XE[nidx(v, col, 0)] = cplx{1.0, 0.0};    // top row incl. corners
        if (onH) XH[nidx(v, col, 0)] = cplx{1.0, 0.0};
    }
    // slabs: 0 k, then 1 zp, 2 th, 3 zp*(zp*th), 4 e+  ->  1 m11, 2 m12, 3 m21, 4 m22
    for (int idx = threadIdx.x; idx < nz * CW; idx += blockDim.x) {
        const int j = idx / CW, c = idx - j * CW;
        cplx kk, ep, zp, th, zt;
        layer_up_terms(bc_column_sigma(v, j, min(col0 + c, v.ny)), omega, v.zLen[j], kk, ep, zp, th, zt);
        const long o = (long)j * ls + c;
        tab[o] = kk; tab[qs + o] = zp; tab[2 * qs + o] = th; tab[3 * qs + o] = zt; tab[4 * qs + o] = ep;
    }
    __syncthreads();
    cplx ztmp = cplx{0.0, 0.0};
    if (serial) ztmp = bc1d_up(nz, tab + qs + threadIdx.x, tab + 2 * qs + threadIdx.x, tab + 3 * qs + threadIdx.x, ls);
    __syncthreads();
    for (int idx = threadIdx.x; idx < nz * CW; idx += blockDim.x) {
        const int j = idx / CW, c = idx - j * CW;
        const bool lastLayer = j + 1 >= nz;
        const long o = (long)j * ls + c;
        const cplx kk = tab[o], kn = tab[lastLayer ? o : o + ls], ep = tab[4 * qs + o];
        cplx m11, m12, m21, m22;
        layer_down_terms(kk, kn, lastLayer, ep, m11, m12, m21, m22);
        tab[qs + o] = m11; tab[2 * qs + o] = m12; tab[3 * qs + o] = m21; tab[4 * qs + o] = m22;
    }
    __syncthreads();
    if (serial) {
        cplx* ea = amp + (long)slot * 2 * nz;
        int dAt = nz;                                     // first layer behind the overflow cut-off
        FwdTop tp;
        cplx lastE, lastH;
        const cplx* T = tab + threadIdx.x;
        bc1d_down(omega, nz, ztmp, T, T + qs, T + 2 * qs, T + 3 * qs, T + 4 * qs, ls, [&](int i, cplx eu, cplx ed, cplx, bool dead) {
            if (isEdge) { ea[2 * i] = eu; ea[2 * i + 1] = ed; if (dead && dAt > i) dAt = i; }
        }, tp, lastE, lastH);
        if (isEdge) { top[slot] = tp; deadAt[slot] = dAt; }
        else {
            if (onE) XE[nidx(v, col, nz)] = lastE;
            if (onH) XH[nidx(v, col, nz)] = lastH;
        }
    }
    if (has0 || hasN) {
        __syncthreads();
        for (int i = threadIdx.x; i < nz; i += blockDim.x) {
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                if (e == 0 ? !has0 : !hasN) continue;
                const int ecol = e == 0 ? 0 : v.ny, sl = (e == 0 || nslot == 1) ? 0 : 1;
                const cplx kj = tab[(long)(i + 1 < nz ? i + 1 : nz - 1) * ls + (ecol - col0)];      // k of the layer below (the last layer: its own)
                cplx oE, oH;
                fwd_outputs(top[sl], amp[((long)sl * nz + i) * 2], amp[((long)sl * nz + i) * 2 + 1], kj, i >= deadAt[sl], oE, oH);
                if (onE) XE[nidx(v, ecol, 1 + i)] = oE;
                if (onH) XH[nidx(v, ecol, 1 + i)] = oH;
            }
        }
    }
    tick_end(v.ticks, TK_BC);
}
// The same in one launch for meshes whose slabs do NOT fit k_bc_fused's LDS (the stress size: 207 layers x 401 columns x 32
// frequencies; until round 6 k_bc_layers wrote and k_bc_forward read a 340 MB table there, 75 + 194 us, the second one 224 waves
// walking 52 blocks of loads one after the other).  LAYER BLOCKS through a two-buffer ring in LDS: wave 0 runs the serial recurrences
// of the workgroup's CW columns (one lane each) on block b while the other waves build the per-layer terms of block b+1, one barrier
// per block.  Bottom -> top (blocks of LBu layers) the producers do the transcendental work (layer_up_terms: k, e+, zp, th,
// zp*(zp*th)) -- the three terms of the impedance recurrence go to the ring, k and e+ to a stash in global memory (32 B per layer and
// column: 85 MB at the stress size, it stays in the Infinity Cache) --; top -> bottom (blocks of LBd layers) they read k, k of the
// layer below and e+ back, BCB_D blocks ahead of their use, and form the amplitude propagation's four terms (layer_down_terms: no
// transcendentals).  Same item functions and the same order of operations per column as k_bc_fused / bc1d_up / bc1d_down (the
// projective pair is rescaled by an exact power of two every four layers: no effect on the quotient).
// LDS: max(ring[2][3][LBu][CW], ring[2][5][LBd][CW]) complex, then the edge column(s)' amplitudes, amp[nslot][nz][2].
// The host makes LBu * CW and LBd * CW <= the number of producer threads (one item per producer and block).
constexpr int BCB_D = 3;
// workgroup barrier that orders LDS only: __syncthreads() also drains the vector-memory queue (s_waitcnt vmcnt(0)), which would make
// every block of the amplitude pass wait for the stash entries it has just requested for three blocks later
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
__global__ __launch_bounds__(1024) void k_bc_blocked(View v, int CW, int nslot, int LBu, int LBd) {
    extern __shared__ __attribute__((aligned(16))) char smem_bcb[];
    tick_begin(v.ticks, TK_BC);
    const int nz = v.nz, bsu = LBu * CW, bsd = LBd * CW, nbu = (nz + LBu - 1) / LBu, nbd = (nz + LBd - 1) / LBd;
    cplx* ring = reinterpret_cast<cplx*>(smem_bcb);
    cplx* amp = ring + 2 * max(3 * bsu, 5 * bsd);
    __shared__ FwdTop top[2];
    __shared__ int deadAt[2];
    const int col0 = blockIdx.x * CW, f = blockIdx.y;
    const bool onE = v.sysOn[f] != 0, onH = v.sysOn[v.nFreq + f] != 0;
    if (!onE && !onH) return;
    const double omega = v.omega[f];
    cplx* XE = v.X + (long)f * v.vstride;
    cplx* XH = v.X + (long)(v.nFreq + f) * v.vstride;
    const long ls = v.ny + 1, qs = (long)nz * ls;
    cplx* stK = v.fwdTab + (long)f * FWD_NQ * qs;         // the stash: k, e+ of every layer and column of this frequency
    cplx* stE = stK + qs;
    const bool has0 = col0 == 0, hasN = col0 <= v.ny && v.ny < col0 + CW;
    const bool chain = threadIdx.x < 64;
    const int ptid = (int)threadIdx.x - 64;
    const int col = col0 + threadIdx.x;
    const bool serial = (int)threadIdx.x < CW && col <= v.ny;
    const bool isEdge = serial && (col == 0 || col == v.ny);
    const int slot = (col == 0 || nslot == 1) ? 0 : 1;    // (one slot: the two edge columns are in different workgroups)
    if (serial) {
        if (onE) XE[nidx(v, col, 0)] = cplx{1.0, 0.0};    // top row incl. corners
        if (onH) XH[nidx(v, col, 0)] = cplx{1.0, 0.0};
    }
    // block b of the impedance recurrence: layers nz-1 - LBu b - t (slabs 0 zp, 1 th, 2 zp*(zp*th)) into buffer b & 1; one item per
    // thread and block, its conductivity and thickness requested one block ahead (up_fetch) of the transcendental work (up_item)
    auto up_fetch = [&](int b, int item, double& sig, double& h) {
        sig = 1.0; h = 1.0;
        const int t = item / CW, c = item - t * CW, j = nz - 1 - b * LBu - t;
        if (b < nbu && item < bsu && j >= 0) { sig = bc_column_sigma(v, j, min(col0 + c, v.ny)); h = v.zLen[j]; }
    };
    auto up_item = [&](int b, int item, double sig, double h) {
        const int t = item / CW, c = item - t * CW, j = nz - 1 - b * LBu - t;
        if (item >= bsu || j < 0) return;
        cplx* B = ring + (b & 1) * 3 * bsu;
        cplx kk, ep, zp, th, zt;
        layer_up_terms(sig, omega, h, kk, ep, zp, th, zt);
        B[item] = zp; B[bsu + item] = th; B[2 * bsu + item] = zt;
        if (col0 + c <= v.ny) { stK[(long)j * ls + col0 + c] = kk; stE[(long)j * ls + col0 + c] = ep; }
    };
    // block b of the amplitude propagation: layers LBd b + t (slabs 0 k of the layer below, 1..4 m11 m12 m21 m22) into buffer b & 1;
    // producer p has item p of every block, its three stash entries are requested BCB_D blocks ahead (dn_load) of their use (dn_store)
    struct DnIn { cplx k, kn, ep; };
    static_assert(BCB_D == 3, "three register slots");
    const int dt_ = ptid >= 0 ? ptid / CW : 0, dc_ = ptid >= 0 ? ptid - dt_ * CW : 0;
    auto dn_load = [&](int b, DnIn& p) {
        const int i = b * LBd + dt_;
        if (b < nbd && ptid < bsd && i < nz) {
            const long o = (long)i * ls + min(col0 + dc_, v.ny);
            p.k = stK[o]; p.ep = stE[o];
            p.kn = stK[i + 1 < nz ? o + ls : o];          // (an index, not a select of the two values: that one became a select of ADDRESSES, the slot's in scratch memory)
        }
    };
    auto dn_store = [&](int b, const DnIn& p) {
        cplx* B = ring + (b & 1) * 5 * bsd;
        const int i = b * LBd + dt_;
        if (b < nbd && ptid < bsd && i < nz) {
            cplx m11, m12, m21, m22;
            layer_down_terms(p.k, p.kn, i + 1 >= nz, p.ep, m11, m12, m21, m22);
            B[ptid] = p.kn; B[bsd + ptid] = m11; B[2 * bsd + ptid] = m12; B[3 * bsd + ptid] = m21; B[4 * bsd + ptid] = m22;
        }
    };
    const cplx one = cplx{1.0, 0.0};
    double sgN, hzN;
    up_fetch(0, threadIdx.x, sgN, hzN);
    up_item(0, threadIdx.x, sgN, hzN);
    if (!chain) up_fetch(1, ptid, sgN, hzN);
    __syncthreads();
    // --- impedance bottom -> top (bc1d_up): Z = N/D carried projectively
    cplx zn = one, zd = one;
    for (int b = 0; b < nbu; ++b) {
        if (!chain) {
            const double s1 = sgN, h1 = hzN;
            up_fetch(b + 2, ptid, sgN, hzN);
            if (b + 1 < nbu) up_item(b + 1, ptid, s1, h1);
        } else if (serial) {
            const cplx* B = ring + (b & 1) * 3 * bsu + threadIdx.x;
            const int cnt = min(LBu, nz - b * LBu);
            if (b == 0) zn = B[0];                        // half-space below the last layer with the last layer's conductivity
            for (int h = 0; h < cnt; h += 4) {
                cplx c1[4], c2[4], c3[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int tt = h + t < cnt ? h + t : cnt - 1;
                    c1[t] = B[tt * CW]; c3[t] = B[bsu + tt * CW]; c2[t] = B[2 * bsu + tt * CW];
                }
#pragma unroll
                for (int t = 0; t < 4; ++t)
                    if (h + t < cnt) {
                        const cplx nn = c1[t] * zn + c2[t] * zd;
                        zd = c1[t] * zd + zn * c3[t];
                        zn = nn;
                    }
                const int e = -ilogb(fmax(fmax(fabs(zd.re), fabs(zd.im)), fmax(fabs(zn.re), fabs(zn.im))));
                if (e > -1000 && e < 1000) {              // (zero / inf / nan: leave alone, the division below reports it)
                    zn = cplx{ldexp(zn.re, e), ldexp(zn.im, e)};
                    zd = cplx{ldexp(zd.re, e), ldexp(zd.im, e)};
                }
            }
        }
        __syncthreads();
    }
    // (the stash entries were written by threads of this workgroup: the barrier above orders them)
    // --- amplitudes top -> bottom (bc1d_down)
    const double omu0 = omega * MU0, iomu0 = 1.0 / omu0;
    cplx eu = one, ed = one, kj = one;
    FwdTop tp;
    bool dead = false;
    int dAt = nz;                                         // first layer behind the overflow cut-off
    cplx* ea = amp + (long)slot * 2 * nz;
    // The chain wave and the producers walk the blocks in loops of their OWN (the same number of barriers in each: the hardware counts
    // arrivals, not program counters): in one loop the producers' three register slots would stay allocated through the chain's steps,
    // and at 128 registers per lane (1 024 threads) the chain spilled.
    if (chain) {
        if (serial) {
            const cplx ztmp = zn / zd;
            kj = stK[col];
            const cplx a = omu0 / (ztmp * kj);
            eu = 0.5 * (one - a); ed = 0.5 * (one + a);
            tp.if0E = crecip(eu + ed); tp.if0H = crecip(((ed - eu) * kj) * iomu0); tp.iomu0 = iomu0;
        }
        lds_barrier();
        for (int b = 0; b < nbd; ++b) {
            if (serial) {
                const cplx* B = ring + (b & 1) * 5 * bsd + threadIdx.x;
                const int cnt = min(LBd, nz - b * LBd);
                for (int h = 0; h < cnt; h += 4) {
                    cplx kn_[4], m11[4], m12[4], m21[4], m22[4];
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const int tt = h + t < cnt ? h + t : cnt - 1;
                        kn_[t] = B[tt * CW]; m11[t] = B[bsd + tt * CW]; m12[t] = B[2 * bsd + tt * CW]; m21[t] = B[3 * bsd + tt * CW]; m22[t] = B[4 * bsd + tt * CW];
                    }
#pragma unroll
                    for (int t = 0; t < 4; ++t)
                        if (h + t < cnt) {
                            const int i = b * LBd + h + t;
                            const cplx nu = m11[t] * eu + m12[t] * ed;
                            const cplx nd = m21[t] * eu + m22[t] * ed;
                            const double e2 = cabs2(nu + nd), e1 = cabs2(eu + ed);   // |.|^2: same ordering as |.|
                            dead = dead || e2 - e1 > 0.0 || isnan(e2);               // overflow cut-off: zero from here down
                            eu = nu; ed = nd; kj = kn_[t];
                            if (isEdge) { ea[2 * i] = eu; ea[2 * i + 1] = ed; if (dead && dAt > i) dAt = i; }
                        }
                }
            }
            lds_barrier();
        }
    } else {
        DnIn p0, p1, p2;                                  // (named, not an array: the slot of a block, b % 3, is a compile-time choice)
        dn_load(0, p0); dn_load(1, p1); dn_load(2, p2);
        dn_store(0, p0);
        dn_load(3, p0);
        lds_barrier();
        // while the chain walks block b: form block b+1 from its slot and request block b+4 into it
        for (int b0 = 0; b0 < nbd; b0 += BCB_D) {
            dn_store(b0 + 1, p1); dn_load(b0 + 1 + BCB_D, p1);
            lds_barrier();
            if (b0 + 1 < nbd) { dn_store(b0 + 2, p2); dn_load(b0 + 2 + BCB_D, p2); lds_barrier(); }
            if (b0 + 2 < nbd) { dn_store(b0 + 3, p0); dn_load(b0 + 3 + BCB_D, p0); lds_barrier(); }
        }
    }
    if (serial) {
        if (isEdge) { top[slot] = tp; deadAt[slot] = dAt; }
        else {
            cplx lastE, lastH;
            fwd_outputs(tp, eu, ed, kj, dead, lastE, lastH);
            if (onE) XE[nidx(v, col, nz)] = lastE;
            if (onH) XH[nidx(v, col, nz)] = lastH;
        }
    }
    if (has0 || hasN) {
        __syncthreads();
        for (int i = threadIdx.x; i < nz; i += blockDim.x) {
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                if (e == 0 ? !has0 : !hasN) continue;
                const int ecol = e == 0 ? 0 : v.ny, sl = (e == 0 || nslot == 1) ? 0 : 1;
                const cplx kjn = stK[(long)(i + 1 < nz ? i + 1 : nz - 1) * ls + ecol];      // k of the layer below (the last layer: its own)
                cplx oE, oH;
                fwd_outputs(top[sl], amp[((long)sl * nz + i) * 2], amp[((long)sl * nz + i) * 2 + 1], kjn, i >= deadAt[sl], oE, oH);
                if (onE) XE[nidx(v, ecol, 1 + i)] = oE;
                if (onH) XH[nidx(v, ecol, 1 + i)] = oH;
            }
        }
    }
    tick_end(v.ticks, TK_BC);
}
__global__ __launch_bounds__(64) void k_sens_layers(View v) {
    int j = blockIdx.x * blockDim.x + threadIdx.x, prof = blockIdx.y, s = blockIdx.z;
    if (j <= v.nz) item_sens_layers(v, s, prof, j);
}
__global__ __launch_bounds__(64) void k_sens_profile(View v) {
    int e = blockIdx.x * blockDim.x + threadIdx.x;
    tick_begin(v.ticks, TK_SENS);
    if (e < 3 * v.S) item_sens_profile(v, e / 3, e % 3);
    tick_end(v.ticks, TK_SENS);
}
// k_sens_layers + k_sens_profile + k_bcsens_pre in ONE launch (end of round 6), one workgroup per (system, profile), every table in LDS:
// as three launches the stages handed their tables on through global memory and walked them row by row -- uniform loads in front of
// every row's arithmetic, on the two CUs per XCD a solve leaves free, behind the persistent kernel's traffic in the L2's queues: 0.4 ms for
// the 192 serial profiles and 4.8 ms for the derivative columns at the stress size, all latency.  Here the layers' terms are formed by the
// whole workgroup (stage 1), lane 0 walks the profile (stage 2) and every lane its derivative column (stage 3) with nothing but LDS reads in
// the loops; what leaves the workgroup is what the gradient's tail reads (dBC, gMn) and the TM boundary values (bcsL / bcsR / bcsB).
// Same item functions in the same order: bitwise the three-kernel results.  LDS: 12 (nz+1) complex + nz doubles.
__global__ __launch_bounds__(256) void k_sens_fused(View v) {
    extern __shared__ __attribute__((aligned(16))) char smem_sf[];
    const int prof = blockIdx.x, s = blockIdx.y, nz = v.nz;
    const long n1 = nz + 1;
    if (!v.sysOn[s]) {
        if (prof == 2) for (int c = threadIdx.x; c < nz; c += blockDim.x) v.gMn[(long)s * nz + c] = cplx{0, 0};
        return;
    }
    tick_begin(v.ticks, TK_SENS);
    cplx* T = reinterpret_cast<cplx*>(smem_sf);           // [5][n1]: ka, 1/ka, exp(i ka h), its inverse, exp(-2 i ka h)
    cplx* sEu = T + 5 * n1;                               // [n1]
    cplx* sEd = sEu + n1;                                 // [n1]
    cplx* sMix = sEd + n1;                                // [4][nz]
    cplx* sDz1 = sMix + 4 * (long)nz;                     // [nz]
    double* sZ = reinterpret_cast<double*>(sDz1 + nz);    // [nz] the layers' thicknesses (a global load in a serial loop waits for the solve's traffic)
    __shared__ cplx z1sh;
    __shared__ int deadsh;
    const bool tm = s >= v.nFreq;
    const double omega = v.omega[s];
    for (int j = threadIdx.x; j <= nz; j += blockDim.x) { // stage 1 (item_sens_layers)
        const int jj = j < nz ? j : nz - 1;
        if (j < nz) sZ[j] = v.zLen[j];
        const double sig = prof == 0 ? v.sigma[(long)jj * v.ny] : (prof == 1 ? v.sigma[(long)jj * v.ny + v.ny - 1] : v.sigMeanA[jj]);
        cplx t[5];
        layer_sens(sig, omega, v.zLen[jj], t);
#pragma unroll
        for (int q = 0; q < 5; ++q) T[q * n1 + j] = t[q];
    }
    __syncthreads();
    if (threadIdx.x == 0) {                               // stage 2 (item_sens_profile)
        cplx* fout = nullptr;
        long fstride = 1;
        if (tm) {                                         // `bc` of getBCderivTM (compJacTMatVec.jl:309,315)
            if (prof == 0) fout = v.bcsL + (long)s * nz;
            else if (prof == 1) fout = v.bcsR + (long)s * nz;
            else { fout = v.bcsB + s; fstride = 0; }      // only the bottom value of the mean profile
        }
        cplx z1;
        deadsh = sens_profile(omega, nz, sZ, tm, T, T + n1, T + 2 * n1, T + 3 * n1, T + 4 * n1, sEu, sEd, sMix, sDz1, &z1, fout, fstride);
        z1sh = z1;
    }
    __syncthreads();
    const cplx z1 = z1sh;
    const int dead = deadsh;
    for (int c = threadIdx.x; c < nz; c += blockDim.x) {  // stage 3 (item_bcsens_pre)
        cplx* out = prof < 2 ? v.dBC + ((long)s * 2 + prof) * nz * nz + c : nullptr;
        const cplx g = bc1d_sens_column(omega, nz, sZ, tm, c, T, T + n1, T + 2 * n1, T + 3 * n1, sEu, sEd, sMix, sDz1, z1, dead, nullptr, 1, out, nz);
        if (prof == 2) v.gMn[(long)s * nz + c] = g;
    }
    tick_end(v.ticks, TK_SENS);
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
    __shared__ double sh[32];
    double a = 0, b = 0;
    for (int p = threadIdx.x; p < v.nData; p += blockDim.x) a += v.misfitPart[p];
    block_sum2(a, b, sh);
    if (threadIdx.x == 0) *out = a;
}
// Between the two solves: per (system, receiver) the impedance (+ its derivatives), then the residual / misfit terms of
// the data that address this receiver and their sum of conj(W'W r) -- one launch instead of three in a row on the
// critical path (a datum belongs to exactly one (system, receiver), so there is no cross-thread dependency; the
// misfit itself, a reduction over all data, is not needed by the adjoint half and is summed after the sources).
// the per-solve records of k_solve_end, written by the first kernel behind a solve instead (one launch less on the critical
// path behind each solve): iteration counts, status, error estimates -> mapped host memory
struct SolveRec { const int* iters; const int* status; const double* errEst; int* recI; double* recE; int kind, S; };
__device__ __forceinline__ void write_solve_rec(const SolveRec& r, int s) {
    r.recI[r.kind * r.S + s] = r.iters[s];
    r.recI[(2 + r.kind) * r.S + s] = r.status[s];
    r.recE[r.kind * r.S + s] = r.errEst[s];
}
// kernels queued speculatively behind a persistent solve (View::gate): true = the solve did not end clean, do nothing
__device__ __forceinline__ bool gate_closed(const View& v) { return v.gate && *(volatile const int*)v.gate != v.gateGen; }
__global__ __launch_bounds__(64) void k_rxall(View v, int wantGrad, SolveRec rec) {
    if (gate_closed(v)) return;        // (64: registers instead of 200 B of spills)
    const int e = TID1;
    tick_begin(v.ticks, TK_RXALL);
    if (rec.recI && e < rec.S) write_solve_rec(rec, e);
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
    tick_end(v.ticks, TK_RXALL);
}
__global__ void k_rxcoef(View v) { int e = TID1; if (e < v.S * v.nRx) item_rxcoef(v, e / v.nRx, e % v.nRx); }
// adjoint sources; workgroup (0,0) also adds up the misfit terms (a reduction nothing on the device waits for: no
// launch of its own on the critical path between the solves)
__global__ __launch_bounds__(128) void k_src(View v, double* misfitOut, int nsrc) {
    if (gate_closed(v)) return;
    // blocks x < nsrc: the sources; the blocks behind them: the receiver-layer Q-terms of the gradient (item_qterm
    // needs nothing from the adjoint solve: here they cost no launch in the gradient tail)
    const int s = blockIdx.y;
    tick_begin(v.ticks, TK_SRC);
    // The system's receiver table in LDS: a node (item_src) / a cell column (item_qterm) asks every receiver whether it lies in
    // the receiver's stencil -- from global memory a dependent load per receiver and thread (12 us for 40 receivers).
    // Same terms in the same order as the item functions.
    constexpr int SRC_RX = 256;
    __shared__ int sN0[SRC_RX];
    __shared__ cplx sCoef[SRC_RX];
    const bool tab = v.nRx <= SRC_RX;
    if (tab) {
        for (int r = threadIdx.x; r < v.nRx; r += blockDim.x) { sN0[r] = v.rxN0[(long)s * v.nRx + r]; sCoef[r] = v.rxCoef[(long)s * v.nRx + r]; }
        __syncthreads();
    }
    if ((int)blockIdx.x >= nsrc) {
        const int ky = (blockIdx.x - nsrc) * blockDim.x + threadIdx.x;
        if (ky < v.ny) {
            if (!tab) item_qterm(v, s, ky);
            else {
                double g = 0.0;
                if (v.sysOn[s])
                    for (int r = 0; r < v.nRx; ++r) {
                        const int o = ky - sN0[r];
                        if (o >= 0 && o < 3) g += (sCoef[r] * v.rxD[((long)s * v.nRx + r) * 11 + 8 + o]).re;
                    }
                v.qPart[(long)s * v.ny + ky] = g;
            }
        }
        return;
    }
    int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e < 2 * (v.ny + 1)) {
        if (!tab) item_src(v, s, e / (v.ny + 1), e % (v.ny + 1));
        else {
            const int row = e / (v.ny + 1), iy = e % (v.ny + 1), iz = v.zid + row;
            cplx acc = cplx{0, 0};
            for (int r = 0; r < v.nRx; ++r) {
                const int o = iy - sN0[r];
                if (o >= 0 && o < 4) acc += sCoef[r] * v.rxD[((long)s * v.nRx + r) * 11 + row * 4 + o];
            }
            const bool interior = iz >= 1 && iz <= v.nz - 1 && iy >= 1 && iy <= v.ny - 1;
            if (interior) v.R[(long)s * v.vstride + nidx(v, iy, iz)] = acc;
            else if (iy == 0) v.srcB[(long)s * 4 + row * 2 + 0] = acc;
            else if (iy == v.ny) v.srcB[(long)s * 4 + row * 2 + 1] = acc;
        }
    }
    if (blockIdx.x == 0 && blockIdx.y == 0) {
        __shared__ double sh[2];
        double a = 0;
        // (eight terms requested per pass: one load per pass and thread made this block the kernel's last, 10 round trips in a row)
        for (int p0 = threadIdx.x; p0 < v.nData; p0 += 8 * 128) {
            double t[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) t[q] = v.misfitPart[min(p0 + q * 128, v.nData - 1)];
#pragma unroll
            for (int q = 0; q < 8; ++q) if (p0 + q * 128 < v.nData) a += t[q];
        }
        a = wave_sum(a);
        if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = a;
        __syncthreads();
        if (threadIdx.x == 0) *misfitOut = sh[0] + sh[1];
    }
    tick_end(v.ticks, TK_SRC);
}
__global__ void k_wb(View v, SolveRec rec) {
    if (gate_closed(v)) return;
    int e = blockIdx.x * blockDim.x + threadIdx.x, s = blockIdx.y;
    tick_begin(v.ticks, TK_WB);
    if (rec.recI && e == 0) write_solve_rec(rec, s);
    if (e < v.nz) item_wside(v, s, e + 1);
    else if (e < v.nz + v.ny) item_colw(v, s, e - v.nz);
    tick_end(v.ticks, TK_WB);
}
// k_wb and k_gradcell in ONE launch (end of round 5): the per-cell P-terms read the two solves' fields only, the boundary weights
// feed k_bcsens_contract -- nothing of one is read by the other, and as two launches of a few microseconds each they were two
// boundaries of the serial tail behind the adjoint solve.  Blocks [0, nwb) are k_wb's (nwbx per system), the rest k_gradcell's
// (ngcx per mode and frequency group).
__global__ __launch_bounds__(128) void k_wb_gradcell(View v, SolveRec rec, int nwbx, int ngcx) {
    if (gate_closed(v)) return;
    const int nwb = nwbx * v.S;
    if ((int)blockIdx.x < nwb) {
        const int bx = blockIdx.x % nwbx, s = blockIdx.x / nwbx;
        const int e = bx * blockDim.x + threadIdx.x;
        tick_begin(v.ticks, TK_WB);
        if (rec.recI && e == 0) write_solve_rec(rec, s);
        if (e < v.nz) item_wside(v, s, e + 1);
        else if (e < v.nz + v.ny) item_colw(v, s, e - v.nz);
        tick_end(v.ticks, TK_WB);
    } else {
        const int idx = blockIdx.x - nwb;
        const int cb = idx % ngcx, mode = (idx / ngcx) & 1, grp = idx / (2 * ngcx);
        const int c = cb * blockDim.x + threadIdx.x;
        if (c < v.nCell) item_gradcell_group(v, mode, grp, c);
        if (threadIdx.x == 0 && idx == 0) { /* (ticks: the slot of k_gradcell stays empty in fused runs) */ }
    }
}
__global__ __launch_bounds__(64) void k_bcsens_pre(View v) {
    int c = blockIdx.x * blockDim.x + threadIdx.x, prof = blockIdx.y, s = blockIdx.z;
    if (c < v.nz) item_bcsens_pre(v, s, prof, c);
}
// Round 6: FOUR lanes per (system, profile, column) -- item_bcsens_contract's nz products in four contiguous quarters of the rows, the
// quarters added in lane order --: with one lane per column the launch is 107 waves walking 13 batches of loads one after the other
// (13 us at the headline size, all latency); the gradient's tail behind the adjoint solve is a chain of such kernels.
constexpr int BCC_L = 4;
__global__ void k_bcsens_contract(View v, double* lfPart) {
    if (gate_closed(v)) return;
    const int t = blockIdx.x * blockDim.x + threadIdx.x, c = t / BCC_L, l = t % BCC_L, prof = blockIdx.y, s = blockIdx.z;
    // (the step-bound maxima k_gradfinal collects behind this launch start from zero)
    if (lfPart && (blockIdx.x | blockIdx.y | blockIdx.z) == 0 && threadIdx.x < LFNB) lfPart[threadIdx.x] = 0.0;
    tick_begin(v.ticks, TK_BCSENS);
    cplx acc = cplx{0.0, 0.0};
    const bool on = c < v.nz && v.sysOn[s];
    if (on) {
        const cplx* D = v.dBC + ((long)s * 2 + prof) * v.nz * v.nz + c;
        const cplx* w = (prof == 0 ? v.wL : v.wR) + (long)s * v.nz;
        const int per = (v.nz + BCC_L - 1) / BCC_L, j0 = l * per, j1 = min(j0 + per, v.nz);
        int j = j0;
        for (; j + 8 <= j1; j += 8) {
            cplx d[8], ww[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) { d[q] = D[(long)(j + q) * v.nz]; ww[q] = w[j + q]; }
#pragma unroll
            for (int q = 0; q < 8; ++q) acc += d[q] * ww[q];
        }
        for (; j < j1; ++j) acc += D[(long)j * v.nz] * w[j];
    }
    // lanes 4c .. 4c+3 are neighbours in a wave (128 threads per workgroup: a multiple of four)
    const double r1 = __shfl_down(acc.re, 1, BCC_L), r2 = __shfl_down(acc.re, 2, BCC_L), r3 = __shfl_down(acc.re, 3, BCC_L);
    const double i1 = __shfl_down(acc.im, 1, BCC_L), i2 = __shfl_down(acc.im, 2, BCC_L), i3 = __shfl_down(acc.im, 3, BCC_L);
    if (c < v.nz && l == 0) {
        const cplx tot = cplx{((acc.re + r1) + r2) + r3, ((acc.im + i1) + i2) + i3};
        const long o = (long)s * v.nz + c;
        if (prof == 0) v.gL[o] = tot; else v.gR[o] = tot;
    }
    tick_end(v.ticks, TK_BCSENS);
}
__global__ void k_gradcell(View v) {
    if (gate_closed(v)) return;
    int c = blockIdx.x * blockDim.x + threadIdx.x, mode = blockIdx.y, grp = blockIdx.z;
    tick_begin(v.ticks, TK_GRADCELL);
    if (c < v.nCell) item_gradcell_group(v, mode, grp, c);
    tick_end(v.ticks, TK_GRADCELL);
}
__global__ __launch_bounds__(64) void k_qterm(View v) {
    int ky = blockIdx.x * blockDim.x + threadIdx.x, s = blockIdx.y;
    if (ky < v.ny) item_qterm(v, s, ky);
}
// final assembly with EIGHT lanes per active cell (round 6; four until then: a latency-bound loop over the systems -- the last kernel of a
// leapfrog step, and the next step's first waits for it), each taking every eighth system / partial sum; the eight partial sums are
// added in lane order
constexpr int GF_L = 8;
__global__ void k_gradfinal(View v, LfMom mom) {
    if (gate_closed(v)) return;
    const int t = TID1, a = t / GF_L, l = t % GF_L;
    tick_begin(v.ticks, TK_GRADFINAL);
    double g = 0.0;
    if (a < v.nAC) {
        const int cell = v.act[a];
        const int ky = cell % v.ny, kz = cell / v.ny;
        for (int q = l; q < 2 * GRAD_NG; q += GF_L) g += v.gPartG[(long)q * v.nCell + cell];
#pragma unroll 4
        for (int s = l; s < v.S; s += GF_L) g += gradfinal_sys(v, s, ky, kz);
        if (kz == v.zid)
            for (int s = l; s < v.S; s += GF_L) g += v.qPart[(long)s * v.ny + ky];
    }
    // lanes 8a .. 8a+7 are neighbours in a wave (the grid is a multiple of 64 threads)
    double gs = g;
#pragma unroll
    for (int o = 1; o < GF_L; ++o) gs += __shfl_down(g, o, GF_L);
    const bool own = a < v.nAC && l == 0;
    double gd = 0.0;
    if (own) { gd = exp(v.m[a]) * gs; v.grad[a] = gd; }
    if (mom.on) {
        // the momentum update of the leapfrog step this gradient belongs to (k_lf_momentum_max's arithmetic, HMCSampler.jl:255-263)
        // and the maxima of |dt*invM*p| for the position update that follows (:237-240; zeroed by k_bcsens_contract)
        const LfView& L = mom.L;
        double mx = 0.0;
        if (own) {
            double p = L.p[a];
            if (mom.done[a] != mom.gen) {
                double acc = 0.0;
                for (long long q = L.wmRow[a]; q < L.wmRow[a + 1]; ++q) {
                    const long long j = L.wmCol[q];
                    acc += L.wmVal[q] * (L.m[j] - L.mref[j]);
                }
                const double gt = gd + mom.lambda * acc;
                p = p - mom.cdt * gt;
                L.p[a] = p;
                mom.done[a] = mom.gen;
            }
            mx = fmax(0.0, fabs(mom.dt * L.invM[a] * p));
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) mx = fmax(mx, __shfl_down(mx, o, 64));
        if ((threadIdx.x & 63) == 0 && mx > 0.0)
            atomicMax(reinterpret_cast<unsigned long long*>(L.part + (blockIdx.x & (LFNB - 1))), (unsigned long long)__double_as_longlong(mx));
    }
    tick_end(v.ticks, TK_GRADFINAL);
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

// g <- g + lambda*Wm*(m - mref) ; p <- p - c*dt*g      (HMCSampler.jl:223-228, 255-263)
// and the partial maxima of |dt*invM*p| (:237-240) for the position update that follows,
// both in one launch (the leapfrog loop: momentum update, then the step bound of the position update that follows it)
__global__ __launch_bounds__(256) void k_lf_momentum_max(LfView L, double lambda, double cdt, double dt) {
    __shared__ double sh[4];
    tick_begin(L.ticks, TK_LF_MOM);
    double mx = 0.0;
    for (int a = blockIdx.x * 256 + threadIdx.x; a < L.n; a += 256 * LFNB) {
        double acc = 0.0;
        for (long long t = L.wmRow[a]; t < L.wmRow[a + 1]; ++t) {
            const long long j = L.wmCol[t];
            acc += L.wmVal[t] * (L.m[j] - L.mref[j]);
        }
        const double g = L.g[a] + lambda * acc;
        const double p = L.p[a] - cdt * g;
        L.p[a] = p;
        mx = fmax(mx, fabs(dt * L.invM[a] * p));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmax(mx, __shfl_down(mx, o, 64));
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = mx;
    __syncthreads();
    if (threadIdx.x == 0) L.part[blockIdx.x] = fmax(fmax(sh[0], sh[1]), fmax(sh[2], sh[3]));
    tick_end(L.ticks, TK_LF_MOM);
}
// m += dm (clamped to max |dm| = 3), reflect at the ln-sigma bounds, flip momentum (:241-247, :515-559)
__global__ void k_lf_step(LfView L, double dt, double lo, double hi) {
    const int a = TID1;
    tick_begin(L.ticks, TK_LF_STEP);
    if (a >= L.n) return;
    lf_step_one(L, a, dt, lo, hi, lf_step_bound(L));
    tick_end(L.ticks, TK_LF_STEP);
}
// mnorm = 0.5*lambda*(m-mref)' Wm (m-mref)   (HMCSampler.jl:389-391): partial sums, then block 0 finishes
__global__ __launch_bounds__(256) void k_lf_mnorm(LfView L, double lambda) {
    __shared__ double sh[32];
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
