// Per-item bodies of the "one thread = one item" kernels of the hot path, written against a
// plain-pointer view of the context's HBM arrays (struct View).  hmcmt_kernels.hip wraps each in
// a __global__ launcher; tests/emul/ instantiates the same bodies on the host for CPU unit tests.
//
// Data layout.  S = 2*nFreq systems, s < nFreq: TE at frequency s, else TM at
// frequency s-nFreq.  Every field-like vector lives on the PADDED NODAL GRID
//     v[s][iz][iy],  iz = 0..nz (NZP = nz+1 rows),  iy = 0..NYP-1,  NYP = roundup(ny+1, 16)
// with iy fastest; columns iy > ny are always zero, boundary nodes (iy=0, iy=ny, iz=0, iz=nz)
// hold Dirichlet values for a forward field and zero for residuals / search directions / adjoint
// fields.  This is the reference's nodal ordering `(iz-1)(ny+1)+iy` (MT2DFwdSolver.jl:232) with a
// padded row stride, so the interior/boundary split of getBoundaryIndex (:227-248) is implicit.
#pragma once
#include "hmcmt_math.h"

namespace hmcmt {

struct View {
    // sizes
    int ny, nz, NYP, NZP, nFreq, S, nRx, nData, nAC, nCell;
    int twist;                     // 1: twisted (two-sided) factorisation of the FDM tridiagonals, see item_pivot
    int zid;                       // node row of the receivers (mt2DTE.jl:66-67), 0-based
    long long* ticks;              // HMCMT_TICKS (measurement only): [2][32] earliest start / latest end of the kernels around the solves
    int dbg;                       // test hooks (hmcmt_debug_flags): bit 1 = leave the boundary-derivative terms B^T v out of the gradient
    const int* gate;               // non-null: the kernel was queued BEHIND a persistent solve before the host knew its outcome and runs only if
    int gateGen;                   //   the solve left *gate == gateGen (every system converged); else it returns at once and the host queues it again
    long vstride;                  // NZP*NYP elements per system
    // mesh (constant)
    const double* yLen;            // [ny]
    const double* zLen;            // [nz]
    const double* omega;           // [S]
    const double* lam;             // [NYP] generalised eigenvalues of the y-operator (0 in the pad)
    const int* sysOn;              // [S] 0 for the systems of a polarisation without data (never solved)
    // model-dependent
    const double* m;               // [nAC] ln(sigma) on active cells
    double* sigma;                 // [nCell]
    const int* cell2act;           // [nCell] active index or -1
    const double* bg;              // [nCell]
    const int* act;                // [nAC] cell id of each active cell
    double* sigMeanA;              // [nz] arithmetic lateral mean (MT1DSensitivity.jl:313)
    double* sigMeanG;              // [nz] geometric lateral mean (the FDM background of both modes)
    // stencil coefficients on the padded nodal grid, per mode: [2][NZP*NYP]
    double* cY;                    // coupling node (iy,iz) <-> (iy+1,iz)
    double* cZ;                    // coupling node (iy,iz) <-> (iy,iz+1)
    double* dK;                    // diagonal of K
    double* dM;                    // node mass D (A = K + i*omega*D)
    // FDM background tridiagonals per mode: [2][NZP]
    double* mzq;                   // multiplies lambda_j
    double* dgz;                   // z-stiffness diagonal
    double* ofz;                   // z-stiffness coupling iz <-> iz+1
    double* mzs;                   // multiplies i*omega
    cplx* invp;                    // [S][NZP][NYP] inverse pivots of the per-(s,j) tridiagonal LDL^T
    // fields
    cplx* X;                       // forward fields (exTE | hxTM), padded nodal layout
    cplx* Lam;                     // adjoint fields
    cplx* R;                       // residual / right-hand side
    // receivers
    const int* rxIdn;              // [nRx] forward interpolation node `id` (0-based upper node)
    const double* rxDy1;           // [nRx]
    const double* rxDy2;           // [nRx]
    const int* rxKL;               // [nRx] linearInterp (sensUtils.jl:133-161)
    const int* rxKR;
    const double* rxWL;
    const double* rxWR;
    cplx* Zrx;                     // [S][nRx]
    int* rxN0;                     // [S][nRx] derivative window start
    cplx* rxD;                     // [S][nRx][11]: d0[4], d1[4], dq[3]
    cplx* rxCoef;                  // [S][nRx] sum of conj(W^T W r) over the data of (s, rx)
    // data
    const int* predSys;            // [nData] system of the p-th masked entry of the full table
    const int* predRx;             // [nData] receiver of it
    const int* datSys;             // [nData] system addressed by (freqID, dtID) of datum p
    const int* datRx;              // [nData] rxID-1
    const int* predKind;           // [nData] what the p-th masked table entry is: 0 impedance, 1 apparent resistivity, 2 phase (degrees)
    const int* datKind;            // [nData] the same for the component dtID of datum p addresses
    const cplx* obs;               // [nData]
    const double* dataW;           // [nData]
    cplx* pred;                    // [nData]
    cplx* vbar;                    // [nData] conj(W^T W (pred-obs))
    double* misfitPart;            // [nData] 0.5*|W(pred-obs)|^2
    double* gPartG;                // [2][GRAD_NG][nCell] P-terms by mode and frequency group (GPU launch)
    const int* srStart;            // [S*nRx+1] CSR of data per (s, rx)
    const int* srList;             // [nData]
    // boundary / gradient work arrays
    cplx* srcB;                    // [S][4]: source on (0,zid), (ny,zid), (0,zid+1), (ny,zid+1)
    cplx* wL;                      // [S][nz]  weights on left-boundary nodes iz = 1..nz
    cplx* wR;                      // [S][nz]
    cplx* colw;                    // [S][ny]
    cplx* gL;                      // [S][nz]  dBC^T w, left column cells
    cplx* gR;                      // [S][nz]
    cplx* gMn;                     // [S][nz]  last row of the mean-profile sensitivity
    cplx* dBC;                     // [S][2][nz rows][nz cols]  d(edge column field at row j+1)/d sigma_c, left / right profile
    cplx* bcsL;                    // [S][nz]  sensitivity-version boundary fields (TM term)
    cplx* bcsR;                    // [S][nz]
    cplx* bcsB;                    // [S]      mean-profile bottom value
    cplx* fwdTab;                  // [S][FWD_NQ][nz][ny+1] per-layer terms of the forward 1-D columns (layer_forward)
    cplx* sensTab;                 // [S][3][5][nz+1]  per-layer terms of the sensitivity profiles
    cplx* sensEu;                  // [S][3][nz+1]     up-going amplitude per row (scratch in stage 2)
    cplx* sensEd;                  // [S][3][nz+1]
    cplx* sensMix;                 // [S][3][4][nz]
    cplx* sensDz1;                 // [S][3][nz]       d z1 / d sigma_c
    cplx* sensZ1;                  // [S][3]
    int* sensDead;                 // [S][3]           cut-off row or nz+1
    double* qPart;                 // [S][ny] Q-term (explicit sigma-dependence of the data functional) per system
    double* gPart;                 // [2][nCell] P-term partial sums per mode
    double* grad;                  // [nAC]
};

HD long nidx(const View& v, int iy, int iz) { return (long)iz * v.NYP + iy; }

// --- sigma = A*exp(m) + bg  (HMCSampler.jl:292-293, HMCUtility.jl:69-77)
HD void item_sigma(const View& v, int cell) {
    int a = v.cell2act[cell];
    v.sigma[cell] = v.bg[cell] + (a >= 0 ? exp(v.m[a]) : 0.0);
}

// --- lateral means of one cell row (arithmetic: MT1DSensitivity.jl:313; geometric: FDM background)
HD void item_rowmean(const View& v, int kz) {
    double sa = 0.0, sl = 0.0;
    for (int ky = 0; ky < v.ny; ++ky) {
        double s = v.sigma[(long)kz * v.ny + ky];
        sa += s;
        sl += log(s);
    }
    v.sigMeanA[kz] = sa / v.ny;
    v.sigMeanG[kz] = exp(sl / v.ny);
}

// --- 5-point stencil of  Grad' diag(AveCF*F*q) Grad  and node mass AveCN*F*s at one node
//     (MT2DFwdSolver.jl:124-135,150-161; MT2DOperators.jl:35-48,118-130).
//     mode 0 (TE): q = 1/mu0, s = sigma ; mode 1 (TM): q = 1/sigma, s = mu0.
HD double cellq(const View& v, int mode, int ky, int kz) {
    return mode == 0 ? 1.0 / MU0 : 1.0 / v.sigma[(long)kz * v.ny + ky];
}
HD double cells(const View& v, int mode, int ky, int kz) {
    return mode == 0 ? v.sigma[(long)kz * v.ny + ky] : MU0;
}
HD double coupY(const View& v, int mode, int iy, int iz) {   // edge (iy,iz)-(iy+1,iz), 1<=iz<=nz-1
    return -(0.5 / v.yLen[iy]) * (v.zLen[iz - 1] * cellq(v, mode, iy, iz - 1) + v.zLen[iz] * cellq(v, mode, iy, iz));
}
HD double coupZ(const View& v, int mode, int iy, int iz) {   // edge (iy,iz)-(iy,iz+1), 1<=iy<=ny-1
    return -(0.5 / v.zLen[iz]) * (v.yLen[iy - 1] * cellq(v, mode, iy - 1, iz) + v.yLen[iy] * cellq(v, mode, iy, iz));
}
HD void item_coef(const View& v, int mode, int iy, int iz, bool doK, bool doM) {
    const long o = (long)mode * v.vstride + nidx(v, iy, iz);
    const bool rowI = iz >= 1 && iz <= v.nz - 1, colI = iy >= 1 && iy <= v.ny - 1;
    if (doK) {
        double cy = 0.0, cz = 0.0, dk = 0.0;
        if (rowI && iy <= v.ny - 1) cy = coupY(v, mode, iy, iz);
        if (colI && iz <= v.nz - 1) cz = coupZ(v, mode, iy, iz);
        if (rowI && colI)
            dk = -(coupY(v, mode, iy, iz) + coupY(v, mode, iy - 1, iz) + coupZ(v, mode, iy, iz) + coupZ(v, mode, iy, iz - 1));
        v.cY[o] = cy; v.cZ[o] = cz; v.dK[o] = dk;
    }
    if (doM) {
        double d = 0.0;
        if (rowI && colI) {
            double ya = v.yLen[iy - 1], yb = v.yLen[iy], za = v.zLen[iz - 1], zb = v.zLen[iz];
            d = 0.25 * (ya * za * cells(v, mode, iy - 1, iz - 1) + yb * za * cells(v, mode, iy, iz - 1) +
                        ya * zb * cells(v, mode, iy - 1, iz) + yb * zb * cells(v, mode, iy, iz));
        }
        v.dM[o] = d;
    }
}

// --- background (laterally averaged) tridiagonal in z for the fast-diagonalisation
//     preconditioner: P = Ty (x) Mz_q + My (x) (Tz_q + i w Mz_s)   
// (values; item_fdm_z stores them, k_pivot also computes them straight into its LDS copy)
HD void fdm_z_values(const View& v, int mode, int iz, double& mzq, double& dgz, double& ofz, double& mzs) {
    if (iz < 1 || iz > v.nz - 1) { mzq = 0; dgz = 0; ofz = 0; mzs = 0; return; }
    double qa, qb, sa, sb;
    // the lateral mean of the background is the geometric mean of sigma for both modes (the arithmetic mean of the coefficient the
    // operator is linear in was tried in round 3 and removed: DESIGN section 9)
    const double* sTE = v.sigMeanG;
    if (mode == 0) { qa = qb = 1.0 / MU0; sa = sTE[iz - 1]; sb = sTE[iz]; }
    else { qa = 1.0 / v.sigMeanG[iz - 1]; qb = 1.0 / v.sigMeanG[iz]; sa = sb = MU0; }
    double za = v.zLen[iz - 1], zb = v.zLen[iz];
    mzq = 0.5 * (za * qa + zb * qb);
    mzs = 0.5 * (za * sa + zb * sb);
    dgz = qa / za + qb / zb;
    ofz = (iz <= v.nz - 2) ? -(qb / zb) : 0.0;
}
HD void item_fdm_z(const View& v, int mode, int iz) {
    const long o = (long)mode * v.NZP + iz;
    fdm_z_values(v, mode, iz, v.mzq[o], v.dgz[o], v.ofz[o], v.mzs[o]);
}

// --- inverse pivots of the (s, j) tridiagonal  d_iz = lam_j*mzq + dgz + i w mzs  in TWISTED form:
//     rows 1..mid are eliminated top-down, rows n..mid+1 bottom-up (two independent recurrences the
//     solve kernels run interleaved, halving their dependent chain); the factor that couples the two
//     halves at rows mid, mid+1 is stored in the unused boundary row iz = 0.
// View.twist selects it at run time: the fused forward kernel (k_fdm_fwd) runs the two halves in the two lane
//     halves of its sweeping wave and halves its serial chain; the stand-alone tridiagonal kernels, which read
//     global memory, are instruction-issue-bound and do not gain (20 us vs 16 us), so they use twist = 0
//     (mid = n: the classic Thomas factorisation) unless they have to follow the fused kernel's pivots.
HD int twist_mid(int n, int twist) { return twist ? (n + 1) / 2 : n; }   // rows 1..mid | mid+1..n   (n = nz-1 >= 2)
// (mzq, dgz, ofz, mzs: the mode's four coefficient rows -- the GPU kernel passes copies staged in LDS: read from global
// memory they cost the serial loop one memory round trip per step)
// ip32 (optional): complex64 copy of the pivots for the mixed-precision FDM stage, written along
HD void item_pivot_tab(const View& v, int s, int j, const double* mzq, const double* dgz, const double* ofz, const double* mzs,
                       float* ip32 = nullptr) {
    const double w = v.omega[s], lam = v.lam[j];
    cplx* ip = v.invp + (long)s * v.vstride + j;
    const int n = v.nz - 1, mid = twist_mid(n, v.twist);
    // the two recurrences are independent: one loop advances both (two dependent chains of complex reciprocals in
    // flight instead of one after the other: k_pivot 58 -> us)
    cplx pt = cplx{0, 0}, pb = cplx{0, 0};
    for (int t = 0; t < mid; ++t) {
        const int it = 1 + t, ib = n - t;
        {                                                  // d'_iz = d_iz - of_{iz-1}^2 / d'_{iz-1}
            cplx d = cplx{lam * mzq[it] + dgz[it], w * mzs[it]};
            if (it > 1) d -= (ofz[it - 1] * ofz[it - 1]) * pt;
            pt = crecip_plain(d);
            ip[(long)it * v.NYP] = pt;
            if (ip32) { ip32[2 * (long)it * v.NYP] = (float)pt.re; ip32[2 * (long)it * v.NYP + 1] = (float)pt.im; }
        }
        if (ib >= mid + 1) {                               // d''_iz = d_iz - of_iz^2 / d''_{iz+1}
            cplx d = cplx{lam * mzq[ib] + dgz[ib], w * mzs[ib]};
            if (ib < n) d -= (ofz[ib] * ofz[ib]) * pb;
            pb = crecip_plain(d);
            ip[(long)ib * v.NYP] = pb;
            if (ip32) { ip32[2 * (long)ib * v.NYP] = (float)pb.re; ip32[2 * (long)ib * v.NYP + 1] = (float)pb.im; }
        }
    }
    // middle coupling of the two normalised halves: x_mid + c x_{mid+1} = y'_mid, x_{mid+1} + c' x_mid = y''_{mid+1}
    // with c = o*ip_mid, c' = o*ip_{mid+1}; store 1/(1 - c c') in the unused boundary row iz = 0
    const double o = ofz[mid];
    const cplx jn = (mid + 1 <= n) ? crecip(cplx{1.0, 0.0} - (o * o) * (pt * pb)) : cplx{1.0, 0.0};   // (pt, pb: the pivots of rows mid, mid+1)
    ip[0] = jn;
    if (ip32) { ip32[0] = (float)jn.re; ip32[1] = (float)jn.im; }
}
HD void item_pivot(const View& v, int s, int j) {
    const int mode = s >= v.nFreq;
    item_pivot_tab(v, s, j, v.mzq + (long)mode * v.NZP, v.dgz + (long)mode * v.NZP, v.ofz + (long)mode * v.NZP,
                   v.mzs + (long)mode * v.NZP);
}

// --- per-layer terms of the 1-D column under boundary node column `col` (0 = left edge, ny = right
//     edge, else the width-weighted mean of the two adjacent cell columns, mt2DTE.jl:127-131)
HD double bc_column_sigma(const View& v, int j, int col) {
    if (col == 0) return v.sigma[(long)j * v.ny];
    if (col == v.ny) return v.sigma[(long)j * v.ny + v.ny - 1];
    const double ya = v.yLen[col - 1], yb = v.yLen[col];
    return (v.sigma[(long)j * v.ny + col - 1] * ya + v.sigma[(long)j * v.ny + col] * yb) / (ya + yb);
}
HD void item_bc_layers(const View& v, int s, int j, int col) {
    if (!v.sysOn[s]) return;
    const bool lastLayer = j + 1 >= v.nz;
    const double sig = bc_column_sigma(v, j, col), sigNext = lastLayer ? sig : bc_column_sigma(v, j + 1, col);
    cplx t[FWD_NQ];
    layer_forward(sig, sigNext, lastLayer, v.omega[s], v.zLen[j], t);
    const long ls = v.ny + 1, qs = (long)v.nz * ls;
    cplx* T = v.fwdTab + (long)s * FWD_NQ * qs + (long)j * ls + col;
    for (int q = 0; q < FWD_NQ; ++q) T[q * qs] = t[q];
}

// The per-layer tables depend on the frequency and the column profile, not on the polarisation: the GPU computes them
// once per frequency (slot f of fwdTab = the TE system's) for the frequencies with at least one polarisation in the data
HD void item_bc_layers_f(const View& v, int f, int j, int col) {
    if (!v.sysOn[f] && !v.sysOn[v.nFreq + f]) return;
    const bool lastLayer = j + 1 >= v.nz;
    const double sig = bc_column_sigma(v, j, col), sigNext = lastLayer ? sig : bc_column_sigma(v, j + 1, col);
    cplx t[FWD_NQ];
    layer_forward(sig, sigNext, lastLayer, v.omega[f], v.zLen[j], t);
    const long ls = v.ny + 1, qs = (long)v.nz * ls;
    cplx* T = v.fwdTab + (long)f * FWD_NQ * qs + (long)j * ls + col;
    for (int q = 0; q < FWD_NQ; ++q) T[q * qs] = t[q];
}

// --- Dirichlet values of the forward problem written into X's boundary nodes
//     (getBoundaryMT2DTE/TM, mt2DTE.jl:100-134, mt2DTM.jl:100-134).  col = 0..ny.
HD void item_bc_forward(const View& v, int s, int col) {
    if (!v.sysOn[s]) return;
    const bool tm = s >= v.nFreq;
    cplx* X = v.X + (long)s * v.vstride;
    X[nidx(v, col, 0)] = cplx{1.0, 0.0};                  // top row incl. corners
    const long ls = v.ny + 1, qs = (long)v.nz * ls;
    const cplx* T = v.fwdTab + (long)s * FWD_NQ * qs + col;
    if (col == 0 || col == v.ny) bc1d_forward_tab(v.omega[s], v.nz, T, qs, ls, tm, X + nidx(v, col, 1), v.NYP);
    else X[nidx(v, col, v.nz)] = bc1d_forward_tab(v.omega[s], v.nz, T, qs, ls, tm, nullptr, 0);
}

// --- y = K u + i w D u at one interior node (u on the padded nodal grid incl. boundary values)
HD cplx stencil_apply(const View& v, int s, const cplx* u, int iy, int iz) {
    const int mode = s >= v.nFreq;
    const long mo = (long)mode * v.vstride, o = nidx(v, iy, iz);
    const double *cY = v.cY + mo, *cZ = v.cZ + mo;
    cplx c = u[o];
    cplx acc = cplx{v.dK[mo + o] * c.re - v.omega[s] * v.dM[mo + o] * c.im,
                    v.dK[mo + o] * c.im + v.omega[s] * v.dM[mo + o] * c.re};
    acc += cY[o] * u[o + 1];
    acc += cY[o - 1] * u[o - 1];
    acc += cZ[o] * u[o + v.NYP];
    acc += cZ[o - v.NYP] * u[o - v.NYP];
    return acc;
}

// --- rhs = -Aio*bc (mt2DTE.jl:44): u = X with zero interior, so K u picks the boundary terms
HD void item_rhs(const View& v, int s, int iy, int iz) {
    const long o = (long)s * v.vstride + nidx(v, iy, iz);
    const bool interior = iz >= 1 && iz <= v.nz - 1 && iy >= 1 && iy <= v.ny - 1;
    if (!interior) { v.R[o] = cplx{0, 0}; return; }
    const int mode = s >= v.nFreq;
    const long mo = (long)mode * v.vstride, n = nidx(v, iy, iz);
    const cplx* X = v.X + (long)s * v.vstride;
    cplx acc = cplx{0, 0};
    if (iy == v.ny - 1) acc += v.cY[mo + n] * X[n + 1];
    if (iy == 1) acc += v.cY[mo + n - 1] * X[n - 1];
    if (iz == v.nz - 1) acc += v.cZ[mo + n] * X[n + v.NYP];
    if (iz == 1) acc += v.cZ[mo + n - v.NYP] * X[n - v.NYP];
    v.R[o] = -acc;
}

// --- impedance and its derivative at one receiver of one system
HD void item_rx(const View& v, int s, int r, bool wantDeriv) {
    if (!v.sysOn[s]) {                                    // polarisation absent from the data set
        v.Zrx[(long)s * v.nRx + r] = cplx{0, 0};
        if (wantDeriv) {
            v.rxN0[(long)s * v.nRx + r] = 0;
            for (int i = 0; i < 11; ++i) v.rxD[((long)s * v.nRx + r) * 11 + i] = cplx{0, 0};
        }
        return;
    }
    const bool tm = s >= v.nFreq;
    const cplx* F0 = v.X + (long)s * v.vstride + nidx(v, 0, v.zid);
    const cplx* F1 = F0 + v.NYP;
    const double* sig1 = v.sigma + (long)v.zid * v.ny;
    const double dz1 = v.zLen[v.zid];
    v.Zrx[(long)s * v.nRx + r] = rx_impedance(tm, v.omega[s], v.ny, F0, F1, v.yLen, sig1, dz1,
                                               v.rxIdn[r], v.rxDy1[r], v.rxDy2[r]);
    if (wantDeriv) {
        cplx* D = v.rxD + ((long)s * v.nRx + r) * 11;
        rx_impedance_deriv(tm, v.omega[s], v.ny, F0, F1, v.yLen, sig1, dz1, v.rxKL[r], v.rxKR[r],
                           v.rxWL[r], v.rxWR[r], &v.rxN0[(long)s * v.nRx + r], D, D + 4, D + 8);
    }
}

// --- residual, misfit terms and conj(W'W r) per datum (HMCSampler.jl:298-304, compJacTMatVec.jl:160)
//     DataType Rho_Pha: the data are rho_a = |Z|^2/(w mu0) and phi = atan2(Im Z, Re Z) in degrees (compMTRespTE,
//     mt2DTE.jl:253-255; compMTRespTM, mt2DTM.jl:236-238), real.  Their sensitivities are the impedance's times
//     2 conj(Z)/(w mu0) and -i (180/pi) conj(Z)/|Z|^2 (dataFuncSens.jl:137-141, :307-311); the residual weights being
//     real, the whole adjoint machinery stays the impedance one with vbar = weight x that factor.
HD void item_resid(const View& v, int p) {
    const int sp = v.predSys[p];
    const cplx z = v.Zrx[(long)sp * v.nRx + v.predRx[p]];
    const int kind = v.predKind[p];
    cplx val = z;
    if (kind == 1) val = cplx{cabs2(z) / (v.omega[sp] * MU0), 0.0};
    else if (kind == 2) val = cplx{atan2(z.im, z.re) * (180.0 / 3.14159265358979323846), 0.0};
    v.pred[p] = val;
    const cplx res = v.dataW[p] * (val - v.obs[p]);
    v.misfitPart[p] = 0.5 * cabs2(res);
    const cplx wr = v.dataW[p] * res;
    const int dk = v.datKind[p];
    if (dk == 0) { v.vbar[p] = conj(wr); return; }
    const int sd = v.datSys[p];
    const cplx zs = v.Zrx[(long)sd * v.nRx + v.datRx[p]];
    if (dk == 1) v.vbar[p] = (wr.re * 2.0 / (v.omega[sd] * MU0)) * conj(zs);
    else {
        const cplx c = conj(zs);                                   // -i conj(Z) = (Im conj Z, -Re conj Z)
        v.vbar[p] = (wr.re * (180.0 / 3.14159265358979323846) / cabs2(zs)) * cplx{c.im, -c.re};
    }
}

// --- per (s, rx): sum of vbar over the data addressing that receiver of that system
HD void item_rxcoef(const View& v, int s, int r) {
    const long k = (long)s * v.nRx + r;
    cplx c = cplx{0, 0};
    for (int t = v.srStart[k]; t < v.srStart[k + 1]; ++t) c += v.vbar[v.srList[t]];
    v.rxCoef[k] = c;
}

// --- adjoint source sVec = L^T conj(v) on node rows zid, zid+1 (compJacTMatVec.jl:208, :279):
//     interior part -> R, boundary part (iy = 0 / ny) -> srcB
HD void item_src(const View& v, int s, int row, int iy) {
    const int iz = v.zid + row;
    cplx acc = cplx{0, 0};
    for (int r = 0; r < v.nRx; ++r) {
        const long k = (long)s * v.nRx + r;
        const int o = iy - v.rxN0[k];
        if (o >= 0 && o < 4) acc += v.rxCoef[k] * v.rxD[k * 11 + row * 4 + o];
    }
    const bool interior = iz >= 1 && iz <= v.nz - 1 && iy >= 1 && iy <= v.ny - 1;
    if (interior) v.R[(long)s * v.vstride + nidx(v, iy, iz)] = acc;
    else if (iy == 0) v.srcB[(long)s * 4 + row * 2 + 0] = acc;
    else if (iy == v.ny) v.srcB[(long)s * 4 + row * 2 + 1] = acc;
}

// --- boundary weights w_b = s_b - (K lambda)_b  (compJacTMatVec.jl:240-242, :312-316)
HD cplx srcB_at(const View& v, int s, int side, int iz) {
    if (iz == v.zid) return v.srcB[(long)s * 4 + side];
    if (iz == v.zid + 1) return v.srcB[(long)s * 4 + 2 + side];
    return cplx{0, 0};
}
HD void item_wside(const View& v, int s, int iz) {          // iz = 1..nz
    const int mode = s >= v.nFreq;
    const long mo = (long)mode * v.vstride;
    const cplx* L = v.Lam + (long)s * v.vstride;
    cplx wl = srcB_at(v, s, 0, iz), wr = srcB_at(v, s, 1, iz);
    if (iz <= v.nz - 1) {
        wl -= v.cY[mo + nidx(v, 0, iz)] * L[nidx(v, 1, iz)];
        wr -= v.cY[mo + nidx(v, v.ny - 1, iz)] * L[nidx(v, v.ny - 1, iz)];
    }
    v.wL[(long)s * v.nz + iz - 1] = wl;
    v.wR[(long)s * v.nz + iz - 1] = wr;
}
HD cplx wbottom(const View& v, int s, int iy) {             // iy = 1..ny-1
    const int mode = s >= v.nFreq;
    const long mo = (long)mode * v.vstride;
    const cplx* L = v.Lam + (long)s * v.vstride;
    return -(v.cZ[mo + nidx(v, iy, v.nz - 1)] * L[nidx(v, iy, v.nz - 1)]);
}
HD void item_colw(const View& v, int s, int ky) {           // MT1DSensitivity.jl:315-328
    cplx c = cplx{0, 0};
    if (ky >= 1) c += wbottom(v, s, ky) * (v.yLen[ky] / (v.yLen[ky - 1] + v.yLen[ky]));
    if (ky + 1 <= v.ny - 1) c += wbottom(v, s, ky + 1) * (v.yLen[ky] / (v.yLen[ky] + v.yLen[ky + 1]));
    v.colw[(long)s * v.ny + ky] = c;
}

// --- 1-D sensitivities (getBCDerivMatrix, MT1DSensitivity.jl:253-333) in three stages.
//     prof 0: left edge column, 1: right edge column, 2: lateral-mean profile (:313-314)
HD void item_sens_layers(const View& v, int s, int prof, int j) {      // j = 0..nz (nz = half-space copy)
    if (!v.sysOn[s]) return;
    const int jj = j < v.nz ? j : v.nz - 1;
    const double sig = prof == 0 ? v.sigma[(long)jj * v.ny] : (prof == 1 ? v.sigma[(long)jj * v.ny + v.ny - 1] : v.sigMeanA[jj]);
    cplx t[5];
    layer_sens(sig, v.omega[s], v.zLen[jj], t);
    const long n1 = v.nz + 1;
    cplx* T = v.sensTab + ((long)s * 3 + prof) * 5 * n1 + j;
    for (int q = 0; q < 5; ++q) T[q * n1] = t[q];
}
HD void item_sens_profile(const View& v, int s, int prof) {
    if (!v.sysOn[s]) return;
    const bool tm = s >= v.nFreq;
    const long n1 = v.nz + 1, sp = (long)s * 3 + prof;
    const cplx* T = v.sensTab + sp * 5 * n1;
    cplx* fout = nullptr;
    long fstride = 1;
    if (tm) {                                             // `bc` of getBCderivTM (compJacTMatVec.jl:309,315)
        if (prof == 0) fout = v.bcsL + (long)s * v.nz;
        else if (prof == 1) fout = v.bcsR + (long)s * v.nz;
        else { fout = v.bcsB + s; fstride = 0; }          // only the bottom value of the mean profile
    }
    v.sensDead[sp] = sens_profile(v.omega[s], v.nz, v.zLen, tm, T, T + n1, T + 2 * n1, T + 3 * n1, T + 4 * n1,
                                  v.sensEu + sp * n1, v.sensEd + sp * n1, v.sensMix + sp * 4 * v.nz,
                                  v.sensDz1 + sp * v.nz, v.sensZ1 + sp, fout, fstride);
}
// dBC^T w, one derivative column c of one profile
HD void item_bcsens(const View& v, int s, int prof, int c) {
    const long o = (long)s * v.nz + c;
    if (!v.sysOn[s]) {
        if (prof == 0) v.gL[o] = cplx{0, 0}; else if (prof == 1) v.gR[o] = cplx{0, 0}; else v.gMn[o] = cplx{0, 0};
        return;
    }
    const bool tm = s >= v.nFreq;
    const long n1 = v.nz + 1, sp = (long)s * 3 + prof;
    const cplx* T = v.sensTab + sp * 5 * n1;
    const cplx* w = prof == 0 ? v.wL + (long)s * v.nz : (prof == 1 ? v.wR + (long)s * v.nz : nullptr);
    const cplx g = bc1d_sens_column(v.omega[s], v.nz, v.zLen, tm, c, T, T + n1, T + 2 * n1, T + 3 * n1,
                                    v.sensEu + sp * n1, v.sensEd + sp * n1, v.sensMix + sp * 4 * v.nz,
                                    v.sensDz1 + sp * v.nz, v.sensZ1[sp], v.sensDead[sp], w, 1);
    if (prof == 0) v.gL[o] = g; else if (prof == 1) v.gR[o] = g; else v.gMn[o] = g;
}

// The same in two steps, so that the serial recurrences are off the critical path (they depend on sigma only):
//   item_bcsens_pre       side stream, beside the solves: the w-independent entries dF(row, c) of the left / right
//                         profile into v.dBC (the mean profile needs its last row only: v.gMn directly)
//   item_bcsens_contract  after the adjoint solve: g[c] = sum over rows of dF(row, c) w[row], rows in the order of
//                         item_bcsens' accumulation (the rows beyond the cut-off hold zeros)
HD void item_bcsens_pre(const View& v, int s, int prof, int c) {
    const long o = (long)s * v.nz + c;
    if (!v.sysOn[s]) { if (prof == 2) v.gMn[o] = cplx{0, 0}; return; }
    const bool tm = s >= v.nFreq;
    const long n1 = v.nz + 1, sp = (long)s * 3 + prof;
    const cplx* T = v.sensTab + sp * 5 * n1;
    cplx* out = prof < 2 ? v.dBC + ((long)s * 2 + prof) * v.nz * v.nz + c : nullptr;
    const cplx g = bc1d_sens_column(v.omega[s], v.nz, v.zLen, tm, c, T, T + n1, T + 2 * n1, T + 3 * n1,
                                    v.sensEu + sp * n1, v.sensEd + sp * n1, v.sensMix + sp * 4 * v.nz,
                                    v.sensDz1 + sp * v.nz, v.sensZ1[sp], v.sensDead[sp], nullptr, 1, out, v.nz);
    if (prof == 2) v.gMn[o] = g;
}
HD void item_bcsens_contract(const View& v, int s, int prof, int c) {     // prof 0 / 1
    const long o = (long)s * v.nz + c;
    cplx acc = cplx{0.0, 0.0};
    if (v.sysOn[s]) {
        const cplx* D = v.dBC + ((long)s * 2 + prof) * v.nz * v.nz + c;
        const cplx* w = (prof == 0 ? v.wL : v.wR) + (long)s * v.nz;
        // (batches of 8 rows requested together; the additions stay in row order)
        int j = 0;
        for (; j + 8 <= v.nz; j += 8) {
            cplx d[8], ww[8];
            HMCMT_UNROLL
            for (int t = 0; t < 8; ++t) { d[t] = D[(long)(j + t) * v.nz]; ww[t] = w[j + t]; }
            HMCMT_UNROLL
            for (int t = 0; t < 8; ++t) acc += d[t] * ww[t];
        }
        for (; j < v.nz; ++j) acc += D[(long)j * v.nz] * w[j];
    }
    if (prof == 0) v.gL[o] = acc; else v.gR[o] = acc;
}

// --- TM field with the sensitivity-version boundary values (compJacTMatVec.jl:307,315):
//     h~ = forward solution on interior nodes, getBCderivTM's bc on boundary nodes
HD cplx tm_field_sens(const View& v, int s, int iy, int iz) {
    if (iz == 0) return cplx{1.0, 0.0};
    if (iy == 0) return v.bcsL[(long)s * v.nz + iz - 1];
    if (iy == v.ny) return v.bcsR[(long)s * v.nz + iz - 1];
    if (iz == v.nz) return v.bcsB[s];
    return v.X[(long)s * v.vstride + nidx(v, iy, iz)];
}

// --- P-terms of J^T v for one cell, summed over the frequencies of one mode
HD double gradcell_freqs(const View& v, int mode, int cell, int f0, int f1) {
    const int ky = cell % v.ny, kz = cell / v.ny;
    const double area = v.yLen[ky] * v.zLen[kz];
    double acc = 0.0;
#pragma unroll 4
    for (int f = f0; f < f1; ++f) {
        const int s = mode * v.nFreq + f;
        if (!v.sysOn[s]) continue;
        const cplx* L = v.Lam + (long)s * v.vstride;
        if (mode == 0) {
            // -i w (1/4 area) sum over the 4 corner nodes e*lambda (compJacTMatVec.jl:83,235)
            const cplx* E = v.X + (long)s * v.vstride;
            cplx sum = cplx{0, 0};
            for (int dz = 0; dz < 2; ++dz)
                for (int dy = 0; dy < 2; ++dy) {
                    long n = nidx(v, ky + dy, kz + dz);
                    sum += E[n] * L[n];
                }
            // Re[-i w a sum] = w a Im(sum)
            acc += v.omega[s] * 0.25 * area * sum.im;
        } else {
            // (area/sig^2) * 1/2 * sum over the 4 edges (grad h~)(grad lambda)
            // (compJacTMatVec.jl:97-99,306-307,314-315); mesh-boundary edges have grad lambda = 0
            cplx h00 = tm_field_sens(v, s, ky, kz), h10 = tm_field_sens(v, s, ky + 1, kz);
            cplx h01 = tm_field_sens(v, s, ky, kz + 1), h11 = tm_field_sens(v, s, ky + 1, kz + 1);
            cplx l00 = L[nidx(v, ky, kz)], l10 = L[nidx(v, ky + 1, kz)];
            cplx l01 = L[nidx(v, ky, kz + 1)], l11 = L[nidx(v, ky + 1, kz + 1)];
            const double iy2 = 1.0 / (v.yLen[ky] * v.yLen[ky]), iz2 = 1.0 / (v.zLen[kz] * v.zLen[kz]);
            cplx sum = ((h10 - h00) * (l10 - l00) + (h11 - h01) * (l11 - l01)) * iy2 +
                       ((h01 - h00) * (l01 - l00) + (h11 - h10) * (l11 - l10)) * iz2;
            const double sg = v.sigma[cell];
            acc += 0.5 * area / (sg * sg) * sum.re;
        }
    }
    return acc;
}
HD void item_gradcell(const View& v, int mode, int cell) {
    v.gPart[(long)mode * v.nCell + cell] = gradcell_freqs(v, mode, cell, 0, v.nFreq);
}
// the same over one of GRAD_NG groups of frequencies (the GPU's launch: 4x the threads of a latency-bound kernel);
// partial sums [mode][group][cell], added up by the final assembly
constexpr int GRAD_NG = 4;
HD void item_gradcell_group(const View& v, int mode, int grp, int cell) {
    const int per = (v.nFreq + GRAD_NG - 1) / GRAD_NG, f0 = grp * per, f1 = f0 + per < v.nFreq ? f0 + per : v.nFreq;
    v.gPartG[((long)mode * GRAD_NG + grp) * v.nCell + cell] = gradcell_freqs(v, mode, cell, f0, f1);
}

// --- Q-term of one system for one receiver-layer cell: Re sum_r conj(v_r) dZ_r/dsigma_c
//     (compJacTMatVec.jl:209, :280)
HD void item_qterm(const View& v, int s, int ky) {
    double g = 0.0;
    if (v.sysOn[s])
        for (int r = 0; r < v.nRx; ++r) {
            const long k = (long)s * v.nRx + r;
            const int o = ky - v.rxN0[k];
            if (o >= 0 && o < 3) g += (v.rxCoef[k] * v.rxD[k * 11 + 8 + o]).re;
        }
    v.qPart[(long)s * v.ny + ky] = g;
}

// --- final assembly of the gradient w.r.t. m = ln(sigma) for one active cell:
//     P-terms + boundary terms + Q-terms, real part, chain rule (compJacTMatVec.jl:244,318,325-327;
//     HMCSampler.jl:306)
// boundary term of one system for one cell (branch-free select, not `continue`, so that the loads of several systems
// are in flight)
HD double gradfinal_sys(const View& v, int s, int ky, int kz) {
    if (v.dbg & 2) return 0.0;
    const long o = (long)s * v.nz + kz;
    cplx b = v.gMn[o] * v.colw[(long)s * v.ny + ky];
    if (ky == 0) b += v.gL[o];
    if (ky == v.ny - 1) b += v.gR[o];
    return v.sysOn[s] ? b.re : 0.0;
}
HD void item_gradfinal(const View& v, int a) {
    const int cell = v.act[a];
    const int ky = cell % v.ny, kz = cell / v.ny;
    double g = v.gPart[cell] + v.gPart[(long)v.nCell + cell];
    // branch-free over the systems (select, not `continue`) so that the loads of several systems are in flight
#pragma unroll 8
    for (int s = 0; s < v.S; ++s) g += gradfinal_sys(v, s, ky, kz);
    if (kz == v.zid)
        for (int s = 0; s < v.S; ++s) g += v.qPart[(long)s * v.ny + ky];
    v.grad[a] = exp(v.m[a]) * g;
}

}  // namespace hmcmt
