// Host-side, once-per-run preparation of the constant tables the kernels need: index maps of the
// data vector, receiver interpolation tables, and the generalised eigen-decomposition of the
// y-direction operator used by the fast-diagonalisation preconditioner.  Plain C++ (no HIP).
#pragma once
#include <cmath>
#include <cstdint>
#include <string>
#include <vector>
#include <algorithm>
#include "hmcmt_math.h"

namespace hmcmt {

struct HostProblem {
    int ny = 0, nz = 0, NYP = 0, NZP = 0, nFreq = 0, S = 0, nRx = 0, nData = 0, nAC = 0, nCell = 0, zid = 0;
    bool compTE = false, compTM = false, rhoPhase = false;
    std::vector<double> yLen, zLen, omega, lam, Vpad, Vtpad, bg, dataW;
    std::vector<double> rxDy1, rxDy2, rxWL, rxWR;
    std::vector<int> cell2act, act, rxIdn, rxKL, rxKR, predSys, predRx, predKind, datSys, datRx, datKind, srStart, srList, sysOn;
    std::vector<cplx> obs;
    std::string error;

    // sensUtils.jl:133-161 (0-based)
    static void linearInterp(double point, const std::vector<double>& x, int& indL, int& indR, double& wL, double& wR) {
        int n = (int)x.size(), ind = 0;
        double best = std::fabs(point - x[0]);
        for (int i = 1; i < n; ++i) {
            double d = std::fabs(point - x[i]);
            if (d < best) { best = d; ind = i; }
        }
        if (point - x[ind] > 0) { indL = ind; indR = ind + 1; }
        else { indL = ind - 1; indR = ind; }
        indL = std::max(std::min(indL, n - 1), 0);
        indR = std::max(std::min(indR, n - 1), 0);
        if (indL == indR) { wL = 0.5; wR = 0.5; return; }
        double xLen = x[indR] - x[indL];
        wL = 1 - (point - x[indL]) / xLen;
        wR = 1 - (x[indR] - point) / xLen;
    }

    // Symmetric tridiagonal eigenproblem by implicit QL with eigenvector accumulation
    // (d: diagonal -> eigenvalues, e: sub-diagonal e[0..n-2], z: n x n row-major, identity on entry
    // -> column k is the k-th eigenvector).
    static bool tqli(std::vector<double>& d, std::vector<double>& e, int n, std::vector<double>& z) {
        e.resize(n);
        e[n - 1] = 0.0;
        for (int l = 0; l < n; ++l) {
            int iter = 0, m;
            do {
                for (m = l; m < n - 1; ++m) {
                    double dd = std::fabs(d[m]) + std::fabs(d[m + 1]);
                    if (std::fabs(e[m]) <= 2.3e-16 * dd) break;
                }
                if (m != l) {
                    if (iter++ == 200) return false;
                    double g = (d[l + 1] - d[l]) / (2.0 * e[l]);
                    double r = std::hypot(g, 1.0);
                    g = d[m] - d[l] + e[l] / (g + (g >= 0 ? std::fabs(r) : -std::fabs(r)));
                    double s = 1.0, c = 1.0, p = 0.0;
                    int i;
                    for (i = m - 1; i >= l; --i) {
                        double f = s * e[i], b = c * e[i];
                        e[i + 1] = (r = std::hypot(f, g));
                        if (r == 0.0) { d[i + 1] -= p; e[m] = 0.0; break; }
                        s = f / r; c = g / r;
                        g = d[i + 1] - p;
                        r = (d[i] - g) * s + 2.0 * c * b;
                        d[i + 1] = g + (p = s * r);
                        g = c * r - b;
                        for (int k = 0; k < n; ++k) {
                            f = z[(size_t)k * n + i + 1];
                            z[(size_t)k * n + i + 1] = s * z[(size_t)k * n + i] + c * f;
                            z[(size_t)k * n + i] = c * z[(size_t)k * n + i] - s * f;
                        }
                    }
                    if (r == 0.0 && i >= l) continue;
                    d[l] -= p; e[l] = g; e[m] = 0.0;
                }
            } while (m != l);
        }
        return true;
    }

    // Ty v = lam My v on the interior y-nodes; Vpad[iy][j] (NYP x NYP, zero outside iy=1..ny-1,
    // j=0..ny-2) with V' My V = I; Vtpad is its transpose.
    bool build_fdm() {
        const int n = ny - 1;
        std::vector<double> my(n), d(n), e(n, 0.0), z((size_t)n * n, 0.0);
        for (int i = 0; i < n; ++i) {
            double ya = yLen[i], yb = yLen[i + 1];
            my[i] = 0.5 * (ya + yb);
            d[i] = (1.0 / ya + 1.0 / yb) / my[i];
            z[(size_t)i * n + i] = 1.0;
        }
        for (int i = 0; i < n - 1; ++i) e[i] = -(1.0 / yLen[i + 1]) / std::sqrt(my[i] * my[i + 1]);
        if (!tqli(d, e, n, z)) { error = "FDM eigen-decomposition did not converge"; return false; }
        lam.assign(NYP, 0.0);
        Vpad.assign((size_t)NYP * NYP, 0.0);
        Vtpad.assign((size_t)NYP * NYP, 0.0);
        for (int j = 0; j < n; ++j) lam[j] = d[j];
        for (int i = 0; i < n; ++i)
            for (int j = 0; j < n; ++j) {
                double v = z[(size_t)i * n + j] / std::sqrt(my[i]);
                Vpad[(size_t)(i + 1) * NYP + j] = v;
                Vtpad[(size_t)j * NYP + (i + 1)] = v;
            }
        return true;
    }

    bool build(int64_t ny_, int64_t nz_, const double* yLen_, const double* zLen_, const double* origin,
               int64_t nFreq_, const double* freqs, int64_t nRx_, const double* rxY, const double* rxZ,
               int64_t nComp, const int64_t* compMode, int64_t nData_, const int64_t* freqID,
               const int64_t* rxID, const int64_t* dtID, const uint8_t* dataID, const double* obs_,
               const double* dataW_, int64_t nAC_, const int64_t* activeIdx, const double* bgModel) {
        if (ny_ < 3 || nz_ < 3 || nFreq_ < 1 || nRx_ < 1 || nData_ < 0 || nAC_ < 1 || nComp < 1) {
            error = "invalid sizes"; return false;
        }
        ny = (int)ny_; nz = (int)nz_; nFreq = (int)nFreq_; nRx = (int)nRx_; nData = (int)nData_; nAC = (int)nAC_;
        S = 2 * nFreq; nCell = ny * nz; NZP = nz + 1; NYP = ((ny + 1 + 15) / 16) * 16;
        yLen.assign(yLen_, yLen_ + ny); zLen.assign(zLen_, zLen_ + nz);
        for (double v : yLen) if (!(v > 0)) { error = "non-positive yLen"; return false; }
        for (double v : zLen) if (!(v > 0)) { error = "non-positive zLen"; return false; }
        omega.resize(S);
        for (int f = 0; f < nFreq; ++f) {
            if (!(freqs[f] > 0)) { error = "non-positive frequency"; return false; }
            omega[f] = omega[nFreq + f] = 2 * 3.14159265358979323846 * freqs[f];   // mt2DTE.jl:37
        }
        // node coordinates (mt2DTE.jl:31-32)
        std::vector<double> yNode(ny + 1), zNode(nz + 1);
        {
            double a = 0; yNode[0] = -origin[0];
            for (int i = 0; i < ny; ++i) { a += yLen[i]; yNode[i + 1] = a - origin[0]; }
            a = 0; zNode[0] = -origin[1];
            for (int i = 0; i < nz; ++i) { a += zLen[i]; zNode[i + 1] = a - origin[1]; }
        }
        // receiver row: first node with |zNode - zRx| < 0.1 (mt2DTE.jl:66-67)
        zid = -1;
        for (int i = 0; i <= nz; ++i) if (std::fabs(zNode[i] - rxZ[0]) < 0.1) { zid = i; break; }
        if (zid < 0 || zid + 1 > nz || zid >= nz) { error = "receiver depth does not coincide with a grid node"; return false; }
        rxIdn.resize(nRx); rxDy1.resize(nRx); rxDy2.resize(nRx);
        rxKL.resize(nRx); rxKR.resize(nRx); rxWL.resize(nRx); rxWR.resize(nRx);
        for (int r = 0; r < nRx; ++r) {
            int id = -1;
            for (int i = 0; i <= ny; ++i) if (yNode[i] > rxY[r]) { id = i; break; }   // mt2DTE.jl:198
            if (id < 1) { error = "The receiver location seems to be out of range!"; return false; }
            rxIdn[r] = id; rxDy1[r] = rxY[r] - yNode[id - 1]; rxDy2[r] = yNode[id] - rxY[r];
            linearInterp(rxY[r], yNode, rxKL[r], rxKR[r], rxWL[r], rxWR[r]);
        }
        // modes
        // component codes: 1 ZXY, 2 ZYX (DataType Impedance, complex data); 3 RhoXY, 4 PhsXY, 5 RhoYX, 6 PhsYX
        // (DataType Rho_Pha, real data: apparent resistivity |Z|^2/(w mu0) and phase in degrees, mt2DTE.jl:253-255)
        std::vector<int> cmode(nComp), ckind(nComp);
        bool anyZ = false, anyRP = false;
        for (int c = 0; c < nComp; ++c) {
            const int code = (int)compMode[c];
            if (code < 1 || code > 6) { error = "compMode must be 1 ZXY, 2 ZYX, 3 RhoXY, 4 PhsXY, 5 RhoYX or 6 PhsYX"; return false; }
            cmode[c] = (code == 1 || code == 3 || code == 4) ? 1 : 2;
            ckind[c] = code <= 2 ? 0 : ((code == 3 || code == 5) ? 1 : 2);
            if (code <= 2) anyZ = true; else anyRP = true;
            if (cmode[c] == 1) compTE = true; else compTM = true;
        }
        if (anyZ && anyRP) { error = "impedance and rho/phase components cannot be mixed in one data set"; return false; }
        rhoPhase = anyRP;
        // full response table per (freq, rx): Impedance [Z_TE][Z_TM], Rho_Pha [rho_TE, phs_TE][rho_TM, phs_TM], for the
        // modes present (MT2DFwdSolver.jl:175-205)
        const int per = ((compTE ? 1 : 0) + (compTM ? 1 : 0)) * (rhoPhase ? 2 : 1);
        const int64_t nMask = (int64_t)nComp * nRx * nFreq;
        if ((int64_t)per * nRx * nFreq != nMask) { error = "dataID length does not match the response table"; return false; }
        predSys.clear(); predRx.clear(); predKind.clear();
        for (int64_t q = 0; q < nMask; ++q) {
            if (!dataID[q]) continue;
            int c = (int)(q % per), r = (int)((q / per) % nRx), f = (int)(q / ((int64_t)per * nRx));
            const int cm = rhoPhase ? c / 2 : c;                       // which of the modes present
            bool tm = compTE ? (cm == 1) : true;
            predSys.push_back((tm ? nFreq : 0) + f);
            predRx.push_back(r);
            predKind.push_back(rhoPhase ? 1 + (c & 1) : 0);
        }
        if ((int)predSys.size() != nData) { error = "dataID selects a different number of entries than nData"; return false; }
        datSys.resize(nData); datRx.resize(nData); datKind.resize(nData);
        obs.resize(nData); dataW.assign(dataW_, dataW_ + nData);
        std::vector<std::vector<int>> lists((size_t)S * nRx);
        for (int p = 0; p < nData; ++p) {
            int f = (int)freqID[p] - 1, r = (int)rxID[p] - 1, c = (int)dtID[p] - 1;
            if (f < 0 || f >= nFreq || r < 0 || r >= nRx || c < 0 || c >= nComp) { error = "data index out of range"; return false; }
            datSys[p] = (cmode[c] == 2 ? nFreq : 0) + f;
            datRx[p] = r;
            datKind[p] = ckind[c];
            obs[p] = cplx{obs_[2 * p], obs_[2 * p + 1]};
            lists[(size_t)datSys[p] * nRx + r].push_back(p);
        }
        srStart.assign((size_t)S * nRx + 1, 0); srList.clear();
        for (size_t k = 0; k < lists.size(); ++k) {
            srStart[k] = (int)srList.size();
            for (int p : lists[k]) srList.push_back(p);
        }
        srStart[lists.size()] = (int)srList.size();
        // active cells
        cell2act.assign(nCell, -1); act.resize(nAC);
        for (int a = 0; a < nAC; ++a) {
            int64_t c = activeIdx[a] - 1;
            if (c < 0 || c >= nCell || cell2act[c] >= 0) { error = "activeIdx out of range or duplicated"; return false; }
            cell2act[c] = a; act[a] = (int)c;
        }
        bg.assign(bgModel, bgModel + nCell);
        sysOn.resize(S);
        for (int s = 0; s < S; ++s) sysOn[s] = (s < nFreq) ? (compTE ? 1 : 0) : (compTM ? 1 : 0);
        return build_fdm();
    }
};

}  // namespace hmcmt
