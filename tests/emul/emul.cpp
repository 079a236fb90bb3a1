// TEST-ONLY host instantiation of the per-item kernel bodies (hmcmt_items.h / hmcmt_math.h) with
// a plain serial driver.  It exists so the kernel arithmetic can be unit-tested against the oracle
// in the GPU-less build container (`pytest -m "not gpu"`).  It is NOT part of the product: the
// package never loads it and libhmcmt_hip.so has no host compute path.
#include <cstdio>
#include <cstring>
#include <vector>
#include "../../hmcmt2d_amd/csrc/hmcmt_host.h"
#include "../../hmcmt2d_amd/csrc/hmcmt_items.h"

using namespace hmcmt;

namespace {

struct Emul {
    HostProblem hp;
    View v;
    std::vector<double> sigma, sigMeanA, sigMeanG, cY, cZ, dK, dM, mzq, dgz, ofz, mzs, misfitPart, gPart, grad, m;
    std::vector<cplx> invp, X, Lam, R, Zrx, rxD, rxCoef, pred, vbar, srcB, wL, wR, colw, gL, gR, gMn, bcsL, bcsR, bcsB;
    std::vector<int> rxN0, sensDead;
    std::vector<double> qPart;
    std::vector<cplx> fwdTab, sensTab, sensEu, sensEd, sensMix, sensDz1, sensZ1;
    std::vector<int> iters;

    void bind() {
        const HostProblem& h = hp;
        v.ny = h.ny; v.nz = h.nz; v.NYP = h.NYP; v.NZP = h.NZP; v.nFreq = h.nFreq; v.S = h.S; v.nRx = h.nRx;
        v.nData = h.nData; v.nAC = h.nAC; v.nCell = h.nCell; v.zid = h.zid; v.dbg = 0; v.vstride = (long)h.NZP * h.NYP;
        const size_t VS = (size_t)v.vstride;
        sigma.assign(h.nCell, 0); sigMeanA.assign(h.nz, 0); sigMeanG.assign(h.nz, 0);
        cY.assign(2 * VS, 0); cZ.assign(2 * VS, 0); dK.assign(2 * VS, 0); dM.assign(2 * VS, 0);
        mzq.assign(2 * h.NZP, 0); dgz.assign(2 * h.NZP, 0); ofz.assign(2 * h.NZP, 0); mzs.assign(2 * h.NZP, 0);
        invp.assign(h.S * VS, cplx{0, 0}); X.assign(h.S * VS, cplx{0, 0}); Lam.assign(h.S * VS, cplx{0, 0});
        R.assign(h.S * VS, cplx{0, 0});
        Zrx.assign((size_t)h.S * h.nRx, cplx{0, 0}); rxN0.assign((size_t)h.S * h.nRx, 0);
        rxD.assign((size_t)h.S * h.nRx * 11, cplx{0, 0}); rxCoef.assign((size_t)h.S * h.nRx, cplx{0, 0});
        pred.assign(h.nData, cplx{0, 0}); vbar.assign(h.nData, cplx{0, 0}); misfitPart.assign(h.nData, 0);
        srcB.assign((size_t)h.S * 4, cplx{0, 0});
        wL.assign((size_t)h.S * h.nz, cplx{0, 0}); wR = wL; gL = wL; gR = wL; gMn = wL; bcsL = wL; bcsR = wL;
        colw.assign((size_t)h.S * h.ny, cplx{0, 0}); bcsB.assign(h.S, cplx{0, 0});
        gPart.assign((size_t)2 * h.nCell, 0); grad.assign(h.nAC, 0); m.assign(h.nAC, 0);
        fwdTab.assign((size_t)h.S * hmcmt::FWD_NQ * h.nz * (h.ny + 1), cplx{0, 0}); sensTab.assign((size_t)h.S * 15 * (h.nz + 1), cplx{0, 0});
        sensEu.assign((size_t)h.S * 3 * (h.nz + 1), cplx{0, 0}); sensEd = sensEu; sensMix.assign((size_t)h.S * 12 * h.nz, cplx{0, 0});
        sensDz1.assign((size_t)h.S * 3 * h.nz, cplx{0, 0}); sensZ1.assign((size_t)h.S * 3, cplx{0, 0}); sensDead.assign((size_t)h.S * 3, 0);
        iters.assign(2 * h.S, 0);
        v.yLen = h.yLen.data(); v.zLen = h.zLen.data(); v.omega = h.omega.data(); v.lam = h.lam.data(); v.sysOn = h.sysOn.data();
        v.m = m.data(); v.sigma = sigma.data(); v.cell2act = h.cell2act.data(); v.bg = h.bg.data(); v.act = h.act.data();
        v.sigMeanA = sigMeanA.data(); v.sigMeanG = sigMeanG.data();
        v.cY = cY.data(); v.cZ = cZ.data(); v.dK = dK.data(); v.dM = dM.data();
        v.mzq = mzq.data(); v.dgz = dgz.data(); v.ofz = ofz.data(); v.mzs = mzs.data(); v.invp = invp.data();
        v.X = X.data(); v.Lam = Lam.data(); v.R = R.data();
        v.rxIdn = h.rxIdn.data(); v.rxDy1 = h.rxDy1.data(); v.rxDy2 = h.rxDy2.data();
        v.rxKL = h.rxKL.data(); v.rxKR = h.rxKR.data(); v.rxWL = h.rxWL.data(); v.rxWR = h.rxWR.data();
        v.Zrx = Zrx.data(); v.rxN0 = rxN0.data(); v.rxD = rxD.data(); v.rxCoef = rxCoef.data();
        v.predSys = h.predSys.data(); v.predRx = h.predRx.data(); v.datSys = h.datSys.data(); v.datRx = h.datRx.data(); v.predKind = h.predKind.data(); v.datKind = h.datKind.data();
        v.obs = h.obs.data(); v.dataW = h.dataW.data(); v.pred = pred.data(); v.vbar = vbar.data();
        v.misfitPart = misfitPart.data(); v.srStart = h.srStart.data(); v.srList = h.srList.data();
        v.srcB = srcB.data(); v.wL = wL.data(); v.wR = wR.data(); v.colw = colw.data();
        v.gL = gL.data(); v.gR = gR.data(); v.gMn = gMn.data(); v.bcsL = bcsL.data(); v.bcsR = bcsR.data();
        v.bcsB = bcsB.data(); v.gPart = gPart.data(); v.grad = grad.data();
        qPart.assign((size_t)h.S * h.ny, 0); v.qPart = qPart.data();
        v.fwdTab = fwdTab.data(); v.sensTab = sensTab.data(); v.sensEu = sensEu.data(); v.sensEd = sensEd.data();
        v.sensMix = sensMix.data(); v.sensDz1 = sensDz1.data(); v.sensZ1 = sensZ1.data(); v.sensDead = sensDead.data();
        // constant halves of the stencils: TE stiffness, TM mass
        for (int iz = 0; iz < v.NZP; ++iz)
            for (int iy = 0; iy <= v.ny; ++iy) { item_coef(v, 0, iy, iz, true, false); item_coef(v, 1, iy, iz, false, true); }
    }

    bool interior(int iy, int iz) const { return iy >= 1 && iy <= v.ny - 1 && iz >= 1 && iz <= v.nz - 1; }

    // z = P^-1 r for system s (kind 2: Jacobi/FDM/Jacobi product form, 1: FDM, 0: Jacobi)
    void precond(int s, int kind, const cplx* r, cplx* z) {
        const int NYP = v.NYP, nz = v.nz;
        if (kind == 2) {
            const int mode = s >= v.nFreq;
            const double wJ = 0.8;
            std::vector<cplx> z0((size_t)v.vstride, cplx{0, 0}), t((size_t)v.vstride, cplx{0, 0}), z1((size_t)v.vstride, cplx{0, 0});
            auto dinv = [&](int iy, int iz) { long mo = (long)mode * v.vstride + nidx(v, iy, iz); return wJ / cplx{v.dK[mo], v.omega[s] * v.dM[mo]}; };
            for (int iz = 1; iz <= nz - 1; ++iz) for (int iy = 1; iy <= v.ny - 1; ++iy) z0[nidx(v, iy, iz)] = dinv(iy, iz) * r[nidx(v, iy, iz)];
            for (int iz = 1; iz <= nz - 1; ++iz) for (int iy = 1; iy <= v.ny - 1; ++iy) t[nidx(v, iy, iz)] = r[nidx(v, iy, iz)] - stencil_apply(v, s, z0.data(), iy, iz);
            precond(s, 1, t.data(), z1.data());
            for (long o = 0; o < v.vstride; ++o) z1[o] += z0[o];
            for (int iz = 1; iz <= nz - 1; ++iz) for (int iy = 1; iy <= v.ny - 1; ++iy) {
                long o = nidx(v, iy, iz);
                z[o] = z1[o] + dinv(iy, iz) * (r[o] - stencil_apply(v, s, z1.data(), iy, iz));
            }
            return;
        }
        if (kind == 0) {
            const int mode = s >= v.nFreq;
            for (int iz = 1; iz <= nz - 1; ++iz)
                for (int iy = 1; iy <= v.ny - 1; ++iy) {
                    long o = nidx(v, iy, iz), mo = (long)mode * v.vstride + o;
                    z[o] = r[o] / cplx{v.dK[mo], v.omega[s] * v.dM[mo]};
                }
            return;
        }
        std::vector<cplx> Y((size_t)v.vstride, cplx{0, 0});
        for (int iz = 1; iz <= nz - 1; ++iz)
            for (int j = 0; j < v.ny - 1; ++j) {
                cplx a = cplx{0, 0};
                for (int iy = 1; iy <= v.ny - 1; ++iy) a += hp.Vpad[(size_t)iy * NYP + j] * r[nidx(v, iy, iz)];
                Y[nidx(v, j, iz)] = a;
            }
        const int mode = s >= v.nFreq;
        const double* ofz_ = v.ofz + (long)mode * v.NZP;
        const cplx* ip = v.invp + (long)s * v.vstride;
        for (int j = 0; j < v.ny - 1; ++j) {
            // twisted solve (see item_pivot): top-down to mid, bottom-up to mid+1, 2x2 in the middle, outwards
            const int n = nz - 1, mid = twist_mid(n, v.twist);
            auto Yr = [&](int iz) -> cplx& { return Y[nidx(v, j, iz)]; };
            cplx pt = cplx{0, 0}, pb = cplx{0, 0};
            for (int iz = 1; iz <= mid; ++iz) { pt = (Yr(iz) - ofz_[iz - 1] * pt) * ip[nidx(v, j, iz)]; Yr(iz) = pt; }
            for (int iz = n; iz >= mid + 1; --iz) { pb = (Yr(iz) - ofz_[iz] * pb) * ip[nidx(v, j, iz)]; Yr(iz) = pb; }
            if (mid + 1 <= n) {
                const cplx c = ofz_[mid] * ip[nidx(v, j, mid)], c2 = ofz_[mid] * ip[nidx(v, j, mid + 1)];
                pt = (pt - c * pb) * ip[nidx(v, j, 0)];
                pb = pb - c2 * pt;
                Yr(mid) = pt; Yr(mid + 1) = pb;
            }
            for (int iz = mid - 1; iz >= 1; --iz) { pt = Yr(iz) - (ofz_[iz] * ip[nidx(v, j, iz)]) * pt; Yr(iz) = pt; }
            for (int iz = mid + 2; iz <= n; ++iz) { pb = Yr(iz) - (ofz_[iz - 1] * ip[nidx(v, j, iz)]) * pb; Yr(iz) = pb; }
        }
        for (int iz = 1; iz <= nz - 1; ++iz)
            for (int iy = 1; iy <= v.ny - 1; ++iy) {
                cplx a = cplx{0, 0};
                for (int j = 0; j < v.ny - 1; ++j) a += hp.Vtpad[(size_t)j * NYP + iy] * Y[nidx(v, j, iz)];
                z[nidx(v, iy, iz)] = a;
            }
    }

    // COCG on system s: solves A x = R[s] into xout (interior only, boundary untouched)
    int cocg(int s, cplx* xout, int kind, double tol, int maxit) {
        const long VS = v.vstride;
        cplx* r = v.R + (long)s * VS;
        std::vector<cplx> x(VS, cplx{0, 0}), z(VS, cplx{0, 0}), p(VS, cplx{0, 0}), q(VS, cplx{0, 0});
        double bb = 0;
        for (int iz = 1; iz <= v.nz - 1; ++iz) for (int iy = 1; iy <= v.ny - 1; ++iy) bb += cabs2(r[nidx(v, iy, iz)]);
        int it = 0;
        if (bb > 0) {
            precond(s, kind, r, z.data());
            p = z;
            cplx rho = cplx{0, 0};
            for (long o = 0; o < VS; ++o) rho += r[o] * z[o];
            for (it = 1; it <= maxit; ++it) {
                cplx pq = cplx{0, 0};
                for (int iz = 1; iz <= v.nz - 1; ++iz)
                    for (int iy = 1; iy <= v.ny - 1; ++iy) {
                        long o = nidx(v, iy, iz);
                        q[o] = stencil_apply(v, s, p.data(), iy, iz);
                        pq += p[o] * q[o];
                    }
                cplx al = rho / pq;
                double xx = 0;
                for (int iz = 1; iz <= v.nz - 1; ++iz)
                    for (int iy = 1; iy <= v.ny - 1; ++iy) {
                        long o = nidx(v, iy, iz);
                        x[o] += al * p[o];
                        r[o] -= al * q[o];
                        xx += cabs2(x[o]);
                    }
                // z = P^-1 r approximates the error A^-1 r: stop on ||z|| <= tol ||x||
                precond(s, kind, r, z.data());
                cplx rho1 = cplx{0, 0};
                double zz = 0;
                for (long o = 0; o < VS; ++o) { rho1 += r[o] * z[o]; zz += cabs2(z[o]); }
                if (zz <= tol * tol * xx) break;
                cplx be = rho1 / rho;
                rho = rho1;
                for (long o = 0; o < VS; ++o) p[o] = z[o] + be * p[o];
            }
        }
        for (int iz = 1; iz <= v.nz - 1; ++iz)
            for (int iy = 1; iy <= v.ny - 1; ++iy) xout[nidx(v, iy, iz)] = x[nidx(v, iy, iz)];
        return it;
    }

    void run(const double* m_in, bool wantGrad, int kind, double tol, int maxit, double* misfit) {
        const View& V = v;
        std::memcpy(m.data(), m_in, sizeof(double) * V.nAC);
        for (int c = 0; c < V.nCell; ++c) item_sigma(V, c);
        for (int kz = 0; kz < V.nz; ++kz) item_rowmean(V, kz);
        for (int iz = 0; iz < V.NZP; ++iz)
            for (int iy = 0; iy <= V.ny; ++iy) { item_coef(V, 0, iy, iz, false, true); item_coef(V, 1, iy, iz, true, false); }
        for (int mode = 0; mode < 2; ++mode) for (int iz = 0; iz < V.NZP; ++iz) item_fdm_z(V, mode, iz);
        for (int s = 0; s < V.S; ++s) for (int j = 0; j < V.ny - 1; ++j) item_pivot(V, s, j);
        std::fill(X.begin(), X.end(), cplx{0, 0});
        for (int s = 0; s < V.S; ++s) for (int j = 0; j < V.nz; ++j) for (int col = 0; col <= V.ny; ++col) item_bc_layers(V, s, j, col);
        for (int s = 0; s < V.S; ++s) for (int col = 0; col <= V.ny; ++col) item_bc_forward(V, s, col);
        for (int s = 0; s < V.S; ++s)
            for (int iz = 0; iz < V.NZP; ++iz) for (int iy = 0; iy <= V.ny; ++iy) item_rhs(V, s, iy, iz);
        for (int s = 0; s < V.S; ++s) iters[s] = V.sysOn[s] ? cocg(s, X.data() + (long)s * V.vstride, kind, tol, maxit) : 0;
        for (int s = 0; s < V.S; ++s) for (int r = 0; r < V.nRx; ++r) item_rx(V, s, r, wantGrad);
        for (int p = 0; p < V.nData; ++p) item_resid(V, p);
        double mf = 0;
        for (int p = 0; p < V.nData; ++p) mf += misfitPart[p];
        *misfit = mf;
        if (!wantGrad) return;
        for (int s = 0; s < V.S; ++s) for (int r = 0; r < V.nRx; ++r) item_rxcoef(V, s, r);
        std::fill(R.begin(), R.end(), cplx{0, 0});
        std::fill(srcB.begin(), srcB.end(), cplx{0, 0});
        for (int s = 0; s < V.S; ++s) for (int row = 0; row < 2; ++row) for (int iy = 0; iy <= V.ny; ++iy) item_src(V, s, row, iy);
        std::fill(Lam.begin(), Lam.end(), cplx{0, 0});
        for (int s = 0; s < V.S; ++s) iters[V.S + s] = V.sysOn[s] ? cocg(s, Lam.data() + (long)s * V.vstride, kind, tol, maxit) : 0;
        for (int s = 0; s < V.S; ++s) {
            for (int iz = 1; iz <= V.nz; ++iz) item_wside(V, s, iz);
            for (int ky = 0; ky < V.ny; ++ky) item_colw(V, s, ky);
        }
        for (int s = 0; s < V.S; ++s) for (int prof = 0; prof < 3; ++prof) {
            for (int j = 0; j <= V.nz; ++j) item_sens_layers(V, s, prof, j);
            item_sens_profile(V, s, prof);
            for (int c = 0; c < V.nz; ++c) item_bcsens(V, s, prof, c);
        }
        for (int mode = 0; mode < 2; ++mode) for (int c = 0; c < V.nCell; ++c) item_gradcell(V, mode, c);
        for (int s = 0; s < V.S; ++s) for (int ky = 0; ky < V.ny; ++ky) item_qterm(V, s, ky);
        for (int a = 0; a < V.nAC; ++a) item_gradfinal(V, a);
    }
};

}  // namespace

extern "C" {

void* emul_create(int64_t ny, int64_t nz, const double* yLen, const double* zLen, const double* origin,
                  int64_t nFreq, const double* freqs, int64_t nRx, const double* rxY, const double* rxZ,
                  int64_t nComp, const int64_t* compMode, int64_t nData, const int64_t* freqID,
                  const int64_t* rxID, const int64_t* dtID, const uint8_t* dataID, const double* obs,
                  const double* dataW, int64_t nAC, const int64_t* activeIdx, const double* bgModel,
                  char* err, int errlen) {
    Emul* e = new Emul();
    if (!e->hp.build(ny, nz, yLen, zLen, origin, nFreq, freqs, nRx, rxY, rxZ, nComp, compMode, nData, freqID,
                     rxID, dtID, dataID, obs, dataW, nAC, activeIdx, bgModel)) {
        std::snprintf(err, errlen, "%s", e->hp.error.c_str());
        delete e;
        return nullptr;
    }
    e->bind();
    return e;
}

void emul_destroy(void* h) { delete (Emul*)h; }

// precond: 0 Jacobi, 1 FDM.  pred: interleaved c128[nData]; grad[nAC]; iters[2*S]
int emul_grad(void* h, const double* m, int wantGrad, int precond, double tol, int maxit,
              double* pred, double* misfit, double* grad, int* iters) {
    Emul* e = (Emul*)h;
    e->run(m, wantGrad != 0, precond, tol, maxit, misfit);
    std::memcpy(pred, e->pred.data(), sizeof(cplx) * e->v.nData);
    if (wantGrad) std::memcpy(grad, e->grad.data(), sizeof(double) * e->v.nAC);
    std::memcpy(iters, e->iters.data(), sizeof(int) * 2 * e->v.S);
    return 0;
}

int emul_dims(void* h, int* out /* NYP, NZP, S, ny, nz, zid */) {
    Emul* e = (Emul*)h;
    out[0] = e->v.NYP; out[1] = e->v.NZP; out[2] = e->v.S; out[3] = e->v.ny; out[4] = e->v.nz; out[5] = e->v.zid;
    return 0;
}

// copies of work arrays for unit comparisons: which = 0 X, 1 Lam, 2 sigma, 3 Vpad, 4 lam, 5 gL, 6 gR, 7 gMn,
// 8 bcsL, 9 bcsR, 10 bcsB, 11 Zrx, 12 rxD, 13 gPart
long emul_get(void* h, int which, double* out, long cap) {
    Emul* e = (Emul*)h;
    const void* src = nullptr; size_t bytes = 0;
    auto setv = [&](const void* p, size_t b) { src = p; bytes = b; };
    switch (which) {
        case 0: setv(e->X.data(), e->X.size() * sizeof(cplx)); break;
        case 1: setv(e->Lam.data(), e->Lam.size() * sizeof(cplx)); break;
        case 2: setv(e->sigma.data(), e->sigma.size() * sizeof(double)); break;
        case 3: setv(e->hp.Vpad.data(), e->hp.Vpad.size() * sizeof(double)); break;
        case 4: setv(e->hp.lam.data(), e->hp.lam.size() * sizeof(double)); break;
        case 5: setv(e->gL.data(), e->gL.size() * sizeof(cplx)); break;
        case 6: setv(e->gR.data(), e->gR.size() * sizeof(cplx)); break;
        case 7: setv(e->gMn.data(), e->gMn.size() * sizeof(cplx)); break;
        case 8: setv(e->bcsL.data(), e->bcsL.size() * sizeof(cplx)); break;
        case 9: setv(e->bcsR.data(), e->bcsR.size() * sizeof(cplx)); break;
        case 10: setv(e->bcsB.data(), e->bcsB.size() * sizeof(cplx)); break;
        case 11: setv(e->Zrx.data(), e->Zrx.size() * sizeof(cplx)); break;
        case 12: setv(e->rxD.data(), e->rxD.size() * sizeof(cplx)); break;
        case 13: setv(e->gPart.data(), e->gPart.size() * sizeof(double)); break;
        default: return -1;
    }
    long n = (long)(bytes / sizeof(double));
    if (out && cap >= n) std::memcpy(out, src, bytes);
    return n;
}

// which = 0: q = A_s p for every system; 1: z = P_fdm^-1 r; 2: z = P_jacobi^-1 r   (padded nodal layout)
int emul_apply(void* h, int which, const double* in, double* out) {
    Emul* e = (Emul*)h;
    const View& v = e->v;
    const cplx* x = (const cplx*)in;
    cplx* y = (cplx*)out;
    std::memset(y, 0, sizeof(cplx) * v.S * v.vstride);
    for (int s = 0; s < v.S; ++s) {
        const cplx* xs = x + (long)s * v.vstride;
        cplx* ys = y + (long)s * v.vstride;
        if (which == 0) {
            for (int iz = 1; iz <= v.nz - 1; ++iz)
                for (int iy = 1; iy <= v.ny - 1; ++iy) ys[nidx(v, iy, iz)] = stencil_apply(v, s, xs, iy, iz);
        } else {
            e->precond(s, which == 1 ? 1 : (which == 3 ? 2 : 0), xs, ys);
        }
    }
    return 0;
}

}  // extern "C"
